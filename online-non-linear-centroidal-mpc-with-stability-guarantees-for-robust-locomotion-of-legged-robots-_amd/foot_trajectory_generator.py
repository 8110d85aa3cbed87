"""Swing-foot reference (host side; feeds the CoM-reference knots only).

Mirrors ``FootTrajectoryGenerator.generate_feet_trajectories_at_time``
(reference: code/foot_trajectory_generator.py:12-114): support foot pinned to
its plan entry, swing foot on a cubic blend between plan[i-1] and plan[i+1] with
a quartic vertical bump of ``step_height``.  ``generate_feet_trajectories_pre``
(reference :117-257) tabulates the same function over the whole run.
"""
import numpy as np


class FootTrajectoryGenerator:
    def __init__(self, initial, footstep_planner, params):
        self.delta = params['world_time_step']
        self.step_height = params['step_height']
        self.initial = initial
        self.footstep_planner = footstep_planner
        self.plan = footstep_planner.plan
        self.first_swing = params['first_swing']

    def generate_feet_trajectories_at_time(self, time):
        fp = self.footstep_planner
        idx = fp.get_step_index_at_time(time)
        t = time - fp.get_start_time(idx)
        step = fp.plan[idx]
        support = step['foot_id']
        swing = 'lfoot' if support == 'rfoot' else 'rfoot'
        z6 = np.zeros(6)

        if idx == 0:                                   # reference :21-35
            return {f: {'pos': self.initial[f]['pos'], 'vel': z6, 'acc': z6}
                    for f in ('lfoot', 'rfoot')}

        sup_pose = np.hstack((step['ang'], step['pos']))
        if t >= step['ss_duration']:                   # double support, reference :38-60
            nxt = fp.plan[idx + 1]
            return {support: {'pos': sup_pose, 'vel': z6, 'acc': z6},
                    swing: {'pos': np.hstack((nxt['ang'], nxt['pos'])), 'vel': z6, 'acc': z6}}

        # single support: cubic in the plane / yaw, quartic in z (reference :63-90)
        p0, p1 = fp.plan[idx - 1]['pos'], fp.plan[idx + 1]['pos']
        a0, a1 = fp.plan[idx - 1]['ang'], fp.plan[idx + 1]['ang']
        T = step['ss_duration']
        A, B = -2 / T**3, 3 / T**2
        s0 = A * t**3 + B * t**2
        s1 = (3 * A * t**2 + 2 * B * t) / self.delta
        s2 = (6 * A * t + 2 * B) / self.delta**2
        pos, vel, acc = p0 + (p1 - p0) * s0, (p1 - p0) * s1, (p1 - p0) * s2
        ang, dang, ddang = a0 + (a1 - a0) * s0, (a1 - a0) * s1, (a1 - a0) * s2
        h = self.step_height
        A, B, C = 16 * h / T**4, -32 * h / T**3, 16 * h / T**2
        pos[2] = A * t**4 + B * t**3 + C * t**2
        vel[2] = (4 * A * t**3 + 3 * B * t**2 + 2 * C * t) / self.delta
        acc[2] = (12 * A * t**2 + 6 * B * t + 2 * C) / self.delta**2
        return {support: {'pos': sup_pose, 'vel': z6, 'acc': z6},
                swing: {'pos': np.hstack((ang, pos)), 'vel': np.hstack((dang, vel)),
                        'acc': np.hstack((ddang, acc))}}

    def generate_feet_trajectories_pre(self):
        """Per-tick table; entry layout ``traj[foot][t][0]['pos']`` as in the reference."""
        sim_time = int(len(self.plan) / self.delta)
        out = {'lfoot': [], 'rfoot': []}
        for time in range(sim_time):
            now = self.generate_feet_trajectories_at_time(time)
            for foot in out:
                out[foot].append([now[foot]])
        return out
