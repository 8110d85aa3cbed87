"""Host-side footstep planner (input side of the centroidal-MPC hot path).

Mirrors the public surface of the reference planner so that the MPC class and
the batched parameter builder can be driven without the reference checkout:

* ``FootstepPlanner(vref, initial_lfoot, initial_rfoot, params)``
  (reference: code/footstep_planner_vertices.py:6-69)
* ``.plan``  list of dicts ``{'pos','ang','ss_duration','ds_duration','foot_id'}``
* ``.position_contacts_ref``  ``{'contact_left','contact_right'}`` arrays (T,6),
  rows ``[ang(3), pos(3)]``  (reference :106-147)
* ``get_step_index_at_time / get_start_time / get_phase_at_time``
  (reference :82-103)

In addition the planner exposes array views (``step_end_times`` ...) that the
batched parameter builder consumes; they are derived from ``.plan`` on demand so
that the MPC's plan write-back (reference centroidal_mpc_vertices.py:669,672)
is always seen.
"""
import numpy as np


class FootstepPlanner:
    def __init__(self, vref, initial_lfoot, initial_rfoot, params):
        ss_default = params['ss_duration']
        ds_default = params['ds_duration']
        dt = params['world_time_step']

        initial_lfoot = np.asarray(initial_lfoot, dtype=np.float64)
        initial_rfoot = np.asarray(initial_rfoot, dtype=np.float64)
        # virtual unicycle starts between the feet (reference :17-18)
        uni_xy = (initial_lfoot[3:5] + initial_rfoot[3:5]) / 2.
        uni_th = (initial_lfoot[2] + initial_rfoot[2]) / 2.
        foot = params['first_swing']
        self.plan = []

        for j, cmd in enumerate(vref):
            ss, ds = ss_default, ds_default
            if j == 0:
                # long initial double support (reference :29-31)
                ss, ds = 0, (ss_default + ds_default) * 2
            if j > 1:
                # unicycle integration, one Euler step per tick (reference :38-43)
                for _ in range(ss + ds):
                    uni_th += cmd[2] * dt
                    c, s = np.cos(uni_th), np.sin(uni_th)
                    rot = np.array([[c, -s], [s, c]])
                    uni_xy = uni_xy + rot @ np.asarray(cmd[:2], dtype=np.float64) * dt
            lateral = 0.1 if foot == 'lfoot' else -0.1
            pos = np.array((uni_xy[0] - np.sin(uni_th) * lateral,
                            uni_xy[1] + np.cos(uni_th) * lateral,
                            0.))
            ang = np.array((0., 0., uni_th))
            self.plan.append({'pos': pos, 'ang': ang, 'ss_duration': ss,
                              'ds_duration': ds, 'foot_id': foot})
            foot = 'rfoot' if foot == 'lfoot' else 'lfoot'

        self.position_contacts_ref = self.gen_pos_contacts_ref_at_time(params)

    # ---- time lookups (reference :82-103) -------------------------------
    def get_step_index_at_time(self, time):
        t = 0
        for i, step in enumerate(self.plan):
            t += step['ss_duration'] + step['ds_duration']
            if t > time:
                return i
        return None

    def get_start_time(self, step_index):
        t = 0
        for i in range(step_index):
            t += self.plan[i]['ss_duration'] + self.plan[i]['ds_duration']
        return t

    def get_phase_at_time(self, time):
        idx = self.get_step_index_at_time(time)
        in_step = time - self.get_start_time(idx)
        return 'ss' if in_step < self.plan[idx]['ss_duration'] else 'ds'

    # ---- per-tick nominal contact table (reference :106-147) ------------
    @staticmethod
    def left_right_plan_indices(index, first_swing):
        """Plan entries that hold the (left, right) contact during step `index`."""
        if index < 2:
            il, ir = (2 * index, 1) if first_swing == 'lfoot' else (1, 2 * index)
        else:
            a, b = index + (index % 2), index + (index - 1) % 2
            il, ir = (a, b) if first_swing == 'lfoot' else (b, a)
        return il, ir

    def gen_pos_contacts_ref_at_time(self, params):
        first_swing = params['first_swing']
        sim_time = int(len(self.plan) / params['world_time_step'])
        left, right = [], []
        for i in range(sim_time):
            il, ir = self.left_right_plan_indices(self.get_step_index_at_time(i), first_swing)
            left.append(np.hstack((self.plan[il]['ang'], self.plan[il]['pos'])))
            right.append(np.hstack((self.plan[ir]['ang'], self.plan[ir]['pos'])))
        return {'contact_left': np.array(left), 'contact_right': np.array(right)}

    # ---- array views for the batched parameter builder ------------------
    def step_end_times(self):
        """Cumulative end time (exclusive) of every step, int64 (len(plan),)."""
        d = np.array([s['ss_duration'] + s['ds_duration'] for s in self.plan], dtype=np.int64)
        return np.cumsum(d)

    def ss_durations(self):
        return np.array([s['ss_duration'] for s in self.plan], dtype=np.int64)

    def support_is_left(self):
        return np.array([s['foot_id'] == 'lfoot' for s in self.plan], dtype=np.int64)

    def plan_positions(self):
        return np.array([s['pos'] for s in self.plan], dtype=np.float64)
