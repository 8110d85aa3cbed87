"""Headless stand-in for the reference's simulator loop (``Hrp4Controller.customPreStep``,
code/simulation.py:193-300) around the drop-in ``centroidal_mpc`` class: BASELINE config 1, the
flat-ground walk, without DART.

Per tick, exactly the reference's call pattern:
  current = retrieve_state()                      (:201)   here: the harness state
  model_state, contact = mpc.solve(current, t)    (:204)
  counter == 1 -> reset counter / swing flag      (:245-248)
  feet = foot_trajectory_generator...at_time(t)   (:227)   sees the plan the MPC has just rewritten
and, instead of inverse dynamics + physics (:276-281, out of scope), a perfect-tracking centroidal
model: the next measured CoM position / velocity is the MPC's x_1.  The angular momentum about the CoM
is NOT controlled by the reference's whole-body QP (it only consumes ``desired['com']``,
inverse_dynamics.py:30-134); in the simulator it is whatever the swinging limbs produce.  The harness
therefore takes it as an exogenous measured signal ``hw_measured[t]`` -- the reference ships such a
recording of its own run (original_code/cuhw.txt, kept as tests/golden/measured_hw_cuhw.txt).  With
``hw_measured=None`` the MPC's own prediction is fed back, which drives |hw_0| to zero and, in late
single support, the reference formulation to the edge of feasibility.
"""
import numpy as np


class WalkHarness:
    def __init__(self, mpc, planner, foot_trajectory_generator, params, initial, hw_measured=None):
        self.mpc, self.planner, self.ftg, self.params = mpc, planner, foot_trajectory_generator, params
        self.hw_measured = None if hw_measured is None else np.asarray(hw_measured, dtype=np.float64)
        self.time = 0
        self.com = np.array(initial['com']['pos'], dtype=np.float64)
        self.dcom = np.array(initial['com']['vel'], dtype=np.float64)
        self.hw = np.zeros(3) if self.hw_measured is None else self.hw_measured[0].copy()
        self.lfoot = np.array(initial['lfoot']['pos'], dtype=np.float64)
        self.rfoot = np.array(initial['rfoot']['pos'], dtype=np.float64)
        self.log = {k: [] for k in ('t', 'com', 'contact', 'status', 'iterations', 'kkt', 'counter',
                                    'mpc_new_contact', 'hw_des', 'swing_target')}

    def retrieve_state(self):
        return {'com': {'pos': self.com.copy(), 'vel': self.dcom.copy()}, 'hw': {'val': self.hw.copy()},
                'lfoot': {'pos': self.lfoot.copy()}, 'rfoot': {'pos': self.rfoot.copy()}}

    def step(self):
        t, mpc = self.time, self.mpc
        current = self.retrieve_state()
        if t % self.params['mpc_rate'] == 0:
            self.mpc_robot_state, self.mpc_contact = mpc.solve(current, t)
        state, contact = self.mpc_robot_state, self.mpc_contact
        counter = state['counter']['val']
        feet = self.ftg.generate_feet_trajectories_at_time(t)
        log = self.log
        log['t'].append(t); log['com'].append(state['com']['pos'].copy()); log['contact'].append(contact)
        log['status'].append(getattr(mpc, 'last_status', 0)); log['iterations'].append(getattr(mpc, 'last_iterations', 0))
        log['kkt'].append(getattr(mpc, 'last_kkt', 0.0)); log['counter'].append(counter)
        log['mpc_new_contact'].append(np.array(state['mpc_new_contact']['val'], dtype=np.float64).copy())
        log['hw_des'].append(state['hw']['val'].copy())
        swing = None if contact == 'ds' else ('rfoot' if contact == 'lfoot' else 'lfoot')
        log['swing_target'].append(None if swing is None else
                                   self.planner.plan[self.planner.get_step_index_at_time(t) + 1]['pos'].copy())
        if counter == 1:                                    # simulation.py:245-248
            state['counter']['val'] = 0
            mpc.reset_update_swing_trj()
        # perfect tracking of the centroidal reference; the feet follow their generated trajectories
        self.com, self.dcom = state['com']['pos'].copy(), state['com']['vel'].copy()
        self.lfoot, self.rfoot = np.array(feet['lfoot']['pos']), np.array(feet['rfoot']['pos'])
        self.time = t + 1
        if self.hw_measured is not None:
            self.hw = self.hw_measured[min(self.time, len(self.hw_measured) - 1)].copy()
        else:
            self.hw = state['hw']['val'].copy()
        return state, contact

    def run(self, ticks):
        for _ in range(ticks):
            self.step()
        return {k: (np.array(v) if k in ('t', 'com', 'status', 'iterations', 'kkt', 'counter', 'hw_des') else v)
                for k, v in self.log.items()}
