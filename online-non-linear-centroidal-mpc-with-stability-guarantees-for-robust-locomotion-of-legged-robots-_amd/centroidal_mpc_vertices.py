"""Drop-in ``centroidal_mpc`` class: same constructor, ``solve(current, t)`` and side effects as the
reference (code/centroidal_mpc_vertices.py:6-686), with the per-tick CasADi/IPOPT solve (:606)
replaced by the batched HIP solver (batch of one).

Kept identical to the reference:
  * ctor signature / params keys read (:7-18, :29, :33)
  * x0 assembly incl. foot positions taken from the plan (:482-509)
  * contact flags and reference sampling (:515-600)  -> ``problem.build_record``
  * warm start = previous primal solution, unshifted (:630-631)
  * ``model_state`` dict object, its keys and value shapes (:358-366, :639-649); the SAME dict
    object is returned on every call (the driver mutates it, simulation.py:247)
  * plan write-back of the predicted landing position (:656-675) and the returned contact (:679-683)
  * failure -> ``RuntimeError`` (the reference's uncaught failure path, :605-614); a solve that stops at the
    acceptable level (status 3: KKT error <= 1e-4, the reference's IPOPT tolerance is 1e-3) is a success
The reference's print() calls are dropped (``verbose=True`` restores a one-line summary).
"""
import numpy as np
import torch

from .problem import ProblemSpec, build_record
from .solver import BatchedCentroidalMPC, STATUS_CONVERGED, STATUS_ACCEPTABLE


class centroidal_mpc:
    _payload = False

    def __init__(self, initial, footstep_planner, params, CoM_ref, contact_trj_l, contact_trj_r,
                 device=None, verbose=False):
        self.params = params
        self.N = params['N']
        self.delta = params['world_time_step'] * params['mpc_rate']
        self.h = params['h']
        self.eta = params['eta']
        self.foot_size = params['foot_size']
        self.mass = params['mass']
        self.g = params['g']
        self.initial = initial
        self.footstep_planner = footstep_planner
        self.mpc_rate = params['mpc_rate']
        self.update_contact_flag = 0
        self.update_swing_trj = 0
        self.verbose = verbose

        self.spec = ProblemSpec.from_params(params, payload=self._payload)
        self.k1, self.k2 = self.spec.k1, self.spec.k2
        self.CoM_ref = CoM_ref
        self.contact_trj_l = contact_trj_l
        self.contact_trj_r = contact_trj_r
        # nominal contact table captured once, like the reference (:77-84)
        self._contacts_ref = {
            'contact_left': footstep_planner.position_contacts_ref['contact_left'],
            'contact_right': footstep_planner.position_contacts_ref['contact_right']}

        self._solver = self._make_solver(device)
        self._warm = None                       # previous solution (device tensor)
        self._state = None                      # solver state of the previous tick (cmpc_solve_batch_state), device tensor
        self.last_status, self.last_iterations, self.last_kkt = None, 0, float('nan')
        self.x = np.zeros(20)
        self.u = np.zeros(self.spec.nu)
        self.x_collect = np.zeros((20, self.N + 1))
        self.current_state = np.zeros(20)

        self.model_state = {'com': {'pos': np.zeros(3), 'vel': np.zeros(3), 'acc': np.zeros(3)},
                            'hw': {'val': np.zeros(3), 'dot': np.zeros(3)},
                            'theta_hat': {'val': np.zeros(3)},
                            'ang_contact_left': {'val': np.zeros(3)},
                            'pos_contact_left': {'val': np.zeros(3)},
                            'ang_contact_right': {'val': np.zeros(3)},
                            'pos_contact_right': {'val': np.zeros(3)},
                            'mpc_new_contact': {'val': np.zeros(3)},
                            'counter': {'val': 0}}

    # ------------------------------------------------------------------------------------
    def _make_solver(self, device):
        """The HIP solver handle (raises without a ROCm GPU: there is no CPU fallback)."""
        solver = BatchedCentroidalMPC(self.spec, device=device)
        self._device = solver.device
        return solver

    def _solve_record(self, rec):
        """opt.solve() of the reference (:606) for one parameter record -> (solution (nsol,) numpy, status,
        iterations, kkt).  Warm start = previous primal solution, unshifted (:630-631), plus the interior point
        method's own state of the previous tick (the central-path point it resumes from, include/cmpc.h)."""
        d_rec = torch.from_numpy(rec[None, :]).to(self._device)
        state_out = self._solver.new_state(1)
        out, status, iters, kkt = self._solver.solve(d_rec, warm=self._warm, state=self._state, state_out=state_out)
        st = int(status.item())
        if st in (STATUS_CONVERGED, STATUS_ACCEPTABLE):
            self._warm = out                                    # set_initial(U*, X*), unshifted
            self._state = state_out
        return out[0].cpu().numpy(), st, int(iters.item()), float(kkt.item())

    def solve(self, current, t):
        N, nu, nv = self.N, self.spec.nu, self.spec.nv
        rec = build_record(
            self.spec, self.footstep_planner, self.CoM_ref, t,
            com=current['com']['pos'][0:3], dcom=current['com']['vel'][0:3], hw=current['hw']['val'][0:3],
            theta_hat=self.model_state['theta_hat']['val'],
            yaw_l=current['lfoot']['pos'][2], yaw_r=current['rfoot']['pos'][2],
            mass=self.mass, mu=0.5, first_swing=self.params['first_swing'], rate=self.mpc_rate,
            contacts_ref=self._contacts_ref)
        self.current_state = rec[0:20].copy()
        self.last_record = rec
        sol, st, n_it, kkt = self._solve_record(rec)
        self.last_status, self.last_iterations, self.last_kkt = st, n_it, kkt
        # The reference returns from opt.solve() on IPOPT's Solve_Succeeded AND Solved_To_Acceptable_Level at
        # tol = 1e-3 (:128); status 3 is a KKT error within spec.acc_tol = 1e-4, tighter than either.  Anything
        # else (locally infeasible, iteration cap) is the reference's RuntimeError path (:605-614).
        if st not in (STATUS_CONVERGED, STATUS_ACCEPTABLE):
            raise RuntimeError(f"centroidal MPC solve failed at t={t}: status {st}, "
                               f"iterations {n_it}, KKT error {kkt:.3e}")
        X = sol[:20 * (N + 1)].reshape(N + 1, 20).T             # opti_state, 20 x (N+1)
        U = sol[20 * (N + 1):].reshape(N, nu).T                 # U, nu x N
        self.x = X[:, 1].copy()
        self.u = U[:, 0].copy()
        self.x_collect = X.copy()
        gl0, gr0 = rec[24 + 17], rec[24 + 18]

        F = self.u[:6 * nv].reshape(2 * nv, 3)
        Vl, Vr = F[:nv].sum(0), F[nv:].sum(0)
        com_acc = (gl0 * Vl + gr0 * Vr) / self.mass + np.array([0, 0, -self.g])
        # hw.dot = (0.01 * f(x_0, ., u_0))[6:9] * delta * mpc_rate   (:283-284, :643)
        tau = np.zeros(3)
        verts = self.spec.vertices()
        x0 = X[:, 0]
        for f, (gam, iy, ip) in enumerate(((gl0, 12, 13), (gr0, 16, 17))):
            c, s = np.cos(x0[iy]), np.sin(x0[iy])
            for j in range(nv):
                rv = np.array([c * verts[j, 0] - s * verts[j, 1], s * verts[j, 0] + c * verts[j, 1], 0.])
                tau += gam * np.cross(x0[ip:ip + 3] + rv - x0[0:3], F[f * nv + j])

        ms = self.model_state
        ms['com']['pos'] = self.x[0:3].copy()
        ms['com']['vel'] = self.x[3:6].copy()
        ms['com']['acc'] = com_acc
        ms['hw']['val'] = self.x[6:9].copy()
        ms['hw']['dot'] = 0.01 * tau * self.delta * self.mpc_rate
        ms['theta_hat']['val'] = self.x[9:12].copy()
        ms['ang_contact_left']['val'] = self.x[12]
        ms['pos_contact_left']['val'] = self.x[13:16].copy()
        ms['ang_contact_right']['val'] = self.x[16]
        ms['pos_contact_right']['val'] = self.x[17:20].copy()
        ms['counter']['val'] = 0
        if self.verbose:
            print(f"time in solve():{t}  iterations {n_it}  kkt {kkt:.2e}")

        planner = self.footstep_planner
        if self.params['update_contact'] == 'YES':
            update_step = 1
            phase_now = planner.get_phase_at_time(t)
            phase_end = planner.get_phase_at_time(t + self.N * self.mpc_rate - update_step)
            if phase_now == 'ss' and phase_end == 'ds' and self.update_contact_flag == 0:
                self.update_contact_flag = 1
                ms['counter']['val'] = self.update_contact_flag
                idx = planner.get_step_index_at_time(t)
                if planner.plan[idx]['foot_id'] == 'lfoot':         # swing foot is the right one
                    planner.plan[idx + 1]['pos'] = self.x_collect[17:20, self.N].copy()
                    ms['mpc_new_contact']['val'] = self.x_collect[17:20, self.N].copy()
                else:
                    planner.plan[idx + 1]['pos'] = self.x_collect[13:16, self.N].copy()
                    ms['mpc_new_contact']['val'] = self.x_collect[13:16, self.N].copy()
            if phase_now == 'ds':
                self.update_contact_flag = 0

        contact = planner.get_phase_at_time(t)
        if contact == 'ss':
            contact = planner.plan[planner.get_step_index_at_time(t)]['foot_id']
        return self.model_state, contact

    def reset_update_swing_trj(self):
        self.update_swing_trj = 0


class centroidal_mpc_payload(centroidal_mpc):
    """Gains of code/centroidal_mpc_vertices_payload.py:27-31 (k1, k2 = 7, 1)."""
    _payload = True
