"""Closed-loop batched rollout: B independent copies of the controller loop, entirely on the GPU.

Per tick (the reference's ``customPreStep``, code/simulation.py:193-212, without DART; the single-instance
twin of this loop is ``walk.WalkHarness`` around the drop-in class):
  1. parameter records from the per-tick tables, the current centroidal state and the instance's OWN contact
     plan (``DeviceRecordBuilder`` = front half of ``centroidal_mpc.solve``, :482-600);
  2. batched solve, warm-started from the previous tick's solution, unshifted (:630-631);
  3. back half of ``solve`` (:614-683): theta_hat is carried from x_1 (:485, :644); the contact-plan
     write-back (:656-675) scatters the predicted landing point x_N of the swing foot into the instance's plan
     (``update_contact = 'YES'``), guarded by the per-instance ``update_contact_flag``; the state is advanced
     to the MPC's own prediction x_1 (perfect-tracking centroidal model standing in for simulator +
     whole-body controller), optionally disturbed by a velocity push.  The angular momentum is either the MPC's
     prediction or an exogenous measured signal ``hw_measured[t] (+ per-instance offset)`` (see walk.py).
  4. optionally the consumer of the tick, the whole-body inverse-dynamics QP (code/inverse_dynamics.py:30-134, called
     from code/simulation.py:214-232 with ``desired['com']`` = the MPC's CoM position / velocity / acceleration): a
     batched ``wbc.BatchedInverseDynamicsQP`` fed by the caller's rigid-body model (``attach_whole_body``), so that the
     whole tick -- records, MPC solve, write-back, QP -- stays on the device.
Everything after the solve is index arithmetic and copies in torch (device memory plumbing); the schedule
(phases, step indices) is shared by the batch, the plan positions are per instance.
"""
import numpy as np
import torch

from .solver import BatchedCentroidalMPC, DeviceRecordBuilder, usable


class BatchedRollout:
    def __init__(self, scene, spec, B, device="cuda:0", mass=None, mu=0.5, update_contact=True,
                 hw_measured=None, hw_offset=None, rate=1):
        self.scene, self.spec, self.B, self.rate = scene, spec, B, rate
        self.solver = BatchedCentroidalMPC(spec, device=device)
        self.device = self.solver.device
        self.builder = DeviceRecordBuilder(scene, device=self.device)
        dev, f64 = self.device, torch.float64
        self.state = torch.zeros((B, 16), dtype=f64, device=dev)
        self.state[:, 14] = scene.params['mass'] if mass is None else torch.as_tensor(mass, dtype=f64, device=dev)
        self.state[:, 15] = torch.as_tensor(mu, dtype=f64, device=dev)
        self.t = torch.zeros(B, dtype=torch.int32, device=dev)
        self.warm = None
        # solver states of the previous / this tick (cmpc_solve_batch_state), swapped every tick
        self._state = [self.solver.new_state(B), self.solver.new_state(B)]
        self.alive = torch.ones(B, dtype=torch.bool, device=dev)
        # per-instance plans and the shared schedule (:656-675)
        self.update_contact = update_contact
        self.plan_pos = torch.from_numpy(scene.plan_pos).to(dev).repeat(B, 1, 1).contiguous()
        self.flag = torch.zeros(B, dtype=torch.bool, device=dev)             # update_contact_flag
        self.counter = torch.zeros(B, dtype=torch.bool, device=dev)          # model_state['counter'] of the last tick
        T, N = scene.T, spec.N
        end = np.minimum(np.arange(T) + N * rate - 1, T - 1)
        cond = scene.is_ss & ~scene.is_ss[end]                                # now 'ss', horizon end 'ds'
        self._cond = torch.from_numpy(cond).to(dev)
        self._is_ds = torch.from_numpy(~scene.is_ss).to(dev)
        self._wb_slot = torch.from_numpy(np.minimum(scene.step_idx + 1, scene.plan_pos.shape[0] - 1).astype(np.int64)).to(dev)
        # support = lfoot -> the swing foot is the right one -> rows 17:20 of x_N, else rows 13:16
        self._wb_row = torch.from_numpy(np.where(scene.support_is_l, 17, 13).astype(np.int64)).to(dev)
        self.hw_measured = None if hw_measured is None else torch.as_tensor(np.asarray(hw_measured), dtype=f64, device=dev)
        self.hw_offset = None if hw_offset is None else torch.as_tensor(np.asarray(hw_offset), dtype=f64, device=dev)
        self._gl = torch.from_numpy(np.ascontiguousarray(scene.gl_tab)).to(dev)
        self._gr = torch.from_numpy(np.ascontiguousarray(scene.gr_tab)).to(dev)
        self._wbc = None

    def attach_whole_body(self, qp, model):
        """Run the whole-body QP inside every tick.  ``qp``: a ``wbc.BatchedInverseDynamicsQP``; ``model(rollout, desired)``
        returns the device tensors (Hq, Fq, M, h, Jc) of its ``solve`` from the caller's rigid-body library (DART in the
        reference, code/inverse_dynamics.py:46-66, :107-111) and ``desired`` = dict(com_pos, com_vel, com_acc (B,3),
        gamma_l, gamma_r (B,)) -- what code/simulation.py:214-232 hands over from the MPC's ``model_state``.  The
        result of the last tick is kept in ``last_wbc`` = (tau (B,24), qdd, f_c, status, iters)."""
        self._wbc = (qp, model)
        self.last_wbc = None

    def desired_com(self, x1, u0, t):
        """``model_state['com']`` of the reference's back half (:633-649) for the batch: position and velocity of x_1 and
        CoM_acc = (gamma_l sum F_l + gamma_r sum F_r) / m + (0, 0, -g) from u_0 and the contact flags at tick t."""
        nv = self.spec.nv
        tl = torch.clamp(t.long(), max=self._gl.shape[0] - 1)
        gl, gr = self._gl[tl], self._gr[tl]
        F = u0[:, :6 * nv].reshape(-1, 2, nv, 3).sum(dim=2)                  # (B, foot, 3)
        acc = (gl[:, None] * F[:, 0] + gr[:, None] * F[:, 1]) / self.state[:, 14:15]
        acc = acc + torch.tensor([0.0, 0.0, -self.spec.g], dtype=torch.float64, device=self.device)
        return {"com_pos": x1[:, 0:3], "com_vel": x1[:, 3:6], "com_acc": acc, "gamma_l": gl, "gamma_r": gr}

    def _hw_at(self, t):
        h = self.hw_measured[torch.clamp(t.long(), max=self.hw_measured.shape[0] - 1)]
        return h if self.hw_offset is None else h + self.hw_offset

    def reset(self, t0, com, dcom, hw=None, theta_hat=None):
        dev, f64 = self.device, torch.float64
        self.t[:] = torch.as_tensor(t0, dtype=torch.int32, device=dev)
        self.state[:, 0:3] = torch.as_tensor(com, dtype=f64, device=dev)
        self.state[:, 3:6] = torch.as_tensor(dcom, dtype=f64, device=dev)
        if hw is not None:
            self.state[:, 6:9] = torch.as_tensor(hw, dtype=f64, device=dev)
        elif self.hw_measured is not None:
            self.state[:, 6:9] = self._hw_at(self.t)
        else:
            self.state[:, 6:9] = 0.0
        self.state[:, 9:12] = 0.0 if theta_hat is None else torch.as_tensor(theta_hat, dtype=f64, device=dev)
        self.state[:, 12:14] = 0.0
        self.warm = None
        self._state[0].zero_()
        self.alive[:] = True
        self.flag[:] = False
        self.plan_pos = torch.from_numpy(self.scene.plan_pos).to(dev).repeat(self.B, 1, 1).contiguous()

    def step(self, push_dv=None):
        """One control tick for every instance.  Returns (x1 (B,20), u0 (B,nu), status (B,))."""
        sp, N = self.spec, self.spec.N
        rec = self.builder.build(sp, self.t, self.state, rate=self.rate,
                                 plan_pos=self.plan_pos if self.update_contact else None)
        s_in, s_out = self._state
        XU, status, iters, kkt = self.solver.solve(rec, warm=self.warm, state=s_in, state_out=s_out)
        ok = usable(status) & self.alive
        self._state = [s_out, s_in]              # (an instance whose solve failed stops for good, see `alive`)
        x1 = XU[:, 20:40]
        u0 = XU[:, 20 * (N + 1):20 * (N + 1) + sp.nu]
        self.last_records, self.last_XU, self.last_status, self.last_iters = rec, XU, status, iters
        if self._wbc is not None:                # the consumer of the tick: whole-body QP on the same stream, no host hop
            qp, model = self._wbc
            self.last_wbc = qp.solve(*model(self, self.desired_com(x1, u0, self.t)))
        # plan write-back (:656-675), per instance
        tl = self.t.long()
        if self.update_contact:
            fire = self._cond[tl] & ~self.flag & ok
            rows = self._wb_row[tl]
            cols = 20 * N + rows[:, None] + torch.arange(3, device=self.device)[None, :]
            landing = torch.gather(XU, 1, cols)                               # x_collect[17:20 or 13:16, N]
            b = torch.nonzero(fire, as_tuple=True)[0]
            self.plan_pos[b, self._wb_slot[tl][b]] = landing[b]
            self.flag = (self.flag | fire) & ~(self._is_ds[tl] & ok)
            self.counter = fire
        # instances whose solve failed stop moving (the reference raises, :605-614); the rest advance
        self.alive = ok
        nxt = self.state.clone()
        nxt[:, 0:12] = x1[:, 0:12]
        if self.hw_measured is not None:
            nxt[:, 6:9] = self._hw_at(self.t + self.rate)
        if push_dv is not None:
            nxt[:, 3:6] += torch.as_tensor(push_dv, dtype=torch.float64, device=self.device)
        self.state = torch.where(ok[:, None], nxt, self.state)
        # x_1 is the state delta = rate * world_time_step ahead, and the reference solves every `rate`-th tick
        # (code/simulation.py:203): schedule time moves with the state
        self.t = torch.where(ok, self.t + self.rate, self.t).to(torch.int32)
        self.warm = XU if self.warm is None else torch.where(ok[:, None], XU, self.warm)
        return x1, u0, status

    def run(self, ticks, push=None):
        """`ticks` control steps; push = (first_tick, last_tick, dv(3)) velocity disturbance per tick.
        Returns the CoM history (ticks+1, B, 3) and the final alive mask."""
        hist = [self.state[:, 0:3].clone()]
        for i in range(ticks):
            dv = push[2] if (push is not None and push[0] <= i <= push[1]) else None
            self.step(dv)
            hist.append(self.state[:, 0:3].clone())
        return torch.stack(hist), self.alive
