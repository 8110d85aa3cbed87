"""Closed-loop batched rollout: B independent copies of the controller loop, entirely on the GPU.

Per tick (the reference's ``customPreStep``, code/simulation.py:193-212, without DART):
  1. parameter records from the per-tick tables and the current centroidal state
     (``DeviceRecordBuilder`` = front half of ``centroidal_mpc.solve``, :482-600);
  2. batched solve, warm-started from the previous tick's solution, unshifted (:630-631);
  3. back half of ``solve`` (:614-649): theta_hat is carried from x_1 (:485, :644) and the state is
     advanced to the MPC's own prediction x_1 (a perfect-tracking centroidal model stands in for the
     simulator + whole-body controller), optionally disturbed by a velocity push.
The contact-plan write-back (:656-675) mutates a per-robot plan and is only done by the single-instance
``centroidal_mpc`` class; here all instances share the nominal plan (``update_contact = 'NO'``).
"""
import torch

from .solver import BatchedCentroidalMPC, DeviceRecordBuilder, usable


class BatchedRollout:
    def __init__(self, scene, spec, B, device="cuda:0", mass=None, mu=0.5):
        self.scene, self.spec, self.B = scene, spec, B
        self.device = torch.device(device)
        self.solver = BatchedCentroidalMPC(spec, device=self.device)
        self.builder = DeviceRecordBuilder(scene, device=self.device)
        self.state = torch.zeros((B, 16), dtype=torch.float64, device=self.device)
        self.state[:, 14] = scene.params['mass'] if mass is None else torch.as_tensor(mass, device=self.device)
        self.state[:, 15] = torch.as_tensor(mu, dtype=torch.float64, device=self.device)
        self.t = torch.zeros(B, dtype=torch.int32, device=self.device)
        self.warm = None
        self.alive = torch.ones(B, dtype=torch.bool, device=self.device)

    def reset(self, t0, com, dcom, hw=None, theta_hat=None):
        self.t[:] = torch.as_tensor(t0, dtype=torch.int32, device=self.device)
        self.state[:, 0:3] = torch.as_tensor(com, dtype=torch.float64, device=self.device)
        self.state[:, 3:6] = torch.as_tensor(dcom, dtype=torch.float64, device=self.device)
        self.state[:, 6:9] = 0.0 if hw is None else torch.as_tensor(hw, dtype=torch.float64, device=self.device)
        self.state[:, 9:12] = 0.0 if theta_hat is None else torch.as_tensor(theta_hat, dtype=torch.float64, device=self.device)
        self.state[:, 12:14] = 0.0
        self.warm = None
        self.alive[:] = True

    def step(self, push_dv=None):
        """One control tick for every instance.  Returns (x1 (B,20), u0 (B,nu), status (B,))."""
        sp = self.spec
        rec = self.builder.build(sp, self.t, self.state)
        XU, status, iters, kkt = self.solver.solve(rec, warm=self.warm)
        ok = usable(status) & self.alive
        x1 = XU[:, 20:40]
        u0 = XU[:, 20 * (sp.N + 1):20 * (sp.N + 1) + sp.nu]
        # instances whose solve failed stop moving (the reference raises, :605-614); the rest advance
        self.alive = ok
        nxt = self.state.clone()
        nxt[:, 0:12] = x1[:, 0:12]
        if push_dv is not None:
            nxt[:, 3:6] += torch.as_tensor(push_dv, dtype=torch.float64, device=self.device)
        self.state = torch.where(ok[:, None], nxt, self.state)
        self.t = torch.where(ok, self.t + 1, self.t).to(torch.int32)
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
