"""Batched whole-body inverse-dynamics QP: the QP that the reference's
``InverseDynamics.get_joint_torques`` (code/inverse_dynamics.py:30-134) assembles and hands to ``QPSolver`` (CasADi conic
+ OSQP, code/utils.py:40-92) once per tick, for B robots at once on the GPU (csrc/wbc_qp.hip through the C ABI of
include/cmpc_wbc.h).  The Jacobians, the mass matrix and the Coriolis forces come from the caller's rigid-body library
(DART in the reference, :46-66, :107-111) as device tensors; nothing here imports the oracle, and there is no CPU
fallback."""
import ctypes

import torch

from . import capi

DOFS, BASE, CONTACT = 30, 6, 12

# weights and gains of code/inverse_dynamics.py:41-44
TASKS = ('lfoot', 'rfoot', 'com', 'torso', 'base', 'joints')
WEIGHTS = {'lfoot': 1., 'rfoot': 1., 'com': 1., 'torso': 1., 'base': 1., 'joints': 1.e-1}
POS_GAINS = {'lfoot': 10., 'rfoot': 10., 'com': 5., 'torso': 10., 'base': 10., 'joints': 10.}
VEL_GAINS = {'lfoot': 5., 'rfoot': 5., 'com': 10., 'torso': 5., 'base': 3., 'joints': 5.}


def assemble_task_cost(J, Jdot, ff, pos_error, vel_error, qd):
    """Hq (B,30,30), Fq (B,30) of code/inverse_dynamics.py:92-103 from batched task Jacobians J[task] (B,r,30), their
    derivatives, feed-forward accelerations and errors (B,r), and the joint velocities qd (B,30).  Plain torch
    (device-memory plumbing: two batched GEMMs per task)."""
    Hq = Fq = None
    for task in TASKS:
        Jt = J[task]
        acc = ff[task] + VEL_GAINS[task] * vel_error[task] + POS_GAINS[task] * pos_error[task] \
            - torch.einsum('brn,bn->br', Jdot[task], qd)
        Ht = WEIGHTS[task] * Jt.transpose(1, 2) @ Jt
        Ft = -WEIGHTS[task] * torch.einsum('brn,br->bn', Jt, acc)
        Hq = Ht if Hq is None else Hq + Ht
        Fq = Ft if Fq is None else Fq + Ft
    return Hq.contiguous(), Fq.contiguous()


class BatchedInverseDynamicsQP:
    """``InverseDynamics`` of the reference for a batch: ``solve`` returns what ``get_joint_torques`` returns
    (``tau[6:]``, :134) for every instance, plus the accelerations and contact wrenches of the QP."""

    def __init__(self, foot_size=0.1, mu=0.5, device=None, tol=1e-9, max_iter=60):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedInverseDynamicsQP needs a ROCm GPU: there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.d, self.mu, self.tol, self.max_iter = foot_size / 2.0, mu, tol, max_iter
        self._lib = capi.load()

    def solve(self, Hq, Fq, M, h, Jc):
        """Hq (B,30,30), Fq (B,30), M (B,30,30), h (B,30), Jc (B,12,30: rows already scaled by the contact flags, :109),
        contiguous fp64 on this GPU -> (tau_actuated (B,24), qdd (B,30), f_c (B,12), status (B,), iters (B,))."""
        B = Hq.shape[0]
        for name, t, shape in (("Hq", Hq, (B, DOFS, DOFS)), ("Fq", Fq, (B, DOFS)), ("M", M, (B, DOFS, DOFS)),
                               ("h", h, (B, DOFS)), ("Jc", Jc, (B, CONTACT, DOFS))):
            if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and tuple(t.shape) == shape
                    and t.device == self.device):
                raise ValueError(f"{name} must be a contiguous fp64 tensor of shape {shape} on {self.device}")
        tau = torch.empty((B, DOFS), dtype=torch.float64, device=self.device)
        qdd = torch.empty((B, DOFS), dtype=torch.float64, device=self.device)
        f = torch.empty((B, CONTACT), dtype=torch.float64, device=self.device)
        status = torch.empty(B, dtype=torch.int32, device=self.device)
        iters = torch.empty(B, dtype=torch.int32, device=self.device)
        if B == 0:
            return tau[:, BASE:], qdd, f, status, iters
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self._lib.cmpc_wbc_qp_solve_batch(self.device.index, B, Hq.data_ptr(), Fq.data_ptr(), M.data_ptr(), h.data_ptr(),
                                               Jc.data_ptr(), self.d, self.mu, self.tol, self.max_iter, tau.data_ptr(),
                                               qdd.data_ptr(), f.data_ptr(), status.data_ptr(), iters.data_ptr(),
                                               ctypes.c_void_p(stream))
        if rc != 0:
            raise RuntimeError(self._lib.cmpc_wbc_last_error().decode())
        return tau[:, BASE:], qdd, f, status, iters
