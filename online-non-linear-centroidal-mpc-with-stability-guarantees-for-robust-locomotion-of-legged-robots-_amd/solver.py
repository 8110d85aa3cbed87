"""Batched solver front end: torch tensors in, torch tensors out, HIP kernels in between.

``BatchedCentroidalMPC`` owns one C-ABI handle (one GPU).  PyTorch is only the allocator
and stream provider; all arithmetic happens in libcmpc_amd.so.  It replaces the
reference's ``self.opt.solve()`` (code/centroidal_mpc_vertices.py:606) for B instances at once.
"""
import ctypes

import torch

from . import capi
from .problem import ProblemSpec, to_cspec

# per-instance outcome (include/cmpc.h): 3 = stopped short of `tol` with a KKT error within `acc_tol`
# (IPOPT's "Solved To Acceptable Level", which CasADi's Opti.solve() returns without raising)
STATUS_CONVERGED, STATUS_MAX_ITER, STATUS_INFEASIBLE, STATUS_ACCEPTABLE = 0, 1, 2, 3
STATUS_NUMERICAL = STATUS_INFEASIBLE


def usable(status):
    """Boolean mask of the instances whose solution a caller may use: converged or acceptable."""
    return (status == STATUS_CONVERGED) | (status == STATUS_ACCEPTABLE)


def _device_of(device):
    """torch.device with an explicit index ('cuda' -> the current device)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise ValueError("the solver runs on ROCm GPUs only (device must be a cuda device)")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


class BatchedCentroidalMPC:
    def __init__(self, spec: ProblemSpec, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedCentroidalMPC needs a ROCm GPU: there is no CPU fallback")
        self.spec = spec
        self.device = _device_of(device)
        self._lib = capi.load()
        self._cspec = to_cspec(spec)
        h = ctypes.c_void_p()
        rc = self._lib.cmpc_create(ctypes.byref(self._cspec), self.device.index, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError("cmpc_create failed: " + self._lib.cmpc_last_error(None).decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cmpc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def workspace_bytes(self, B):
        return self._lib.cmpc_workspace_bytes(ctypes.byref(self._cspec), B)

    def solve(self, records, warm=None, out=None, state=None, state_out=None):
        """records (B, nrec) fp64 on this GPU -> (XU (B, nsol), status, iters, kkt), all on the GPU.

        Asynchronous on torch's current stream.  ``warm`` (B, nsol) is the previous solution
        (initial guess and proximal centre), the batched form of ``opt.set_initial``
        (code/centroidal_mpc_vertices.py:630-631).  Closed-loop ticks additionally hand the solver state over:
        ``state`` (B, nstate) = the previous tick's ``state_out`` (``new_state()`` for the first tick), and this
        tick's state is written to ``state_out`` (a different tensor): ``cmpc_solve_batch_state``.
        """
        sp = self.spec
        if not (records.is_cuda and records.dtype == torch.float64 and records.is_contiguous()):
            raise ValueError("records must be a contiguous fp64 CUDA tensor")
        if records.device != self.device:
            raise ValueError(f"records live on {records.device}, this handle on {self.device}")
        if records.dim() != 2 or records.shape[1] != sp.nrec:
            raise ValueError(f"records must have shape (B, {sp.nrec})")
        B = records.shape[0]
        if warm is not None:
            if not (warm.is_cuda and warm.dtype == torch.float64 and warm.is_contiguous()
                    and tuple(warm.shape) == (B, sp.nsol) and warm.device == self.device):
                raise ValueError(f"warm must be a contiguous fp64 CUDA tensor of shape (B, {sp.nsol})")
        for name, t in (("state", state), ("state_out", state_out)):
            if t is not None and not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
                                      and tuple(t.shape) == (B, sp.nstate) and t.device == self.device):
                raise ValueError(f"{name} must be a contiguous fp64 CUDA tensor of shape (B, {sp.nstate})")
        if state is not None and state_out is not None:
            a, b, nb = state.data_ptr(), state_out.data_ptr(), B * sp.nstate * 8
            if a < b + nb and b < a + nb:                      # (views of one buffer included; the C entry point checks too)
                raise ValueError("state and state_out must not overlap")
        if out is None:
            out = torch.empty((B, sp.nsol), dtype=torch.float64, device=records.device)
        status = torch.empty(B, dtype=torch.int32, device=records.device)
        iters = torch.empty(B, dtype=torch.int32, device=records.device)
        kkt = torch.empty(B, dtype=torch.float64, device=records.device)
        if B == 0:
            return out, status, iters, kkt
        stream = torch.cuda.current_stream(records.device).cuda_stream
        rc = self._lib.cmpc_solve_batch_state(self._h, B, records.data_ptr(),
                                              warm.data_ptr() if warm is not None else None,
                                              state.data_ptr() if state is not None else None,
                                              out.data_ptr(),
                                              state_out.data_ptr() if state_out is not None else None,
                                              status.data_ptr(), iters.data_ptr(), kkt.data_ptr(),
                                              ctypes.c_void_p(stream))
        if rc != 0:
            raise RuntimeError("cmpc_solve_batch failed: " + self._lib.cmpc_last_error(self._h).decode())
        return out, status, iters, kkt

    def new_state(self, B):
        """An empty solver state for B instances (barrier word 0 = "no state": the first tick starts cold)."""
        return torch.zeros((B, self.spec.nstate), dtype=torch.float64, device=self.device)

    def last_kernel_name(self):
        """Name of the solver kernel the last solve launched, as the library reports it (cmpc_last_kernel_name)."""
        return self._lib.cmpc_last_kernel_name(self._h).decode()

    def last_kernel_ms(self):
        """Duration of the last solve's kernel (HIP events on the launch stream); synchronises."""
        ms = ctypes.c_float()
        rc = self._lib.cmpc_last_kernel_ms(self._h, ctypes.byref(ms))
        if rc != 0:
            raise RuntimeError(self._lib.cmpc_last_error(self._h).decode())
        return ms.value


class DeviceRecordBuilder:
    """Device-side front half of ``centroidal_mpc.solve`` (code/centroidal_mpc_vertices.py:482-600): the
    per-tick tables of a ``workloads.Scene`` are uploaded once, then records for a whole batch are
    gathered on the GPU (``cmpc_build_records``) - no host loop, no host-to-device copy per tick."""

    def __init__(self, scene, device=None):
        import numpy as np
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceRecordBuilder needs a ROCm GPU: there is no CPU fallback")
        self.device = _device_of(device)
        self._lib = capi.load()
        T = scene.T
        arrs = [np.ascontiguousarray(a[:T], dtype=np.float64) for a in
                (scene.com_tab, scene.pose_l, scene.pose_r, scene.gl_tab, scene.gr_tab, scene.cur_l, scene.cur_r)]
        h = ctypes.c_void_p()
        rc = self._lib.cmpc_tables_create(self.device.index, T, *[a.ctypes.data_as(ctypes.c_void_p) for a in arrs],
                                          ctypes.byref(h))
        if rc != 0:
            raise RuntimeError("cmpc_tables_create failed: " + self._lib.cmpc_last_error(None).decode())
        self._h, self.T = h, T
        self.n_steps = int(scene.plan_pos.shape[0])
        sl, sr = (np.ascontiguousarray(a[:T], dtype=np.int32) for a in (scene.slot_l, scene.slot_r))
        rc = self._lib.cmpc_tables_set_plan_slots(self._h, self.n_steps, sl.ctypes.data_as(ctypes.c_void_p),
                                                  sr.ctypes.data_as(ctypes.c_void_p))
        if rc != 0:
            raise RuntimeError("cmpc_tables_set_plan_slots failed: " + self._lib.cmpc_last_error(None).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cmpc_tables_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build(self, spec, t, state, rate=1, out=None, plan_pos=None):
        """t (B,) int32 and state (B, 16) fp64 on the GPU -> records (B, nrec) on the GPU.  plan_pos
        (B, n_steps, 3): per-instance contact plans (the x0 foot positions of :493-509 come from them)."""
        if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
            raise ValueError("t must be a contiguous int32 CUDA tensor")
        if t.device != self.device or state.device != self.device:
            raise ValueError(f"t / state must live on {self.device}")
        B = t.shape[0]
        if not (state.is_cuda and state.dtype == torch.float64 and state.is_contiguous()
                and tuple(state.shape) == (B, 16)):
            raise ValueError("state must be a contiguous fp64 CUDA tensor of shape (B, 16)")
        if out is None:
            out = torch.empty((B, spec.nrec), dtype=torch.float64, device=t.device)
        if plan_pos is not None and not (plan_pos.is_cuda and plan_pos.dtype == torch.float64 and plan_pos.is_contiguous()
                                         and tuple(plan_pos.shape) == (B, self.n_steps, 3) and plan_pos.device == self.device):
            raise ValueError(f"plan_pos must be a contiguous fp64 tensor of shape (B, {self.n_steps}, 3) on {self.device}")
        stream = torch.cuda.current_stream(t.device).cuda_stream
        rc = self._lib.cmpc_build_records_planned(self._h, spec.N, rate, B, t.data_ptr(), state.data_ptr(),
                                                  plan_pos.data_ptr() if plan_pos is not None else None,
                                                  out.data_ptr(), ctypes.c_void_p(stream))
        if rc != 0:
            raise RuntimeError("cmpc_build_records failed: " + self._lib.cmpc_last_error(None).decode())
        return out
