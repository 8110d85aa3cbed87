"""The DEVICE source of the solver (csrc/cmpc_kernel.hpp) executed on the CPU by the thread-per-lane
harness in tests/emu, against the C oracle.  This is a test of the kernel's logic and barrier
placement, not a product path."""
import ctypes

import numpy as np
import pytest

import build as _b
from conftest import oracle_spec, rel_inf
from cmpc_amd import workloads as wl


@pytest.fixture(scope="module")
def emu():
    return ctypes.CDLL(_b.build_emu())


def _emu_solve(emu, cs, rec, warm=None):
    B = rec.shape[0]
    nsol = 20 * (cs.N + 1) + (6 * cs.nv + 8) * cs.N
    out, st, it, kk = np.zeros((B, nsol)), np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B)
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    rec = np.ascontiguousarray(rec)
    assert emu.cmpc_emu_solve_batch(ctypes.byref(cs), B, p(rec), p(warm), p(out), p(st), p(it), p(kk)) == 0
    return out, st, it, kk


def test_lds_budget(emu):
    assert emu.cmpc_emu_lds_bytes(4) <= 40 * 1024           # 4 workgroups per CU (160 KiB LDS)
    assert emu.cmpc_emu_lds_bytes(8) <= 80 * 1024           # 2 workgroups per CU


@pytest.mark.parametrize("name,N,B", [("perturbed", 3, 2), ("payload", 3, 1), ("randomized", 2, 1)])
def test_kernel_source_matches_oracle(emu, oracle, name, N, B):
    # the harness runs 64 OS threads per instance and every broadcast is a barrier: keep the cases small
    # (one of them with two instances, so that slab reuse between instances is exercised)
    spec, rec = wl.make_workload(name, B=B, N=N)
    cs = oracle_spec(oracle, spec)
    got, st, it, kk = _emu_solve(emu, cs, rec)
    ref, st_ref, it_ref, _ = oracle.solve_batch(cs, rec)
    assert (st == st_ref).all() and (st == 0).all()
    assert np.abs(it - it_ref).max() <= 1
    assert rel_inf(got, ref).max() < 1e-9


def test_kernel_source_eight_vertex_patch(emu, oracle):
    spec, rec = wl.make_workload("long_horizon", B=1, N=2)
    cs = oracle_spec(oracle, spec)
    got, st, it, kk = _emu_solve(emu, cs, rec)
    ref, st_ref, it_ref, _ = oracle.solve_batch(cs, rec)
    assert st[0] == 0 and st_ref[0] == 0
    assert rel_inf(got, ref).max() < 1e-9


def test_kernel_source_warm_start_and_garbage_memory(emu, oracle, monkeypatch):
    """Warm start path; LDS and scratch pre-filled with NaN (no read of uninitialised memory)."""
    monkeypatch.setenv("CMPC_EMU_FILL", "nan")
    spec, rec = wl.make_workload("perturbed", B=1, N=3, scale=0.5)
    cs = oracle_spec(oracle, spec)
    cold, st, _, _ = oracle.solve_batch(cs, rec)
    assert st[0] == 0
    got, st, it, kk = _emu_solve(emu, cs, rec, warm=cold)
    ref, st_ref, it_ref, _ = oracle.solve_batch(cs, rec, warm=cold)
    assert st[0] == 0 and st_ref[0] == 0
    assert rel_inf(got, ref).max() < 1e-9
