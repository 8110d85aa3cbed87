"""TEST HARNESS ONLY -- child process of tests/test_sanitizers.py: runs the device source of the solver
(host emulation, tests/emu) and the C oracle, both built with AddressSanitizer + UBSan, on one small
instance per kernel variant.  LDS and the scratch slab are heap buffers of exactly the device sizes, so an
access past either allocation aborts the process."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cmpc_amd  # noqa: E402,F401
from cmpc_amd import workloads as wl  # noqa: E402
from oracle import oracle_lib as ol  # noqa: E402
from conftest import oracle_spec, rel_inf  # noqa: E402

emu = ctypes.CDLL(sys.argv[1])
ol._LIB = ctypes.CDLL(sys.argv[2])
p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
for name, N, B, warm_start, pair in (("long_horizon", 2, 1, False, False), ("long_horizon", 2, 1, True, False),
                                     ("perturbed", 2, 2, False, False), ("perturbed", 3, 2, False, True)):
    os.environ["CMPC_EMU_PAIR"] = "1" if pair else "0"      # the pipelined pair of waves: two LDS images + exchange words
    spec, rec = wl.make_workload(name, B=B, N=N)
    cs = oracle_spec(ol, spec)
    rec = np.ascontiguousarray(rec)
    ref, st_ref, it_ref, _ = ol.solve_batch(cs, rec)
    warm = np.ascontiguousarray(ref) if warm_start else None
    if warm_start:
        ref, st_ref, it_ref, _ = ol.solve_batch(cs, rec, warm=warm)
    nsol = 20 * (N + 1) + (6 * spec.nv + 8) * N
    out, st, it, kk = np.zeros((B, nsol)), np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B)
    assert emu.cmpc_emu_solve_batch(ctypes.byref(cs), B, p(rec), p(warm), p(out), p(st), p(it), p(kk)) == 0
    err = rel_inf(out, ref).max()
    print(name, "nv", spec.nv, "pair" if pair else "", "warm" if warm_start else "cold", "status", st.tolist(), "err", err)
    assert np.isin(st, (0, 3)).all() and err < 1e-6
print("sanitized run ok")
