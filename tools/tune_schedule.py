"""Developer script: effect of the barrier-schedule constants on the iteration count (C oracle compiled with -D
overrides into /tmp; the product constants live in oracle/cmpc_oracle.c and csrc/cmpc_kernel.hpp).
usage: python tools/tune_schedule.py "MU_INIT=100 MU_FACTOR=0.1" "MU_INIT=10 KAPPA_EPS=30" ...   [env B=1024 WL=randomized]"""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cmpc_amd
from cmpc_amd import workloads as wl
from oracle import oracle_lib as ol

B = int(os.environ.get("B", 1024)); WL = os.environ.get("WL", "randomized"); N = int(os.environ.get("N", 20))
spec, rec = wl.make_workload(WL, B=B, N=N)
for variant in sys.argv[1:] or [""]:
    so = f"/tmp/oracle_{abs(hash(variant))}.so"
    defs = ["-D" + d for d in variant.split()]
    subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-fPIC", "-shared", "-o", so,
                           os.path.join(ROOT, "oracle", "cmpc_oracle.c"), "-lm"] + defs)
    lib = ctypes.CDLL(so)
    cs = ol.default_spec(N=spec.N, nv=spec.nv, tol=spec.tol, max_iter=spec.max_iter, k1=spec.k1, k2=spec.k2, prox=spec.prox, acc_tol=spec.acc_tol)
    out = np.zeros((B, ol.nsol(cs))); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kkt = np.zeros(B)
    t0 = time.time()
    lib.cmpc_oracle_solve_batch(ctypes.byref(cs), B, ol._p(rec), None, ol._p(out), ol._p(st), ol._p(it), ol._p(kkt), 0)
    dt = time.time() - t0
    print(f"{variant or '(default)':50s} iters mean {it.mean():6.2f} p90 {np.percentile(it, 90):4.0f} max {it.max():3d}  "
          f"status 0/1/2/3: {[(st == c).mean().round(4) for c in range(4)]}  {dt:.1f}s", flush=True)
