"""Generates tests/golden/solver_kat_*.npz: parameter records and the oracle's tightly converged
solutions for a handful of instances of every synthetic config.  These pin the ORACLE and the HIP
solver against accidental changes; they are NOT reference outputs (the reference's CasADi/IPOPT
stack cannot run here: parity unpinned).   usage: python tests/golden/make_solver_kat.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import cmpc_amd  # noqa: E402,F401
from cmpc_amd import workloads as wl  # noqa: E402
from oracle import oracle_lib as ol  # noqa: E402

for name, B, N in (("perturbed", 6, 20), ("payload", 4, 20), ("randomized", 4, 20), ("perturbed", 4, 10)):
    spec, rec = wl.make_workload(name, B=B, N=N, scale=0.5)
    cs = ol.default_spec(N=spec.N, nv=spec.nv, tol=1e-10, max_iter=300, k1=spec.k1, k2=spec.k2)
    sol, st, it, kkt = ol.solve_batch(cs, rec)
    keep = (st == 0) & (it <= 30)        # well-conditioned instances only: hard ones are path dependent at 1e-6
    out = os.path.join(HERE, f"solver_kat_{name}_N{N}.npz")
    np.savez_compressed(out, records=rec[keep], solutions=sol[keep], k1=spec.k1, k2=spec.k2, N=N, nv=spec.nv)
    print(out, int(keep.sum()), "instances, iterations", it[keep])
