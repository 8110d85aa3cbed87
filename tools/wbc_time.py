"""Developer script (GPU box): wall time of the batched whole-body QP at B = 65536 (bench.py's `wbc_qp` workload), best and
median of 6 launches after a warm-up.  usage: [CMPC_LIB_PATH=...] python tools/wbc_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmpc_amd  # noqa: F401
from cmpc_amd import wbc, workloads as wl
Bq, uniq = 65536, 1024
mats = [torch.from_numpy(np.ascontiguousarray(np.tile(a, (Bq // uniq,) + (1,) * (a.ndim - 1)))).cuda()
        for a in wl.wbc_synthetic(uniq, seed=20250715)]
qp = wbc.BatchedInverseDynamicsQP(foot_size=0.1, mu=0.5, device="cuda:0")
ms = []
for i in range(7):
    torch.cuda.synchronize(); t = time.perf_counter()
    _, _, _, st, it = qp.solve(*mats)
    torch.cuda.synchronize()
    if i: ms.append((time.perf_counter() - t) * 1e3)
print(f"wbc_qp B={Bq}: best {min(ms):.1f} ms, median {np.median(ms):.1f} ms = {Bq / np.median(ms):.0f} k QPs/s; converged "
      f"{float((st == 0).double().mean()):.4f}, mean iterations {float(it.double().mean()):.2f}  ({os.environ.get('CMPC_LIB_PATH', 'shipped')})")
