"""Developer script (GPU box): the batched whole-body QP (wbc_qp_kernel) at B = 65536 for rocprofv3 --
three launches of the workload bench.py's `wbc_qp` leg times.  usage: rocprofv3 --kernel-trace --stats -- python3 tools/wbc_profile.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmpc_amd  # noqa: F401
from cmpc_amd import wbc, workloads as wl
Bq, uniq = 65536, 1024
mats = [torch.from_numpy(np.ascontiguousarray(np.tile(a, (Bq // uniq,) + (1,) * (a.ndim - 1)))).cuda()
        for a in wl.wbc_synthetic(uniq, seed=20250715)]
qp = wbc.BatchedInverseDynamicsQP(foot_size=0.1, mu=0.5, device="cuda:0")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    _, _, _, st, it = qp.solve(*mats)
torch.cuda.synchronize()
print("converged", float((st == 0).double().mean()), "mean iterations", float(it.double().mean()))
