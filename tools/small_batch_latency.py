"""Developer script (GPU box): cold-solve latency of small batches (HIP events on the launch stream, best and median of n
solves after two warm-ups).  usage: [KERNEL=auto|single|pair] python tools/small_batch_latency.py [workload] [B ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cmpc_amd  # noqa: F401
from cmpc_amd import workloads as wl
from cmpc_amd.solver import BatchedCentroidalMPC

name = sys.argv[1] if len(sys.argv) > 1 else "randomized"
sizes = [int(v) for v in sys.argv[2:]] or [1, 16, 256]
for B in sizes:
    spec, rec = wl.make_workload(name, B=B)
    spec.kernel = {"auto": 0, "single": 1, "pair": 2}[os.environ.get("KERNEL", "auto")]     # cmpc_spec.kernel
    s = BatchedCentroidalMPC(spec, device="cuda:0")
    d = torch.from_numpy(rec).to("cuda:0")
    ms = []
    for i in range(12):
        _, st, it, _ = s.solve(d)
        torch.cuda.synchronize()
        if i >= 2:
            ms.append(s.last_kernel_ms())
    it = it.cpu().numpy()
    print(f"{name} B={B}: kernel best {min(ms):.2f} ms, median {np.median(ms):.2f} ms; iterations mean {it.mean():.1f} max {it.max()}"
          f"  ({s.last_kernel_name()})", flush=True)
