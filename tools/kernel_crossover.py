"""Developer script (GPU box): serial throughput of the one-wave kernel and of the pipelined pair over batch sizes, to place
the library's switch (cmpc_create: pair_max_batch).  usage: python tools/kernel_crossover.py [workload] [B ...]"""
import dataclasses, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cmpc_amd  # noqa: F401
from cmpc_amd import workloads as wl
from cmpc_amd.solver import BatchedCentroidalMPC

name = sys.argv[1] if len(sys.argv) > 1 else "randomized"
sizes = [int(v) for v in sys.argv[2:]] or [1024, 2048, 3072, 4096, 5120, 6144, 7168, 8192]
for B in sizes:
    spec, rec = wl.make_workload(name, B=B)
    d = torch.from_numpy(rec).to("cuda:0")
    row = []
    for kern in (1, 2):
        s = BatchedCentroidalMPC(dataclasses.replace(spec, kernel=kern), device="cuda:0")
        ms = []
        for i in range(5):
            s.solve(d); torch.cuda.synchronize()
            if i >= 1:
                ms.append(s.last_kernel_ms())
        row.append(float(np.median(ms)))
    print(f"{name} B={B}: one wave {row[0]:.1f} ms ({B / row[0]:.1f} k/s)   pair {row[1]:.1f} ms ({B / row[1]:.1f} k/s)   -> {'pair' if row[1] < row[0] else 'one wave'}", flush=True)
