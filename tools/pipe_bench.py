"""Developer script: throughput of K back-to-back batches, one stream vs two alternating handles/streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cmpc_amd
from cmpc_amd import workloads as wl
from cmpc_amd.solver import BatchedCentroidalMPC

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
spec, rec = wl.make_workload("randomized", B=B, N=20)
d_rec = torch.from_numpy(rec).to("cuda:0")
for nstream in (1, 2, 3):
    solvers = [BatchedCentroidalMPC(spec, device="cuda:0") for _ in range(nstream)]
    streams = [torch.cuda.Stream() for _ in range(nstream)]
    outs = [torch.empty((B, spec.nsol), dtype=torch.float64, device="cuda:0") for _ in range(nstream)]
    for i in range(nstream):
        with torch.cuda.stream(streams[i]):
            solvers[i].solve(d_rec, out=outs[i])
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(K):
        j = i % nstream
        with torch.cuda.stream(streams[j]):
            solvers[j].solve(d_rec, out=outs[j])
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"streams {nstream}: {K} batches of {B} in {dt*1e3:.1f} ms -> {K*B/dt:.0f} solves/s ({dt/K*1e3:.1f} ms per batch)")
    del solvers
