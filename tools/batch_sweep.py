"""Developer script: solves/s of one launch as a function of the batch size (one GPU, cold start)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cmpc_amd
from cmpc_amd import workloads as wl
from cmpc_amd.solver import BatchedCentroidalMPC

name = sys.argv[1] if len(sys.argv) > 1 else "randomized"
for B in (1, 16, 256, 4096, 8192, 65536):
    spec, rec = wl.make_workload(name, B=B, N=20)
    s = BatchedCentroidalMPC(spec, device="cuda:0")
    d = torch.from_numpy(rec).to("cuda:0")
    s.solve(d); torch.cuda.synchronize()
    out, st, it, kkt = s.solve(d)
    ms = s.last_kernel_ms()
    print(f"{name} B={B:6d}: kernel {ms:9.2f} ms  {B / ms * 1e3:9.0f} solves/s  {ms / B:8.3f} ms/solve  "
          f"converged {float((st == 0).double().mean()):.3f}  mean iters {float(it.double().mean()):.1f}")
    del s
