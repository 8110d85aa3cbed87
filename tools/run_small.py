"""Developer script: one solve of a small batch (for rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cmpc_amd
from cmpc_amd import workloads as wl
from cmpc_amd.solver import BatchedCentroidalMPC
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spec, rec = wl.make_workload("perturbed", B=B, N=20)
spec.max_iter = 30
s = BatchedCentroidalMPC(spec, device="cuda:0")
out, st, it, kkt = s.solve(torch.from_numpy(rec).to("cuda:0")); torch.cuda.synchronize()
print("kernel ms", s.last_kernel_ms(), "iterations", int(it.sum().item()))
