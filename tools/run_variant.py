import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cmpc_amd
from cmpc_amd import workloads as wl, capi
capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "lib_%s.so" % os.environ.get("CMPC_VARIANT", "BASE"))
from cmpc_amd.solver import BatchedCentroidalMPC
spec, rec = wl.make_workload("perturbed", B=256, N=20)
spec.max_iter = 20
s = BatchedCentroidalMPC(spec, device="cuda:0")
out, st, it, kkt = s.solve(torch.from_numpy(rec).to("cuda:0")); torch.cuda.synchronize()
print("kernel ms", s.last_kernel_ms(), "iterations", int(it.sum().item()) + 256)
