"""Developer script: per-iteration cycles of kernel variants with one phase compiled out (timing only)."""
import os, sys, ctypes, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmpc_amd
from cmpc_amd import workloads as wl, capi
import cmpc_amd.solver as sv
spec, rec = wl.make_workload("perturbed", B=256, N=20)
spec.max_iter = 30
d_rec = torch.from_numpy(rec).to("cuda:0")
here = os.path.dirname(os.path.abspath(__file__))
for path in sorted(glob.glob(os.path.join(here, "variants", "lib_*.so"))):
    capi._lib = None; capi.LIB_PATH = path
    s = sv.BatchedCentroidalMPC(spec, device="cuda:0")
    out, st, it, kkt = s.solve(d_rec); torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 16)()
    s._lib.cmpc_profile_read(s._h, buf)
    tot = float(sum(buf)); its = float(it.sum().item())
    print("%-16s kernel %.1f ms  iterations %d  cycles/instance-iteration %.0f" % (os.path.basename(path), s.last_kernel_ms(), its, tot / max(its, 1)))
