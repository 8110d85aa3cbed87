"""Developer script: per-phase share of the solve kernel's cycles (diagnostic build with in-kernel s_memtime ticks,
build.build_hip_profile()).  usage (GPU box): python tools/phase_profile.py [workload] [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CMPC_LIB_PATH"] = os.environ.get("CMPC_PROF_LIB") or os.path.join(ROOT, "tools", "libcmpc_amd_prof.so")
import numpy as np, torch
import cmpc_amd
from cmpc_amd import workloads as wl, capi
from cmpc_amd.solver import BatchedCentroidalMPC
name = sys.argv[1] if len(sys.argv) > 1 else "randomized"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
spec, rec = wl.make_workload(name, B=B)
if spec.N > 20:
    spec.max_iter = 150
s = BatchedCentroidalMPC(spec, device="cuda:0")
d = torch.from_numpy(rec).to("cuda:0")
s.solve(d); torch.cuda.synchronize()
lib = capi.load()
buf = (ctypes.c_longlong * 28)()
lib.cmpc_profile_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert lib.cmpc_profile_read(s._h, buf) == 0
_, _, iters, _ = s.solve(d); torch.cuda.synchronize()
assert lib.cmpc_profile_read(s._h, buf) == 0
n_it = float(iters.sum().item())
v = np.array(list(buf), dtype=np.float64)
names = {24: "stage iterates: load + LDS commit", 11: "geometry", 12: "inequality rows", 25: "barrier weights", 26: "gradient / residual",
         0: "slab stores of the evaluation", 1: "Hessian rows: diagonal", 9: "Hessian rows: row roles and coefficients", 15: "Hessian rows: force columns", 23: "Hessian rows: velocity and state columns", 13: "P b", 10: "G'PG: T = P[B A]", 14: "G'PG: M += [B A]'T", 
         19: "Cholesky: trailing write-back of the previous block + block load", 20: "Cholesky: pivot chain + in-block updates", 21: "Cholesky: block store", 22: "MFMA trailing update",  8: "(factor tail)",
         2: "backward vectors: m", 3: "backward vectors: l", 5: "backward vectors: p", 4: "factor store",
         16: "forward sweep: loads", 17: "forward sweep: du, slack directions", 18: "forward sweep: dx", 6: "(forward tail + reductions)", 7: "step application"}
tot = v.sum()
print(f"cycles per instance-iteration (s_memtime ticks): {tot / n_it:.0f}   kernel {s.last_kernel_ms():.1f} ms")
for i in np.argsort(-v):
    if v[i] > 0:
        print(f"{100 * v[i] / tot:5.1f} %  {v[i] / n_it:8.0f} cyc/it  tick {i:2d}  {names.get(int(i), '')}")
