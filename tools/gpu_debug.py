"""Developer script: GPU vs oracle vs host-emulation on selected instances."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmpc_amd
from cmpc_amd import workloads as wl
from cmpc_amd.solver import BatchedCentroidalMPC
from oracle import oracle_lib as ol
import build as _b

B, N = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 20
spec, rec = wl.make_workload("perturbed", B=B, N=N)
solver = BatchedCentroidalMPC(spec, device="cuda:0")
out, status, iters, kkt = solver.solve(torch.from_numpy(rec).to("cuda:0"))
torch.cuda.synchronize()
got = out.cpu().numpy(); st = status.cpu().numpy(); it = iters.cpu().numpy()
cs = ol.default_spec(N=spec.N, nv=spec.nv, tol=spec.tol, max_iter=spec.max_iter, k1=spec.k1, k2=spec.k2, prox=spec.prox)
ref, st_ref, it_ref, kkt_ref = ol.solve_batch(cs, rec)
both = (st == 0) & (st_ref == 0)
rel = np.where(both, np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1), -1)
sel = [i for i in np.argsort(rel)[::-1] if it[i] == it_ref[i]][:2]
print("selected", sel, rel[sel], it[sel], it_ref[sel])
emu = ctypes.CDLL(_b.build_emu())
for i in sel:
    r1 = rec[i:i + 1].copy()
    eo = np.zeros((1, ol.nsol(cs))); es = np.zeros(1, np.int32); ei = np.zeros(1, np.int32); ek = np.zeros(1)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    t0 = time.time()
    emu.cmpc_emu_solve_batch(ctypes.byref(cs), 1, p(r1), None, p(eo), p(es), p(ei), p(ek))
    print(f"inst {i}: emu {ei[0]} its ({time.time() - t0:.0f}s) gpu {it[i]} oracle {it_ref[i]}")
    print("   |emu-oracle| %.3e  |gpu-oracle| %.3e  |gpu-emu| %.3e" % (np.abs(eo[0] - ref[i]).max(), np.abs(got[i] - ref[i]).max(), np.abs(got[i] - eo[0]).max()))
    for name, w in (("gpu", got[i]), ("oracle", ref[i])):
        c, dfc, g, act = ol.evaluate(cs, rec[i], w)
        print("   %-6s cost %.10e max|defect| %.3e max ineq %.3e" % (name, c, np.abs(dfc).max(), g[act != 0].max()))
