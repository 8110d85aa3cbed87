"""Developer script: compare GPU and oracle iterates after k iterations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmpc_amd
from cmpc_amd import workloads as wl
from cmpc_amd.solver import BatchedCentroidalMPC
from oracle import oracle_lib as ol

B, N = 64, 20
spec, rec = wl.make_workload("perturbed", B=B, N=N)
Nn, nu = spec.N, spec.nu
def groups(v):
    X = v[:20 * (Nn + 1)].reshape(Nn + 1, 20); U = v[20 * (Nn + 1):].reshape(Nn, nu)
    return {"com": X[:, 0:3], "vel": X[:, 3:6], "hw": X[:, 6:9], "th": X[:, 9:12], "feet": X[:, 12:20],
            "force": U[:, :nu - 8], "footvel": U[:, nu - 8:]}
for mi in (1, 2, 5):
    spec.max_iter = mi
    solver = BatchedCentroidalMPC(spec, device="cuda:0")
    out, status, iters, kkt = solver.solve(torch.from_numpy(rec).to("cuda:0"))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    cs = ol.default_spec(N=spec.N, nv=spec.nv, tol=spec.tol, max_iter=mi, k1=spec.k1, k2=spec.k2, prox=spec.prox)
    ref, st_ref, it_ref, kkt_ref = ol.solve_batch(cs, rec)
    d = np.abs(got - ref).max(axis=1)
    print(f"max_iter={mi}: max abs diff over batch: max {d.max():.3e} median {np.median(d):.3e}; kkt gpu/oracle inst0 {kkt.cpu().numpy()[0]:.6e} {kkt_ref[0]:.6e}")
    i = int(np.argmax(d))
    ga, gb = groups(got[i]), groups(ref[i])
    for k in ga:
        dd = np.abs(ga[k] - gb[k]); idx = np.unravel_index(dd.argmax(), dd.shape)
        print("   inst %d %-8s maxabs diff %.3e (scale %.3e) at stage %d comp %d" % (i, k, dd.max(), np.abs(gb[k]).max(), idx[0], idx[1]))
