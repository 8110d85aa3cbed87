"""Developer script: GPU vs C oracle on EVERY instance of a batch (default: the bench batch, 8192 randomized, N = 20).
usage (GPU box): python tools/full_parity.py [workload] [B]   -- the workload's own horizon and vertex count"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmpc_amd
from cmpc_amd import workloads as wl
from cmpc_amd.solver import BatchedCentroidalMPC
from oracle import oracle_lib as ol

name = sys.argv[1] if len(sys.argv) > 1 else "randomized"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
spec, rec = wl.make_workload(name, B=B)
if spec.N > 20:
    spec.max_iter = 150                                     # as bench.py and the parity tests run the long horizon
out, st, it, kkt = BatchedCentroidalMPC(spec, device="cuda:0").solve(torch.from_numpy(rec).to("cuda:0"))
torch.cuda.synchronize()
got, st, it = out.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy()
cs = ol.default_spec(N=spec.N, nv=spec.nv, tol=spec.tol, max_iter=spec.max_iter, k1=spec.k1, k2=spec.k2, prox=spec.prox,
                     acc_tol=spec.acc_tol)
t0 = time.time(); ref, st_r, it_r, _ = ol.solve_batch(cs, rec); dt = time.time() - t0
print('status counts gpu', np.bincount(st, minlength=4).tolist(), 'oracle', np.bincount(st_r, minlength=4).tolist(),
      'usable verdict equal', int((np.isin(st, (0, 3)) == np.isin(st_r, (0, 3))).sum()), 'of', B)
both = np.isin(st, (0, 3)) & np.isin(st_r, (0, 3))
err = np.abs(got[both] - ref[both]).max(1) / np.abs(ref[both]).max(1)
print(f"{name} B={B}: oracle {dt:.1f} s; status equal {int((st == st_r).sum())}/{B}; usable on both {int(both.sum())}")
print("rel-inf error: median %.2e  p90 %.2e  p99 %.2e  max %.2e;  within 1e-4: %.4f  within 1e-6: %.4f"
      % (np.median(err), np.quantile(err, .9), np.quantile(err, .99), err.max(), (err < 1e-4).mean(), (err < 1e-6).mean()))
print("iterations equal: %.3f, |diff| <= 1: %.3f" % ((it == it_r).mean(), (np.abs(it - it_r) <= 1).mean()))
# the pairs further apart than the north-star tolerance: same optimum (objective, dynamics defect)?
idx = np.where(both)[0][err > 1e-4]
df, dd = [], []
for i in idx:
    f_g, def_g, _, _ = ol.evaluate(cs, rec[i], got[i])
    f_r, def_r, _, _ = ol.evaluate(cs, rec[i], ref[i])
    df.append(abs(f_g - f_r) / max(1.0, abs(f_r))); dd.append(max(np.abs(def_g).max(), np.abs(def_r).max()))
if len(idx):
    df, dd = np.array(df), np.array(dd)
    print("%d pairs beyond 1e-4: relative objective difference median %.1e p90 %.1e max %.1e; max dynamics defect %.1e"
          % (len(idx), np.median(df), np.quantile(df, .9), df.max(), dd.max()))
