import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cmpc_amd
from cmpc_amd import workloads as wl
from cmpc_amd.solver import BatchedCentroidalMPC
from oracle import oracle_lib as ol
import nlp_batch
from conftest import oracle_spec
spec, rec = wl.make_workload("randomized", B=8192, N=20)
s = BatchedCentroidalMPC(spec, device="cuda:0")
out, st, it, kkt = s.solve(torch.from_numpy(rec).to("cuda:0")); torch.cuda.synchronize()
got, st, it, kkt = out.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), kkt.cpu().numpy()
conv = st == 0
r = nlp_batch.residuals(spec, rec[conv], got[conv])
idx = np.where(conv)[0]
for key in r:
    print(key, "max %.3e" % r[key].max(), "n>1e-5:", int((r[key] > 1e-5).sum()))
bad = idx[np.argsort(r["swing_force"])[-5:]]
print("worst swing-force instances", bad, "iters", it[bad], "kkt", kkt[bad], "status", st[bad])
ref, st_ref, it_ref, kkt_ref = ol.solve_batch(oracle_spec(ol, spec), rec[bad])
print("oracle status", st_ref, "iters", it_ref, "kkt", kkt_ref)
rr = nlp_batch.residuals(spec, rec[bad], ref)
print("oracle swing force", rr["swing_force"], "gpu", r["swing_force"][np.argsort(r["swing_force"])[-5:]])
print("rel diff", np.abs(got[bad] - ref).max(axis=1) / np.abs(ref).max(axis=1))
b = bad[-1]
stt = rec[b, 24:].reshape(20, 19)
print("gamma_l", stt[:, 17], "gamma_r", stt[:, 18], rec[b, 22:24], "mu", rec[b, 21], "mass", rec[b, 20])
U = got[b, 420:].reshape(20, 32)
print("U stage with max swing force:"); k = np.argmax(np.abs(U[:, :24] * np.repeat(np.stack([1 - stt[:, 17]] * 4 + [1 - stt[:, 18]] * 4, 1), 3, axis=1)).max(1)); print(k, U[k, :24].reshape(8, 3))
