"""Developer script (GPU box): one workload solved by the HIP kernel and by the C oracle; prints the outcome cross-table,
the pairs whose status differs and the pairs further apart than 1e-4 with their iteration counts, KKT errors and
objective values.  usage: python tools/gpu_pairs.py [workload] [B] [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cmpc_amd  # noqa: F401
from cmpc_amd import workloads as wl
from cmpc_amd.solver import BatchedCentroidalMPC
from oracle import oracle_lib as ol

name = sys.argv[1] if len(sys.argv) > 1 else "randomized"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N = int(sys.argv[3]) if len(sys.argv) > 3 else None
spec, rec = wl.make_workload(name, B=B, N=N)
if spec.N > 20:
    spec.max_iter = 150
s = BatchedCentroidalMPC(spec, device="cuda:0")
out, st, it, kkt = s.solve(torch.from_numpy(rec).cuda())
torch.cuda.synchronize()
out, st, it, kkt = out.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), kkt.cpu().numpy()
cs = ol.default_spec(N=spec.N, nv=spec.nv, tol=spec.tol, max_iter=spec.max_iter, k1=spec.k1, k2=spec.k2, prox=spec.prox, acc_tol=spec.acc_tol)
ref, st_r, it_r, kkt_r = ol.solve_batch(cs, rec)
print(f"{name} B={B} N={spec.N}: gpu its mean {it.mean():.2f} max {it.max()}  oracle its mean {it_r.mean():.2f} max {it_r.max()}")
print("status gpu x oracle:")
for a in range(4):
    print("  ", a, [int(((st == a) & (st_r == b)).sum()) for b in range(4)])
err = np.abs(out - ref).max(axis=1) / np.abs(ref).max(axis=1)
idx = np.where((st != st_r) | ((err > 1e-4) & np.isin(st, (0, 3)) & np.isin(st_r, (0, 3))))[0]
for i in idx[:40]:
    fg = ol.evaluate(cs, rec[i], out[i])[0]; fr = ol.evaluate(cs, rec[i], ref[i])[0]
    print(f"  {i:5d} gpu st {st[i]} it {it[i]:3d} kkt {kkt[i]:.1e} | oracle st {st_r[i]} it {it_r[i]:3d} kkt {kkt_r[i]:.1e} | rel-inf {err[i]:.1e} | f gpu {fg:.9e} oracle {fr:.9e} rel gap {abs(fg - fr) / abs(fr):.1e}")
top = np.argsort(-it)[:10]
print("longest on the gpu:", [(int(i), int(it[i]), int(st[i]), int(it_r[i]), int(st_r[i])) for i in top], "(index, its, status, oracle its, oracle status)")
