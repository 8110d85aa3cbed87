"""Parity statistics of the HIP solver against the C oracle for every test configuration (run on the GPU box):
status cross-table, rel-inf error quantiles for pairs that both met the tight tolerance and for all usable
pairs, and the explanation check (same objective / feasible) for every pair beyond 1e-4."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cmpc_amd
from cmpc_amd import workloads as wl
from cmpc_amd.solver import BatchedCentroidalMPC
from oracle import oracle_lib as ol
from conftest import oracle_spec, rel_inf

CASES = [("perturbed", 256, 20, 1), ("payload", 512, 20, 1), ("randomized", 512, 20, 1), ("perturbed", 128, 10, 1),
         ("perturbed", 64, 3, 1), ("perturbed", 32, 40, 1), ("long_horizon", 64, 10, 1), ("long_horizon", 512, 40, 1),
         ("perturbed", 128, 10, 10), ("perturbed", 64, 20, 10)]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if f"{c[0]}-{c[1]}-{c[2]}-{c[3]}" in sys.argv[1:]]
for name, B, N, rate in CASES:
    spec, rec = wl.make_workload(name, B=B, N=N, rate=rate)
    if N > 20:
        spec.max_iter = 150
    cs = oracle_spec(ol, spec)
    s = BatchedCentroidalMPC(spec, device="cuda:0")
    out, st, it, kkt = s.solve(torch.from_numpy(rec).to("cuda:0"))
    torch.cuda.synchronize()
    got, st, it = out.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy()
    ref, st_ref, it_ref, kkt_ref = ol.solve_batch(cs, rec, nthreads=16)
    print(f"== {name} B={B} N={N} rate={rate}: iters gpu {it.mean():.1f} oracle {it_ref.mean():.1f}")
    tab = np.zeros((4, 4), int)
    for a, b in zip(st, st_ref):
        tab[a, b] += 1
    print("   status gpu(rows) x oracle(cols):", tab.tolist())
    both = np.isin(st, (0, 3)) & np.isin(st_ref, (0, 3))
    tight = (st == 0) & (st_ref == 0)
    for label, m in (("tight", tight), ("usable", both)):
        e = rel_inf(got[m], ref[m])
        if len(e):
            print(f"   {label:6s} n={m.sum():4d} median {np.median(e):.1e} q90 {np.quantile(e, .9):.1e} q97 {np.quantile(e, .97):.1e} "
                  f"max {e.max():.1e}  beyond 1e-4: {(e >= 1e-4).sum()}")
    idx = np.where(both)[0][rel_inf(got[both], ref[both]) >= 1e-4]
    for i in idx:
        f_g, d_g, q_g, a_g = ol.evaluate(cs, rec[i], got[i])
        f_r, d_r, q_r, a_r = ol.evaluate(cs, rec[i], ref[i])
        print(f"      outlier {i}: st {st[i]}/{st_ref[i]} err {rel_inf(got[i], ref[i])[0]:.1e} dJ/J {(f_g - f_r) / max(1, abs(f_r)):+.1e} "
              f"defect {np.abs(d_g).max():.1e} ineq {q_g[a_g == 1].max():+.1e}  kkt {float(kkt[i]):.1e}/{kkt_ref[i]:.1e}")
