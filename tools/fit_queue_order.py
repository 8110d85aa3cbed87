"""Fit of the queue-order predictor (csrc/cmpc_order_fit.h): least squares of the interior-point iteration count on six
numbers of the parameter record (cmpc_amd/queue_order.py: features).  Iteration counts come from the C oracle (the same
algorithm as the kernel; counts agree to a step or two), so the fit needs no GPU.

The fit seed is NOT a BASELINE seed: bench.py's workloads (seeds 20250711 ... 20250714) are held out, and the
script reports the replayed makespan on them next to the in-sample one.

usage: python tools/fit_queue_order.py [--seed 424242] [--batch 4096] [--write]      (--write rewrites the header)"""
import argparse, heapq, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cmpc_amd  # noqa: F401
from cmpc_amd import workloads as wl, queue_order as qo
from oracle import oracle_lib as ol

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=424242)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--workload", default="randomized")
ap.add_argument("--held-out", default="randomized:8192,payload:4096,perturbed:4096")
ap.add_argument("--write", action="store_true")
args = ap.parse_args()
assert args.seed not in [c[0] for c in wl.CONFIGS.values()], "fit on a seed that is not a BASELINE seed"


def solve(name, B, seed=None):
    spec, rec = wl.make_workload(name, B=B, seed=seed)
    cs = ol.default_spec(N=spec.N, nv=spec.nv, tol=spec.tol, max_iter=spec.max_iter, k1=spec.k1, k2=spec.k2, prox=spec.prox,
                         acc_tol=spec.acc_tol)
    _, st, it, _ = ol.solve_batch(cs, rec)
    return spec, rec, it.astype(np.int64), st


def makespan(it, order, slots):
    h = [0] * slots
    heapq.heapify(h)
    end = 0
    for i in order:
        t = heapq.heappop(h) + it[i]
        end = max(end, t)
        heapq.heappush(h, t)
    return end


t0 = time.time()
spec, rec, it, st = solve(args.workload, args.batch, args.seed)
X = qo.features(rec, spec)
coef, *_ = np.linalg.lstsq(X, it.astype(float), rcond=None)
r2 = 1.0 - ((X @ coef - it) ** 2).sum() / ((it - it.mean()) ** 2).sum()
old = qo.coefficients()[0]
print(f"fit on {args.workload}, seed {args.seed}, B = {args.batch}: mean iterations {it.mean():.2f}, R^2 {r2:.3f}  ({time.time() - t0:.0f} s)")
print("  coefficients", dict(zip(qo.NAMES, np.round(coef, 3))))
print("  header now  ", dict(zip(qo.NAMES, np.round(old, 3))))
origin = float(np.floor(np.quantile(X @ coef, 0.02)))
print(f"  bucket origin {origin} (2 % quantile of the predictions, rounded down); 64 buckets of half an iteration")
for item in [f"{args.workload}:{args.batch}:{args.seed}"] + args.held_out.split(","):
    parts = item.split(":")
    name, B = parts[0], int(parts[1])
    seed = int(parts[2]) if len(parts) > 2 else None
    sp, rc, itx, _ = (spec, rec, it, st) if seed == args.seed else solve(name, B, seed)
    slots = min(B, 256 * (6 if sp.nv == 4 else 2))
    bal = itx.sum() / slots
    F = qo.features(rc, sp)
    res = []
    for lab, o in (("input order", np.arange(B)),
                   ("header", np.argsort(-qo.bucket_of(F @ old), kind="stable")),
                   ("this fit", np.argsort(-qo.bucket_of(F @ coef, origin), kind="stable")),
                   ("perfect", np.argsort(-itx))):
        res.append(f"{lab} {makespan(itx, o, slots) / bal:.3f}")
    print(f"  {'in-sample ' if seed == args.seed else 'HELD OUT  '}{name:12s} B {B:5d} seed {seed or wl.CONFIGS[name][0]}: makespan / balanced: " + ", ".join(res)
          + f"; corr {np.corrcoef(F @ coef, itx)[0, 1]:.2f}")
if args.write:
    lines = open(qo.FIT_HEADER).read().splitlines()
    out = []
    for ln in lines:
        if ln.startswith("// fit:") or ln.startswith("//      cold start"):
            continue
        if ln.startswith("#ifndef CMPC_ORDER_FIT_H"):
            out.append(f"// fit: {time.strftime('%Y-%m-%d')}, tools/fit_queue_order.py --seed {args.seed} --batch {args.batch}: least squares on {args.batch} instances of the")
            out.append(f"//      `{args.workload}` workload drawn with seed {args.seed} (NOT a BASELINE seed: bench.py's batches are held out), C oracle, cold start")
            out.append(f"//      rolled out under the initial inputs, MU_INIT 100, MU_FACTOR 0.1, tol {spec.tol:g}; mean {it.mean():.2f} iterations, R^2 {r2:.2f}")
        matched = False
        for n, c in list(zip(qo.NAMES, coef)) + [("BUCKET_ORIGIN", origin)]:
            if ln.startswith(f"#define CMPC_ORDER_{n} "):
                tail = ln[ln.index("//"):] if "//" in ln else ""
                out.append(f"#define CMPC_ORDER_{n} {c:.3f}" + ("   " + tail if tail else ""))
                matched = True
        if not matched:
            out.append(ln)
    open(qo.FIT_HEADER, "w").write("\n".join(out) + "\n")
    print("wrote", qo.FIT_HEADER)
