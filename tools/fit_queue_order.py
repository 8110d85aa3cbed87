"""Fit of the queue-order predictor (csrc/cmpc_order_fit.h): least squares of the cost of a solve -- its interior-point
iterations plus 3/4 of the sweeps' worth of stages its failed factorisations went through, i.e. what the kernel spends on
it -- on eighteen numbers of the parameter record (cmpc_amd/queue_order.py: features).  Counts come from the C oracle (the
same algorithm as the kernel; they agree to a step or two), so the fit needs no GPU.

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
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--workload", default="randomized")
ap.add_argument("--held-out", default="randomized:8192,randomized:8192:777,payload:4096,long_horizon:2048,long_horizon:2048:777")
ap.add_argument("--write", action="store_true")
args = ap.parse_args()
assert args.seed not in [c[0] for c in wl.CONFIGS.values()], "fit on a seed that is not a BASELINE seed"


def solve(name, B, seed=None):
    """(spec, records, cost in 1/100 iterations, iterations, status) of a workload solved by the oracle."""
    spec, rec = wl.make_workload(name, B=B, seed=seed)
    if spec.N > 20:
        spec.max_iter = 150
    cs = ol.default_spec(N=spec.N, nv=spec.nv, tol=spec.tol, max_iter=spec.max_iter, k1=spec.k1, k2=spec.k2, prox=spec.prox,
                         acc_tol=spec.acc_tol)
    import ctypes
    out = np.zeros((B, ol.nsol(cs))); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kkt = np.zeros(B)
    nreg = np.zeros(B, np.int32); waste = np.zeros(B)
    ol.lib().cmpc_oracle_solve_batch_stats(ctypes.byref(cs), B, ol._p(rec), None, ol._p(out), ol._p(st), ol._p(it), ol._p(kkt),
                                           ol._p(nreg), ol._p(waste), 0)
    return spec, rec, it + 0.75 * waste, it.astype(np.int64), st


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
spec, rec, cost, it, st = solve(args.workload, args.batch, args.seed)
X = qo.features(rec, spec)
# A column that repeats another on the fit set (the N = 20 workload has at most one contact switch in its horizon: `n_switch`
# IS `switch`) makes the design matrix rank deficient, and least squares then splits the weight between the two by minimum
# norm -- an arbitrary -4 iterations per EXTRA switch on horizons that have two (round-4 advisor).  Such a column gets the
# coefficient 0 and the fit runs on the others; any other rank deficiency stops the script.
keep = np.ones(X.shape[1], bool)
for j in range(X.shape[1]):
    for i in range(j):
        if keep[i] and np.array_equal(X[:, i], X[:, j]):
            keep[j] = False
            print(f"  column `{qo.NAMES[j]}` repeats `{qo.NAMES[i]}` on the fit set: coefficient 0")
rank = np.linalg.matrix_rank(X[:, keep])
assert rank == keep.sum(), f"design matrix still rank deficient ({rank} of {keep.sum()} columns): refusing to fit"
coef = np.zeros(X.shape[1])
coef[keep], *_ = np.linalg.lstsq(X[:, keep], cost, rcond=None)
r2 = 1.0 - ((X @ coef - cost) ** 2).sum() / ((cost - cost.mean()) ** 2).sum()
old, old_origin = qo.coefficients()
print(f"fit on {args.workload}, seed {args.seed}, B = {args.batch}: mean iterations {it.mean():.2f}, mean cost {cost.mean():.2f}, R^2 {r2:.3f}  ({time.time() - t0:.0f} s)")
print("  coefficients", dict(zip(qo.NAMES, np.round(coef, 3))))
print("  header now  ", dict(zip(qo.NAMES, np.round(old, 3))))
origin = float(np.floor(np.quantile(X @ coef, 0.02)))
print(f"  bucket origin {origin} (2 % quantile of the predictions, rounded down); 64 buckets of half an iteration")
rows = []
for item in [f"{args.workload}:{args.batch}:{args.seed}"] + [x for x in args.held_out.split(",") if x]:
    parts = item.split(":")
    name, B = parts[0], int(parts[1])
    seed = int(parts[2]) if len(parts) > 2 else None
    sp, rc, cx, itx, _ = (spec, rec, cost, it, st) if seed == args.seed else solve(name, B, seed)
    slots = min(B, 256 * (7 if sp.nv == 4 else 2))
    c100 = np.round(100 * cx).astype(np.int64)
    bal = c100.sum() / slots
    F = qo.features(rc, sp)
    res = []
    for lab, o in (("input order", np.arange(B)),
                   ("header", np.argsort(-qo.bucket_of(F @ old, old_origin), kind="stable")),
                   ("this fit", np.argsort(-qo.bucket_of(F @ coef, origin), kind="stable")),
                   ("perfect", np.argsort(-c100))):
        res.append(f"{lab} {makespan(c100, o, slots) / bal:.3f}")
    print(f"  {'in-sample ' if seed == args.seed else 'HELD OUT  '}{name:12s} B {B:5d} seed {seed or wl.CONFIGS[name][0]}: makespan / balanced: " + ", ".join(res)
          + f"; corr {np.corrcoef(F @ coef, cx)[0, 1]:.2f}")
if args.write:
    text = open(qo.FIT_HEADER).read()
    import re
    text = re.sub(r"// fit:.*?\n(?=#ifndef)", "", text, flags=re.S)
    note = (f"// fit: {time.strftime('%Y-%m-%d')}, tools/fit_queue_order.py --seed {args.seed} --batch {args.batch}: least squares on {args.batch} instances of the\n"
            f"//      `{args.workload}` workload drawn with seed {args.seed} (NOT a BASELINE seed: bench.py's batches are held out), C oracle, cold start\n"
            f"//      rolled out under the initial inputs, MU_INIT 100, MU_FACTOR 0.1, tol {spec.tol:g}; mean {it.mean():.2f} iterations, mean cost {cost.mean():.2f}, R^2 {r2:.2f}\n")
    text = text.replace("#ifndef CMPC_ORDER_FIT_H", note + "#ifndef CMPC_ORDER_FIT_H", 1)
    text = re.sub(r"#define CMPC_ORDER_COEF \{[^}]*\}", "#define CMPC_ORDER_COEF {" + ", ".join(f"{c:.4f}" for c in coef) + "}", text)
    text = re.sub(r"#define CMPC_ORDER_BUCKET_ORIGIN [-+0-9.eE]+", f"#define CMPC_ORDER_BUCKET_ORIGIN {origin:.3f}", text)
    open(qo.FIT_HEADER, "w").write(text)
    print("wrote", qo.FIT_HEADER)
