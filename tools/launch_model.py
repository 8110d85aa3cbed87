"""Developer script: what a change of the ALGORITHM (oracle compiled with -D overrides) does to a launch, without a GPU.
Solves a workload with the C oracle, replays the kernel's ticket queue on the measured iteration counts in the shipped
queue order (decreasing predicted count, csrc/cmpc_order_fit.h) and prints mean / tail iterations, outcome fractions,
the makespan against the balanced bound, and two figures proportional to what bench.py reports: instances per unit of
makespan (all instances/s) and the converged share of it (`value`).
usage: python tools/launch_model.py "COLD_ROLLOUT=1" "COLD_ROLLOUT=1 NOPROG_ANY_LEVEL=1" ...  [env B=8192 WL=randomized N=20 SEED=]"""
import ctypes, heapq, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cmpc_amd  # noqa: F401
from cmpc_amd import workloads as wl
from cmpc_amd import queue_order as qo
from oracle import oracle_lib as ol

B = int(os.environ.get("B", 8192)); WL = os.environ.get("WL", "randomized"); N = os.environ.get("N")
SEED = os.environ.get("SEED")
spec, rec = wl.make_workload(WL, B=B, N=int(N) if N else None, seed=int(SEED) if SEED else None)
if spec.N > 20:
    spec.max_iter = 150
slots = min(B, 256 * int(os.environ.get("SLOTS_PER_CU", 7 if spec.nv == 4 else 2)))
pred = qo.predicted_iterations(rec, spec)
order = np.argsort(-qo.bucket_of(pred), kind="stable")


def makespan(it, order):
    h = [0] * slots
    heapq.heapify(h)
    end = 0
    for i in order:
        t = heapq.heappop(h) + it[i]
        end = max(end, t)
        heapq.heappush(h, t)
    return end


base = None
for variant in sys.argv[1:] or [""]:
    so = f"/tmp/oracle_lm_{abs(hash(variant))}.so"
    subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-fPIC", "-shared", "-o", so,
                           os.path.join(ROOT, "oracle", "cmpc_oracle.c"), "-lm"] + ["-D" + d for d in variant.split()])
    lib = ctypes.CDLL(so)
    cs = ol.default_spec(N=spec.N, nv=spec.nv, tol=spec.tol, max_iter=spec.max_iter, k1=spec.k1, k2=spec.k2, prox=spec.prox,
                         acc_tol=spec.acc_tol)
    out = np.zeros((B, ol.nsol(cs))); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kkt = np.zeros(B)
    nreg = np.zeros(B, np.int32); waste = np.zeros(B)
    lib.cmpc_oracle_solve_batch_stats(ctypes.byref(cs), B, ol._p(rec), None, ol._p(out), ol._p(st), ol._p(it), ol._p(kkt), ol._p(nreg),
                                      ol._p(waste), 0)
    # cost of a solve in units of 1/100 iteration: a factorisation retry repeats the matrix sweep (3/4 of an iteration) down to
    # the failing stage; `waste` = the sweeps' worth of stages factorised in vain
    itl = np.maximum(it.astype(np.int64), 1) * 100 + np.round(75 * waste).astype(np.int64)
    mk, bal = makespan(itl, order), itl.sum() / slots
    conv = (st == 0).mean()
    rate, val = B / mk, B / mk * conv
    base = base or (rate, val)
    print(f"{variant or '(default)':44s} retries/solve {nreg.mean():5.2f} (= {waste.mean():4.2f} sweeps) its mean {it.mean():5.2f} q90 {np.quantile(it, .9):3.0f} q99 {np.quantile(it, .99):3.0f} max {it.max():3d} | "
          f"conv {conv:.4f} acc {(st == 3).mean():.4f} s2 {(st == 2).mean():.4f} cap {(st == 1).mean():.4f} | makespan {mk} = "
          f"{mk / bal:.3f} x balanced | all {rate / base[0]:.3f} value {val / base[1]:.3f} (relative to the first variant)"
          f" | corr(pred, its) {np.corrcoef(pred, it)[0, 1]:.2f}"
          + (f" | acceptable: kkt q50 {np.quantile(kkt[st == 3], .5):.1e} q90 {np.quantile(kkt[st == 3], .9):.1e} max {kkt[st == 3].max():.1e}"
             if (st == 3).any() else ""), flush=True)
