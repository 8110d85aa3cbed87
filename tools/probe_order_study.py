"""Developer study (CPU, C oracle): what would a PROBE buy the queue order of a large launch?  Every instance is run for
K iterations first (the oracle with max_iter = K: its verdict and KKT error at that point), the total iteration count is
then predicted from the record's features AND what the probe shows, and the launch is replayed as one queue with 1792
slots: probe pieces (K iterations each) first, then the remainders longest-predicted first.  Fit on one seed, replay on
others.  usage: python tools/probe_order_study.py [B] [K ...]"""
import dataclasses, heapq, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import importlib
sys.path.insert(0, os.path.join(ROOT, "tests"))
cmpc_amd = importlib.import_module("cmpc_amd") if "cmpc_amd" in sys.modules else __import__("cmpc_amd")
from cmpc_amd import workloads as wl, queue_order as qo
from oracle import oracle_lib as ol
from conftest import oracle_spec

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
Ks = [int(x) for x in sys.argv[2:]] or [4, 6, 8, 10]
SLOTS = 1792
FIT_SEED, TEST_SEEDS = 424242, (20250713, 777, 31337)


def run(seed, K):
    spec, rec = wl.make_workload("randomized", B=B, seed=seed)
    cs = oracle_spec(ol, spec)
    _, st, it, kkt = ol.solve_batch(cs, rec)
    cK = oracle_spec(ol, dataclasses.replace(spec, max_iter=K))
    _, stK, itK, kktK = ol.solve_batch(cK, rec)
    return spec, rec, it.astype(float), st, kktK, stK, itK


def design(spec, rec, kktK, stK, itK, K):
    F = qo.features(rec, spec)
    lk = np.log10(np.clip(kktK, 1e-12, 1e6))
    done = (itK < K + 1) & (stK != 1)             # finished inside the probe
    return np.column_stack([F, lk, lk ** 2, lk * F[:, 1], lk * F[:, 4], done.astype(float)])


def makespan_two_phase(it, pred_total, K):
    """one queue: every instance's first min(K, it) iterations in input order, then the remainders by predicted remainder"""
    h = [0.0] * SLOTS
    heapq.heapify(h)
    end = 0.0
    for i in range(len(it)):
        t = heapq.heappop(h) + min(K, it[i]); end = max(end, t); heapq.heappush(h, t)
    rem = np.maximum(it - K, 0)
    order = np.argsort(-(pred_total - K), kind="stable")
    for i in order:
        if rem[i] > 0:
            t = heapq.heappop(h) + rem[i]; end = max(end, t); heapq.heappush(h, t)
    return end


def makespan(it, order):
    h = [0.0] * SLOTS
    heapq.heapify(h)
    end = 0.0
    for i in order:
        t = heapq.heappop(h) + it[i]; end = max(end, t); heapq.heappush(h, t)
    return end


for K in Ks:
    spec, rec, it, st, kktK, stK, itK = run(FIT_SEED, K)
    X = design(spec, rec, kktK, stK, itK, K)
    coef, *_ = np.linalg.lstsq(X, it, rcond=None)
    print(f"K = {K}: fit on seed {FIT_SEED}: correlation record-only {np.corrcoef(qo.predicted_iterations(rec, spec), it)[0, 1]:.3f}, "
          f"with the probe {np.corrcoef(X @ coef, it)[0, 1]:.3f}")
    for seed in TEST_SEEDS:
        spec, rec, it, st, kktK, stK, itK = run(seed, K)
        pred0 = qo.predicted_iterations(rec, spec)
        pred1 = design(spec, rec, kktK, stK, itK, K) @ coef
        bal = it.sum() / SLOTS
        m0 = makespan(it, np.argsort(-qo.bucket_of(pred0), kind="stable"))
        m1 = makespan_two_phase(it, pred1, K)
        mp = makespan(it, np.argsort(-it))
        mpp = makespan_two_phase(it, it, K)
        print(f"   seed {seed}: corr {np.corrcoef(pred0, it)[0, 1]:.3f} -> {np.corrcoef(pred1, it)[0, 1]:.3f};  makespan / balanced: shipped {m0 / bal:.3f}, "
              f"probe + re-sort {m1 / bal:.3f}, perfect knowledge {mp / bal:.3f} (after a probe: {mpp / bal:.3f})")
