"""Developer script (GPU box): what the deal of a batch over the ranks by predicted cost (cmpc_amd.dist.shard_order, SURVEY
8e) buys without an 8-GPU box.  One B = 65536 launch per seed gives every instance's iteration count; each of the G
shards is then replayed through the launch's own ticket queue (queue order by predicted cost, `slots` resident
workgroups) and the step time of the G-GPU job is the WORST shard's makespan.  Contiguous shards against the deal;
efficiency = balanced bound of the whole batch / worst shard's makespan.
A batch drawn i.i.d. gains nothing from the deal (its contiguous shards are already alike: the worst shard is decided by
stragglers nobody predicted); a batch that ARRIVES ORDERED -- scenarios generated hardest first, robots grouped by task --
does, which `--sorted` shows by ordering the records by predicted cost before they are cut into shards.
usage: python tools/deal_replay.py [--sorted] [workload] [G] [seed ...]"""
import heapq, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cmpc_amd  # noqa: F401
from cmpc_amd import workloads as wl, queue_order as qo, dist as cdist
from cmpc_amd.solver import BatchedCentroidalMPC

presorted = "--sorted" in sys.argv
if presorted:
    sys.argv.remove("--sorted")
name = sys.argv[1] if len(sys.argv) > 1 else "randomized"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
seeds = [int(v) for v in sys.argv[3:]] or [None, 777, 31337, 11, 12]
per_gpu = {"randomized": 8192, "payload": 8192, "perturbed": 8192, "long_horizon": 2048}[name]
B = per_gpu * G
slots = 256 * (7 if name != "long_horizon" else 2)


def makespan(it, pred):
    order = np.argsort(-qo.bucket_of(pred), kind="stable")
    h = [0] * slots
    heapq.heapify(h)
    end = 0
    for i in order:
        t = heapq.heappop(h) + int(it[i])
        end = max(end, t)
        heapq.heappush(h, t)
    return end


rows = []
for seed in seeds:
    spec, rec = wl.make_workload(name, B=B, seed=seed)
    if presorted:                                   # the batch arrives hardest first
        rec = np.ascontiguousarray(rec[np.argsort(-qo.predicted_iterations(rec, spec), kind="stable")])
    if spec.N > 20:
        spec.max_iter = 150
    s = BatchedCentroidalMPC(spec, device="cuda:0")
    _, st, it, _ = s.solve(torch.from_numpy(rec).cuda())
    torch.cuda.synchronize()
    it = it.cpu().numpy()
    pred = qo.predicted_iterations(rec, spec)
    balanced = it.sum() / (G * slots)
    cont = [makespan(it[lo:hi], pred[lo:hi]) for lo, hi in (cdist.shard_bounds(B, G, r) for r in range(G))]
    order = cdist.shard_order(rec, spec, G)
    deal = [makespan(it[cdist.dealt_rows(order, G, r)], pred[cdist.dealt_rows(order, G, r)]) for r in range(G)]
    rows.append((seed, balanced, cont, deal))
    print(f"seed {seed if seed is not None else wl.CONFIGS[name][0]}: balanced {balanced:.1f}  contiguous shards: worst {max(cont)} (min {min(cont)}) -> efficiency "
          f"{balanced / max(cont):.3f}   dealt: worst {max(deal)} (min {min(deal)}) -> efficiency {balanced / max(deal):.3f}   "
          f"mean iterations {it.mean():.2f}, longest {it.max()}", flush=True)
ec = np.mean([r[1] / max(r[2]) for r in rows]); ed = np.mean([r[1] / max(r[3]) for r in rows])
print(f"{name}{' (batch ordered by predicted cost)' if presorted else ''}, {G} GPUs x {per_gpu}, {slots} slots per GPU, {len(rows)} seeds: expected weak-scaling efficiency against the balanced bound "
      f"{ec:.3f} with contiguous shards, {ed:.3f} with the deal; against ONE GPU's own makespan on a {per_gpu}-instance shard "
      f"(the N = 1 bench): step time ratio worst / mean shard = {np.mean([max(r[2]) / np.mean(r[2]) for r in rows]):.3f} contiguous, "
      f"{np.mean([max(r[3]) / np.mean(r[3]) for r in rows]):.3f} dealt")
