"""Developer script: how much of a launch is drain?  Solves a workload once, then replays the ticket queue on the host
from the measured iteration counts (env SEED: another draw of the workload): makespan (in instance-iterations) of the shipped order (contact-switch class first),
of the input order and of longest-first with perfect knowledge, against the balanced bound sum / slots and the longest
instance.  usage (GPU box): python tools/tail_study.py [workload] [B]"""
import heapq, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cmpc_amd
from cmpc_amd import workloads as wl
from cmpc_amd.solver import BatchedCentroidalMPC

name = sys.argv[1] if len(sys.argv) > 1 else "long_horizon"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
spec, rec = wl.make_workload(name, B=B, seed=int(os.environ["SEED"]) if os.environ.get("SEED") else None)
if spec.N > 20:
    spec.max_iter = 150
s = BatchedCentroidalMPC(spec, device="cuda:0")
d = torch.from_numpy(rec).to("cuda:0")
s.solve(d); torch.cuda.synchronize()
_, status, iters, _ = s.solve(d); torch.cuda.synchronize()
ms = s.last_kernel_ms()
it = iters.cpu().numpy().astype(np.int64); st = status.cpu().numpy()
N = spec.N
fl = np.stack([rec[:, 24 + 19 * np.arange(N) + 17], rec[:, 24 + 19 * np.arange(N) + 18]], -1)
fl = np.concatenate([fl, rec[:, None, 22:24]], 1)
sw = (np.diff(fl, axis=0 if fl.ndim == 1 else 1) != 0).any(axis=(1, 2))
slots = min(B, 256 * int(os.environ.get("SLOTS_PER_CU", 7 if spec.nv == 4 else 2)))   # resident workgroups (cmpc_hip.hip)


def makespan(order):
    h = [0] * slots
    heapq.heapify(h)
    end = 0
    for i in order:
        t = heapq.heappop(h) + it[i]
        end = max(end, t)
        heapq.heappush(h, t)
    return end


idx = np.arange(B)
switch_first = np.concatenate([idx[sw], idx[~sw][::-1]])          # the queue order of rounds 2-3 (switch class first)
# the shipped order (csrc/cmpc_hip.hip, cmpc_order_bucket; coefficients: csrc/cmpc_order_fit.h through cmpc_amd/queue_order.py):
# decreasing predicted iteration count, 64 buckets of half an iteration
from cmpc_amd import queue_order as qo
pred = qo.predicted_iterations(rec, spec)
bucket = qo.bucket_of(pred)
shipped = np.argsort(-bucket, kind="stable")
print(f"{name} B={B} N={N} nv={spec.nv}: kernel {ms:.1f} ms, slots {slots}, mean its {it.mean():.2f}, max {it.max()}, "
      f"switch class {sw.mean():.2%}")
print("iterations: quantiles 50/90/99/100 =", [int(np.quantile(it, q)) for q in (0.5, 0.9, 0.99, 1.0)])
for cls, m in (("switch", sw), ("no switch", ~sw)):
    if m.any():
        print(f"  {cls:10s} n={m.sum():5d} mean {it[m].mean():6.2f} q99 {int(np.quantile(it[m], 0.99))} max {it[m].max()}")
bal = it.sum() / slots
print(f"balanced bound {bal:.1f} instance-iterations per slot; longest instance {it.max()}")
print(f"predicted against measured iteration count: correlation {np.corrcoef(pred, it)[0, 1]:.2f}")
for lab, o in (("input order", idx), ("switch class first (rounds 2-3)", switch_first), ("shipped (predicted count, 64 buckets)", shipped),
               ("longest first (perfect knowledge)", np.argsort(-it))):
    mk = makespan(o)
    print(f"  {lab:36s} makespan {mk:5d} = {mk / bal:.3f} x balanced")
print(f"time per instance-iteration at the shipped makespan: {ms / makespan(shipped) * 1e3:.1f} us")
by = {int(v): int((st == v).sum()) for v in np.unique(st)}
print("status", by, " mean its by status", {int(v): round(float(it[st == v].mean()), 1) for v in np.unique(st)},
      " max its by status", {int(v): int(it[st == v].max()) for v in np.unique(st)})
