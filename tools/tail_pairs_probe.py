"""Developer probe (GPU box): does it pay to run the END of a large batch's queue through the pipelined pair?  The
predicted-cheapest T instances go to a pair-kernel handle on a LOW-priority stream, the rest to the one-wave kernel on a
high-priority one, both launched together: the pair's workgroups are dispatched as the one-wave launch drains.
usage: python tools/tail_pairs_probe.py [workload] [B] [T ...]"""
import dataclasses, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cmpc_amd  # noqa: F401
from cmpc_amd import workloads as wl, queue_order as qo
from cmpc_amd.solver import BatchedCentroidalMPC

name = sys.argv[1] if len(sys.argv) > 1 else "randomized"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
Ts = [int(x) for x in sys.argv[3:]] or [0, 256, 512, 768, 1024, 1536]
spec, rec = wl.make_workload(name, B=B, N=20)
pred = qo.predicted_iterations(rec, spec)
order = np.argsort(-pred, kind="stable")           # longest predicted first
hi = torch.cuda.Stream(priority=-1)
lo = torch.cuda.Stream(priority=0)
single = BatchedCentroidalMPC(dataclasses.replace(spec, kernel=1), device="cuda:0")
pair = BatchedCentroidalMPC(dataclasses.replace(spec, kernel=2), device="cuda:0")
for T in Ts:
    head = torch.from_numpy(np.ascontiguousarray(rec[order[:B - T]])).cuda()
    tail = torch.from_numpy(np.ascontiguousarray(rec[order[B - T:]])).cuda() if T else None
    times = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(hi):
            r1 = single.solve(head)
        if T:
            with torch.cuda.stream(lo):
                r2 = pair.solve(tail)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    ms = 1e3 * np.median(times[1:])
    its = float(r1[2].double().mean())
    print(f"{name} B={B} tail T={T:5d}: {ms:7.2f} ms per step = {B / ms:7.1f} k instances/s   (one-wave part: mean iterations {its:.1f}"
          + (f", pair part {float(r2[2].double().mean()):.1f})" if T else ")"))
