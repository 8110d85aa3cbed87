"""Developer experiment (GPU box): would the drain of a large launch shrink if the tail of the queue ran as pairs?
The batch is put in queue order on the host; the first B - X records go to the one-wave kernel on one stream, the last X
to the pair kernel on a second stream launched right behind it (its workgroups get onto the CUs as the first kernel's
retire).  Wall time of both against one launch of the whole batch.  usage: python tools/drain_pairs_probe.py [B] [X ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cmpc_amd  # noqa: F401
from cmpc_amd import workloads as wl, queue_order as qo
from cmpc_amd.solver import BatchedCentroidalMPC

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
xs = [int(v) for v in sys.argv[2:]] or [0, 512, 1024, 1536, 2048]
spec, rec = wl.make_workload("randomized", B=B)
order = np.argsort(-qo.bucket_of(qo.predicted_iterations(rec, spec)), kind="stable")
rec = np.ascontiguousarray(rec[order])
dev = "cuda:0"
import dataclasses
sa = BatchedCentroidalMPC(dataclasses.replace(spec, kernel=1), device=dev)      # cmpc_spec.kernel: one wavefront per instance
sb = BatchedCentroidalMPC(dataclasses.replace(spec, kernel=2), device=dev)      # the pipelined pair
d = torch.from_numpy(rec).to(dev)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for X in xs:
    ts = []
    for rep in range(6):
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(s1)
        with torch.cuda.stream(s1):
            ra = sa.solve(d[:B - X])
            e1.record(s1)
        if X:
            s2.wait_event(e0)
            with torch.cuda.stream(s2):
                rb = sb.solve(d[B - X:])
                e2.record(s2)
        torch.cuda.synchronize()
        t = max(e0.elapsed_time(e1), e0.elapsed_time(e2) if X else 0.0)
        if rep >= 2:
            ts.append((t, e0.elapsed_time(e1), e0.elapsed_time(e2) if X else 0.0))
    best = min(ts)
    print(f"B={B} tail as pairs X={X}: both done after {best[0]:.1f} ms (one-wave part {best[1]:.1f}, pair part {best[2]:.1f}); "
          f"{B / best[0]:.1f} k instances/s", flush=True)
