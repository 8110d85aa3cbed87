"""Developer script (GPU box): where do the one-wave kernel and the pipelined pair first differ?  Both are run with an
iteration cap of 1, 2, 3 ... on one instance and the returned iterates compared word by word.
usage: python tools/pair_diff.py [workload] [N] [index]"""
import dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cmpc_amd  # noqa: F401
from cmpc_amd import workloads as wl
from cmpc_amd.solver import BatchedCentroidalMPC

name = sys.argv[1] if len(sys.argv) > 1 else "randomized"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
idx = int(sys.argv[3]) if len(sys.argv) > 3 else 0
spec, rec = wl.make_workload(name, B=idx + 1, N=N)
rec = rec[idx:idx + 1]
d = torch.from_numpy(rec).cuda()
nu = spec.nu
for cap in (1, 2, 3, 4, 6, 8, 12, 16, 24):
    res = []
    for kern in (1, 2):
        s = BatchedCentroidalMPC(dataclasses.replace(spec, kernel=kern, max_iter=cap), device="cuda:0")
        out, st, it, kkt = s.solve(d)
        torch.cuda.synchronize()
        res.append((out.cpu().numpy()[0], int(st[0]), int(it[0]), float(kkt[0])))
    a, b = res[0][0], res[1][0]
    diff = np.abs(a - b)
    i = int(diff.argmax())
    nX = 20 * (N + 1)
    where = f"X stage {i // 20} var {i % 20}" if i < nX else f"U stage {(i - nX) // nu} var {(i - nX) % nu}"
    nd = int((a != b).sum())
    print(f"cap {cap:2d}: status {res[0][1]}/{res[1][1]} iters {res[0][2]}/{res[1][2]} kkt {res[0][3]:.3e}/{res[1][3]:.3e}  "
          f"words that differ {nd}, max |diff| {diff.max():.3e} at {where} (value {a[i]:.6e})")
    if nd:
        first = np.flatnonzero(a != b)[:12]
        print("   first differing words:", [(f"X{j // 20}.{j % 20}" if j < nX else f"U{(j - nX) // nu}.{(j - nX) % nu}") for j in first])
