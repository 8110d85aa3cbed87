"""Developer script (GPU box): dump the slab of a B = 1 launch (developer build, cmpc_debug_slab) for two libraries and
report which per-stage blocks differ.  usage: python tools/slab_diff.py libA.so libB.so workload N index cap"""
import ctypes, dataclasses, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "--dump":
    import torch
    import cmpc_amd  # noqa: F401
    from cmpc_amd import workloads as wl, capi
    from cmpc_amd.solver import BatchedCentroidalMPC
    name, N, idx, cap, kern, out = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    spec, rec = wl.make_workload(name, B=idx + 1, N=N)
    s = BatchedCentroidalMPC(dataclasses.replace(spec, kernel=kern, max_iter=cap), device="cuda:0")
    o, st, it, kkt = s.solve(torch.from_numpy(rec[idx:idx + 1]).cuda())
    torch.cuda.synchronize()
    lib = capi.load()
    lib.cmpc_debug_slab.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    p, n = ctypes.c_void_p(), ctypes.c_size_t()
    assert lib.cmpc_debug_slab(s._h, ctypes.byref(p), ctypes.byref(n)) == 0
    buf = torch.empty(n.value, dtype=torch.float64, device="cuda:0")
    import torch.cuda
    hip = ctypes.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), p, ctypes.c_size_t(n.value * 8), 3) == 0
    np.save(out, buf.cpu().numpy())
    sys.exit(0)

la, lb, name, N, idx, cap = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
ka = int(os.environ.get("KA", "1")); kb = int(os.environ.get("KB", "1"))
for lib, k, out in ((la, ka, "/tmp/slab_a.npy"), (lb, kb, "/tmp/slab_b.npy")):
    subprocess.check_call([sys.executable, __file__, "--dump", name, str(N), str(idx), str(cap), str(k), out],
                          env=dict(os.environ, CMPC_LIB_PATH=os.path.abspath(lib)))
a, b = np.load("/tmp/slab_a.npy"), np.load("/tmp/slab_b.npy")
NZ, NXA, NU, NI, NTRI = 60, 28, 32, 55, 1830
gM = 0; gAL = 1832; gGH = gAL + NZ; gB = gGH + 192; gPB = gB + NXA; gPV = gPB + NXA; gL = gPV + NXA; gG = gL + NU; gL1 = gG + NI; gPV1 = gL1 + NU
STAGE = (gPV1 + NXA + 7) // 8 * 8
blocks = [("M", gM, NTRI), ("AL", gAL, NZ), ("GH", gGH, 192), ("B", gB, NXA), ("PV", gPV, NXA), ("L", gL, NU), ("G", gG, NI), ("L1", gL1, NU), ("PV1", gPV1, NXA)]
for k in range(N, -1, -1):
    row = []
    for nm, o, n in blocks:
        x, y = a[k * STAGE + o:k * STAGE + o + n], b[k * STAGE + o:k * STAGE + o + n]
        neq = (x != y) & ~(np.isnan(x) & np.isnan(y))
        if neq.any():
            i = int(np.flatnonzero(neq)[0])
            row.append(f"{nm}:{int(neq.sum())} (first word {i}: {x[i]:.17g} vs {y[i]:.17g})")
    print(f"stage {k:2d}:", "; ".join(row) if row else "equal")
    x, y = a[k * STAGE:k * STAGE + NTRI], b[k * STAGE:k * STAGE + NTRI]
    neq = np.flatnonzero(x != y)
    if neq.size:
        rc = []
        for w in neq:
            r = int((np.sqrt(8 * w + 1) - 1) / 2)
            rc.append((r, int(w - r * (r + 1) // 2), float(abs(x[w] - y[w]) / max(abs(x[w]), 1e-300))))
        print("   M words (row, col, rel diff):", [(r, c, f"{d:.1e}") for r, c, d in rc[:80]])
p = (N + 1) * STAGE
for nm, n in (("x", (N + 1) * NXA), ("lam", (N + 1) * NXA), ("dx", (N + 1) * NXA), ("lamn", (N + 1) * NXA), ("u", (N + 1) * NU), ("du", (N + 1) * NU),
              ("upx", (N + 1) * NU), ("s", (N + 1) * NI), ("z", (N + 1) * NI), ("ds", (N + 1) * NI), ("dz", (N + 1) * NI)):
    x, y = a[p:p + n], b[p:p + n]
    neq = (x != y) & ~(np.isnan(x) & np.isnan(y))
    print(f"{nm}: {int(neq.sum())} words differ")
    p += n
