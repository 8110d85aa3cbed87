"""Developer script: GPU vs oracle on a synthetic config, plus timing.  usage: gpu_check.py [config] [B] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmpc_amd
from cmpc_amd import workloads as wl
from cmpc_amd.solver import BatchedCentroidalMPC
from oracle import oracle_lib as ol

name = sys.argv[1] if len(sys.argv) > 1 else "perturbed"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = int(sys.argv[3]) if len(sys.argv) > 3 else None
ncheck = int(sys.argv[4]) if len(sys.argv) > 4 else min(B, 256)
spec, rec = wl.make_workload(name, B=B, N=N)
if os.environ.get("CMPC_MAX_ITER"):
    spec.max_iter = int(os.environ["CMPC_MAX_ITER"])
solver = BatchedCentroidalMPC(spec, device="cuda:0")
d_rec = torch.from_numpy(rec).to("cuda:0")
t0 = time.time()
out, status, iters, kkt = solver.solve(d_rec)
torch.cuda.synchronize()
t1 = time.time()
ms = solver.last_kernel_ms()
print(f"{name} B={B} N={spec.N} nv={spec.nv}: first call wall {t1 - t0:.3f}s kernel {ms:.2f} ms -> {B / ms * 1e3:.0f} solves/s")
for rep in range(3):
    out, status, iters, kkt = solver.solve(d_rec)
    ms = solver.last_kernel_ms()
    print(f"  rep {rep}: kernel {ms:.2f} ms -> {B / ms * 1e3:.0f} solves/s")
st = status.cpu().numpy(); it = iters.cpu().numpy()
print("status counts", np.bincount(st, minlength=3), "iters mean %.1f p50 %d p90 %d max %d" % (it.mean(), np.median(it), np.percentile(it, 90), it.max()))
cs = ol.default_spec(N=spec.N, nv=spec.nv, tol=spec.tol, max_iter=spec.max_iter, k1=spec.k1, k2=spec.k2, prox=spec.prox)
t0 = time.time()
ref, st_ref, it_ref, kkt_ref = ol.solve_batch(cs, rec[:ncheck])
dt = time.time() - t0
print(f"oracle: {ncheck} instances in {dt:.2f}s ({ncheck / dt:.1f} solves/s, {os.cpu_count()} cpus)")
got = out.cpu().numpy()[:ncheck]
both = (st[:ncheck] == 0) & (st_ref == 0)
print("status agree:", int((st[:ncheck] == st_ref).sum()), "/", ncheck, " iters equal:", int((it[:ncheck] == it_ref).sum()))
if both.any():
    rel = np.abs(got[both] - ref[both]).max(axis=1) / np.abs(ref[both]).max(axis=1)
    print("rel-inf error over converged: max %.3e median %.3e" % (rel.max(), np.median(rel)))
    worst = np.argsort(rel)[-3:]
    print("worst", worst, rel[worst])
# ---- diagnostics ----
out2, status2, iters2, kkt2 = solver.solve(d_rec)
torch.cuda.synchronize()
print("deterministic:", bool((out2 == out).all().item()), "iters equal:", bool((iters2 == iters).all().item()))
Nn, nu = spec.N, spec.nu
def groups(v):
    X = v[:20 * (Nn + 1)].reshape(Nn + 1, 20); U = v[20 * (Nn + 1):].reshape(Nn, nu)
    return {"com": X[:, 0:3], "vel": X[:, 3:6], "hw": X[:, 6:9], "th": X[:, 9:12], "feet": X[:, 12:20],
            "force": U[:, :nu - 8], "footvel": U[:, nu - 8:]}
if both.any():
    idxs = np.where(both)[0]
    order = np.argsort(rel)
    for tag, i in (("median", idxs[order[len(order) // 2]]), ("worst", idxs[order[-1]])):
        ga, gb = groups(got[i]), groups(ref[i])
        print(tag, "inst", i, "iters gpu/oracle", it[i], it_ref[i], "kkt", kkt.cpu().numpy()[i], kkt_ref[i])
        for k in ga:
            d = np.abs(ga[k] - gb[k])
            print("   %-8s maxabs diff %.3e (scale %.3e) at stage %d" % (k, d.max(), np.abs(gb[k]).max(), np.unravel_index(d.argmax(), d.shape)[0]))
# ---- phase profile (diagnostic build) ----
if os.environ.get("CMPC_PROF"):
    import ctypes
    from cmpc_amd import capi
    capi._lib = None
    capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcmpc_amd_prof.so")
    ps = BatchedCentroidalMPC(spec, device="cuda:0")
    ps.solve(d_rec); torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 28)()
    ps._lib.cmpc_profile_read(ps._h, buf)
    tot = float(sum(buf)); names = ["eval_rest", "build_H", "bwd_m", "bwd_lsolve", "store", "bwd_p", "vec_fwd", "step", "chol", "schur", "GtPG_T", "load+geom", "ineq", "Pb", "GtPG_Mupd", "-", "fwd_load", "fwd_backsub", "fwd_dx", "fwd_lam", "-", "chol_panels", "chol_mfma", "-", "load_stage", "eval_weights", "eval_grad", "-"]
    print("phase cycles (sum over instances):", {n: "%.1f%%" % (100 * b / tot) for n, b in zip(names, buf)})
    print("cycles per instance-iteration: %.0f" % (tot / it.sum()))
