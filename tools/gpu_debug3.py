"""Developer script: full primal-dual iterate of instance 0 after k iterations, GPU vs oracle."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmpc_amd
from cmpc_amd import workloads as wl, capi
from oracle import oracle_lib as ol
capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcmpc_amd_prof.so")
from cmpc_amd.solver import BatchedCentroidalMPC
inst = int(sys.argv[1]) if len(sys.argv) > 1 else 43
spec, rec = wl.make_workload("perturbed", B=64, N=20)
rec = rec[inst:inst + 1].copy()
N, nx, ni = spec.N, 28, 55
nz, nu = 60, 32
n0 = 2 * (N + 1) * nx + 2 * (N + 1) * ni + 8
per = nz + 3 * nx + 3 * nu
n = n0 + (N + 1) * per
for mi in (2,):
    spec.max_iter = 50
    import cmpc_amd.problem as _pb
    _orig = _pb.to_cspec
    def _tc(sp, _mi=mi):
        c = _orig(sp); c.reserved = _mi; return c
    import cmpc_amd.solver as _sv
    _sv.to_cspec = _tc
    solver = BatchedCentroidalMPC(spec, device="cuda:0")
    out, status, iters, kkt = solver.solve(torch.from_numpy(rec).to("cuda:0"))
    torch.cuda.synchronize()
    buf = np.zeros(n)
    big = np.zeros(61000)
    solver._lib.cmpc_debug_read(solver._h, big.ctypes.data_as(ctypes.c_void_p), 61000)
    buf = big[:n].copy(); geo = big[60000:60000 + 210]; GHg = big[60210:60210 + 180].reshape(3, 60)
    cs = ol.default_spec(N=N, nv=4, tol=spec.tol, max_iter=50, reserved=mi, k1=spec.k1, k2=spec.k2, prox=spec.prox)
    full = np.zeros(30000); o = np.zeros(ol.nsol(cs))
    ol.lib().cmpc_oracle_solve_full(ctypes.byref(cs), rec.ctypes.data_as(ctypes.c_void_p), None, o.ctypes.data_as(ctypes.c_void_p), full.ctypes.data_as(ctypes.c_void_p))
    a = (N + 1) * nx; b = (N + 1) * ni
    parts = {"x": (0, a, nx), "lam": (a, 2 * a, nx), "s": (2 * a, 2 * a + b, ni), "z": (2 * a + b, 2 * a + 2 * b, ni)}
    print("stopped before applying step", mi, ":  [mu reg_last ap ad nreg] gpu", buf[n0-8:n0-3], "oracle", full[n0-8:n0-3])
    for name, (lo, hi, w) in parts.items():
        g, r = buf[lo:hi].reshape(N + 1, w), full[lo:hi].reshape(N + 1, w)
        d = np.abs(g - r); idx = np.unravel_index(d.argmax(), d.shape)
        print("   %-4s max abs diff %.3e (scale %.3e) at stage %d index %d: gpu %.12e oracle %.12e" % (name, d.max(), np.abs(r).max(), idx[0], idx[1], g[idx], r[idx]))
        if name == "lam":
            pst = d.max(axis=1); print("      per-stage lam diff:", " ".join("%.1e" % v for v in pst))
            perc = d.max(axis=0); print("      per-comp lam diff:", " ".join("%.1e" % v for v in perc))
    G = buf[n0:].reshape(N + 1, per); R = full[n0:n].reshape(N + 1, per)
    secs = {"h": (0, nz), "b": (nz, nz + nx), "l": (nz + nx, nz + nx + nu), "p": (nz + nx + nu, nz + 2 * nx + nu),
            "du": (nz + 2 * nx + nu, nz + 2 * nx + 2 * nu), "dx": (nz + 2 * nx + 2 * nu, nz + 3 * nx + 2 * nu), "Ldiag": (nz + 3 * nx + 2 * nu, per)}
    for name, (lo, hi) in secs.items():
        d = np.abs(G[:, lo:hi] - R[:, lo:hi]); idx = np.unravel_index(d.argmax(), d.shape)
        print("   last-step %-3s max abs diff %.3e (scale %.3e) at stage %d idx %d | per-stage:" % (name, d.max(), np.abs(R[:, lo:hi]).max(), idx[0], idx[1]), " ".join("%.0e" % v for v in d.max(axis=1)))
    VR = geo[:24].reshape(8, 3); MISC = geo[24:88]; XK = geo[88:116]; UK = geo[116:148]; BV = geo[148:176]; XN1 = geo[176:204]
    F = UK[:24].reshape(8, 3)
    # recompute on host
    vloc = np.array([[0.125, 0.065], [0.125, -0.065], [-0.125, -0.065], [-0.125, 0.065]])
    r = np.zeros((8, 3))
    for f in range(2):
        for j in range(4):
            p = XK[13 + 4 * f:16 + 4 * f]
            r[f * 4 + j] = p + np.array([vloc[j, 0], vloc[j, 1], 0]) - XK[0:3]
    gl, gr = rec[0, 24 + 19 * 5 + 17], rec[0, 24 + 19 * 5 + 18]
    tau = gl * np.cross(r[:4], F[:4]).sum(0) + gr * np.cross(r[4:], F[4:]).sum(0)
    print("   stage5 VR diff", np.abs(VR - r).max(), "tau gpu", MISC[6:9], "host", tau, "gam", gl, gr)
    print("   b_h gpu", BV[6:9], "host", XK[6:9] + 0.01 * tau - XN1[6:9])
    st = rec[0, 24:].reshape(N, 19); print("   gamma_l", st[:, 17].astype(int), "gamma_r", st[:, 18].astype(int), rec[0, 22:24])
    names = ['cx','cy','cz','vx','vy','vz','hx','hy','hz','tx','ty','tz','yl','plx','ply','plz','yr','prx','pry','prz'] + ['fp%d' % i for i in range(8)]
    for kk in (18, 17, 16, 15):
        Pg = big[50000 + (kk - 15) * 1000: 50000 + (kk - 15) * 1000 + 784].reshape(28, 28); Po = full[n0 - 8 + 20000 + (kk - 15) * 1000: n0 - 8 + 20000 + (kk - 15) * 1000 + 784].reshape(28, 28)
        d = np.abs(Pg - Po); print("   P_%d max abs diff %.3e (scale %.3e)" % (kk, d.max(), np.abs(Po).max()))
        if d.max() > 1e-9 * np.abs(Po).max():
            rows = np.where(d.max(axis=1) > 0.01 * d.max())[0]; print("      rows with large diff:", [names[r] for r in rows])
            i, j = np.unravel_index(d.argmax(), d.shape); print("      worst (%s,%s): gpu %.10e oracle %.10e" % (names[i], names[j], Pg[i, j], Po[i, j]))
    d_ = 0.01; m_ = rec[0, 20]
    def skew(a): return np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    GHh = np.zeros((3, 60))
    gam = [gl, gr]
    for v in range(8):
        GHh[:, 3 * v:3 * v + 3] = d_ * gam[v // 4] * skew(r[v])
    Fsum = [F[:4].sum(0), F[4:].sum(0)]
    GHh[:, 32:35] = d_ * (gl * skew(Fsum[0]) + gr * skew(Fsum[1]))
    GHh[:, 32 + 13:32 + 16] = -d_ * gl * skew(Fsum[0]); GHh[:, 32 + 17:32 + 20] = -d_ * gr * skew(Fsum[1])
    dv = np.array([[-vloc[j, 1], vloc[j, 0], 0] for j in range(4)])
    GHh[:, 32 + 12] = d_ * gl * np.cross(dv, F[:4]).sum(0); GHh[:, 32 + 16] = d_ * gr * np.cross(dv, F[4:]).sum(0)
    dd = np.abs(GHg - GHh); print("   GH stage5 max diff %.3e (scale %.3e) at" % (dd.max(), np.abs(GHh).max()), np.unravel_index(dd.argmax(), dd.shape))
    print("   GH c cols gpu", GHg[:, 32:35].ravel(), "host", GHh[:, 32:35].ravel())
