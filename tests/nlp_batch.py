"""Batched numpy evaluation of the NLP's constraint functions (nv = 4), used for size-independent
property checks at full batch sizes.  Follows code/centroidal_mpc_vertices.py:185-271, :371-461."""
import numpy as np


def split(sol, N, nu):
    B = sol.shape[0]
    return sol[:, :20 * (N + 1)].reshape(B, N + 1, 20), sol[:, 20 * (N + 1):].reshape(B, N, nu)


def residuals(spec, rec, sol):
    """dict of worst violations per instance: dynamics defect, cone, unilateral, height, box, Lyapunov."""
    N, nu, nv = spec.N, spec.nu, spec.nv
    assert nv == 4
    B = rec.shape[0]
    X, U = split(sol, N, nu)
    m, mu = rec[:, 20], rec[:, 21]
    st = rec[:, 24:].reshape(B, N, 19)
    gl = np.concatenate([st[:, :, 17], rec[:, 22:23]], axis=1)
    gr = np.concatenate([st[:, :, 18], rec[:, 23:24]], axis=1)
    d, g, k1, k2 = spec.delta, spec.g, spec.k1, spec.k2
    verts = spec.vertices()
    out = {k: np.zeros(B) for k in ("x0", "defect", "cone", "unilateral", "height", "box", "lyapunov", "swing_force")}
    out["x0"] = np.abs(X[:, 0] - rec[:, :20]).max(axis=1)
    for k in range(N):
        x, u, xn = X[:, k], U[:, k], X[:, k + 1]
        F = u[:, :24].reshape(B, 8, 3)
        gam = np.stack([gl[:, k]] * 4 + [gr[:, k]] * 4, axis=1)
        tau = np.zeros((B, 3))
        for f, (iy, ip) in enumerate(((12, 13), (16, 17))):
            c, s = np.cos(x[:, iy]), np.sin(x[:, iy])
            for j in range(4):
                rv = np.stack([c * verts[j, 0] - s * verts[j, 1], s * verts[j, 0] + c * verts[j, 1], np.zeros(B)], axis=1)
                r = x[:, ip:ip + 3] + rv - x[:, 0:3]
                tau += gam[:, f * 4 + j, None] * np.cross(r, F[:, f * 4 + j])
        V = (gam[:, :, None] * F).sum(axis=1) / m[:, None]
        grav = np.array([0, 0, -g])
        f_ = np.zeros((B, 20))
        f_[:, 0:3] = x[:, 3:6]
        f_[:, 3:6] = grav + V
        f_[:, 6:9] = tau
        z1 = x[:, 0:3] - st[:, k, 0:3]
        z2 = k1 * z1 + x[:, 3:6] - st[:, k, 3:6]
        f_[:, 9:12] = z2 / m[:, None]
        f_[:, 12] = (1 - gl[:, k]) * u[:, 30]
        f_[:, 13:16] = (1 - gl[:, k, None]) * u[:, 24:27]
        f_[:, 16] = (1 - gr[:, k]) * u[:, 31]
        f_[:, 17:20] = (1 - gr[:, k, None]) * u[:, 27:30]
        out["defect"] = np.maximum(out["defect"], np.abs(x + d * f_ - xn).max(axis=1))
        fx, fy, fz = F[..., 0], F[..., 1], F[..., 2]
        cone = np.maximum(np.abs(fx), np.abs(fy)) - mu[:, None] * fz
        out["cone"] = np.maximum(out["cone"], (gam * cone).max(axis=1))
        out["unilateral"] = np.maximum(out["unilateral"], (gam * -fz).max(axis=1))
        # a foot in the air carries nothing -- except on its first airborne stage, where the rate term
        # gamma[k-1]*(fz[k]-fz[k-1])^2 (reference :343-351) still ties fz to the stance value
        if k >= 1:
            gprev = np.stack([gl[:, k - 1]] * 4 + [gr[:, k - 1]] * 4, axis=1)
            air = (1 - gam) * (1 - gprev)
            out["swing_force"] = np.maximum(out["swing_force"], (air[:, :, None] * np.abs(F)).max(axis=(1, 2)))
        if k >= 1:
            out["height"] = np.maximum(out["height"], x[:, 2] - spec.cz_max)
        # Lyapunov row (:202-220), written on the next state as in the reference
        z1n = xn[:, 0:3] - st[:, k, 0:3]
        z2n = k1 * z1n + xn[:, 3:6] - st[:, k, 3:6]
        un = -(k1 + k2) * z2n + k1 * k1 * z1n - grav + st[:, k, 6:9] - x[:, 9:12] / m[:, None]
        lyap = (-k1 * (z1n * z1n).sum(1) - k2 * (z2n * z2n).sum(1) + (z1n * z2n).sum(1) + (z2n * (V - un)).sum(1))
        out["lyapunov"] = np.maximum(out["lyapunov"], lyap)
        for ip, ref, gg in ((13, st[:, k, 9:12], gl[:, k + 1]), (17, st[:, k, 12:15], gr[:, k + 1])):
            dd = np.abs(xn[:, ip:ip + 3] - ref) * gg[:, None] - np.array(spec.box)[None, :]
            out["box"] = np.maximum(out["box"], dd.max(axis=1))
    hw0, hw1 = (X[:, 0, 6:9] ** 2).sum(1), (X[:, 1, 6:9] ** 2).sum(1)
    out["contraction"] = hw1 - hw0
    return out
