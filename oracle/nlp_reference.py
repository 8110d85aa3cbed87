"""TEST INFRASTRUCTURE ONLY -- literal restatement of the reference NLP (torch fp64, autograd).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything under oracle/.  Nothing here is on the product path.

PARITY UNPINNED: the reference solves this NLP with CasADi+IPOPT, which cannot be
run in the build environment (no casadi/ipopt, no network), and the reference
ships no solver outputs usable as golden vectors.  This file is a
line-by-line reading of the reference's problem statement; it anchors the C
oracle (oracle/cmpc_oracle.c) and through it the HIP solver.

What it restates (reference: code/centroidal_mpc_vertices.py):
  layout of U / state          :135-166
  equality constraints          :185-190   (x_0 = x0, forward Euler on centroidal_dynamic)
  Lyapunov rows                 :193-220   (UNSUBSTITUTED: written on X[:,i+1] as in the reference)
  angular-momentum contraction  :223-224
  CoM height                    :229-230
  friction cone / unilateral    :235-254
  contact-location box          :258-271
  cost                          :275-353
  centroidal_dynamic            :371-461

Build-defined additions (documented in DESIGN.md): per-instance mass and mu,
`nv` vertices per foot, a proximal term 0.5*rho*||U-U_prox||^2 that selects one
point of the reference problem's non-unique optimal set, and IPOPT's
bound_relax_factor (inequalities relaxed by 1e-8).

Inequalities are returned in the form g(w) <= 0, rows whose contact flag is 0
(reference rows that read 0<=0) are dropped.
"""
import numpy as np
import torch

NX = 20


class Spec:
    """Problem constants shared by a batch (reference values as defaults)."""

    def __init__(self, N=20, nv=4, delta=0.01, g=9.81, k1=4.0, k2=0.1, w_rate=1.0,
                 prox=1e-4, relax=1e-8, foot_length=0.25, foot_width=0.13):
        self.N, self.nv = N, nv
        self.delta, self.g, self.k1, self.k2 = delta, g, k1, k2
        self.w_rate, self.prox, self.relax = w_rate, prox, relax
        self.w_hw, self.w_cxy, self.w_foot, self.w_force = 1000.0, 1.0, 1000.0, 10.0
        self.w_cz_const = 2000.0
        self.cz_max = 0.76
        self.box = (0.01, 0.005, 0.00005)
        L, W = foot_length / 2, foot_width / 2
        corners = [[L, W, 0.], [L, -W, 0.], [-L, -W, 0.], [-L, W, 0.]]   # reference :55-60
        if nv == 4:
            self.verts = np.array(corners)
        elif nv == 8:                                                    # corners + edge midpoints
            mids = [[L, 0, 0.], [0, -W, 0.], [-L, 0, 0.], [0, W, 0.]]
            self.verts = np.array(corners + mids)
        else:
            raise ValueError("nv must be 4 or 8")
        self.nu = 6 * nv + 8
        self.nrec = 24 + 19 * N

    def w_cz(self, i):
        half = self.w_cz_const / 2
        return (self.w_cz_const - half) * np.exp(-i) + half             # reference :301-305


def unpack_record(spec, rec):
    """Parameter record (see DESIGN.md 'parameter record') -> dict of arrays."""
    N = spec.N
    rec = np.asarray(rec, dtype=np.float64)
    st = rec[24:24 + 19 * N].reshape(N, 19)
    gl = np.concatenate([st[:, 17], rec[22:23]])
    gr = np.concatenate([st[:, 18], rec[23:24]])
    return dict(x0=rec[0:20], mass=rec[20], mu=rec[21], com_ref=st[:, 0:9].T.copy(),
                pl_ref=st[:, 9:12].T.copy(), pr_ref=st[:, 12:15].T.copy(),
                yl_ref=st[:, 15].copy(), yr_ref=st[:, 16].copy(), gl=gl, gr=gr)


def pack_record(spec, x0, mass, mu, com_ref, pl_ref, pr_ref, yl_ref, yr_ref, gl, gr):
    N = spec.N
    rec = np.zeros(spec.nrec)
    rec[0:20] = x0
    rec[20], rec[21], rec[22], rec[23] = mass, mu, gl[N], gr[N]
    st = rec[24:].reshape(N, 19)
    st[:, 0:9] = np.asarray(com_ref).T
    st[:, 9:12] = np.asarray(pl_ref).T
    st[:, 12:15] = np.asarray(pr_ref).T
    st[:, 15], st[:, 16] = yl_ref, yr_ref
    st[:, 17], st[:, 18] = gl[:N], gr[:N]
    return rec


def _t(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64)


def _cross(a, b):
    return torch.stack((a[1] * b[2] - a[2] * b[1],
                        a[2] * b[0] - a[0] * b[2],
                        a[0] * b[1] - a[1] * b[0]))


def dynamics(spec, par, state, com_ref, gam_l, gam_r, inp):
    """centroidal_dynamic, reference :371-461.  state (20,), inp (nu,) torch."""
    nv, m = spec.nv, par['mass']
    grav = _t([0., 0., -spec.g])
    com, vel, th = state[0:3], state[3:6], state[9:12]
    yaw_l, p_l, yaw_r, p_r = state[12], state[13:16], state[16], state[17:20]
    F = inp[0:6 * nv].reshape(2 * nv, 3)
    Fl, Fr = F[:nv], F[nv:]
    vel_l, vel_r = inp[6 * nv:6 * nv + 3], inp[6 * nv + 3:6 * nv + 6]
    om_l, om_r = inp[6 * nv + 6], inp[6 * nv + 7]
    Vl = Fl.sum(0) * gam_l
    Vr = Fr.sum(0) * gam_r
    z1 = com - com_ref[0:3]
    z2 = spec.k1 * z1 + (vel - com_ref[3:6])

    def verts(pos, yaw):
        c, s = torch.cos(yaw), torch.sin(yaw)
        out = []
        for v in spec.verts:
            out.append(torch.stack((c * v[0] - s * v[1], s * v[0] + c * v[1],
                                    torch.zeros((), dtype=torch.float64) + v[2])) + pos)
        return out

    tl = sum(_cross(pv - com, f) for pv, f in zip(verts(p_l, yaw_l), Fl)) * gam_l
    tr = sum(_cross(pv - com, f) for pv, f in zip(verts(p_r, yaw_r), Fr)) * gam_r
    ddcom = grav + (Vl + Vr + th * 0) / m
    return torch.cat((vel, ddcom, tl + tr, z2 / m,
                      ((1 - gam_l) * om_l).reshape(1), (1 - gam_l) * vel_l,
                      ((1 - gam_r) * om_r).reshape(1), (1 - gam_r) * vel_r))


def split(spec, w):
    """w = [X (20,(N+1)) column-major, U (nu,N) column-major]  (reference layout)."""
    N, nu = spec.N, spec.nu
    X = w[:NX * (N + 1)].reshape(N + 1, NX).T
    U = w[NX * (N + 1):].reshape(N, nu).T
    return X, U


def cost(spec, par, w, u_prox=None):
    """reference :275-353 (+ proximal term)."""
    N, nv = spec.N, spec.nv
    X, U = split(spec, w)
    cr, plr, prr = _t(par['com_ref']), _t(par['pl_ref']), _t(par['pr_ref'])
    ylr, yrr, gl, gr = _t(par['yl_ref']), _t(par['yr_ref']), _t(par['gl']), _t(par['gr'])
    J = torch.zeros((), dtype=torch.float64)
    for i in range(N):
        F = U[0:6 * nv, i].reshape(2 * nv, 3)
        Fl, Fr = F[:nv], F[nv:]
        avg_l = Fl.sum(0) * gl[i] * gl[i] / nv          # reference :211-215,277-279
        avg_r = Fr.sum(0) * gr[i] * gr[i] / nv
        J = J + spec.w_hw * (X[6:9, i] ** 2).sum()
        J = J + spec.w_cxy * (X[0, i + 1] - cr[0, i]) ** 2 + spec.w_cxy * (X[1, i + 1] - cr[1, i]) ** 2
        J = J + spec.w_cz(i) * (X[2, i + 1] - cr[2, i]) ** 2
        J = J + spec.w_foot * (((X[13:16, i + 1] - plr[:, i]) * gl[i + 1]) ** 2).sum()
        J = J + spec.w_foot * (((X[17:20, i + 1] - prr[:, i]) * gr[i + 1]) ** 2).sum()
        J = J + spec.w_foot * ((X[12, i + 1] - ylr[i]) * gl[i + 1]) ** 2
        J = J + spec.w_foot * ((X[16, i + 1] - yrr[i]) * gr[i + 1]) ** 2
        J = J + spec.w_force * gl[i] * ((avg_l[None, :] - Fl) ** 2).sum()
        J = J + spec.w_force * gr[i] * ((avg_r[None, :] - Fr) ** 2).sum()
        J = J + spec.w_force * (1 - gl[i]) * (Fl ** 2).sum()
        J = J + spec.w_force * (1 - gr[i]) * (Fr ** 2).sum()
    for i in range(N - 1):                               # reference :343-351
        d = (U[0:6 * nv, i + 1] - U[0:6 * nv, i]).reshape(2 * nv, 3)[:, 2]
        J = J + spec.w_rate * gl[i] * (d[:nv] ** 2).sum() + spec.w_rate * gr[i] * (d[nv:] ** 2).sum()
    if spec.prox > 0:
        up = torch.zeros_like(U) if u_prox is None else _t(u_prox)
        J = J + 0.5 * spec.prox * ((U - up) ** 2).sum()
    return J


def equalities(spec, par, w):
    """reference :185-190.  Returns (20*(N+1),)."""
    N = spec.N
    X, U = split(spec, w)
    cr, gl, gr = _t(par['com_ref']), _t(par['gl']), _t(par['gr'])
    out = [X[:, 0] - _t(par['x0'])]
    for i in range(N):
        out.append(X[:, i] + spec.delta * dynamics(spec, par, X[:, i], cr[:, i], gl[i], gr[i], U[:, i])
                   - X[:, i + 1])
    return torch.cat(out)


def inequalities(spec, par, w):
    """All rows g(w) <= 0 (relaxed by spec.relax), inactive-contact rows dropped."""
    N, nv, m, mu = spec.N, spec.nv, par['mass'], par['mu']
    k1, k2 = spec.k1, spec.k2
    X, U = split(spec, w)
    cr, plr, prr = _t(par['com_ref']), _t(par['pl_ref']), _t(par['pr_ref'])
    gl, gr = par['gl'], par['gr']
    grav = _t([0., 0., -spec.g])
    rows = []
    for i in range(N):                                   # Lyapunov, reference :202-220
        F = U[0:6 * nv, i].reshape(2 * nv, 3)
        z1 = X[0:3, i + 1] - cr[0:3, i]
        z2 = k1 * z1 + (X[3:6, i + 1] - cr[3:6, i])
        u_n = -(k1 + k2) * z2 + k1 * k1 * z1 - grav + cr[6:9, i] - X[9:12, i] / m
        V = (F[:nv].sum(0) * gl[i] + F[nv:].sum(0) * gr[i]) / m
        rows.append((-k1 * (z1 * z1).sum() - k2 * (z2 * z2).sum() + (z1 * z2).sum()
                     + (z2 * (V - u_n)).sum()).reshape(1))
    rows.append(((X[6:9, 1] ** 2).sum() - (X[6:9, 0] ** 2).sum()).reshape(1))   # :223-224
    for i in range(N):
        rows.append((X[2, i] - spec.cz_max).reshape(1))  # :230
        F = U[0:6 * nv, i].reshape(2 * nv, 3)
        for foot, gam in ((F[:nv], gl[i]), (F[nv:], gr[i])):
            if gam == 0:
                continue
            for f in foot:                               # :236-254
                rows.append(torch.stack((f[0] - mu * f[2], -f[0] - mu * f[2],
                                         f[1] - mu * f[2], -f[1] - mu * f[2], -f[2])) * gam)
    for i in range(N):                                   # :258-271
        for p, ref, gam in ((X[13:16, i + 1], plr[:, i], gl[i + 1]), (X[17:20, i + 1], prr[:, i], gr[i + 1])):
            if gam == 0:
                continue
            for ax in range(3):
                d = (p[ax] - ref[ax]) * gam
                rows.append(torch.stack((d - spec.box[ax], -d - spec.box[ax])))
    return torch.cat(rows) - spec.relax
