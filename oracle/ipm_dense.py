"""TEST INFRASTRUCTURE ONLY -- slow, independent full-space interior-point solver.

Solves the literal NLP of oracle/nlp_reference.py with autograd derivatives and
dense KKT factorisations (no Riccati, no stage structure, no hand derivatives).
It exists to cross-check oracle/cmpc_oracle.c, which shares its algorithmic
structure with the HIP solver.  Never imported by the product path.

Algorithm: monotone-barrier primal-dual interior point (Fiacco-McCormick barrier
schedule as in IPOPT's default mode), exact Lagrangian Hessian with inertia
correction (delta_w * I), l1-merit backtracking line search.
"""
import numpy as np
import scipy.linalg as sla
import torch

from . import nlp_reference as nlp


def _inertia_ok(K, nw, ne):
    _, d, _ = sla.ldl(K)
    ev = np.linalg.eigvalsh(d)
    return int((ev > 0).sum()) == nw and int((ev < 0).sum()) == ne


def solve(spec, par, w0=None, u_prox=None, tol=1e-9, max_iter=300, verbose=False,
          mu0=100.0, hessian="exact", linesearch=True):
    """Returns dict(w, iters, kkt, status, lam, z, s).  status 0 = converged."""
    N = spec.N
    nw = nlp.NX * (N + 1) + spec.nu * N
    w = np.zeros(nw) if w0 is None else np.array(w0, dtype=np.float64)

    def evaluate(wv):
        wt = torch.tensor(wv)
        return (nlp.cost(spec, par, wt, u_prox).item(), nlp.equalities(spec, par, wt).numpy(),
                nlp.inequalities(spec, par, wt).numpy())

    _, c0, g0 = evaluate(w)
    ne, ni = c0.size, g0.size
    s = np.maximum(-g0, 1e-2)
    mu = mu0
    z = mu / s
    lam = np.zeros(ne)
    nu_pen = 1.0
    reg_last = 0.0
    status, kkt, it = 1, np.inf, 0
    for it in range(max_iter):
        wt = torch.tensor(w, requires_grad=True)
        f = nlp.cost(spec, par, wt, u_prox)
        gradf = torch.autograd.grad(f, wt)[0].numpy()
        f = f.item()
        Jc = torch.autograd.functional.jacobian(lambda v: nlp.equalities(spec, par, v), wt.detach(),
                                                vectorize=True).numpy()
        Jg = torch.autograd.functional.jacobian(lambda v: nlp.inequalities(spec, par, v), wt.detach(),
                                                vectorize=True).numpy()
        _, c, g = evaluate(w)
        lam_t, z_t = torch.tensor(lam), torch.tensor(z)

        def lagr(v):
            L = nlp.cost(spec, par, v, u_prox)
            if hessian == "exact":
                L = L + (lam_t * nlp.equalities(spec, par, v)).sum() + (z_t * nlp.inequalities(spec, par, v)).sum()
            return L

        H = torch.autograd.functional.hessian(lagr, wt.detach(), vectorize=True).numpy()
        rd = gradf + Jc.T @ lam + Jg.T @ z
        rg = g + s
        comp = s * z
        sd = max(100.0, (np.abs(lam).sum() + np.abs(z).sum()) / (ne + ni)) / 100.0
        e_d, e_p, e_c0 = np.abs(rd).max() / sd, max(np.abs(c).max(), np.abs(rg).max()), np.abs(comp).max() / sd
        kkt = max(e_d, e_p, e_c0)
        if verbose:
            print(f"it {it:3d} f={f:.8e} d={e_d:.2e} p={e_p:.2e} c={e_c0:.2e} mu={mu:.1e} reg={reg_last:.1e} zmax={z.max():.2e}")
        if kkt < tol:
            status = 0
            break
        while mu > tol / 10 and max(e_d, e_p, np.abs(comp - mu).max() / sd) < 10 * mu:
            mu = max(tol / 10, min(0.1 * mu, mu ** 1.5))
        Sig = z / s
        Hb = H + Jg.T @ (Sig[:, None] * Jg)
        reg = 0.0
        while True:
            K = np.block([[Hb + reg * np.eye(nw), Jc.T], [Jc, np.zeros((ne, ne))]])
            if _inertia_ok(K, nw, ne):
                break
            reg = (1e-4 if reg_last == 0 else max(1e-20, reg_last / 3)) if reg == 0 else \
                  reg * (100 if reg_last == 0 else 8)
            if reg > 1e20:
                return dict(w=w, iters=it, kkt=kkt, status=2, lam=lam, z=z, s=s)
        if reg > 0:
            reg_last = reg
        rhs_w = -(gradf + Jg.T @ (mu / s + Sig * rg))
        sol = sla.lu_solve(sla.lu_factor(K), np.concatenate([rhs_w, -c]))
        dw, lam_new = sol[:nw], sol[nw:]
        ds = -rg - Jg @ dw
        dz = (mu - comp - z * ds) / s

        def max_step(v, dv, tau):
            neg = dv < 0
            return 1.0 if not neg.any() else min(1.0, (tau * v[neg] / -dv[neg]).min())

        tau = max(0.99, 1 - mu)
        ap, ad = max_step(s, ds, tau), max_step(z, dz, tau)
        # l1 merit on (w, s)
        infeas = np.abs(c).sum() + np.abs(rg).sum()
        dphi_obj = gradf @ dw - mu * (ds / s).sum()
        quad = dw @ (Hb + reg * np.eye(nw)) @ dw
        if infeas > 0:
            nu_need = (dphi_obj + 0.5 * max(quad, 0.0)) / (0.9 * infeas)
            if nu_pen < nu_need:
                nu_pen = nu_need + 1.0
        dphi = dphi_obj - nu_pen * infeas

        def merit(wv, sv):
            fv, cv, gv = evaluate(wv)
            return fv - mu * np.log(sv).sum() + nu_pen * (np.abs(cv).sum() + np.abs(gv + sv).sum())

        phi0 = merit(w, s)
        a = ap
        for _ in range(30 if linesearch else 0):
            if merit(w + a * dw, s + a * ds) <= phi0 + 1e-8 * a * dphi + 1e-12 * abs(phi0):
                break
            a *= 0.5
        w = w + a * dw
        s = s + a * ds
        lam = lam + a * (lam_new - lam)
        z = z + ad * dz
        # keep z within a factor of mu/s (IPOPT's kappa_sigma safeguard)
        z = np.clip(z, mu / s / 1e10, mu / s * 1e10)
        if verbose and a < ap:
            print(f"      (step {a:.2e} of {ap:.2e})")
    return dict(w=w, iters=it, kkt=kkt, status=status, lam=lam, z=z, s=s)


def refine_active_set(spec, par, r, u_prox=None, s_max=1e-9, rounds=40, verbose=False):
    """Primal-dual active-set refinement of an interior-point answer `r` (the dict `solve` returns): Newton's method on the
    KKT conditions with a working set of inequality rows held as equalities, the others dropped, and the working set
    corrected between Newton solves -- a violated row comes in, a row with a negative multiplier goes out, and of a group
    of rows that cannot all hold as equalities (the five rows of a vertex at the apex of its friction cone, 5e-9 apart
    through the 1e-8 relaxation) the one with the smallest multiplier goes out.  Without the barrier's z / s conditioning the
    iteration converges to the stationary point the interior point was approaching, to rounding level -- which is what
    pins the directions the NLP leaves almost flat (curvature = the 1e-4 proximal weight: a KKT error of 1e-9 determines
    them to 1e-5 only).  Still dense, still autograd derivatives of the literal restatement; nothing of the C oracle or of
    the HIP solver.  Returns dict(w, lam, z, kkt, active, ok): `ok` CERTIFIES the answer whatever path led to it --
    stationarity and the working rows to 1e-10 (scaled as the solvers scale their KKT error), every other row satisfied
    to 1e-12, every multiplier non-negative."""
    w, lam = np.array(r["w"], dtype=np.float64), np.array(r["lam"], dtype=np.float64)
    z0, s0 = np.asarray(r["z"]), np.asarray(r["s"])
    # first guess.  Complementarity s z = mu: an active row has s = mu / z* -> 0; a row inactive by a hair keeps its slack and
    # z = mu / s, which at mu = 1e-10 is still larger than s: rows with a slack above s_max start outside the working set.
    active = (z0 > s0) & (s0 < s_max)
    zfull = np.where(active, z0, 0.0)
    nw = w.size
    kkt, ok = np.inf, False

    def derivatives(w, lam, za, active):
        wt = torch.tensor(w, requires_grad=True)
        gradf = torch.autograd.grad(nlp.cost(spec, par, wt, u_prox), wt)[0].numpy()
        Jc = torch.autograd.functional.jacobian(lambda v: nlp.equalities(spec, par, v), wt.detach(), vectorize=True).numpy()
        Jg = torch.autograd.functional.jacobian(lambda v: nlp.inequalities(spec, par, v), wt.detach(), vectorize=True).numpy()
        lam_t, z_t, idx = torch.tensor(lam), torch.tensor(za), torch.tensor(np.flatnonzero(active))

        def lagr(v):
            return (nlp.cost(spec, par, v, u_prox) + (lam_t * nlp.equalities(spec, par, v)).sum()
                    + (z_t * nlp.inequalities(spec, par, v)[idx]).sum())
        H = torch.autograd.functional.hessian(lagr, wt.detach(), vectorize=True).numpy()
        return gradf, Jc, Jg, H

    for rnd in range(rounds):
        za = zfull[active]
        for it in range(3):
            gradf, Jc, Jg, H = derivatives(w, lam, za, active)
            c = nlp.equalities(spec, par, torch.tensor(w)).numpy()
            g = nlp.inequalities(spec, par, torch.tensor(w)).numpy()
            A = np.vstack([Jc, Jg[active]])
            rd = gradf + Jc.T @ lam + Jg[active].T @ za
            rp = np.concatenate([c, g[active]])
            sd = max(100.0, (np.abs(lam).sum() + np.abs(za).sum()) / max(1, lam.size + za.size)) / 100.0
            kkt = max(np.abs(rd).max() / sd, np.abs(rp).max())
            if kkt < 1e-13:
                break
            K = np.block([[H, A.T], [A, np.zeros((A.shape[0], A.shape[0]))]])
            sol = np.linalg.lstsq(K, -np.concatenate([rd, rp]), rcond=1e-14)[0]
            w = w + sol[:nw]
            lam = lam + sol[nw:nw + lam.size]
            za = za + sol[nw + lam.size:]
        zfull = np.zeros_like(z0)
        zfull[active] = za
        g = nlp.inequalities(spec, par, torch.tensor(w)).numpy()
        c = nlp.equalities(spec, par, torch.tensor(w)).numpy()
        viol = np.flatnonzero(~active & (g > 1e-12))
        neg = np.flatnonzero(active & (zfull < -1e-10))
        stuck = np.flatnonzero(active & (np.abs(g) > 1e-11))
        if verbose:
            print(f"   refine round {rnd}: kkt {kkt:.2e} working rows {int(active.sum())}, violated outside {viol.size}, "
                  f"negative multipliers {neg.size}, rows that cannot hold {stuck.size}", flush=True)
        if viol.size == 0 and neg.size == 0 and stuck.size == 0 and kkt < 1e-10 and np.abs(c).max() < 1e-10:
            ok = True
            break
        if viol.size:                                           # most violated row first, one per vertex group at most
            active[viol[np.argmax(g[viol])]] = True
        elif neg.size:
            active[neg[np.argmin(zfull[neg])]] = False
        elif stuck.size:
            active[stuck[np.argmin(zfull[stuck])]] = False
        # (else: the working set stands, Newton goes on)
    z = zfull
    return dict(w=w, lam=lam, z=z, kkt=kkt, active=active, ok=ok)
