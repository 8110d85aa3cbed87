"""TEST INFRASTRUCTURE ONLY -- rigorous infeasibility certificate from the first stage of the NLP.

With x_0 fixed (reference code/centroidal_mpc_vertices.py:185), every constraint that involves only
(x_0, u_0, x_1) is a constraint on the 6*nv contact-force components of u_0 alone:

  * Lyapunov row i = 0 (:202-220): CoM[:,1] = c_0 + d v_0 is data, dCoM[:,1] = v_0 + d (g + V) is affine
    in V = sum_f gamma_f sum_j f_j / m, and the row is a quadratic in V with leading coefficient
    (k1 d^2 + d) |V|^2 > 0, i.e. a convex (ball) constraint;
  * contraction |hw_1|^2 <= |hw_0|^2 (:223-224) with hw_1 = hw_0 + d * sum gamma (p_j - c_0) x f_j:
    convex quadratic in the forces;
  * friction cone and unilateral rows (:236-254): linear.

The first-stage feasibility problem  min t  s.t.  g_i(u_0) <= t  is therefore CONVEX; a positive
optimum t* proves the whole NLP infeasible (the other stages can only remove points).  It is solved
here with scipy (SLSQP on the epigraph form, from several starts) and does not share code with the C
oracle or the HIP solver.  A non-positive t* proves nothing about the later stages.
"""
import numpy as np
import scipy.optimize as so

from . import nlp_reference as nlp


def _rows(spec, par):
    """Returns callables (lyap(F), contr(F)) and the linear rows A F <= 0 for the stance feet,
    F = stacked force components of the stance vertices (swing-foot forces do not enter any row)."""
    nv, m, d, k1, k2 = spec.nv, par['mass'], spec.delta, spec.k1, spec.k2
    x0 = par['x0']
    c0, v0, hw0, th0 = x0[0:3], x0[3:6], x0[6:9], x0[9:12]
    cr = par['com_ref'][:, 0]
    gam = (par['gl'][0], par['gr'][0])
    grav = np.array([0., 0., -spec.g])
    arms = []                                   # lever arm of every stance vertex
    for f, (iy, ip) in enumerate(((12, 13), (16, 17))):
        if gam[f] == 0:
            continue
        c, s = np.cos(x0[iy]), np.sin(x0[iy])
        for v in spec.verts:
            pv = np.array([c * v[0] - s * v[1], s * v[0] + c * v[1], v[2]]) + x0[ip:ip + 3]
            arms.append(pv - c0)
    arms = np.array(arms)
    nf = arms.shape[0]
    z1 = c0 + d * v0 - cr[0:3]

    def lyap(F):
        V = F.reshape(nf, 3).sum(0) / m
        z2 = k1 * z1 + v0 + d * (grav + V) - cr[3:6]
        un = -(k1 + k2) * z2 + k1 * k1 * z1 - grav + cr[6:9] - th0 / m
        return -k1 * z1 @ z1 - k2 * z2 @ z2 + z1 @ z2 + z2 @ (V - un) - spec.relax

    def contr(F):
        tau = np.cross(arms, F.reshape(nf, 3)).sum(0)
        h1 = hw0 + d * tau
        return h1 @ h1 - hw0 @ hw0 - spec.relax

    mu = par['mu']
    A = np.zeros((5 * nf, 3 * nf))
    for j in range(nf):
        for r, (ax, sg) in enumerate(((0, 1.), (0, -1.), (1, 1.), (1, -1.))):
            A[5 * j + r, 3 * j + ax] = sg
            A[5 * j + r, 3 * j + 2] = -mu
        A[5 * j + 4, 3 * j + 2] = -1.0
    return lyap, contr, A, nf


def min_violation(spec, par, starts=3, seed=0):
    """t* = min over u_0 of the largest first-stage row (scaled: Lyapunov row / 1, contraction / 1,
    cone rows / 100 N).  t* > 0 certifies infeasibility of the NLP.  Returns (t*, F*)."""
    lyap, contr, A, nf = _rows(spec, par)
    m = par['mass']
    n = 3 * nf
    rng = np.random.default_rng(seed)
    best = (np.inf, None)
    cons = [{'type': 'ineq', 'fun': lambda y: y[n] - lyap(y[:n])},
            {'type': 'ineq', 'fun': lambda y: y[n] - contr(y[:n])},
            {'type': 'ineq', 'fun': lambda y: y[n] - (A @ y[:n]) / 100.0 + spec.relax}]
    for s in range(starts):
        F0 = np.zeros(n)
        F0[2::3] = m * spec.g / nf * (1.0 + (0.3 * rng.uniform(-1, 1, nf) if s else 0.0))
        t0 = max(lyap(F0), contr(F0), (A @ F0).max() / 100.0) + 1e-3
        r = so.minimize(lambda y: y[n], np.concatenate([F0, [t0]]), jac=lambda y: np.eye(n + 1)[n],
                        constraints=cons, method='SLSQP', options={'maxiter': 500, 'ftol': 1e-14})
        F = r.x[:n]
        t = max(lyap(F), contr(F), (A @ F).max() / 100.0 - spec.relax)      # true value at the returned point
        if t < best[0]:
            best = (t, F)
        if best[0] <= 0:
            break
    return best


def certify(spec, rec, margin=1e-7):
    """True if the first stage alone proves the instance infeasible."""
    par = nlp.unpack_record(spec, rec)
    t, _ = min_violation(spec, par)
    return t > margin, t
