"""TEST INFRASTRUCTURE ONLY -- infeasibility certificate from the first stage of the NLP.

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

What SLSQP returns is only an UPPER bound on t* (the largest row at some point).  The certificate is
therefore taken from the dual side: for multipliers lam >= 0 with sum 1,
    q(lam) = min_F sum_i lam_i g_i(F)  <=  t*            (weak duality),
and every g_i is a convex quadratic or linear, so q(lam) is the minimum of ONE convex quadratic, computed in
closed form.  `certify` derives lam from the stationarity conditions at SLSQP's point (non-negative least
squares over the active rows) and reports infeasible only if this LOWER bound is positive -- whatever the
optimiser did, a stalled run can only lose the certificate, never forge one.  (The bound is stated over the
ball |F| <= F_MAX = 1e5 N, see `dual_lower_bound`.)
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


#: the dual bound is taken over contact forces with |F| <= F_MAX newtons (250x the robot's weight): the residual
#: slope of the Lagrangian along its flat directions (rounding in the fitted multipliers) is charged against it
F_MAX = 1e5


def _quadratic(fun, n):
    """(Q, c, d) of an exactly quadratic fun(F) = 1/2 F'QF + c'F + d, by differences (exact up to rounding)."""
    h = 10.0                                           # forces are O(100 N)
    d = fun(np.zeros(n))
    E = np.eye(n) * h
    fp = np.array([fun(E[i]) for i in range(n)]); fm = np.array([fun(-E[i]) for i in range(n)])
    c = (fp - fm) / (2 * h)
    Q = np.zeros((n, n))
    for i in range(n):
        Q[i, i] = (fp[i] + fm[i] - 2 * d) / (h * h)
        for j in range(i):
            Q[i, j] = Q[j, i] = (fun(E[i] + E[j]) - fp[i] - fp[j] + d) / (h * h)
    return Q, c, d


def dual_lower_bound(spec, par, F):
    """Lower bound on t* = min_F max_i g_i(F) from multipliers fitted at F (see the module docstring);
    -inf when the fitted multipliers give an unbounded Lagrangian."""
    lyap, contr, A, nf = _rows(spec, par)
    n = 3 * nf
    QL, cL, dL = _quadratic(lyap, n)
    QC, cC, dC = _quadratic(contr, n)
    Al, bl = A / 100.0, -spec.relax * np.ones(A.shape[0])
    vals = np.concatenate([[lyap(F), contr(F)], Al @ F + bl])
    grads = np.vstack([QL @ F + cL, QC @ F + cC, Al])
    act = np.where(vals >= vals.max() - 1e-6)[0]
    # directions in which neither quadratic row has curvature (forces that change neither the net force nor the
    # torque): the multipliers of the linear rows must cancel there, so those components are weighted up
    wq, Vq = np.linalg.eigh(QL + QC)
    Nn = Vq[:, wq <= 1e-9 * wq.max()]
    # lam >= 0, sum lam = 1, sum lam_i grad_i ~ 0 over the active rows
    scale = max(1.0, np.abs(grads[act]).max())
    M = np.vstack([grads[act].T / scale, 1e3 * (Nn.T @ grads[act].T) / scale, 1e3 * np.ones((1, act.size))])
    lam_a, _ = so.nnls(M, np.concatenate([np.zeros(n + Nn.shape[1]), [1e3]]))
    if lam_a.sum() <= 0:
        return -np.inf
    lam = np.zeros(vals.size); lam[act] = lam_a / lam_a.sum()
    Q = lam[0] * QL + lam[1] * QC
    c = lam[0] * cL + lam[1] * cC + Al.T @ lam[2:]
    d = lam[0] * dL + lam[1] * dC + bl @ lam[2:]
    w, V = np.linalg.eigh(Q)
    if w.min() < -1e-9 * max(1.0, w.max()):
        return -np.inf                                 # cannot happen for these rows; guards the differences
    pos = w > 1e-10 * max(1.0, w.max())
    cv = V.T @ c
    # along the flat directions the Lagrangian is linear with a slope of rounding size (the fitted multipliers are
    # not exact); it is bounded over the ball |F| <= F_MAX, which is what the bound is stated for
    flat = np.linalg.norm(cv[~pos]) * F_MAX
    return d - 0.5 * np.sum(cv[pos] ** 2 / w[pos]) - flat


def certify(spec, rec, margin=1e-7):
    """(certified, bound): True only if a dual LOWER bound on the first-stage violation exceeds `margin`,
    which proves the instance infeasible; `bound` is that lower bound (the primal value SLSQP reached is an
    upper bound and is used only to find multipliers)."""
    par = nlp.unpack_record(spec, rec)
    t, F = min_violation(spec, par)
    if not (t > margin) or F is None:
        return False, t
    lb = dual_lower_bound(spec, par, F)
    return bool(lb > margin), lb
