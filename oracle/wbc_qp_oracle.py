"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, fp64) of the reference's whole-body inverse-dynamics QP
 and of the interior point method the HIP kernel csrc/wbc_qp.hip runs on it.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/.

PARITY UNPINNED: the reference solves this QP with CasADi's conic interface + OSQP (code/utils.py:40-92, max_iter 1000,
OSQP's default tolerances 1e-3), neither of which is available here, on matrices that come out of DART (code/
inverse_dynamics.py:46-66, :107-111) -- also unavailable -- and it holds no recorded QP data.  What can be pinned is
pinned: the QP is convex, so a point that satisfies its KKT conditions IS its solution whatever algorithm found it;
`kkt_full` checks the conditions of the reference's own 72-variable statement literally (tests/test_wbc_qp.py), which an
OSQP answer satisfies to OSQP's tolerance.

Reference statement (code/inverse_dynamics.py:92-134), n = dofs = 30 for HRP-4:
    x = [qdd (n), tau (n), f_c (12)]
    min  1/2 qdd' Hq qdd + Fq' qdd + 1/2 1e-6 |f_c|^2          Hq, Fq: the task sums of :92-103
    s.t. M qdd - S tau - Jc' f_c = -h                           S = blockdiag(0_6, I_{n-6})  (:107-111)
         [A 0; 0 A] f_c <= 0                                    A: 8 x 6 CoP / friction rows (:113-129), d = foot_size/2
tau[0:6] appears nowhere (zero cost, zero column of S): the statement leaves it free and the reference returns
tau[6:] only.  The actuated rows of the dynamics define tau[6:] = M_a qdd + h_a - Jc_a' f_c and constrain nothing, so
the problem in (qdd, f_c) with the six floating-base rows  M_b qdd - Jc_b' f_c = -h_b  is the same problem;
`solve` works on that 42-variable form, `kkt_full` checks the result against the 72-variable one.
"""
import numpy as np

ND, NB, NC, NI = 30, 6, 12, 16          # dofs, floating-base dofs, contact wrench dims, inequality rows
F_REG = 1e-6                             # :105
TOL, MAX_ITER, MU0 = 1e-9, 60, 10.0


def wrench_rows(d, mu):
    """The 8 x 6 block of :116-123 (a wrench is [moment(3), force(3)]); rows act as A w <= 0."""
    return np.array([[1, 0, 0, 0, 0, -d], [-1, 0, 0, 0, 0, -d], [0, 1, 0, 0, 0, -d], [0, -1, 0, 0, 0, -d],
                     [0, 0, 0, 1, 0, -mu], [0, 0, 0, -1, 0, -mu], [0, 0, 0, 0, 1, -mu], [0, 0, 0, 0, -1, -mu]], dtype=np.float64)


def ineq_matrix(d, mu):
    A = np.zeros((NI, NC))
    A[0:8, 0:6] = wrench_rows(d, mu)
    A[8:16, 6:12] = wrench_rows(d, mu)
    return A


def solve(Hq, Fq, M, h, Jc, d, mu_f, tol=TOL, max_iter=MAX_ITER, verbose=False):
    """One QP.  Hq (30,30) Fq (30,) M (30,30) h (30,) Jc (12,30) -> dict(qdd, f, tau (30,), nu, s, z, status, iters).
    Primal-dual interior point, monotone barrier schedule, fraction-to-the-boundary steps -- the algorithm of the
    centroidal MPC solver (oracle/cmpc_oracle.c) on a convex problem (no inertia correction needed)."""
    n = ND + NC
    H = np.zeros((n, n)); H[:ND, :ND] = Hq; H[ND:, ND:] = F_REG * np.eye(NC)
    F = np.concatenate([Fq, np.zeros(NC)])
    Ae = np.hstack([M[:NB, :], -Jc[:, :NB].T]); be = -h[:NB]
    Ai = np.hstack([np.zeros((NI, ND)), ineq_matrix(d, mu_f)])
    x = np.zeros(n); nu = np.zeros(NB)
    s = np.maximum(-(Ai @ x), 1.0); mu = MU0; z = mu / s
    status, it = 1, 0
    for it in range(max_iter + 1):
        rd = H @ x + F + Ae.T @ nu + Ai.T @ z
        rp = Ae @ x - be
        rg = Ai @ x + s
        sd = max(100.0, (np.abs(nu).sum() + np.abs(z).sum()) / (NB + NI)) / 100.0
        e_d, e_p, e_c = np.abs(rd).max() / sd, max(np.abs(rp).max(), np.abs(rg).max()), np.abs(s * z).max() / sd
        e_cmu = np.abs(s * z - mu).max() / sd
        kkt = max(e_d, e_p, e_c)
        if verbose:
            print(f"it {it:2d} d={e_d:.2e} p={e_p:.2e} c={e_c:.2e} mu={mu:.1e}")
        if not np.isfinite(kkt):                # NaN / Inf input: numerical failure (the kernel: status 2, zero outputs)
            status = 2
            break
        if kkt <= tol:
            status = 0
            break
        if it == max_iter:
            break
        while mu > tol / 10 and max(e_d, e_p, e_cmu) < 10 * mu:
            mu = max(tol / 10, min(0.1 * mu, mu ** 1.5))
        sig = z / s
        K = np.zeros((n + NB, n + NB))
        K[:n, :n] = H + Ai.T @ (sig[:, None] * Ai)
        K[n:, :n] = Ae; K[:n, n:] = Ae.T
        rhs = np.concatenate([-(H @ x + F) - Ai.T @ (mu / s + sig * rg), -rp])
        sol = ldl_solve(K, rhs, n)
        dx, nu_new = sol[:n], sol[n:]
        ds = -rg - Ai @ dx
        dz = (mu - s * z - z * ds) / s
        tau_ = max(0.99, 1 - mu)
        ap = min(1.0, (tau_ * s[ds < 0] / -ds[ds < 0]).min()) if (ds < 0).any() else 1.0
        ad = min(1.0, (tau_ * z[dz < 0] / -dz[dz < 0]).min()) if (dz < 0).any() else 1.0
        x = x + ap * dx; s = s + ap * ds; nu = nu + ap * (nu_new - nu); z = z + ad * dz
    if status != 0:                             # the reference's QPSolver.solve returns zeros on failure (code/utils.py:85-92)
        x = np.zeros(n)
    qdd, f = x[:ND], x[ND:]
    tau = np.zeros(ND)
    if status == 0:
        tau[NB:] = M[NB:, :] @ qdd + h[NB:] - Jc[:, NB:].T @ f
    return dict(qdd=qdd, f=f, tau=tau, nu=nu, s=s, z=z, status=status, iters=it, kkt=kkt)


def ldl_solve(K, rhs, n_pos):
    """K = L D L' without pivoting (quasi-definite: the first n_pos pivots positive, the rest negative) -- the
    factorisation the kernel runs, restated densely."""
    m = K.shape[0]
    L = np.tril(K).astype(np.float64).copy()
    dvec = np.zeros(m)
    for j in range(m):
        dvec[j] = L[j, j]
        assert (dvec[j] > 0) == (j < n_pos), "wrong inertia"
        col = L[j + 1:, j].copy()
        L[j + 1:, j] = col / dvec[j]
        L[j, j] = 1.0
        for k in range(j + 1, m):
            L[k:, k] -= L[k:, j] * col[k - j - 1]
    y = rhs.astype(np.float64).copy()
    for j in range(m):
        y[j + 1:] -= L[j + 1:, j] * y[j]
    y /= dvec
    for j in range(m - 1, -1, -1):
        y[:j] -= L[j, :j] * y[j]
    return y


def solve_batch(Hq, Fq, M, h, Jc, d, mu_f, **kw):
    B = Hq.shape[0]
    out = [solve(Hq[b], Fq[b], M[b], h[b], Jc[b], d, mu_f, **kw) for b in range(B)]
    return {k: np.array([o[k] for o in out]) for k in out[0]}


def kkt_full(Hq, Fq, M, h, Jc, d, mu_f, qdd, tau, f):
    """KKT residuals of the reference's 72-variable QP (module docstring) at (qdd, tau, f): least-squares multipliers
    on the active inequality rows.  Returns dict(stationarity, equality, ineq_violation, comp)."""
    n = ND
    nv = 2 * n + NC
    H = np.zeros((nv, nv)); H[:n, :n] = Hq; H[2 * n:, 2 * n:] = F_REG * np.eye(NC)
    F = np.concatenate([Fq, np.zeros(n + NC)])
    S = np.zeros((n, n)); S[NB:, NB:] = np.eye(n - NB)
    Aeq = np.hstack([M, -S, -Jc.T]); beq = -h
    Ain = np.hstack([np.zeros((NI, 2 * n)), ineq_matrix(d, mu_f)])
    x = np.concatenate([qdd, tau, f])
    g = H @ x + F
    gi = Ain @ x
    act = gi > -1e-7 * max(1.0, np.abs(f).max())
    # multipliers: g + Aeq' lam + Ain_act' z = 0, z >= 0 (non-negative least squares on z, free lam)
    import scipy.optimize as so
    A = np.hstack([Aeq.T, -Aeq.T, Ain[act].T])
    sol, res = so.nnls(A, -g)
    z = sol[2 * n:]
    return dict(stationarity=res / max(1.0, np.abs(g).max()), equality=np.abs(Aeq @ x - beq).max(),
                ineq_violation=max(gi.max(), 0.0), n_active=int(act.sum()), z_min=float(z.min()) if z.size else 0.0)
