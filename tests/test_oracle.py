"""The C oracle against (a) the literal torch restatement of the reference NLP, (b) finite
differences of its own functions, (c) an independent dense solver, (d) known answers that follow
from the formulation.  CPU only."""
import numpy as np
import pytest
import torch

from conftest import oracle_spec, rel_inf
from cmpc_amd import workloads as wl
from cmpc_amd.problem import ProblemSpec
from oracle import nlp_reference as nlp


def _instance(N, idx=3, name="perturbed", scale=1.0):
    spec, rec = wl.make_workload(name, B=idx + 1, N=N, scale=scale)
    return spec, rec[idx]


def _rollout(lspec, par, U):
    """X from U through the LITERAL dynamics (reference :185-190)."""
    N = lspec.N
    X = torch.zeros((20, N + 1), dtype=torch.float64)
    X[:, 0] = torch.tensor(par['x0'])
    cr = torch.tensor(par['com_ref'])
    for i in range(N):
        X[:, i + 1] = X[:, i] + lspec.delta * nlp.dynamics(lspec, par, X[:, i], cr[:, i], par['gl'][i], par['gr'][i], U[:, i])
    return X


@pytest.mark.parametrize("nv,name", [(4, "perturbed"), (4, "randomized"), (8, "long_horizon")])
def test_function_values_match_literal_restatement(oracle, nv, name):
    N = 6
    spec, rec = _instance(N, name=name)
    lspec = nlp.Spec(N=N, nv=nv, k1=spec.k1, k2=spec.k2)
    par = nlp.unpack_record(lspec, rec)
    rng = np.random.default_rng(0)
    U = torch.tensor(rng.normal(size=(lspec.nu, N)) * 5.0)
    U[2:6 * nv:3, :] += 40.0
    X = _rollout(lspec, par, U)
    w = torch.cat((X.T.reshape(-1), U.T.reshape(-1)))
    cs = oracle_spec(oracle, spec)
    cost, defect, ineq, act = oracle.evaluate(cs, rec, w.numpy())
    assert abs(cost - nlp.cost(lspec, par, w).item()) <= 1e-10 * abs(cost)
    assert np.abs(defect).max() < 1e-12                       # the rollout satisfies the C dynamics too
    lit = list(nlp.inequalities(lspec, par, w).numpy())
    # the reference also writes CoM_z[0] <= 0.76 on the (fixed) initial state (:230, i = 0); with x_0
    # eliminated that row is a constant and the oracle does not carry it
    const_row = par['x0'][2] - lspec.cz_max - lspec.relax
    lit.pop(int(np.argmin(np.abs(np.array(lit) - const_row))))
    lit = np.sort(np.array(lit))
    mine = np.sort(ineq[act != 0])
    assert lit.shape == mine.shape                            # same number of live rows
    assert np.abs(lit - mine).max() <= 1e-9 * max(1.0, np.abs(lit).max())
    # off the dynamics manifold the (unsubstituted) equality rows must still agree
    w2 = w.clone(); w2[20:] += torch.tensor(rng.normal(size=w.numel() - 20) * 0.01)
    _, defect2, _, _ = oracle.evaluate(cs, rec, w2.numpy())
    eq = nlp.equalities(lspec, par, w2).numpy()[20:].reshape(N, 20)
    assert np.abs(defect2 + eq).max() < 1e-12


@pytest.mark.parametrize("k", [0, 1, 2, 5])
def test_analytic_derivatives_match_finite_differences(oracle, k):
    N = 6
    spec, rec = _instance(N, idx=1)
    cs = oracle_spec(oracle, spec)
    nu, nx, ni = spec.nu, 20 + 2 * spec.nv, 15 + 10 * spec.nv
    rng = np.random.default_rng(k)
    x = np.concatenate([rec[:20] + rng.normal(size=20) * 0.05, rng.normal(size=2 * spec.nv) * 10 + 40])
    u = rng.normal(size=nu) * 5
    u[2:6 * spec.nv:3] += 40
    lamn, zmul = rng.normal(size=nx) * 10, rng.uniform(0.1, 2.0, size=ni)
    x0n2 = 0.3

    def lagr_grad(z):
        r = oracle.stage(cs, rec, k, z[nu:], z[:nu], lamn, zmul, x0n2)
        g = r['grad'] + r['Jg'].T @ (zmul * r['act'])
        if k < N:
            g = g + r['G'].T @ lamn
        return g, r

    z0 = np.concatenate([u, x])
    g0, r0 = lagr_grad(z0)
    nz = nu + nx
    Gfd, Jfd, Hfd, cfd = np.zeros((nx, nz)), np.zeros((ni, nz)), np.zeros((nz, nz)), np.zeros(nz)
    for j in range(nz):
        h = 1e-6 * max(1.0, abs(z0[j]))
        zp, zm = z0.copy(), z0.copy()
        zp[j] += h; zm[j] -= h
        gp, rp = lagr_grad(zp); gm, rm = lagr_grad(zm)
        Gfd[:, j] = (rp['xn'] - rm['xn']) / (2 * h)
        Jfd[:, j] = (rp['g'] - rm['g']) / (2 * h)
        Hfd[:, j] = (gp - gm) / (2 * h)
        cfd[j] = (rp['cost'] - rm['cost']) / (2 * h)
    var = np.ones(nz, bool)
    if k == 0:
        var[nu:] = False                                     # x_0 is data: its columns are not modelled
    if k < N:
        assert np.abs(Gfd - r0['G']).max() < 1e-6 * max(1, np.abs(r0['G']).max())
    assert np.abs((Jfd - r0['Jg'])[:, var]).max() < 1e-5 * max(1, np.abs(r0['Jg']).max())
    assert np.abs((cfd - r0['grad'])[var]).max() < 1e-5 * max(1, np.abs(r0['grad']).max())
    H = r0['H']
    assert np.abs(H - H.T).max() < 1e-9 * max(1, np.abs(H).max())
    assert np.abs((Hfd - H)[np.ix_(var, var)]).max() < 2e-5 * max(1, np.abs(H).max())


def test_optimum_matches_independent_dense_solver(oracle):
    """Riccati/IPM oracle vs the autograd + dense-KKT solver on the literal NLP (no shared code)."""
    from oracle import ipm_dense
    N = 3
    spec, rec = _instance(N, idx=2)
    lspec = nlp.Spec(N=N, nv=4)
    par = nlp.unpack_record(lspec, rec)
    cs = oracle_spec(oracle, spec, tol=1e-10, max_iter=200)
    ref, st, it, kkt = oracle.solve(cs, rec)
    assert st == 0
    X0 = np.tile(par['x0'][:, None], (1, N + 1))
    U0 = ref[20 * (N + 1):].reshape(N, lspec.nu).T * 0.9      # any start: the optimum is unique
    w0 = np.concatenate([X0.T.reshape(-1), U0.T.reshape(-1)])
    res = ipm_dense.solve(lspec, par, w0=w0, tol=1e-10, linesearch=False, max_iter=120)
    assert res['status'] == 0
    assert rel_inf(res['w'], ref)[0] < 1e-7


def _standing_record(spec, both=True):
    N = spec.N
    rec = np.zeros(spec.nrec)
    rec[0:3] = [0., 0., 0.72]
    rec[13:16] = [0., 0.1, 0.]
    rec[17:20] = [0., -0.1, 0.]
    rec[20], rec[21] = wl.HRP4_MASS, 0.5
    rec[22], rec[23] = 1.0, 1.0 if both else 0.0
    st = rec[24:].reshape(N, 19)
    st[:, 0:3] = [0., 0., 0.72]
    st[:, 9:12] = [0., 0.1, 0.]
    st[:, 12:15] = [0., -0.1, 0.]
    st[:, 17], st[:, 18] = 1.0, 1.0 if both else 0.0
    return rec


def test_known_answer_symmetric_standing(oracle):
    """Double support, CoM on its reference between the feet, no momentum: eight equal vertical
    forces m g / 8, nothing moves (symmetry of the formulation)."""
    spec = ProblemSpec(N=8)
    rec = _standing_record(spec)
    sol, st, it, kkt = oracle.solve(oracle_spec(oracle, spec, tol=1e-10, max_iter=200), rec)
    assert st == 0
    N, nu = spec.N, spec.nu
    X = sol[:20 * (N + 1)].reshape(N + 1, 20)
    U = sol[20 * (N + 1):].reshape(N, nu)
    F = U[:, :24].reshape(N, 8, 3)
    fz = wl.HRP4_MASS * 9.81 / 8
    assert np.abs(F[..., 2] - fz).max() < 1e-3 * fz
    assert np.abs(F[..., :2]).max() < 1e-6
    assert np.abs(X[:, 0:3] - [0., 0., 0.72]).max() < 1e-5 and np.abs(X[:, 3:12]).max() < 1e-4   # proximal bias ~1e-6
    assert np.abs(U[:, 24:]).max() == 0.0                     # stance-foot velocities stay at the proximal centre


def test_known_answer_single_support_swing_forces_vanish(oracle):
    spec = ProblemSpec(N=6)
    rec = _standing_record(spec, both=False)
    rec[1] = 0.09                                            # CoM over the stance (left) foot
    rec[24:].reshape(6, 19)[:, 1] = 0.09
    sol, st, it, kkt = oracle.solve(oracle_spec(oracle, spec, tol=1e-10, max_iter=200), rec)
    assert st == 0
    U = sol[20 * 7:].reshape(6, spec.nu)
    assert np.abs(U[:, 12:24]).max() < 1e-9                   # right (swing) foot carries nothing
    assert abs(U[:, 2:12:3].sum(axis=1).mean() - wl.HRP4_MASS * 9.81) < 0.5


def test_batch_api_equals_single_and_is_thread_independent(oracle):
    spec, rec = wl.make_workload("perturbed", B=12, N=8)
    cs = oracle_spec(oracle, spec)
    a = oracle.solve_batch(cs, rec, nthreads=1)
    b = oracle.solve_batch(cs, rec, nthreads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    one = oracle.solve(cs, rec[5])
    assert np.array_equal(one[0], a[0][5]) and one[2] == a[2][5]


def test_status_codes(oracle):
    spec, rec = wl.make_workload("perturbed", B=4, N=8)
    cs = oracle_spec(oracle, spec, max_iter=3)
    _, st, it, _ = oracle.solve_batch(cs, rec)
    assert (st == 1).all() and (it == 3).all()                # CMPC_MAX_ITER
    bad = rec[0].copy()
    bad[2] = 0.9                                              # CoM above the 0.76 m height bound: infeasible
    _, st, it, kkt = oracle.solve(oracle_spec(oracle, spec, max_iter=150), bad)
    assert st != 0


def test_warm_start_is_initial_guess_and_proximal_centre(oracle):
    spec, rec = wl.make_workload("perturbed", B=2, N=8, scale=0.5)
    cs = oracle_spec(oracle, spec, tol=1e-10, max_iter=200)
    cold, st, it_cold, _ = oracle.solve(cs, rec[0])
    assert st == 0
    warm, st, it_warm, _ = oracle.solve(cs, rec[0], warm=cold)
    assert st == 0
    # re-solving from the optimum with the optimum as proximal centre stays (almost) put ...
    assert rel_inf(warm, cold)[0] < 1e-3
    # ... but is not identical: the proximal term no longer pulls the inputs towards zero
    assert np.abs(warm - cold).max() > 0


def test_locally_infeasible_verdicts_are_certified_by_the_first_stage(oracle):
    """Classification of the non-converged instances of the bench workload (BASELINE config 4, random pushes up
    to 100 N): every instance the solver gives up on as locally infeasible (status 2) is PROVEN infeasible by the
    convex first-stage problem of oracle/stage0_feasibility.py (scipy SLSQP on the epigraph form; shares no code
    with the solvers), and no instance the solver solves is."""
    from cmpc_amd import workloads as wl
    from oracle import nlp_reference as nlp, stage0_feasibility as s0
    spec, rec = wl.make_workload("randomized", B=384)
    out, st, it, kkt = oracle.solve_batch(oracle_spec(oracle, spec), rec)
    ns = nlp.Spec(N=spec.N)
    bad = np.where(st == 2)[0]
    assert 4 <= len(bad) <= 30 and (st == 1).sum() == 0           # ~3 % of the draws, none at the iteration cap
    for i in bad:
        cert, t = s0.certify(ns, rec[i])
        assert cert, (i, t)
    good = np.where(np.isin(st, (0, 3)))[0][:40]
    assert not any(s0.certify(ns, rec[i])[0] for i in good)
    assert np.isin(st, (0, 3)).sum() + len(bad) == len(st)         # feasible => solved: 100 % of the feasible draws


@pytest.mark.parametrize("over", [dict(acc_tol=1e-12), dict(tol=1e-3)])
def test_status_3_always_carries_a_written_solution(oracle, over):
    """Round-2 advice: with acc_tol < ACC_FACTOR * tol the acceptable-level counter used to end runs whose iterates
    had never been saved (status 3, kkt = inf, `out` untouched).  Every iterate the counter counts is saved now."""
    from cmpc_amd import workloads as wl
    spec, rec = wl.make_workload("randomized", B=192, N=10)
    cs = oracle.default_spec(N=10, nv=4, tol=1e-8, max_iter=100)
    for k, v in over.items():
        setattr(cs, k, v)
    got, st, it, kkt = oracle.solve_batch(cs, rec)          # `out` starts as zeros
    acc = st == 3
    assert np.isfinite(kkt[acc]).all()
    assert np.isfinite(got[acc]).all() and (np.abs(got[acc]).max(axis=1) > 0).all()
