"""Solver state across closed-loop ticks (cmpc_solve_batch_state): the interior point method resumes from the
central-path point the previous tick's solve passed through at its last barrier value >= 1e-2 instead of restarting at
mu = 100 from the boundary solution.  CPU tier: the C oracle on a stretch of the flat-ground
walk, and the device source (host emulation) against the oracle tick by tick."""
import ctypes

import numpy as np
import pytest

import build as _b
from conftest import rel_inf
from cmpc_amd import workloads as wl
from cmpc_amd.problem import ProblemSpec

HW = np.loadtxt(__file__.rsplit("/", 1)[0] + "/golden/measured_hw_cuhw.txt")


def _loop(solve, N, t0, ticks, use_state):
    """Closed loop with perfect tracking (walk.WalkHarness without the drop-in class): returns per-tick records,
    solutions, iteration counts."""
    spec, sc = ProblemSpec(N=N), wl.scene()
    com, dcom = (a[0] for a in sc.nominal_state(np.array([t0])))
    theta, warm, state = np.zeros(3), None, None
    recs, sols, its, sts = [], [], [], []
    for t in range(t0, t0 + ticks):
        rec = sc.build_records(spec, np.array([t]), com[None], dcom[None], HW[t][None], theta[None], np.zeros(1),
                               np.zeros(1), np.full(1, wl.HRP4_MASS), np.full(1, 0.5))
        sol, state_out, st, it = solve(rec, warm, state if use_state else None)
        assert st[0] in (0, 3), (t, st)
        X = sol[0, :20 * (N + 1)].reshape(N + 1, 20)
        com, dcom, theta = X[1, 0:3].copy(), X[1, 3:6].copy(), X[1, 9:12].copy()
        warm, state = sol, state_out
        recs.append(rec[0]); sols.append(sol[0]); its.append(int(it[0])); sts.append(int(st[0]))
    _loop.last_status = np.array(sts)           # (per-tick outcome of the loop just run: 0 converged, 3 acceptable)
    return np.array(recs), np.array(sols), np.array(its)


def test_state_resume_saves_iterations_and_keeps_the_solution(oracle):
    N = 10
    cs = oracle.default_spec(N=N, nv=4, tol=1e-8, max_iter=100)

    def solve(rec, warm, state):
        out, so, st, it, _ = oracle.solve_batch_state(cs, rec, warm=warm, state=state)
        return out, so, st, it

    _, sol_p, it_p = _loop(solve, N, 230, 60, use_state=False)      # primal warm start only (round 2)
    recs, sol_s, it_s = _loop(solve, N, 230, 60, use_state=True)    # resumed from the solver state
    assert it_s[1:].mean() <= 0.8 * it_p[1:].mean(), (it_s.mean(), it_p.mean())
    assert it_s[1:].mean() <= 12.0
    # same closed loop: the CoM paths of the two loops stay together (every solve returns its optimum to ~1e-8; along
    # the flat directions of the problem -- force split, swing-foot timing -- the two loops drift apart by up to a few
    # 1e-4 over 60 ticks of feedback, which is why the comparison is on the quantity the loop feeds back)
    assert np.abs(sol_s[:, 20:26] - sol_p[:, 20:26]).max() < 1e-6
    # and tick by tick, from identical inputs, the resumed solve returns the plain solve's optimum: the same point to
    # 1e-4 rel-inf (north star), or the same optimum seen from two points of its flat valley (equal objective, no
    # dynamics defect: the criterion of tests/test_gpu_parity.py::_explain_outliers)
    for i in (5, 17, 31, 44, 59):
        warm = sol_s[i - 1][None]
        a = oracle.solve_batch(cs, recs[i][None], warm=warm)[0]
        up = sol_s[i - 1][20 * (N + 1):]
        f_a, d_a, _, _ = oracle.evaluate(cs, recs[i], a[0], uprox=up)
        f_s, d_s, _, _ = oracle.evaluate(cs, recs[i], sol_s[i], uprox=up)
        assert rel_inf(sol_s[i][None], a)[0] < 1e-3
        assert abs(f_a - f_s) <= 1e-8 * abs(f_a) and np.abs(d_s).max() < 1e-7


def test_empty_state_is_the_plain_entry_point(oracle):
    spec, rec = wl.make_workload("perturbed", B=6, N=10)
    cs = oracle.default_spec(N=10, nv=4, tol=1e-8, max_iter=100)
    a, st_a, it_a, _ = oracle.solve_batch(cs, rec)
    b, state, st_b, it_b, _ = oracle.solve_batch_state(cs, rec, state=np.zeros((6, oracle.nstate(cs))))
    assert np.array_equal(a, b) and np.array_equal(it_a, it_b)
    ok = np.isin(st_b, (0, 3))
    mu = state[:, -8 - 2 * (10 + 1)]                                # barrier word (the contact flags of the N+1 nodes follow)
    assert ((mu[ok] >= 1e-7) & (mu[ok] <= 100.0)).all()            # a state was written for every solved instance
    # the word after it: what the solve took (the launch of the next tick queues its instances by it)
    assert np.array_equal(state[:, -7 - 2 * (10 + 1)], it_b.astype(np.float64))


@pytest.fixture(scope="module")
def emu():
    return ctypes.CDLL(_b.build_emu())


def test_device_source_resumes_like_the_oracle(emu, oracle):
    N = 10
    cs = oracle.default_spec(N=N, nv=4, tol=1e-8, max_iter=100)
    nsol, nst = oracle.nsol(cs), oracle.nstate(cs)
    p = lambda a: None if a is None else np.ascontiguousarray(a).ctypes.data_as(ctypes.c_void_p)

    def emu_solve(rec, warm, state):
        out, so = np.zeros((1, nsol)), np.zeros((1, nst))
        st, it, kk = np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1)
        rec = np.ascontiguousarray(rec)
        w = None if warm is None else np.ascontiguousarray(warm)
        s_in = None if state is None else np.ascontiguousarray(state)
        assert emu.cmpc_emu_solve_batch_state(ctypes.byref(cs), 1, p(rec), p(w), p(s_in), p(out), p(so), p(st), p(it), p(kk)) == 0
        return out, so, st, it

    def ora_solve(rec, warm, state):
        out, so, st, it, _ = oracle.solve_batch_state(cs, rec, warm=warm, state=state)
        return out, so, st, it

    _, sol_e, it_e = _loop(emu_solve, N, 255, 6, use_state=True)     # touch-down inside the horizon
    st_e = _loop.last_status
    _, sol_o, it_o = _loop(ora_solve, N, 255, 6, use_state=True)
    st_o = _loop.last_status
    # (end game: rounding order.  Where one of the two ended at the acceptable level its count includes the progress
    # watch's window of 12 iterations: only ticks both converged on are compared)
    tight = (st_e == 0) & (st_o == 0)
    assert tight.sum() >= 4 and np.abs(it_e - it_o)[tight].max() <= 2 and abs(it_e[tight].sum() - it_o[tight].sum()) <= 3, (it_e, it_o)
    assert rel_inf(sol_e, sol_o).max() < 1e-6
    assert it_e[1:].mean() < it_e[0]


def test_stale_state_is_given_up_early_and_the_two_attempts_share_the_budget(oracle):
    """A state written for a tick far away (another contact phase, pushed CoM) does not fit this tick's problem: the
    resumed attempt is abandoned once it has stayed at the state's barrier value for twenty iterations (or ended
    without a usable point) and the plain solve follows with what is left of max_iter -- `iters` counts both and never
    exceeds max_iter + 1 (include/cmpc.h).  Round 3 let the resumed attempt crawl to the cap first."""
    sc = wl.scene()
    spec = ProblemSpec(N=10)

    def rec_at(t, push=0.0):
        com, dcom = sc.nominal_state(np.array([t]))
        return sc.build_records(spec, np.array([t]), com, dcom + push, HW[t][None], np.zeros((1, 3)), np.zeros(1),
                                np.zeros(1), np.full(1, wl.HRP4_MASS), np.full(1, 0.5))
    for max_iter in (100, 30):
        cs = oracle.default_spec(N=10, nv=4, tol=1e-8, max_iter=max_iter)
        a, s1, st_a, _, _ = oracle.solve_batch_state(cs, rec_at(120))
        assert st_a[0] == 0
        for t, push in ((640, 0.08), (655, -0.1), (1130, 0.05)):
            far = rec_at(t, push)
            b, _, st_b, it_b, _ = oracle.solve_batch_state(cs, far, warm=a, state=s1)
            c, st_c, it_c, _ = oracle.solve_batch(cs, far, warm=a)
            assert it_b[0] <= max_iter + 1, (t, it_b, max_iter)
            if max_iter == 100:
                assert st_b[0] in (0, 3) and st_c[0] in (0, 3)
                assert it_b[0] <= it_c[0] + 22, (t, it_b, it_c)
                assert rel_inf(b, c)[0] < 1e-3
