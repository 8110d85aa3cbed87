"""BASELINE config 1, the flat-ground walk (code/simulation.py:193-212 driving
code/centroidal_mpc_vertices.py:480-683), on the CPU tier: the drop-in class's host logic (parameter
build, unpack, plan write-back :656-675, contact bookkeeping) driven through ``walk.WalkHarness`` with the
C oracle standing in for the HIP solver.  The same harness runs the real HIP solver in
tests/test_gpu_walk.py."""
import os

import numpy as np
import pytest

from conftest import oracle_spec
from cmpc_amd import workloads as wl
from cmpc_amd.centroidal_mpc_vertices import centroidal_mpc
from cmpc_amd.foot_trajectory_generator import FootTrajectoryGenerator
from cmpc_amd.footstep_planner_vertices import FootstepPlanner
from cmpc_amd.problem import ProblemSpec
from cmpc_amd.walk import WalkHarness

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def measured_hw():
    """Angular momentum about the CoM recorded by the reference in its own simulator run under this
    controller (original_code/cuhw.txt): 1962 ticks x 3."""
    return np.loadtxt(os.path.join(GOLD, "measured_hw_cuhw.txt"))


def make_oracle_backed(oracle):
    class OracleBackedMPC(centroidal_mpc):
        """TEST ONLY: the drop-in class with opt.solve() answered by the C oracle."""

        def _make_solver(self, device):
            self._cs = oracle_spec(oracle, self.spec)
            return None

        def _solve_record(self, rec):
            out, st, it, kkt = oracle.solve(self._cs, rec, warm=self._warm)
            if st in (0, 3):
                self._warm = out
            return out, st, it, kkt
    return OracleBackedMPC


def build_walk(mpc_cls, N=10, hw=None, mpc_rate=1, **kw):
    sc = wl.scene()
    params = wl.default_params(N=N, mpc_rate=mpc_rate)
    planner = FootstepPlanner(wl.VREF, wl.LFOOT0, wl.RFOOT0, params)      # a fresh plan: the MPC rewrites it
    ftg = FootTrajectoryGenerator(sc.initial, planner, params)
    mpc = mpc_cls(sc.initial, planner, params, sc.com_ref, None, None, **kw)
    return WalkHarness(mpc, planner, ftg, params, sc.initial, hw_measured=hw), planner, ftg, mpc, sc


def check_walk_log(log, planner, nominal_plan, sc, N, ticks):
    """Assertions shared by the CPU (oracle) and GPU (HIP) walks."""
    assert np.isin(log['status'], (0, 3)).all()                       # every tick solved
    # returned contact = phase / support foot of the plan (:679-683)
    for t, c in zip(log['t'], log['contact']):
        ph = planner.get_phase_at_time(int(t))
        assert c == ('ds' if ph == 'ds' else planner.plan[planner.get_step_index_at_time(int(t))]['foot_id'])
    # write-back (:656-675): once per step, at the first single-support tick whose horizon end is in double
    # support; steps are 100 ticks (70 ss + 30 ds) from t = 200, so at t = 261, 361, ...
    fired = log['t'][log['counter'] == 1]
    want = np.array([s for s in range(261, ticks, 100)])
    assert np.array_equal(fired, want)
    for t in fired:
        idx = planner.get_step_index_at_time(int(t))
        new = log['mpc_new_contact'][int(t)]
        assert np.array_equal(planner.plan[idx + 1]['pos'], new)       # the plan holds the MPC's landing point
        d = new - nominal_plan[idx + 1]
        assert np.abs(d[0]) <= 0.01 + 1e-6 and np.abs(d[1]) <= 0.005 + 1e-6 and np.abs(d[2]) <= 5e-5 + 1e-6   # box :259-271
        assert np.abs(d).max() > 0.0                                    # the solver's landing point, not the nominal entry
    # the CoM follows its reference through lift-off, single support, touch-down
    ref = sc.com_tab[log['t'] + 1, 0:3]
    err = np.abs(log['com'] - ref)
    assert err[:, 0].max() < 0.03 and err[:, 1].max() < 0.03 and err[:, 2].max() < 0.01
    assert np.abs(log['hw_des']).max() < 2.0                            # |h_w| of the reference's own runs <= 1.9


def test_walk_records_sampled_open_loop(oracle):
    """Nominal-trajectory records (workloads.walk_records) over three full steps, measured momentum from the
    reference's recording: every sampled tick is solvable, double support, early and late single support,
    lift-off and touch-down alike."""
    hw = measured_hw()
    for N in (10, 20):
        spec = ProblemSpec(N=N)
        ticks = np.arange(0, 620, 7)
        rec = wl.walk_records(spec, ticks, hw=hw[ticks])
        out, st, it, kkt = oracle.solve_batch(oracle_spec(oracle, spec), rec)
        assert np.isin(st, (0, 3)).all(), (ticks[~np.isin(st, (0, 3))], st)
        X = out[:, :20 * (N + 1)].reshape(-1, N + 1, 20)
        assert np.abs(X[:, 1, 0:3] - rec[:, 24:27]).max() < 0.03        # x_1 stays on the reference


def test_walk_records_zero_momentum_late_single_support_is_infeasible(oracle):
    """Known property of the reference formulation: with h_w(0) = 0 exactly the contraction
    row |h_w(1)| <= |h_w(0)| (:223-224) forbids any torque in the first step, and in late single support
    the Lyapunov rows (:202-220) cannot be met with a force through the CoM.  Certified by the first-stage
    convex feasibility problem (oracle/stage0_feasibility.py), which shares no code with the solvers."""
    from oracle import nlp_reference as nlp, stage0_feasibility as s0
    spec = ProblemSpec(N=10)
    ticks = np.array([255, 262, 355, 362, 455])
    rec = wl.walk_records(spec, ticks)
    _, st, _, _ = oracle.solve_batch(oracle_spec(oracle, spec), rec)
    ns = nlp.Spec(N=10)
    cert = [s0.certify(ns, r)[0] for r in rec]
    assert all(cert) and (st == 2).all()
    hw = measured_hw()
    rec = wl.walk_records(spec, ticks, hw=hw[ticks])
    assert not any(s0.certify(ns, r)[0] for r in rec)


def test_closed_loop_walk_with_plan_write_back(oracle):
    """Three and a half steps of the flat-ground walk in closed loop, N = 10 as shipped
    (simulation.py:34), measured momentum from the reference's recording."""
    ticks = 580
    walk, planner, ftg, mpc, sc = build_walk(make_oracle_backed(oracle), N=10, hw=measured_hw())
    nominal = [p['pos'].copy() for p in planner.plan]
    assert ftg.plan is planner.plan                                     # foot_trajectory_generator.py:9
    log = walk.run(ticks)
    check_walk_log(log, planner, nominal, sc, 10, ticks)
    # the swing-foot generator sees the rewritten plan: its touch-down pose is the MPC's landing point
    idx = planner.get_step_index_at_time(561)
    swing = 'rfoot' if planner.plan[idx]['foot_id'] == 'lfoot' else 'lfoot'
    touch = ftg.generate_feet_trajectories_at_time(planner.get_start_time(idx) + 75)
    assert np.array_equal(touch[swing]['pos'][3:6], planner.plan[idx + 1]['pos'])
    assert mpc.update_contact_flag == 0 or planner.get_phase_at_time(ticks - 1) == 'ss'
    # x0 of later ticks takes the foot positions from the rewritten plan (:493-509)
    rec = mpc.last_record
    feet0 = {tuple(np.round(rec[13:16], 12)), tuple(np.round(rec[17:20], 12))}
    rewritten = {tuple(np.round(p['pos'], 12)) for p, n in zip(planner.plan, nominal) if not np.array_equal(p['pos'], n)}
    assert feet0 & rewritten


def test_drop_in_outputs_follow_reference_formulas(oracle):
    """hw.dot (:283-284, :643) and com.acc (:633-636) of the returned dict, and the aliasing contract."""
    walk, planner, ftg, mpc, sc = build_walk(make_oracle_backed(oracle), N=10, hw=measured_hw())
    for _ in range(215):
        state, contact = walk.step()
    assert state is mpc.model_state
    N, nu, m, g = 10, 32, mpc.mass, mpc.g
    rec = mpc.last_record
    X, U = mpc.x_collect, None
    u0 = mpc.u
    gl, gr = rec[24 + 17], rec[24 + 18]
    F = u0[:24].reshape(8, 3)
    acc = (gl * F[:4].sum(0) + gr * F[4:].sum(0)) / m + np.array([0, 0, -g])
    assert np.allclose(state['com']['acc'], acc, rtol=0, atol=1e-12) and state['com']['acc'].shape == (3,)
    # hw.dot = (0.01 * f(x_0, u_0))[6:9] * delta * mpc_rate with f from the literal restatement of
    # centroidal_dynamic (oracle/nlp_reference.py, :371-461)
    import torch
    from oracle import nlp_reference as nlp
    ns = nlp.Spec(N=N)
    par = nlp.unpack_record(ns, rec)
    f = nlp.dynamics(ns, par, torch.tensor(X[:, 0]), torch.tensor(par['com_ref'][:, 0]), gl, gr, torch.tensor(u0)).numpy()
    assert np.allclose(state['hw']['dot'], 0.01 * f[6:9] * mpc.delta * mpc.mpc_rate, rtol=1e-12, atol=1e-14)
    assert np.array_equal(state['com']['pos'], X[0:3, 1]) and np.array_equal(state['hw']['val'], X[6:9, 1])
    assert state['ang_contact_left']['val'] == X[12, 1] and np.array_equal(state['pos_contact_right']['val'], X[17:20, 1])


def run_rate_10_walk(mpc_cls, ticks):
    """mpc_rate = 10 (simulation.py:203: one solve every tenth tick; delta = 0.1 s, k1, k2 = 5, 0.2, no force-rate cost,
    :11, :27-31, :339-341): the horizon spans a second, i.e. a whole step."""
    walk, planner, ftg, mpc, sc = build_walk(mpc_cls, N=10, hw=measured_hw(), mpc_rate=10)
    assert (mpc.spec.delta, mpc.spec.k1, mpc.spec.k2, mpc.spec.w_rate) == (0.1, 5.0, 0.2, 0.0)
    log = walk.run(ticks)
    assert np.isin(log['status'], (0, 3)).all()
    solved = log['t'] % 10 == 0
    # x_1 is the state one MPC step (ten ticks) ahead
    ref = sc.com_tab[log['t'][solved] + 10, 0:3]
    assert np.abs(log['com'][solved] - ref).max() < 0.06
    return log


def test_closed_loop_walk_rate_10(oracle):
    run_rate_10_walk(make_oracle_backed(oracle), 400)


def test_closed_loop_ticks_through_the_emulated_kernel(oracle):
    """The DEVICE source of the solver (tests/emu: 64 lane threads) answering opt.solve() for the drop-in class
    around a touch-down, write-back tick included (t = 261): the kernel's logic on the reference's own use case,
    in the CPU tier.  Start state and plan come from an oracle-driven walk up to t = 250."""
    import ctypes
    import build as _b
    emu = ctypes.CDLL(_b.build_emu())

    class EmuBackedMPC(centroidal_mpc):
        """TEST ONLY: opt.solve() answered by the host emulation of the device source."""

        def _make_solver(self, device):
            self._cs = oracle_spec(oracle, self.spec)
            return None

        def _solve_record(self, rec):
            nsol = self.spec.nsol
            out, st, it, kk = np.zeros((1, nsol)), np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1)
            p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
            r = np.ascontiguousarray(rec[None])
            w = None if self._warm is None else np.ascontiguousarray(self._warm[None])
            assert emu.cmpc_emu_solve_batch(ctypes.byref(self._cs), 1, p(r), p(w), p(out), p(st), p(it), p(kk)) == 0
            if st[0] in (0, 3):
                self._warm = out[0].copy()
            return out[0], int(st[0]), int(it[0]), float(kk[0])

    hw = measured_hw()
    lead, planner_o, _, mpc_o, sc = build_walk(make_oracle_backed(oracle), N=10, hw=hw)
    lead.run(250)
    walk, planner, ftg, mpc, _ = build_walk(EmuBackedMPC, N=10, hw=hw)
    walk.time, walk.com, walk.dcom, walk.hw = 250, lead.com.copy(), lead.dcom.copy(), lead.hw.copy()
    mpc.model_state['theta_hat']['val'] = mpc_o.model_state['theta_hat']['val'].copy()
    mpc._warm = mpc_o._warm.copy()
    ref = make_oracle_backed(oracle)
    for _ in range(22):                                       # t = 250 ... 271
        lead_state = lead.step()[0]
        state, contact = walk.step()
        assert mpc.last_status in (0, 3)
        assert np.abs(state['com']['pos'] - lead_state['com']['pos']).max() < 1e-6      # same closed loop as the oracle's
    fired = np.array(walk.log['t'])[np.array(walk.log['counter']) == 1]
    assert fired.tolist() == [261]
    idx = planner.get_step_index_at_time(261)
    assert np.abs(planner.plan[idx + 1]['pos'] - planner_o.plan[idx + 1]['pos']).max() < 1e-7
