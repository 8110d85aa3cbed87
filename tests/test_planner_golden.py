"""Host logic against the reference's own fixtures (code/Debug/*, see tests/golden/README.md)."""
import os

import numpy as np

from cmpc_amd import workloads as wl
from cmpc_amd.problem import ProblemSpec, build_record, contact_flags

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_plan_positions_match_reference_dump(scene):
    lines = [l.strip() for l in open(os.path.join(GOLD, "MPC_pose_contact_ref.txt"))]
    gold = np.array([float(x) for x in lines if x and x != "end"]).reshape(20, 3)
    assert np.array_equal(scene.planner.plan_positions(), gold)          # bit-exact


def test_swing_trajectories_match_reference_dump(scene):
    pre = scene.ftg.generate_feet_trajectories_pre()
    for foot, name in (("lfoot", "pos_lfoot_pre_trj.txt"), ("rfoot", "pos_rfoot_pre_trj.txt")):
        gold = np.loadtxt(os.path.join(GOLD, name))
        mine = np.array([pre[foot][t][0]['pos'][3:6] for t in range(gold.shape[0])])
        assert np.array_equal(mine, gold)                                  # bit-exact


def test_plan_structure(scene):
    pl = scene.planner
    assert len(pl.plan) == 20
    assert pl.plan[0]['ss_duration'] == 0 and pl.plan[0]['ds_duration'] == 200
    assert [s['foot_id'] for s in pl.plan[:4]] == ['rfoot', 'lfoot', 'rfoot', 'lfoot']
    assert pl.get_step_index_at_time(0) == 0 and pl.get_step_index_at_time(199) == 0
    assert pl.get_step_index_at_time(200) == 1 and pl.get_phase_at_time(200) == 'ss'
    assert pl.get_phase_at_time(269) == 'ss' and pl.get_phase_at_time(270) == 'ds'
    assert pl.get_step_index_at_time(10 ** 6) is None
    assert pl.position_contacts_ref['contact_left'].shape == (2000, 6)


def test_contact_flags_and_tables(scene):
    pl = scene.planner
    for t in (0, 150, 199, 200, 269, 270, 305, 1234, 1700):
        gl, gr = contact_flags(pl, t, 20)
        assert np.array_equal(gl, scene.gl_tab[t:t + 21]) and np.array_equal(gr, scene.gr_tab[t:t + 21])
        assert set(np.unique(np.concatenate([gl, gr]))) <= {0.0, 1.0}
        assert np.all(gl + gr >= 1)                                         # never both feet in the air


def test_vectorised_builder_equals_scalar_front_half(scene):
    spec = ProblemSpec(N=20)
    rng = np.random.default_rng(5)
    ts = rng.integers(0, scene.t_max(20), size=40)
    B = len(ts)
    com, dcom, hw, th = rng.normal(size=(B, 3)), rng.normal(size=(B, 3)), rng.normal(size=(B, 3)), rng.normal(size=(B, 3))
    yl, yr = rng.normal(size=B), rng.normal(size=B)
    mass, mu = rng.uniform(30, 50, B), rng.uniform(0.3, 0.9, B)
    batch = scene.build_records(spec, ts, com, dcom, hw, th, yl, yr, mass, mu)
    for b, t in enumerate(ts):
        one = build_record(spec, scene.planner, scene.com_ref, int(t), com[b], dcom[b], hw[b], th[b], yl[b], yr[b],
                           mass[b], mu[b])
        assert np.array_equal(one, batch[b])


def test_vectorised_builder_rate_10(scene):
    """mpc_rate = 10 (:548-600 with rate 10): references sampled at t + (1+i)*10, contact flags at t + i*10."""
    from cmpc_amd.problem import ProblemSpec as PS
    spec = PS.from_params(wl.default_params(N=10, mpc_rate=10))
    assert (spec.delta, spec.k1, spec.k2, spec.w_rate) == (0.1, 5.0, 0.2, 0.0)          # :11, :27-31, :339-341
    rng = np.random.default_rng(6)
    ts = rng.integers(0, scene.t_max(10, 10), size=24)
    B = len(ts)
    com, dcom, hw, th = (rng.normal(size=(B, 3)) for _ in range(4))
    batch = scene.build_records(spec, ts, com, dcom, hw, th, np.zeros(B), np.zeros(B), np.full(B, 40.0), np.full(B, 0.5), rate=10)
    for b, t in enumerate(ts):
        one = build_record(spec, scene.planner, scene.com_ref, int(t), com[b], dcom[b], hw[b], th[b], 0.0, 0.0, 40.0, 0.5, rate=10)
        assert np.array_equal(one, batch[b])
        st = one[24:].reshape(10, 19)
        assert st[3, 0] == scene.com_ref['pos_x'][int(t) + 40] and st[3, 17] == scene.gl_tab[int(t) + 30]


def test_record_layout(scene):
    spec = ProblemSpec(N=10)
    t = 640
    rec = build_record(spec, scene.planner, scene.com_ref, t, [1, 2, 3], [4, 5, 6], [7, 8, 9], [.1, .2, .3], 0.4, 0.5, 41.0, 0.6)
    assert rec.shape == (24 + 19 * 10,)
    assert list(rec[0:12]) == [1, 2, 3, 4, 5, 6, 7, 8, 9, .1, .2, .3] and rec[12] == 0.4 and rec[16] == 0.5
    assert rec[20] == 41.0 and rec[21] == 0.6
    st = rec[24:].reshape(10, 19)
    for i in range(10):
        tt = t + 1 + i
        assert st[i, 0] == scene.com_ref['pos_x'][tt] and st[i, 4] == scene.com_ref['vel_y'][tt]
        assert st[i, 8] == scene.com_ref['acc_z'][tt]
        assert np.array_equal(st[i, 9:12], scene.pose_l[tt, 3:6]) and st[i, 16] == scene.pose_r[tt, 2]
    # foot positions in x0 come from the plan, not from the measured feet (reference :493-509)
    idx = scene.planner.get_step_index_at_time(t - 70)
    assert np.array_equal(rec[13:16], scene.planner.plan[idx + (idx - 1) % 2]['pos'])
    assert np.array_equal(rec[17:20], scene.planner.plan[idx + (idx % 2)]['pos'])


def test_com_reference_shape_and_continuity(scene):
    ref = scene.com_ref
    assert len(ref['pos_x']) == 1971 and len(ref['pos_y']) == 2000
    assert np.all(np.asarray(ref['pos_z']) == 0.72)
    px = np.asarray(ref['pos_x'])
    assert np.abs(np.diff(px)).max() < 3e-3                                  # C0 and smooth
    assert abs(px[-1] - 2.07) < 2e-2


def test_workload_generators_are_deterministic():
    for name in wl.CONFIGS:
        s1, r1 = wl.make_workload(name, B=16)
        s2, r2 = wl.make_workload(name, B=16)
        assert np.array_equal(r1, r2) and r1.shape == (16, s1.nrec)
        assert np.isfinite(r1).all()


def test_quintic_spline_solves_the_reference_system_and_is_minimum_norm(scene):
    """CoM reference knots -> coefficients (code/functions.py:129-157): the reference hands the 4n-1 equality
    rows of :135-149 in 6n unknowns to IPOPT with a zero objective from p = 0.  What is checked here, by
    execution: (i) the rows are exactly those of the reference (position at both ends of every segment,
    zero end velocities, C1 and C2 continuity, zero initial acceleration), (ii) the coefficients satisfy all
    of them to 1e-12, (iii) they are THE minimum-norm solution: orthogonal to the null space of the system, so
    no feasible point is shorter -- the point a regularised Newton step from the origin lands on."""
    from cmpc_amd import functions as fn
    import scipy.linalg as sla
    knot_x, knot_y, seq_x, seq_y = fn.compute_knot(scene.ftg, scene.planner)
    for knots in (knot_x, knot_y):
        n = len(knots)
        A, b = fn.quintic_constraints(knots)
        assert A.shape == (4 * n - 1, 6 * n) and np.linalg.matrix_rank(A) == 4 * n - 1
        # (i) row semantics, written out independently of quintic_constraints
        p = fn.quintic_spline(knots).reshape(n, 6)
        val = lambda i, s: sum(p[i, j] * s ** j for j in range(6))
        d1 = lambda i, s: sum(j * p[i, j] * s ** (j - 1) for j in range(1, 6))
        d2 = lambda i, s: sum(j * (j - 1) * p[i, j] * s ** (j - 2) for j in range(2, 6))
        for i in range(n - 1):
            assert abs(val(i, 0.0) - knots[i]) < 1e-12 and abs(val(i, 1.0) - knots[i + 1]) < 1e-12
            assert abs(d1(i, 1.0) - d1(i + 1, 0.0)) < 1e-12 and abs(d2(i, 1.0) - d2(i + 1, 0.0)) < 1e-12
        assert abs(d1(0, 0.0)) < 1e-12 and abs(d1(n - 1, 0.0)) < 1e-12 and abs(d2(0, 0.0)) < 1e-12
        # (ii) all 4n-1 rows
        x = p.ravel()
        assert np.abs(A @ x - b).max() < 1e-12
        # (iii) minimum norm: x in range(A'), i.e. orthogonal to null(A); any other solution is longer
        Z = sla.null_space(A)
        assert Z.shape[1] == 2 * n + 1 and np.abs(Z.T @ x).max() < 1e-12
        rng = np.random.default_rng(0)
        for _ in range(5):
            other = x + Z @ rng.normal(size=Z.shape[1])
            assert np.abs(A @ other - b).max() < 1e-9 and np.linalg.norm(other) > np.linalg.norm(x)
        # and the closed form x = A'(AA')^-1 b
        assert np.abs(x - A.T @ np.linalg.solve(A @ A.T, b)).max() < 1e-10
