"""BASELINE config 1 on the GPU: the flat-ground walk (code/simulation.py:193-212 driving
code/centroidal_mpc_vertices.py:480-683) through the drop-in class with the HIP solver, and as B parallel
closed-loop rollouts with per-instance contact plans."""
import numpy as np
import pytest
import torch

from conftest import oracle_spec, rel_inf
from cmpc_amd import workloads as wl
from cmpc_amd.problem import ProblemSpec
from test_walk import build_walk, check_walk_log, measured_hw, run_rate_10_walk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device: the HIP extension must run, there is no fallback")


def test_flat_ground_walk_drop_in_class(gpu, oracle):
    """t = 0 ... 700: five steps with lift-off, early and late single support, touch-down, contact switch and
    the plan write-back, N = 10 as shipped; measured momentum = the reference's own recording.  Every tick must
    solve; sampled ticks are re-solved by the oracle from the same record and warm start."""
    import centroidal_mpc_vertices
    ticks = 700
    walk, planner, ftg, mpc, sc = build_walk(centroidal_mpc_vertices.centroidal_mpc, N=10, hw=measured_hw())
    nominal = [p['pos'].copy() for p in planner.plan]
    cs = oracle_spec(oracle, mpc.spec)
    worst, n_cmp = 0.0, 0
    for t in range(ticks):
        sample = (t % 23 == 0) or t in (199, 200, 261, 262, 269, 270, 271, 299, 300)
        warm = None if (mpc._warm is None or not sample) else mpc._warm[0].cpu().numpy()
        walk.step()
        if sample:
            got = np.concatenate([mpc.x_collect.T.ravel(), np.zeros(0)])
            ref, st, _, _ = oracle.solve(cs, mpc.last_record, warm=warm)
            assert st in (0, 3)
            X = ref[:20 * 11].reshape(11, 20).T
            err = np.abs(mpc.x_collect - X).max() / np.abs(X).max()
            if err >= 1e-4:                                   # flat valley: same objective, same defect
                up = warm[20 * 11:] if warm is not None else None
                full = np.concatenate([mpc.x_collect.T.ravel(), mpc._warm[0].cpu().numpy()[20 * 11:]])
                f_g, d_g, _, _ = oracle.evaluate(cs, mpc.last_record, full, uprox=up)
                f_r, d_r, _, _ = oracle.evaluate(cs, mpc.last_record, ref, uprox=up)
                assert abs(f_g - f_r) <= 1e-7 * max(1.0, abs(f_r)) and np.abs(d_g).max() < 1e-7
            else:
                worst = max(worst, err)
            n_cmp += 1
    assert n_cmp >= 30 and worst < 1e-4
    log = {k: (np.array(v) if k in ('t', 'com', 'status', 'iterations', 'kkt', 'counter', 'hw_des') else v)
           for k, v in walk.log.items()}
    check_walk_log(log, planner, nominal, sc, 10, ticks)
    assert log['iterations'].mean() < 30


def test_flat_ground_walk_payload_gains_and_rate_10(gpu, oracle):
    """The payload variant (k1, k2 = 7, 1; centroidal_mpc_vertices_payload.py:27-31) over two steps, and the
    mpc_rate = 10 variant of the nominal class (delta = 0.1, k1, k2 = 5, 0.2, no force-rate cost; :11, :27-31,
    :339-341) called every tenth tick as simulation.py:203 does."""
    import centroidal_mpc_vertices_payload
    walk, planner, ftg, mpc, sc = build_walk(centroidal_mpc_vertices_payload.centroidal_mpc, N=10, hw=measured_hw())
    assert (mpc.spec.k1, mpc.spec.k2) == (7.0, 1.0)
    log = walk.run(420)
    assert np.isin(log['status'], (0, 3)).all()
    ref = sc.com_tab[log['t'] + 1, 0:3]
    assert np.abs(log['com'] - ref).max() < 0.03
    import centroidal_mpc_vertices
    run_rate_10_walk(centroidal_mpc_vertices.centroidal_mpc, 600)


def test_builder_reads_per_instance_plans_bit_exact(gpu, scene):
    from cmpc_amd.footstep_planner_vertices import FootstepPlanner
    from cmpc_amd.problem import build_record
    from cmpc_amd.solver import DeviceRecordBuilder
    rng = np.random.default_rng(5)
    spec = ProblemSpec(N=10)
    B = 257
    t = rng.integers(0, scene.t_max(10) + 1, size=B).astype(np.int32)
    t[:4] = [199, 200, 269, 270]
    state = rng.normal(size=(B, 16))
    plans = np.repeat(scene.plan_pos[None], B, axis=0) + rng.normal(scale=1e-3, size=(B,) + scene.plan_pos.shape)
    bld = DeviceRecordBuilder(scene, device="cuda:0")
    got = bld.build(spec, torch.from_numpy(t).to("cuda:0"), torch.from_numpy(state).to("cuda:0"),
                    plan_pos=torch.from_numpy(plans).to("cuda:0")).cpu().numpy()
    params = wl.default_params(N=10)
    for b in range(0, B, 8):
        planner = FootstepPlanner(wl.VREF, wl.LFOOT0, wl.RFOOT0, params)
        for i, p in enumerate(planner.plan):
            p['pos'] = plans[b, i].copy()
        want = build_record(spec, planner, scene.com_ref, int(t[b]), state[b, 0:3], state[b, 3:6], state[b, 6:9],
                            state[b, 9:12], state[b, 12], state[b, 13], state[b, 14], state[b, 15],
                            contacts_ref=scene.planner.position_contacts_ref)
        assert np.array_equal(got[b], want), b


def test_batched_rollout_equals_single_instance_loops(gpu, scene):
    """B parallel closed loops with per-instance plans against B loops of the single-instance drop-in class:
    states, plans and write-back ticks bit for bit (the batch adds nothing but parallelism)."""
    import centroidal_mpc_vertices
    from cmpc_amd.footstep_planner_vertices import FootstepPlanner
    from cmpc_amd.rollout import BatchedRollout
    hwm = measured_hw()
    spec = ProblemSpec(N=10)
    B, t0, ticks = 6, 240, 75                                # through a write-back (t = 261) and a touch-down (270)
    rng = np.random.default_rng(21)
    com, dcom = scene.nominal_state(np.full(B, t0))
    com = com + rng.uniform(-0.004, 0.004, size=(B, 3))
    off = rng.normal(0, 0.05, size=(B, 3))
    ro = BatchedRollout(scene, spec, B, device="cuda:0", hw_measured=hwm, hw_offset=off)
    ro.reset(t0, com, dcom)
    fired = np.zeros((ticks, B), bool)
    for i in range(ticks):
        ro.step()
        fired[i] = ro.counter.cpu().numpy()
    assert ro.alive.all().item() and fired.any(axis=0).all()
    state_b, plan_b = ro.state.cpu().numpy(), ro.plan_pos.cpu().numpy()
    params = wl.default_params(N=10)
    for b in range(B):
        planner = FootstepPlanner(wl.VREF, wl.LFOOT0, wl.RFOOT0, params)
        mpc = centroidal_mpc_vertices.centroidal_mpc(scene.initial, planner, params, scene.com_ref, None, None)
        c, v = com[b].copy(), dcom[b].copy()
        for i in range(ticks):
            t = t0 + i
            cur = {'com': {'pos': c, 'vel': v}, 'hw': {'val': hwm[t] + off[b]},
                   'lfoot': {'pos': np.zeros(6)}, 'rfoot': {'pos': np.zeros(6)}}
            ms, _ = mpc.solve(cur, t)
            assert bool(ms['counter']['val']) == bool(fired[i, b])
            c, v = ms['com']['pos'].copy(), ms['com']['vel'].copy()
        assert np.array_equal(state_b[b, 0:3], c) and np.array_equal(state_b[b, 3:6], v)
        assert np.array_equal(state_b[b, 9:12], ms['theta_hat']['val'])
        assert np.array_equal(plan_b[b], np.stack([p['pos'] for p in planner.plan]))
        assert not np.array_equal(plan_b[b], scene.plan_pos)


def test_batched_rollout_walk_256(gpu, scene):
    """Config 1 as 256 parallel walks with perturbed initial states and momentum offsets, three steps."""
    from cmpc_amd.rollout import BatchedRollout
    hwm = measured_hw()
    spec = ProblemSpec(N=10)
    B, t0, ticks = 256, 150, 380
    rng = np.random.default_rng(22)
    com, dcom = scene.nominal_state(np.full(B, t0))
    com = com + rng.uniform(-0.003, 0.003, size=(B, 3))
    dcom = dcom + rng.normal(0, 0.01, size=(B, 3))
    ro = BatchedRollout(scene, spec, B, device="cuda:0", hw_measured=hwm, hw_offset=rng.normal(0, 0.05, size=(B, 3)))
    ro.reset(t0, com, dcom)
    hist, alive = ro.run(ticks)
    hist = hist.cpu().numpy()
    assert alive.all().item()                                  # every tick solved for every instance
    ref = np.stack([scene.com_tab[t0 + i, 0:3] for i in range(ticks + 1)])
    err = np.abs(hist - ref[:, None, :])
    assert err[..., :2].max() < 0.05 and err[..., 2].max() < 0.02
    moved = (ro.plan_pos.cpu().numpy() != scene.plan_pos[None]).any(axis=2)   # (B, n_steps)
    assert (moved.sum(axis=1) == 3).all()                      # three write-backs per instance (t = 261, 361, 461)


def test_whole_tick_on_the_device_mpc_then_whole_body_qp(gpu, scene):
    """SURVEY 8f rows 2 + 4 together: records -> MPC solve -> plan write-back -> whole-body QP, every tick, for 64 robots,
    all on the device (code/simulation.py:193-232 per robot: customPreStep solves the MPC, then get_joint_torques solves
    the QP with desired['com'] from the MPC's model_state).  The rigid-body model is a synthetic stand-in of the
    reference's shapes (DART is not available): fixed task Jacobians per robot, the CoM task's feed-forward acceleration
    = the MPC's CoM_acc (:633-636).  Every QP must converge, follow the commanded CoM acceleration, and equal the numpy
    oracle of the QP on the same matrices."""
    from cmpc_amd import wbc
    from cmpc_amd.rollout import BatchedRollout
    from oracle import wbc_qp_oracle as wq
    spec = ProblemSpec(N=10)
    B = 64
    rng = np.random.default_rng(5)
    # (the angular momentum about the CoM is the reference's own recording plus a per-robot offset, as in the walk tests:
    # with hw = 0 exactly, late single support is infeasible)
    ro = BatchedRollout(scene, spec, B, device="cuda:0", hw_measured=measured_hw(), hw_offset=rng.normal(0, 0.05, size=(B, 3)))
    Hq0, Fq0, M, h, Jc = (torch.from_numpy(a).cuda() for a in wl.wbc_synthetic(B, seed=77))
    Jcom = torch.from_numpy(rng.normal(0, 0.4, size=(B, 3, 30))).cuda()
    Jcom[:, :, 3:6] += torch.eye(3, dtype=torch.float64, device="cuda:0")
    w_com = 50.0                                                            # the CoM task on top of the posture tasks
    Hq = (Hq0 + w_com * Jcom.transpose(1, 2) @ Jcom).contiguous()
    seen = []

    def model(rollout, desired):
        # contact Jacobian rows scaled by the contact flags (code/inverse_dynamics.py:109); CoM task gradient from the MPC
        g = torch.cat([desired["gamma_l"][:, None].expand(-1, 6), desired["gamma_r"][:, None].expand(-1, 6)], dim=1)
        Fq = (Fq0 - w_com * torch.einsum("brn,br->bn", Jcom, desired["com_acc"])).contiguous()
        Jc_t = (Jc * g[:, :, None]).contiguous()
        seen.append((Fq, Jc_t, desired["com_acc"].clone()))
        return Hq, Fq, M, h, Jc_t
    qp = wbc.BatchedInverseDynamicsQP(foot_size=0.1, mu=0.5, device="cuda:0")
    ro.attach_whole_body(qp, model)
    t0 = 255                                                                # single support, touch-down inside the run
    com, dcom = scene.nominal_state(np.full(B, t0))
    ro.reset(t0, com + rng.uniform(-0.003, 0.003, size=(B, 3)), dcom)
    for i in range(12):
        _, _, status = ro.step()
        tau, qdd, fc, st_q, it_q = ro.last_wbc
        assert bool(np.isin(status.cpu().numpy(), (0, 3)).all()) and bool((st_q == 0).all())
        assert tau.shape == (B, 24) and bool(torch.isfinite(tau).all())
        Fq, Jc_t, acc = seen[-1]
        # the commanded CoM acceleration is followed (the CoM task dominates its directions)
        got = torch.einsum("brn,bn->br", Jcom, qdd)
        assert float((got - acc).abs().max()) < 0.35 * max(1.0, float(acc.abs().max()))
        # a foot in the air carries no wrench (its Jacobian rows are zero: 1e-6 |f|^2 alone holds f at 0)
        air_l = seen[-1][1][:, 0:6].abs().sum(dim=(1, 2)) == 0
        assert float(fc[air_l][:, 0:6].abs().max()) < 1e-6 if bool(air_l.any()) else True
    # last tick against the numpy oracle of the QP, first three robots
    for b in range(3):
        ref = wq.solve(Hq[b].cpu().numpy(), Fq[b].cpu().numpy(), M[b].cpu().numpy(), h[b].cpu().numpy(), Jc_t[b].cpu().numpy(), 0.05, 0.5)
        assert ref["status"] == 0
        assert np.abs(qdd[b].cpu().numpy() - ref["qdd"]).max() < 1e-6 * max(1.0, np.abs(ref["qdd"]).max())
        assert np.abs(tau[b].cpu().numpy() - ref["tau"][6:]).max() < 1e-6 * max(1.0, np.abs(ref["tau"]).max())
