"""Solver state across closed-loop ticks on the GPU (cmpc_solve_batch_state) against the C oracle, tick by tick."""
import numpy as np
import pytest
import torch

from conftest import oracle_spec, rel_inf
from cmpc_amd import workloads as wl
from cmpc_amd.problem import ProblemSpec
from cmpc_amd.solver import BatchedCentroidalMPC
from test_warm_state import _loop

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,t0,ticks", [(10, 255, 12), (20, 240, 8)])
def test_hip_solver_resumes_like_the_oracle(oracle, N, t0, ticks):
    spec = ProblemSpec(N=N)
    solver = BatchedCentroidalMPC(spec, device="cuda:0")
    cs = oracle_spec(oracle, spec)

    def gpu_solve(rec, warm, state):
        d = torch.from_numpy(np.ascontiguousarray(rec)).cuda()
        w = None if warm is None else torch.from_numpy(np.ascontiguousarray(warm)).cuda()
        s_in = None if state is None else torch.from_numpy(np.ascontiguousarray(state)).cuda()
        s_out = solver.new_state(1)
        out, st, it, _ = solver.solve(d, warm=w, state=s_in, state_out=s_out)
        return out.cpu().numpy(), s_out.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy()

    def ora_solve(rec, warm, state):
        out, so, st, it, _ = oracle.solve_batch_state(cs, rec, warm=warm, state=state)
        return out, so, st, it

    _, sol_g, it_g = _loop(gpu_solve, N, t0, ticks, use_state=True)
    st_g = _loop.last_status
    _, sol_o, it_o = _loop(ora_solve, N, t0, ticks, use_state=True)
    st_o = _loop.last_status
    # (the end game of a solve is sensitive to rounding order: a step more or less here and there.  Where one of the two
    # ended at the acceptable level its count includes the progress watch's window of 12 iterations: iteration counts are
    # compared on the ticks both converged on, and there must be few others)
    tight = (st_g == 0) & (st_o == 0)
    assert tight.sum() >= ticks - 2, (st_g, st_o)
    # (per tick: within 4 iterations, one tick of a run within 8 -- round 5, r05c: the late-single-support tick t = 265 of the
    # N = 10 run took 17 iterations on the GPU against the oracle's 12, both converged, after G'PG moved to registers with
    # explicit fused multiply-adds; rounds 3-4 measured <= 4 on every tick.  The sum over the run is held to 10 % as before.)
    dit = np.abs(it_g - it_o)[tight]
    assert (dit > 4).sum() <= 1 and dit.max() <= 8, (it_g, it_o)
    assert abs(int(it_g[tight].sum()) - int(it_o[tight].sum())) <= max(6, 0.1 * it_o[tight].sum()), (it_g, it_o)
    assert np.abs(sol_g[:, 20:26] - sol_o[:, 20:26]).max() < 1e-6          # the fed-back CoM state of every tick
    assert np.median(rel_inf(sol_g, sol_o)) < 1e-8
    assert it_g[1:].mean() < 0.7 * it_g[0]                                   # resumed ticks are cheaper than the cold first one


def test_state_batch_matches_single_instances_and_plain_entry_point(oracle):
    spec, rec = wl.make_workload("perturbed", B=96, N=10)
    solver = BatchedCentroidalMPC(spec, device="cuda:0")
    d = torch.from_numpy(rec).cuda()
    s1, s2 = solver.new_state(96), solver.new_state(96)
    a, st_a, it_a, _ = solver.solve(d)
    b, st_b, it_b, _ = solver.solve(d, state=s1, state_out=s2)               # empty state = the plain entry point
    assert torch.equal(a, b) and torch.equal(it_a, it_b)
    assert torch.equal(s2[:, -7 - 2 * 11], it_b.double())                   # the state carries what the solve took (queue order of the next launch)
    # every instance again from its own state and solution: a handful of iterations, the same optimum
    c, st_c, it_c, _ = solver.solve(d, warm=b, state=s2, state_out=s1)
    w, st_w, it_w, _ = solver.solve(d, warm=b)                              # the same problem (proximal centre b) the plain way
    ok = (st_w == 0) & (st_c == 0)
    assert ok.float().mean() > 0.9 and it_c[ok].float().mean() < 0.6 * it_w[ok].float().mean()
    assert np.median(rel_inf(c[ok].cpu().numpy(), w[ok].cpu().numpy())) < 1e-6
    # and the batch is the sum of its instances (bitwise)
    for i in (0, 17, 95):
        ci, _, it_i, _ = solver.solve(d[i:i + 1], warm=b[i:i + 1], state=s2[i:i + 1].clone(), state_out=solver.new_state(1))
        assert torch.equal(ci[0], c[i]) and int(it_i[0]) == int(it_c[i])


def test_state_edge_cases_stale_state_nan_record_and_horizon_limits(oracle):
    """(a) a state written for a tick far away (another contact phase, pushed CoM) must not do harm: the solve ends with
    the plain solve's optimum, by its own iterations or through the fallback; (b) a NaN record (out-of-range tick of
    the device builder) ends with status 2 and leaves an invalid state behind; (c) the smallest and the largest horizon
    of the build (N = 1, N = CMPC_MAX_N = 64) take the state path too."""
    sc = wl.scene()
    spec = ProblemSpec(N=10)
    solver = BatchedCentroidalMPC(spec, device="cuda:0")
    HW = np.loadtxt(__file__.rsplit("/", 1)[0] + "/golden/measured_hw_cuhw.txt")

    def rec_at(t, dcom_push=0.0):
        com, dcom = sc.nominal_state(np.array([t]))
        return sc.build_records(spec, np.array([t]), com, dcom + dcom_push, HW[t][None], np.zeros((1, 3)), np.zeros(1),
                                np.zeros(1), np.full(1, wl.HRP4_MASS), np.full(1, 0.5))

    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    s0, s1, s2 = solver.new_state(1), solver.new_state(1), solver.new_state(1)
    a, st_a, it_a, _ = solver.solve(dev(rec_at(120)), state=s0, state_out=s1)            # double support, t = 120
    assert int(st_a[0]) == 0 and float(s1[0, -8 - 2 * 11]) > 0
    far = dev(rec_at(640, dcom_push=0.08))                                               # single support, five steps later, pushed
    b, st_b, it_b, _ = solver.solve(far, warm=a, state=s1, state_out=s2)
    c, st_c, it_c, _ = solver.solve(far, warm=a)
    assert int(st_b[0]) in (0, 3) and int(st_c[0]) in (0, 3)
    cs = oracle_spec(oracle, spec)
    up = a[0].cpu().numpy()[20 * 11:]
    f_b = oracle.evaluate(cs, rec_at(640, 0.08)[0], b[0].cpu().numpy(), uprox=up)[0]
    f_c = oracle.evaluate(cs, rec_at(640, 0.08)[0], c[0].cpu().numpy(), uprox=up)[0]
    assert abs(f_b - f_c) <= 1e-7 * abs(f_c) and rel_inf(b.cpu().numpy(), c.cpu().numpy())[0] < 1e-3
    # fallback: both attempts are counted and share max_iter; a state that does not fit is given up after twenty iterations
    assert int(it_b[0]) <= spec.max_iter + 1 and int(it_b[0]) <= int(it_c[0]) + 22
    # (b)
    bad = rec_at(300); bad[0, 30] = np.nan
    s3 = solver.new_state(1)
    _, st_n, _, _ = solver.solve(dev(bad), warm=a, state=s1, state_out=s3)
    assert int(st_n[0]) == 2 and float(s3[0, -8 - 2 * 11]) == 0.0
    # (c)
    for N in (1, 64):
        sp = ProblemSpec(N=N)
        so = BatchedCentroidalMPC(sp, device="cuda:0")
        com, dcom = sc.nominal_state(np.array([230]))
        r = sc.build_records(sp, np.array([230]), com, dcom, HW[230][None], np.zeros((1, 3)), np.zeros(1), np.zeros(1),
                             np.full(1, wl.HRP4_MASS), np.full(1, 0.5))
        t0, t1 = so.new_state(1), so.new_state(1)
        x, st_x, it_x, _ = so.solve(dev(r), state=t0, state_out=t1)
        y, st_y, it_y, _ = so.solve(dev(r), warm=x, state=t1, state_out=t0)
        ref, st_r, _, _ = oracle.solve_batch(oracle_spec(oracle, sp), r)
        assert int(st_x[0]) in (0, 3) and int(st_y[0]) in (0, 3) and st_r[0] in (0, 3)
        assert rel_inf(x.cpu().numpy(), ref)[0] < 1e-4
        # N = 1 resumes in a handful of iterations; at N = 64 the end game needs inertia corrections of 1 ... 50 and
        # hovers at ~1e-8 for as long as it is allowed to, resumed or not (oracle: cold 43 iterations "acceptable", the
        # same tolerance-level point): only the outcome is asserted there
        assert N == 64 or int(it_y[0]) < int(it_x[0])


def test_state_path_of_the_eight_vertex_two_wave_kernel(oracle):
    """The 8-vertex solver (two waves per instance) through the state entry point: an empty state is the plain solve
    bit for bit, a resumed re-solve of every instance reaches the same optimum as the oracle's resumed solve in a
    fraction of the iterations, and a batch larger than the resident grid (512 workgroups) is the sum of its
    instances."""
    spec, rec = wl.make_workload("long_horizon", B=640, N=10)
    solver = BatchedCentroidalMPC(spec, device="cuda:0")
    cs = oracle_spec(oracle, spec)
    d = torch.from_numpy(rec).cuda()
    s1, s2 = solver.new_state(640), solver.new_state(640)
    a, st_a, it_a, _ = solver.solve(d)
    b, st_b, it_b, _ = solver.solve(d, state=s1, state_out=s2)
    assert torch.equal(a, b) and torch.equal(it_a, it_b) and torch.equal(st_a, st_b)
    c, st_c, it_c, _ = solver.solve(d, warm=b, state=s2, state_out=s1)
    ok = ((st_b == 0) & (st_c == 0)).cpu().numpy()
    assert ok.mean() > 0.85 and it_c.cpu().numpy()[ok].mean() < 0.6 * it_b.cpu().numpy()[ok].mean()
    # the oracle from the kernel's own state and solution (first 24 instances)
    n = 24
    out_o, _, st_o, it_o, _ = oracle.solve_batch_state(cs, rec[:n], warm=b[:n].cpu().numpy(), state=s2[:n].cpu().numpy())
    both = ok[:n] & (st_o == 0)
    assert both.sum() >= 16
    assert np.median(rel_inf(c[:n].cpu().numpy()[both], out_o[both])) < 1e-7
    assert np.abs(it_c[:n].cpu().numpy()[both] - it_o[both]).max() <= 4
    for i in (0, 333, 639):
        ci, _, it_i, _ = solver.solve(d[i:i + 1], warm=b[i:i + 1], state=s2[i:i + 1].clone(), state_out=solver.new_state(1))
        assert torch.equal(ci[0], c[i]) and int(it_i[0]) == int(it_c[i])


def test_state_in_and_state_out_must_not_overlap():
    """The kernel invalidates state_out before it reads state_in: overlapping ranges are refused by the C entry point
    (views of one buffer included), not solved from a half-overwritten state."""
    import ctypes
    from cmpc_amd import capi
    spec, rec = wl.make_workload("perturbed", B=4, N=10)
    solver = BatchedCentroidalMPC(spec, device="cuda:0")
    d = torch.from_numpy(rec).cuda()
    buf = torch.zeros((5, spec.nstate), dtype=torch.float64, device="cuda:0")
    with pytest.raises(ValueError, match="overlap"):
        solver.solve(d, state=buf[0:4], state_out=buf[1:5])
    out = torch.empty((4, spec.nsol), dtype=torch.float64, device="cuda:0")
    st, it = torch.empty(4, dtype=torch.int32, device="cuda:0"), torch.empty(4, dtype=torch.int32, device="cuda:0")
    kk = torch.empty(4, dtype=torch.float64, device="cuda:0")
    lib = capi.load()
    rc = lib.cmpc_solve_batch_state(solver._h, 4, d.data_ptr(), None, buf[0:4].data_ptr(), out.data_ptr(), buf[1:5].data_ptr(),
                                    st.data_ptr(), it.data_ptr(), kk.data_ptr(), None)
    assert rc != 0 and b"overlap" in lib.cmpc_last_error(solver._h)
    rc = lib.cmpc_solve_batch_state(solver._h, 4, d.data_ptr(), None, buf[0:4].data_ptr(), out.data_ptr(), buf[0:4].data_ptr(),
                                    st.data_ptr(), it.data_ptr(), kk.data_ptr(), None)
    assert rc != 0


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["single", "pair"])
def test_failed_resumed_attempts_keep_their_acceptable_point(oracle, kernel):
    """512 instances resumed from ANOTHER instance's solver state with a budget of 30 iterations (the resumed attempt goes
    stale or nowhere, the plain attempt gets what is left): the HIP kernels end like the oracle -- nothing that ends at the
    cap or "numerical" had an acceptable point to fall back on, the verdict usable / not agrees on (nearly) every instance,
    and as many instances end "acceptable" with the whole budget spent (the saved point of the failed attempt among them;
    tests/test_emu_kernel.py has the three instances the rule was written for)."""
    import dataclasses
    spec, rec = wl.make_workload("randomized", B=512, N=20)
    cs0 = oracle_spec(oracle, spec)
    _, state, st0, _, _ = oracle.solve_batch_state(cs0, rec)
    state = state[np.roll(np.arange(512), 1)]
    spec = dataclasses.replace(spec, max_iter=30, kernel={"single": 1, "pair": 2}[kernel])
    cs = oracle_spec(oracle, spec)
    ref, _, st_r, it_r, kk_r = oracle.solve_batch_state(cs, rec, state=state)
    s = BatchedCentroidalMPC(spec, device="cuda:0")
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    so = s.new_state(512)
    out, st, it, kk = s.solve(dev(rec), state=dev(state), state_out=so)
    torch.cuda.synchronize()
    st, it, kk, out = st.cpu().numpy(), it.cpu().numpy(), kk.cpu().numpy(), out.cpu().numpy()
    assert it.max() <= spec.max_iter + 1
    bad = np.isin(st, (1, 2))
    assert bad.any() and kk[bad].min() > spec.acc_tol             # no acceptable point was given up
    ok, ok_r = np.isin(st, (0, 3)), np.isin(st_r, (0, 3))
    assert (ok != ok_r).sum() <= 0.05 * 512, (int((ok != ok_r).sum()),)
    at_cap, at_cap_r = (st == 3) & (it >= spec.max_iter), (st_r == 3) & (it_r >= spec.max_iter)
    assert at_cap.sum() >= 0.7 * at_cap_r.sum() and kk[at_cap].max() <= spec.acc_tol
    both = ok & ok_r & (st == st_r)
    assert np.median(rel_inf(out[both], ref[both])) < 1e-6
