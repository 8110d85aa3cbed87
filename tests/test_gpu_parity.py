"""Parity tests proper: the HIP solver, called through the C ABI, against the C oracle on the same
seeded inputs, against the committed golden vectors, and -- at BASELINE.json's full sizes -- through
size-independent properties.  Tolerance (north star): 1e-4 rel-inf; what is asserted is tighter."""
import dataclasses
import glob
import os

import numpy as np
import pytest
import torch

from conftest import oracle_spec, rel_inf, group_rel_inf
from cmpc_amd import workloads as wl
from cmpc_amd.problem import ProblemSpec
import nlp_batch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REL_TOL = 1e-4            # BASELINE.json north star: solutions within 1e-4 rel-inf of the reference formulation


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device: the HIP extension must run, there is no fallback")
    from cmpc_amd.solver import BatchedCentroidalMPC
    return BatchedCentroidalMPC


def _solve(gpu, spec, rec, warm=None, kernel=None):
    """kernel: None = the library's own choice; "single" / "pair" = cmpc_spec.kernel fixed when the handle is created."""
    if kernel is not None:
        spec = dataclasses.replace(spec, kernel=KERNELS[kernel])
    s = gpu(spec, device="cuda:0")
    w = None if warm is None else torch.from_numpy(np.ascontiguousarray(warm)).to("cuda:0")
    out, st, it, kkt = s.solve(torch.from_numpy(np.ascontiguousarray(rec)).to("cuda:0"), warm=w)
    torch.cuda.synchronize()
    if kernel is not None and rec.shape[0] > 0:
        assert s.last_kernel_name() == KERNEL_NAMES[(kernel, spec.nv)], s.last_kernel_name()
    return out.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), kkt.cpu().numpy()


# ---------------------------------------------------------------------------------------------------------------------
# LEVELS.  Written once, from the problem's own constants, before the round's first GPU run of this file; not to be moved
# after a red run (a level that has to move is recorded with before / after and the run that forced it in DESIGN.md 2).
#
#   TOL = 1e-8      scaled KKT tolerance of a "converged" point (status 0; after the polish step at most 100 TOL)
#   ACC = 1e-4      acceptable level (status 3): such a point reports its own final scaled KKT error kappa <= ACC
#   RHO = 1e-4      proximal weight = the curvature of the objective along the directions the NLP leaves flat
#
# Two points that both satisfy the KKT conditions to kappa agree
#   * in objective to first order in kappa (through the multipliers, scale <= 10): |dJ| / J <= 10 kappa,
#   * in their constraint residuals to kappa,
#   * along a flat direction only to (multiplier scale) x kappa / RHO.
# kappa is what the solvers report per instance, capped at KAPPA_CAP = 1e-5: the largest final error of a usable point on
# the full bench batch is 6e-6 (BENCH_r04, `kkt.max`); nothing is granted beyond what the data shows.
# A pair of status-0 points gets NO kappa allowance: objective to OBJ_TIGHT, residuals to RES_TIGHT, displacement within
# 100 TOL / RHO = 1e-2 at the very most (measured max over 16 384 tight pairs: 2e-4).
TOL, ACC, RHO = 1e-8, 1e-4, 1e-4
KAPPA_CAP = 1e-5
OBJ_TIGHT, RES_TIGHT = 1e-7, 1e-7
FLAT_TIGHT = 100 * TOL / RHO                     # 1e-2


def pair_allowance(st_a, st_b, kkt_a, kkt_b):
    """(objective, residual, displacement) a pair of usable answers may differ by, from the table above."""
    if st_a == 0 and st_b == 0:
        return OBJ_TIGHT, RES_TIGHT, FLAT_TIGHT
    kap = min(KAPPA_CAP, max(float(kkt_a) if st_a == 3 else 0.0, float(kkt_b) if st_b == 3 else 0.0))
    return max(OBJ_TIGHT, 10 * kap), max(RES_TIGHT, kap), max(FLAT_TIGHT, 10 * kap / RHO)


def _explain_outliers(oracle, cs, spec, rec, got, ref, idx, st, st_ref, kkt, kkt_ref, uprox=None, obj_tol=OBJ_TIGHT):
    """Every pair of solutions further apart than the north-star tolerance must be the SAME optimum seen from two points
    of a flat valley: equal objective value, dynamics satisfied, inequalities satisfied, displacement within what the
    curvature RHO allows at the pair's KKT error (`pair_allowance`).  Nothing else may pass."""
    nU = 20 * (spec.N + 1)
    for i in idx:
        up = None if uprox is None else uprox[i, nU:]
        a_obj, a_res, a_flat = pair_allowance(st[i], st_ref[i], kkt[i], kkt_ref[i])
        f_g, def_g, ineq_g, act_g = oracle.evaluate(cs, rec[i], got[i], uprox=up)
        f_r, def_r, ineq_r, act_r = oracle.evaluate(cs, rec[i], ref[i], uprox=up)
        assert abs(f_g - f_r) <= max(obj_tol, a_obj) * max(1.0, abs(f_r)), (i, f_g, f_r, st[i], st_ref[i], kkt[i], kkt_ref[i])
        assert np.abs(def_g).max() < a_res, (i, np.abs(def_g).max(), a_res)
        assert ineq_g[act_g == 1].max() <= a_res, (i, ineq_g[act_g == 1].max(), a_res)
        assert rel_inf(got[i], ref[i])[0] <= a_flat, (i, rel_inf(got[i], ref[i])[0], a_flat)


# Population levels per problem class: (median over usable pairs, median over tight pairs, q90 over tight pairs, share of
# pairs beyond 1e-4, objective agreement demanded of those).  Measured: tools/parity_report.py, profiles/r04m_parity_report.txt.
#   nominal   delta = 0.01 s, N <= 20 (the north-star case): rounding level, outliers = flat valleys
#   long      N = 40: rounding errors of 40 stages add up; same rule for the outliers
#   rate10    mpc_rate = 10 (delta = 0.1 s, horizons of 1-2 s over several steps, no force-rate cost): a fifth of
#             the instances stop at the acceptable level (KKT ~1e-6) in one solver or the other (the exact
#             Hessian is indefinite at the stationary point itself); one pair in ~60 sits in two different local minima
#             (objectives 8e-6 apart, both converged) -- the NLP is not convex
LEVELS = {"nominal": (1e-9, 1e-9, 1e-6, 0.06, OBJ_TIGHT), "long": (1e-6, 1e-9, 1e-4, 0.10, OBJ_TIGHT),
          "rate10": (1e-4, 1e-6, 1e-3, 0.40, 2e-5)}          # (share beyond 1e-4 measured: 0.24 at N = 10, 0.33 at N = 20)

# Both solver kernels of the 4-vertex solver are held to every oracle-parity test: the one-wavefront kernel (what
# `bench.py` times at B = 8192) and the pipelined pair (what small batches run), chosen through cmpc_spec.kernel.
KERNELS = {"single": 1, "pair": 2}
KERNEL_NAMES = {("single", 4): "cmpc_solve_kernel<4, 1>", ("pair", 4): "cmpc_solve_pair_kernel<4, 2>",
                ("single", 8): "cmpc_solve_kernel<8, 2>"}
CASES = [("perturbed", 256, 20, 1), ("payload", 512, 20, 1), ("randomized", 512, 20, 1), ("perturbed", 128, 10, 1),
         ("perturbed", 64, 3, 1), ("perturbed", 32, 40, 1), ("long_horizon", 64, 10, 1), ("long_horizon", 512, 40, 1),
         ("perturbed", 128, 10, 10), ("perturbed", 64, 20, 10)]
# (the N = 40, 8-vertex case ran 48 instances in rounds 2-4: too few to carry a 10 % share -- the same population gives 3 of 47
# and 5 of 47 beyond 1e-4 with two kernels whose rate at 512 instances is 5.3 % and 5.7 %; round 5, gpurun_out/r05n -- it
# runs 512 now, the level is the table's)
CASES_BY_KERNEL = [c + (k,) for c in CASES for k in (("single",) if c[0] == "long_horizon" else ("single", "pair"))]


@pytest.mark.parametrize("name,B,N,rate,kernel", CASES_BY_KERNEL)
def test_parity_with_oracle(gpu, oracle, name, B, N, rate, kernel):
    spec, rec = wl.make_workload(name, B=B, N=N, rate=rate)
    if N > 20:
        spec.max_iter = 150                                   # long horizons take more iterations
    med_all, med_tight, q90_tight, share, obj_tol = LEVELS["rate10" if rate == 10 else "long" if N > 20 else "nominal"]
    cs = oracle_spec(oracle, spec)
    got, st, it, kkt = _solve(gpu, spec, rec, kernel=kernel)
    ref, st_ref, it_ref, kkt_ref = oracle.solve_batch(cs, rec)
    ok_g, ok_r = np.isin(st, (0, 3)), np.isin(st_ref, (0, 3))
    # same verdict (usable: converged / acceptable, or not) on (nearly) every instance; whether a slowly
    # converging instance ends as 0 or as 3 depends on rounding (eight iterates in a row within 1e-6)
    assert (ok_g != ok_r).sum() <= max(2, 0.03 * B)
    both = ok_g & ok_r
    assert both.mean() >= (0.85 if rate == 1 else 0.7)        # rate 10: more infeasible draws
    err = rel_inf(got[both], ref[both])
    tight = (st == 0) & (st_ref == 0)
    err_t = rel_inf(got[tight], ref[tight])
    assert np.median(err) < med_all and np.median(err_t) < med_tight and np.quantile(err_t, 0.9) < q90_tight
    assert err_t.max() <= FLAT_TIGHT                          # hard ceiling on every pair of converged points
    # north star: within 1e-4 rel-inf.  Pairs beyond it are allowed only if they are explained (same optimum,
    # flat valley) -- checked for every one of them -- and they are few.
    out = np.where(both)[0][err >= REL_TOL]
    assert len(out) <= share * both.sum(), (len(out), int(both.sum()))
    _explain_outliers(oracle, cs, spec, rec, got, ref, out, st, st_ref, kkt, kkt_ref, obj_tol=obj_tol)


def test_parity_sample_from_a_launch_beyond_the_pair_threshold(gpu, oracle):
    """The library's own choice at the bench's batch size (B = 8192 > 14 instances per CU) is the one-wavefront kernel with
    its ticket queue running well past the resident grid: a sample of that launch -- early and late queue positions --
    against the oracle, same levels as above."""
    spec, rec = wl.make_workload("randomized", B=8192, N=20)
    s = gpu(spec, device="cuda:0")
    out, st, it, kkt = s.solve(torch.from_numpy(rec).to("cuda:0"))
    torch.cuda.synchronize()
    assert s.last_kernel_name() == "cmpc_solve_kernel<4, 1>"
    got, st, kkt = out.cpu().numpy(), st.cpu().numpy(), kkt.cpu().numpy()
    idx = np.concatenate([np.arange(0, 8192, 32), np.arange(8000, 8192)])
    cs = oracle_spec(oracle, spec)
    ref, st_ref, _, kkt_ref = oracle.solve_batch(cs, rec[idx])
    ok = np.isin(st[idx], (0, 3)) & np.isin(st_ref, (0, 3))
    assert (np.isin(st[idx], (0, 3)) != np.isin(st_ref, (0, 3))).sum() <= 2 and ok.mean() > 0.9
    err = rel_inf(got[idx][ok], ref[ok])
    med_all, _, _, share, obj_tol = LEVELS["nominal"]
    assert np.median(err) < med_all
    far = np.where(ok)[0][err >= REL_TOL]
    assert len(far) <= share * ok.sum()
    _explain_outliers(oracle, cs, spec, rec[idx], got[idx], ref, far, st[idx], st_ref, kkt[idx], kkt_ref, obj_tol=obj_tol)


def _kernels_of(path):
    return ("single", "pair") if int(np.load(path)["nv"]) == 4 else ("single",)


@pytest.mark.parametrize("path,kernel", [(p, k) for p in sorted(glob.glob(os.path.join(GOLD, "solver_kat_*.npz"))) for k in _kernels_of(p)],
                         ids=lambda v: os.path.basename(v) if v.endswith(".npz") else v)
def test_golden_vectors(gpu, path, kernel):
    kat = np.load(path)
    spec = ProblemSpec(N=int(kat["N"]), nv=int(kat["nv"]), k1=float(kat["k1"]), k2=float(kat["k2"]),
                       tol=1e-10, max_iter=300)
    got, st, it, kkt = _solve(gpu, spec, kat["records"], kernel=kernel)
    assert np.isin(st, (0, 3)).all() and kkt.max() <= 1e-8
    assert rel_inf(got, kat["solutions"]).max() < 1e-5


@pytest.mark.parametrize("path,kernel", [(p, k) for p in sorted(glob.glob(os.path.join(GOLD, "independent_pin_*.npz"))) for k in _kernels_of(p)],
                         ids=lambda v: os.path.basename(v)[16:-4] if v.endswith(".npz") else v)
def test_independent_pins(gpu, oracle, path, kernel):
    """The HIP solver against solutions computed without the C oracle (dense interior point on the literal torch
    restatement, scipy trust-constr): tests/test_independent_pins.py, tests/golden/make_independent_pins.py."""
    from test_independent_pins import check_against_pin
    pin = np.load(path)
    # (a pin is a solution to 1e-9: the solver is asked for a point of that quality -- an "acceptable" exit only within
    # 1e-8, not at the default acceptable level 1e-4, which determines the flat directions to 1e-4 / 1e-4 = 1 only)
    spec = ProblemSpec(N=int(pin["N"]), nv=int(pin["nv"]), k1=float(pin["k1"]), k2=float(pin["k2"]), tol=1e-9, max_iter=200,
                       acc_tol=1e-8)
    got, st, it, kkt = _solve(gpu, spec, pin["record"][None, :], kernel=kernel)
    # (status 0: the tolerance was met, the polish step that follows may leave up to 100 * tol; status 3: within acc_tol)
    assert (st[0] == 0 and kkt[0] < 1e-7) or (st[0] == 3 and kkt[0] <= 1e-8), (st, kkt)
    cs = oracle_spec(oracle, spec)
    check_against_pin(pin, got[0], lambda w: oracle.evaluate(cs, pin["record"], w), kkt=float(kkt[0]))


@pytest.mark.parametrize("kernel", ["single", "pair"])
def test_flat_directions_of_the_pinned_optimum_converge_with_the_tolerance(gpu, oracle, kernel):
    """tests/test_independent_pins.py::flat_convergence for the HIP solver: on the N = 20 touch-down pin (refined to 2e-11)
    the foot-velocity group moves towards the pin as the tolerance is tightened from 1e-9 to 1e-10."""
    from test_independent_pins import flat_convergence
    pin = np.load(os.path.join(GOLD, "independent_pin_N20_t255_switch.npz"))

    def solve_at(tol, acc_tol):
        spec = ProblemSpec(N=int(pin["N"]), nv=int(pin["nv"]), k1=float(pin["k1"]), k2=float(pin["k2"]), tol=tol, max_iter=200, acc_tol=acc_tol)
        got, st, it, kkt = _solve(gpu, spec, pin["record"][None, :], kernel=kernel)
        assert st[0] in (0, 3), (st, kkt)
        return got[0]
    flat_convergence(solve_at, pin)


@pytest.mark.parametrize("kernel", ["single", "pair"])
def test_edge_cases_empty_single_and_ragged_batches(gpu, oracle, kernel):
    spec, rec = wl.make_workload("perturbed", B=67, N=10, scale=0.5)
    ref, st_ref, _, kkt_ref = oracle.solve_batch(oracle_spec(oracle, spec), rec)
    got0 = _solve(gpu, spec, rec[:0], kernel=kernel)
    assert got0[0].shape == (0, spec.nsol) and got0[1].shape == (0,)
    cs = oracle_spec(oracle, spec)
    for B in (1, 2, 63, 65, 67):                             # around the wavefront width
        got, st, _, kkt = _solve(gpu, spec, rec[:B], kernel=kernel)
        ok = np.isin(st, (0, 3)) & np.isin(st_ref[:B], (0, 3))
        err = rel_inf(got[ok], ref[:B][ok])
        # all but the occasional flat-direction instance (curvature = the 1e-4 proximal weight against a KKT tolerance of
        # 1e-8: a displacement of 1e-4 is within the tolerance) agree to rounding level; a pair further apart than the
        # north-star tolerance must be the same optimum -- equal objective, feasible, within the displacement its KKT
        # error allows (`pair_allowance`; for two converged points 1e-2 at the very most) -- and there may be one in a batch
        assert ok.mean() > 0.9 and np.median(err) < 1e-9
        far = np.where(ok)[0][err >= REL_TOL]
        assert len(far) <= max(1, 0.03 * B)
        _explain_outliers(oracle, cs, spec, rec[:B], got, ref[:B], far, st, st_ref[:B], kkt, kkt_ref[:B])


@pytest.mark.parametrize("kernel", ["single", "pair"])
def test_batch_composition_does_not_change_results(gpu, kernel):
    """Instances are independent: results are bitwise identical whatever else is in the batch, in
    whatever order, on every run (ticket order and slab reuse must not leak between instances)."""
    spec, rec = wl.make_workload("randomized", B=3000, N=20)  # > resident grid of either kernel: slabs are reused
    a, st_a, it_a, _ = _solve(gpu, spec, rec, kernel=kernel)
    b, st_b, it_b, _ = _solve(gpu, spec, rec, kernel=kernel)
    assert np.array_equal(a, b) and np.array_equal(it_a, it_b)
    perm = np.random.default_rng(0).permutation(rec.shape[0])
    c, st_c, it_c, _ = _solve(gpu, spec, rec[perm], kernel=kernel)
    assert np.array_equal(c, a[perm]) and np.array_equal(st_c, st_a[perm])
    d, _, _, _ = _solve(gpu, spec, rec[100:164], kernel=kernel)
    assert np.array_equal(d, a[100:164])


def test_batch_composition_eight_vertex_kernel_beyond_its_resident_grid(gpu, oracle):
    """The same for cmpc_solve_kernel<8> (BASELINE config 5: N = 40, 8 vertices per foot) with more instances than
    its 512 resident slots (2 workgroups per CU): ticket and slab reuse at NV = 8, which no test exercised in round 2.
    Bitwise batch-composition independence, and a sample against the oracle."""
    spec, rec = wl.make_workload("long_horizon", B=1152)
    spec.max_iter = 150
    assert spec.nv == 8 and spec.N == 40
    a, st_a, it_a, kkt_a = _solve(gpu, spec, rec)
    assert np.isin(st_a, (0, 3)).mean() > 0.95 and (st_a == 1).sum() == 0
    perm = np.random.default_rng(1).permutation(rec.shape[0])
    c, st_c, it_c, _ = _solve(gpu, spec, rec[perm])
    assert np.array_equal(c, a[perm]) and np.array_equal(st_c, st_a[perm]) and np.array_equal(it_c, it_a[perm])
    d, st_d, _, _ = _solve(gpu, spec, rec[1000:1040])        # alone (first use of their slabs) = as part of the big batch
    assert np.array_equal(d, a[1000:1040])
    idx = np.arange(1100, 1124)                                # instances drawn late from the queue: reused slabs
    ref, st_ref, _, _ = oracle.solve_batch(oracle_spec(oracle, spec), rec[idx])
    both = np.isin(st_a[idx], (0, 3)) & np.isin(st_ref, (0, 3))
    assert both.mean() > 0.9
    err = rel_inf(a[idx][both], ref[both])
    med_all, _, _, share, _ = LEVELS["long"]                  # the parity levels of this problem class (above)
    assert np.median(err) < med_all and (err < REL_TOL).mean() >= 1.0 - share - 0.05


def test_eight_vertex_two_wave_kernel_edge_cases(gpu, oracle):
    """cmpc_solve_kernel<8, 2> (128-lane workgroups) at the edges: batch sizes 0, 1, 2, 3 and a ragged 67, the smallest
    and the largest horizon of the build (N = 1, N = CMPC_MAX_N = 64), against the oracle."""
    spec, rec = wl.make_workload("long_horizon", B=67, N=10)
    assert spec.nv == 8
    ref, st_ref, it_ref, _ = oracle.solve_batch(oracle_spec(oracle, spec), rec)
    assert _solve(gpu, spec, rec[:0])[0].shape == (0, spec.nsol)
    for B in (1, 2, 3, 67):
        got, st, it, _ = _solve(gpu, spec, rec[:B])
        ok = np.isin(st, (0, 3)) & np.isin(st_ref[:B], (0, 3))
        err = rel_inf(got[ok], ref[:B][ok])
        assert ok.mean() > 0.9 and np.median(err) < 1e-8 and (err < REL_TOL).mean() > 0.9
        # (the end game of a few instances is sensitive to rounding order: iteration counts agree on nine in ten)
        assert (np.abs(it[ok] - it_ref[:B][ok]) <= 1).mean() >= 0.9 or B < 10
    for N, B in ((1, 5), (64, 2)):
        spec, rec = wl.make_workload("long_horizon", B=B, N=N)
        spec.max_iter = 150
        got, st, it, _ = _solve(gpu, spec, rec)
        ref, st_ref, it_ref, _ = oracle.solve_batch(oracle_spec(oracle, spec), rec)
        assert np.array_equal(np.isin(st, (0, 3)), np.isin(st_ref, (0, 3))) and np.isin(st, (0, 3)).any()
        ok = np.isin(st, (0, 3))
        for i in np.flatnonzero(ok):                       # same optimum: objective and feasibility (flat valleys at N = 64)
            f_g, d_g, _, _ = oracle.evaluate(oracle_spec(oracle, spec), rec[i], got[i])
            f_r, _, _, _ = oracle.evaluate(oracle_spec(oracle, spec), rec[i], ref[i])
            assert abs(f_g - f_r) <= 1e-6 * max(1.0, abs(f_r)) and np.abs(d_g).max() < 1e-7


@pytest.mark.parametrize("kernel", ["single", "pair"])
def test_two_handles_on_two_streams_overlap_without_interference(gpu, kernel):
    """bench.py alternates consecutive batches over two solver handles on two HIP streams (the straggler
    tail of one launch overlaps the next).  Handles share nothing: the overlapped results are bitwise the
    serial ones."""
    spec, rec_a = wl.make_workload("randomized", B=2600, N=20)   # > resident grid, so the launches really overlap
    rec_b = rec_a[::-1].copy()
    ref_a, st_a, _, _ = _solve(gpu, spec, rec_a, kernel=kernel)
    ref_b, st_b, _, _ = _solve(gpu, spec, rec_b, kernel=kernel)
    d_a, d_b = (torch.from_numpy(r).to("cuda:0") for r in (rec_a, rec_b))
    solvers = [gpu(dataclasses.replace(spec, kernel=KERNELS[kernel]), device="cuda:0") for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    outs = []
    for j, d in enumerate((d_a, d_b)):
        with torch.cuda.stream(streams[j]):
            outs.append(solvers[j].solve(d))
    torch.cuda.synchronize()
    assert np.array_equal(outs[0][0].cpu().numpy(), ref_a) and np.array_equal(outs[0][1].cpu().numpy(), st_a)
    assert np.array_equal(outs[1][0].cpu().numpy(), ref_b) and np.array_equal(outs[1][1].cpu().numpy(), st_b)


@pytest.mark.parametrize("kernel", ["single", "pair"])
def test_warm_start_parity_and_speedup(gpu, oracle, kernel):
    spec, rec = wl.make_workload("perturbed", B=64, N=20, scale=0.5)
    cs = oracle_spec(oracle, spec)
    cold, st0, it0, _ = oracle.solve_batch(cs, rec)
    got, st, it, kkt = _solve(gpu, spec, rec, warm=cold, kernel=kernel)
    ref, st_ref, it_ref, kkt_ref = oracle.solve_batch(cs, rec, warm=cold)
    both = np.isin(st, (0, 3)) & np.isin(st_ref, (0, 3)) & np.isin(st0, (0, 3))
    assert both.mean() > 0.9
    # re-centred proximal term: curvature along the flat directions is the 1e-4 proximal weight, so with
    # a KKT tolerance of 1e-8 two correct solvers may differ by ~1e-4 there.  Asserted: the bulk agrees to
    # rounding level, 90 % within the north-star tolerance, and EVERY pair of solutions has the same
    # objective value and dynamics defect (same optimum, different point of the flat valley).
    err = rel_inf(got[both], ref[both])
    assert np.median(err) < 1e-9 and np.quantile(err, 0.9) < REL_TOL
    _explain_outliers(oracle, cs, spec, rec, got, ref, np.where(both)[0], st, st_ref, kkt, kkt_ref, uprox=cold)


def test_full_size_properties_domain_randomised(gpu):
    """BASELINE config 4 shard (65536 / 8 GPUs): properties that need no oracle."""
    spec, rec = wl.make_workload("randomized", B=8192, N=20)
    got, st, it, kkt = _solve(gpu, spec, rec)
    conv = np.isin(st, (0, 3))
    assert conv.mean() > 0.9                                 # the rest are reported as max-iter / locally infeasible
    assert np.isfinite(got).all()
    r = nlp_batch.residuals(spec, rec[conv], got[conv])
    assert r["x0"].max() == 0.0                              # x_0 is copied, not solved for
    assert r["defect"].max() < 1e-7                          # forward-Euler dynamics hold
    assert r["cone"].max() < 1e-5 and r["unilateral"].max() < 1e-5
    assert r["height"].max() < 1e-6 and r["box"].max() < 1e-6
    assert r["lyapunov"].max() < 1e-5 and r["contraction"].max() < 1e-6
    assert r["swing_force"].max() < 1e-6                     # feet in the air carry nothing
    assert (kkt[st == 0] <= 100 * spec.tol).all() and (kkt[st == 3] <= spec.acc_tol).all() and (it[conv] <= spec.max_iter).all()


def test_shipped_queue_order_is_not_worse_than_the_input_order(gpu):
    """The queue-order predictor is a least-squares fit (csrc/cmpc_order_fit.h, tools/fit_queue_order.py) on a seed that is
    not a BASELINE seed; it silently goes stale when the barrier schedule, the tolerance or the start changes.  Replayed
    on the iteration counts of a real launch of the bench workload (held out from the fit), the shipped order must not
    lose against the order the instances came in."""
    import heapq
    from cmpc_amd import queue_order as qo
    spec, rec = wl.make_workload("randomized", B=8192)
    _, st, it, _ = _solve(gpu, spec, rec)
    slots = 256 * 7

    def makespan(order):
        h = [0] * slots
        heapq.heapify(h)
        end = 0
        for i in order:
            t = heapq.heappop(h) + int(it[i])
            end = max(end, t)
            heapq.heappush(h, t)
        return end
    shipped = np.argsort(-qo.bucket_of(qo.predicted_iterations(rec, spec)), kind="stable")
    m_in, m_ship = makespan(np.arange(8192)), makespan(shipped)
    assert m_ship <= m_in, (m_ship, m_in)
    assert np.corrcoef(qo.predicted_iterations(rec, spec), it)[0, 1] > 0.4


def test_drop_in_class_matches_reference_call_sites(gpu, oracle, scene):
    """``centroidal_mpc(initial, planner, params, CoM_ref, trj_l, trj_r).solve(current, t)`` exactly as
    code/simulation.py:143-150 and :204 call it, for consecutive ticks with warm start."""
    import copy
    import centroidal_mpc_vertices
    from cmpc_amd.footstep_planner_vertices import FootstepPlanner
    from cmpc_amd.problem import build_record
    params = wl.default_params(N=10)
    planner = FootstepPlanner(wl.VREF, wl.LFOOT0, wl.RFOOT0, params)
    mpc = centroidal_mpc_vertices.centroidal_mpc(scene.initial, planner, params, scene.com_ref, None, None)
    spec = mpc.spec
    cs = oracle_spec(oracle, spec)
    warm, theta = None, np.zeros(3)
    for t in (205, 206, 207):                                # early single support (left foot down)
        com, dcom = scene.nominal_state(np.array([t]))
        current = {'com': {'pos': com[0] + [0.004, -0.003, 0.0], 'vel': dcom[0]}, 'hw': {'val': np.array([0.02, -0.01, 0.0])},
                   'lfoot': {'pos': wl.LFOOT0}, 'rfoot': {'pos': wl.RFOOT0}}
        state, contact = mpc.solve(copy.deepcopy(current), t)
        assert state is mpc.model_state and contact == 'lfoot'
        rec = build_record(spec, planner, scene.com_ref, t, current['com']['pos'], current['com']['vel'],
                           current['hw']['val'], theta, wl.LFOOT0[2], wl.RFOOT0[2], params['mass'])
        ref, st, _, _ = oracle.solve(cs, rec, warm=warm)
        assert st == 0
        warm = ref
        X = ref[:20 * 11].reshape(11, 20)
        theta = X[1, 9:12].copy()
        assert np.abs(state['com']['pos'] - X[1, 0:3]).max() < 1e-8
        assert np.abs(state['hw']['val'] - X[1, 6:9]).max() < 1e-7
        assert np.abs(state['theta_hat']['val'] - X[1, 9:12]).max() < 1e-9
        assert state['com']['acc'].shape == (3,) and state['counter']['val'] == 0
    with pytest.raises(RuntimeError):
        bad = copy.deepcopy(current); bad['com']['pos'] = np.array([0.3, 0.0, 0.95])   # above the height bound
        mpc.solve(bad, 208)


def test_device_record_builder_is_bit_exact(gpu, scene):
    """Gather kernel vs the host builder (itself pinned to the scalar front half of ``solve`` and to
    the reference's fixtures): integer / copy work, so the bar is bit-exact."""
    from cmpc_amd.solver import DeviceRecordBuilder
    rng = np.random.default_rng(3)
    for N, rate in ((10, 1), (20, 1), (40, 1), (10, 10), (20, 10)):     # rate 10: the reference's mpc_rate = 10 sampling
        spec = ProblemSpec(N=N)
        B = 3001
        t = rng.integers(0, scene.t_max(N, rate) + 1, size=B).astype(np.int32)
        t[:3] = [0, 199, scene.t_max(N, rate)]               # edges of the tick range
        state = rng.normal(size=(B, 16))
        want = scene.build_records(spec, t, state[:, 0:3], state[:, 3:6], state[:, 6:9], state[:, 9:12], state[:, 12],
                                   state[:, 13], state[:, 14], state[:, 15], rate=rate)
        bld = DeviceRecordBuilder(scene, device="cuda:0")
        got = bld.build(spec, torch.from_numpy(t).to("cuda:0"), torch.from_numpy(state).to("cuda:0"), rate=rate)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), want)
    # out-of-range ticks are flagged, not read out of bounds
    bad = torch.tensor([scene.T, -1], dtype=torch.int32, device="cuda:0")
    out = bld.build(spec, bad, torch.zeros((2, 16), dtype=torch.float64, device="cuda:0"))
    assert torch.isnan(out).all()
    assert bld.build(spec, bad[:0], torch.zeros((0, 16), dtype=torch.float64, device="cuda:0")).shape == (0, spec.nrec)


def test_device_builder_serves_the_reference_generator_tables(gpu):
    """SURVEY 8f row 3 (code/functions.py:11-124, :129-248): the CoM reference generator stays on the host (it runs once,
    offline), and what the hot path consumes of it are the nine per-tick lists.  Fed with the lists of a walk that is
    NOT the shipped one (other velocity commands, so another plan, other spline coefficients), the device builder must
    hand every solver record exactly the words `functions.references` returned for ticks t + 1 ... t + N -- read here
    straight from the generator's dict, not from the scene's table -- and, read back tick by tick, the reference is
    still the generator's C2 quintic spline (no jump in position, velocity or acceleration at the knots)."""
    from cmpc_amd import workloads as wl
    from cmpc_amd.functions import references
    from cmpc_amd.footstep_planner_vertices import FootstepPlanner
    from cmpc_amd.foot_trajectory_generator import FootTrajectoryGenerator
    from cmpc_amd.solver import DeviceRecordBuilder
    sc = wl.Scene()
    vref = [(0.1, 0.0, 0)] * 6 + [(0.2, 0.0, 0)] * 6 + [(0.05, 0.0, 0)] * 5 + [(0.0, 0, 0)] * 3
    sc.planner = FootstepPlanner(vref, wl.LFOOT0, wl.RFOOT0, sc.params)
    sc.ftg = FootTrajectoryGenerator(sc.initial, sc.planner, sc.params)
    sc.com_ref = references(sc.ftg, sc.planner)
    sc.refresh_tables()
    assert not np.array_equal(sc.com_tab[:1500], wl.scene().com_tab[:1500])        # really another walk
    keys = ('pos_x', 'pos_y', 'pos_z', 'vel_x', 'vel_y', 'vel_z', 'acc_x', 'acc_y', 'acc_z')
    gen = np.stack([np.asarray(sc.com_ref[k], dtype=np.float64)[:sc.T] for k in keys], axis=1)
    bld = DeviceRecordBuilder(sc, device="cuda:0")
    for N in (10, 20):
        spec = ProblemSpec(N=N)
        t = np.arange(0, sc.t_max(N) + 1, dtype=np.int32)                           # every admissible tick
        state = np.zeros((t.shape[0], 16))
        got = bld.build(spec, torch.from_numpy(t).to("cuda:0"), torch.from_numpy(state).to("cuda:0")).cpu().numpy()
        st = got[:, 24:].reshape(t.shape[0], N, 19)
        want = gen[t[:, None] + 1 + np.arange(N)[None, :]]                          # ticks t + (1 + i), i < N (:548-577)
        assert np.array_equal(st[:, :, 0:9], want)
    # the table as the device serves it, tick by tick (stage 0 of the record of tick t is tick t + 1)
    tab = st[:, 0, 0:9]
    for a in (0, 1):                           # x, y: no jump at the knots in position, velocity (per step time, :222) or
        pos, vel, acc = tab[:, a], tab[:, 3 + a], tab[:, 6 + a]                   # acceleration (per tick^2, :243)
        assert np.abs(np.diff(pos)).max() < 3e-3 and np.abs(np.diff(vel)).max() < 1e-2 and np.abs(np.diff(acc)).max() < 1e-4
    assert np.all(tab[:, 2] == 0.72) and np.all(tab[:, 5] == 0.0) and np.all(tab[:, 8] == 0.0)   # functions.py:97-99


def test_batched_closed_loop_rollout(gpu, scene):
    """Config-1 analogue as parallel rollouts: the nominal walk through the first step (double support,
    lift-off, early single support) with measured-momentum-like perturbations, and a pushed copy."""
    from cmpc_amd.rollout import BatchedRollout
    spec = ProblemSpec(N=10)
    B = 16
    ro = BatchedRollout(scene, spec, B, device="cuda:0")
    t0 = 180
    com, dcom = scene.nominal_state(np.full(B, t0))
    rng = np.random.default_rng(11)
    com = com + rng.uniform(-0.003, 0.003, size=(B, 3))
    hw = rng.normal(0, 0.05, size=(B, 3))
    ro.reset(t0, com, dcom, hw=hw)
    ticks = 40
    hist, alive = ro.run(ticks, push=(10, 14, [0.0, 0.004, 0.0]))
    hist = hist.cpu().numpy()
    assert alive.all().item()                                 # every tick solved for every instance
    ref = np.stack([scene.com_tab[t0 + i, 0:3] for i in range(ticks + 1)])
    err = np.abs(hist - ref[:, None, :])
    assert err[..., :2].max() < 0.05 and err[..., 2].max() < 0.02   # the Lyapunov controller holds the reference
    assert int(ro.t[0].item()) == t0 + ticks


def test_pair_kernel_is_bitwise_the_single_wave_kernel(gpu):
    """Small batches run two waves per instance (cmpc_solve_pair_kernel: evaluation of stage k - 1 beside the Riccati step of
    stage k).  Same arithmetic, same order: everything the call returns must be bit for bit what the one-wave kernel
    returns -- B = 1, a batch larger than the pair kernel's resident grid (several instances through one workgroup),
    cold and resumed from the solver state, the payload gains, the horizon limits of the build."""
    def both(spec, rec, warm=None, state=None):
        res = []
        for kern in ("single", "pair"):                        # cmpc_spec.kernel, fixed when the handle is created
            s = gpu(dataclasses.replace(spec, kernel=KERNELS[kern]), device="cuda:0")
            d = torch.from_numpy(np.ascontiguousarray(rec)).cuda()
            so = s.new_state(rec.shape[0])
            out, st, it, kkt = s.solve(d, warm=warm, state=state, state_out=so)
            torch.cuda.synchronize()
            assert s.last_kernel_name() == KERNEL_NAMES[(kern, 4)]
            res.append((out.clone(), st.clone(), it.clone(), kkt.clone(), so.clone(), s.last_kernel_ms()))
        for x, y in zip(res[0][:5], res[1][:5]):
            assert torch.equal(x, y)
        return res
    for name, B, N in (("randomized", 1, 20), ("randomized", 1100, 20), ("payload", 96, 10), ("perturbed", 5, 1), ("perturbed", 3, 64)):
        spec, rec = wl.make_workload(name, B=B, N=N)
        a = both(spec, rec)
        assert np.isin(a[0][1].cpu().numpy(), (0, 3)).mean() > 0.9
        b = both(spec, rec, warm=a[0][0], state=a[0][4])       # resumed
        if B == 1 and N == 20:
            assert a[1][5] < 0.85 * a[0][5], (a[0][5], a[1][5])  # and an instance alone on the GPU finishes sooner
    # the entry point picks the pair kernel by itself when the batch does not fill the GPU: same results either way
    spec, rec = wl.make_workload("randomized", B=64, N=20)
    auto = gpu(spec, device="cuda:0")
    o1, s1, i1, _ = auto.solve(torch.from_numpy(rec).cuda())
    single = gpu(dataclasses.replace(spec, kernel=KERNELS["single"]), device="cuda:0")
    o2, s2, i2, _ = single.solve(torch.from_numpy(rec).cuda())
    torch.cuda.synchronize()
    assert auto.last_kernel_name() == KERNEL_NAMES[("pair", 4)] and single.last_kernel_name() == KERNEL_NAMES[("single", 4)]
    assert torch.equal(o1, o2) and torch.equal(i1, i2)
    assert auto.last_kernel_ms() < 0.9 * single.last_kernel_ms()
