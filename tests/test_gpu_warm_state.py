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
    _, sol_o, it_o = _loop(ora_solve, N, t0, ticks, use_state=True)
    # (the end game of a solve is sensitive to rounding order: a step more or less here and there)
    assert np.abs(it_g - it_o).max() <= 4 and abs(int(it_g.sum()) - int(it_o.sum())) <= max(6, 0.1 * it_o.sum()), (it_g, it_o)
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
