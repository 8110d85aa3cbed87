"""Whole-body inverse-dynamics QP (reference code/inverse_dynamics.py:92-134, code/utils.py:40-92).
CPU tier: the numpy oracle against the KKT conditions of the reference's own 72-variable statement (the QP is convex:
a KKT point IS the solution, whatever found it), the host-side cost assembly against a literal loop, the C ABI exports.
GPU tier: the HIP kernel against the oracle through the C ABI."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from cmpc_amd import capi, wbc, workloads as wl
from oracle import wbc_qp_oracle as wq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("contact", ["ds", "lfoot", "rfoot"])
def test_oracle_solution_satisfies_the_reference_statement(contact):
    Hq, Fq, M, h, Jc = wl.wbc_synthetic(6, seed=3, contact=contact)
    for b in range(6):
        r = wq.solve(Hq[b], Fq[b], M[b], h[b], Jc[b], 0.05, 0.5)
        assert r["status"] == 0 and r["iters"] <= 40
        k = wq.kkt_full(Hq[b], Fq[b], M[b], h[b], Jc[b], 0.05, 0.5, r["qdd"], r["tau"], r["f"])
        assert k["stationarity"] < 1e-6 and k["equality"] < 1e-8 and k["ineq_violation"] < 1e-9
        # the statement's rows, literally: CoP inside the foot, friction pyramid, unilateral normal force
        for w in (r["f"][0:6], r["f"][6:12]):
            assert abs(w[0]) <= 0.05 * w[5] + 1e-8 and abs(w[1]) <= 0.05 * w[5] + 1e-8
            assert abs(w[3]) <= 0.5 * w[5] + 1e-8 and abs(w[4]) <= 0.5 * w[5] + 1e-8 and w[5] >= -1e-9
        assert np.all(r["tau"][:6] == 0.0)
    # a foot in the air carries (next to) nothing: its Jacobian rows are zero, only the 1e-6 regulariser sees it
    if contact != "ds":
        air = slice(6, 12) if contact == "lfoot" else slice(0, 6)
        assert np.abs(r["f"][air]).max() < 1.0


def test_solution_is_the_minimiser_among_feasible_perturbations():
    """Independent of the KKT algebra: no feasible point nearby has a lower cost."""
    Hq, Fq, M, h, Jc = wl.wbc_synthetic(1, seed=11)
    r = wq.solve(Hq[0], Fq[0], M[0], h[0], Jc[0], 0.05, 0.5)
    cost = lambda q, f: 0.5 * q @ Hq[0] @ q + Fq[0] @ q + 0.5 * wq.F_REG * f @ f
    base = cost(r["qdd"], r["f"])
    Ae = np.hstack([M[0][:6, :], -Jc[0][:, :6].T])
    Ai = wq.ineq_matrix(0.05, 0.5)
    N_ = np.linalg.svd(Ae)[2][6:].T                        # null space of the floating-base rows
    rng = np.random.default_rng(0)
    for _ in range(200):
        dx = N_ @ rng.normal(0, 1e-2, size=N_.shape[1])
        q, f = r["qdd"] + dx[:30], r["f"] + dx[30:]
        if (Ai @ f <= 0).all():
            assert cost(q, f) >= base - 1e-9 * abs(base)


def test_cost_assembly_matches_the_literal_loop():
    rng = np.random.default_rng(5)
    B, rows = 3, {'lfoot': 6, 'rfoot': 6, 'com': 3, 'torso': 3, 'base': 3, 'joints': 30}
    J = {k: rng.normal(size=(B, r, 30)) for k, r in rows.items()}
    Jd = {k: rng.normal(size=(B, r, 30)) for k, r in rows.items()}
    ff, pe, ve = ({k: rng.normal(size=(B, r)) for k, r in rows.items()} for _ in range(3))
    qd = rng.normal(size=(B, 30))
    t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    Hq, Fq = wbc.assemble_task_cost(t(J), t(Jd), t(ff), t(pe), t(ve), torch.from_numpy(qd))
    for b in range(B):
        H, F = np.zeros((30, 30)), np.zeros(30)
        for task in wbc.TASKS:                                # code/inverse_dynamics.py:92-103
            H += wbc.WEIGHTS[task] * J[task][b].T @ J[task][b]
            F += -wbc.WEIGHTS[task] * J[task][b].T @ (ff[task][b] + wbc.VEL_GAINS[task] * ve[task][b]
                                                      + wbc.POS_GAINS[task] * pe[task][b] - Jd[task][b] @ qd[b])
        assert np.allclose(Hq[b].numpy(), H, rtol=1e-12, atol=1e-12) and np.allclose(Fq[b].numpy(), F, rtol=1e-12, atol=1e-12)


def test_header_symbols_are_exported():
    text = open(os.path.join(ROOT, "include", "cmpc_wbc.h")).read()
    names = sorted(set(re.findall(r"\b(cmpc_wbc_[a-z_]+)\s*\(", text)))
    assert set(names) == set(capi.WBC_SYMBOLS)
    raw = ctypes.CDLL(capi.LIB_PATH)
    for n in names:
        assert hasattr(raw, n)


@pytest.mark.gpu
@pytest.mark.parametrize("contact,B", [("ds", 300), ("lfoot", 64), ("rfoot", 64)])
def test_hip_kernel_matches_the_oracle(contact, B):
    Hq, Fq, M, h, Jc = wl.wbc_synthetic(B, seed=21, contact=contact)
    ref = wq.solve_batch(Hq[:48], Fq[:48], M[:48], h[:48], Jc[:48], 0.05, 0.5)
    qp = wbc.BatchedInverseDynamicsQP(foot_size=0.1, mu=0.5, device="cuda:0")
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    tau, qdd, f, st, it = qp.solve(dev(Hq), dev(Fq), dev(M), dev(h), dev(Jc))
    torch.cuda.synchronize()
    assert (st == 0).all() and int(it.max()) <= 45
    tau, qdd, f = tau.cpu().numpy(), qdd.cpu().numpy(), f.cpu().numpy()
    assert (ref["status"] == 0).all()
    scale = lambda a: np.maximum(np.abs(a).max(axis=1, keepdims=True), 1.0)
    assert (np.abs(tau[:48] - ref["tau"][:, 6:]) / scale(ref["tau"])).max() < 1e-6
    assert (np.abs(qdd[:48] - ref["qdd"]) / scale(ref["qdd"])).max() < 1e-6
    assert (np.abs(f[:48] - ref["f"]) / scale(ref["f"])).max() < 1e-6
    assert np.abs(it.cpu().numpy()[:48] - ref["iters"]).max() <= 2
    # every instance of the batch (beyond the oracle's sample): KKT conditions of the reference statement
    for b in (48, B // 2, B - 1):
        t30 = np.concatenate([np.zeros(6), tau[b]])
        k = wq.kkt_full(Hq[b], Fq[b], M[b], h[b], Jc[b], 0.05, 0.5, qdd[b], t30, f[b])
        assert k["stationarity"] < 1e-6 and k["equality"] < 1e-7 and k["ineq_violation"] < 1e-8
    # batch composition does not matter (instances are independent)
    t2 = qp.solve(dev(Hq[7:9]), dev(Fq[7:9]), dev(M[7:9]), dev(h[7:9]), dev(Jc[7:9]))[0].cpu().numpy()
    assert np.array_equal(t2, tau[7:9])


@pytest.mark.gpu
def test_hip_kernel_failure_path_returns_zeros_like_the_reference():
    """code/utils.py:85-92: QPSolver.solve returns zeros when the solver fails.  An indefinite task Hessian (wrong-inertia
    pivot) and a NaN input (non-finite KKT error) must end with status 2 and finite, zero outputs; the good instances
    of the same batch are untouched."""
    Hq, Fq, M, h, Jc = wl.wbc_synthetic(4, seed=5)
    Hq[1] = -Hq[1]                                  # indefinite: the first pivot is negative
    Fq[2, 3] = np.nan                               # non-finite residual
    qp = wbc.BatchedInverseDynamicsQP(foot_size=0.1, mu=0.5, device="cuda:0")
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    tau, qdd, f, st, it = qp.solve(dev(Hq), dev(Fq), dev(M), dev(h), dev(Jc))
    torch.cuda.synchronize()
    assert st.tolist() == [0, 2, 2, 0]
    for b in (1, 2):
        assert float(tau[b].abs().max()) == 0.0 and float(qdd[b].abs().max()) == 0.0 and float(f[b].abs().max()) == 0.0
    assert torch.isfinite(tau).all() and torch.isfinite(qdd).all() and torch.isfinite(f).all()
    bad = wq.solve(Hq[2], Fq[2], M[2], h[2], Jc[2], 0.05, 0.5)               # the oracle takes the same exit
    assert bad["status"] == 2 and not bad["qdd"].any() and not bad["tau"].any()
    ref = wq.solve(Hq[3], Fq[3], M[3], h[3], Jc[3], 0.05, 0.5)
    assert np.abs(qdd[3].cpu().numpy() - ref["qdd"]).max() < 1e-6 * max(1.0, np.abs(ref["qdd"]).max())
