"""Algorithm-independent anchor of the solver: tests/golden/independent_pin_*.npz were computed WITHOUT the C
oracle and without the HIP solver (tests/golden/make_independent_pins.py: a dense full-space interior point on
the literal torch restatement of the NLP with autograd derivatives, and scipy's trust-constr on the same
restatement) for ticks of the flat-ground walk that cover double support, lift-off, early / mid / late single
support and touch-down, nominal and payload gains, N = 10 and N = 20.  The C oracle must reproduce them here;
the HIP solver reproduces them in tests/test_gpu_parity.py::test_independent_pins."""
import glob
import os

import numpy as np
import pytest

from conftest import rel_inf, group_rel_inf

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLD, "independent_pin_*.npz")))


def check_against_pin(pin, sol, evaluate):
    """`sol`: a solver's answer for pin['record'];  evaluate(w) -> (cost, defects, ineq, act)."""
    N = int(pin["N"])
    anchors = 0
    if float(pin["ipm_dense_kkt"]) <= 1e-7:                     # dense interior point, full Newton steps (no safeguards:
        # it may hover just above its 1e-9 tolerance, or fail outright -- then the other solver anchors the case)
        assert rel_inf(sol, pin["sol_ipm_dense"])[0] < 1e-5 and group_rel_inf(sol, pin["sol_ipm_dense"], N, 32)[0] < 1e-5
        anchors += 1
    if "ipm_dense_ls_status" in pin.files and int(pin["ipm_dense_ls_status"]) == 0:   # same, l1 line search
        assert rel_inf(sol, pin["sol_ipm_dense_ls"])[0] < 1e-5
        anchors += 1
    if "sol_trust_constr" in pin.files:
        tc = pin["sol_trust_constr"]
        f_s, f_t = evaluate(sol)[0], evaluate(tc)[0]
        if int(pin["trust_constr_status"]) in (1, 2) and float(pin["trust_constr_optimality"]) < 1e-8:
            assert rel_inf(sol, tc)[0] < 1e-5               # scipy's trust-region interior point, converged
            anchors += 1
        else:
            # stopped at its iteration / trust-radius limit with first-order optimality 1e-4 ... 1e-6: same basin,
            # and nothing it found is better than the solver's point
            assert rel_inf(sol, tc)[0] < 1e-2 and f_s <= f_t + 1e-9 * abs(f_t)
            assert abs(f_s - f_t) <= 1e-6 * abs(f_t)
    assert anchors >= 1
    return anchors


def test_pin_files_cover_the_walk():
    assert len(FILES) >= 10
    whats = " ".join(str(np.load(f)["what"]) for f in FILES)
    for phase in ("double support", "lift-off", "early single support", "late single support", "touch-down", "payload"):
        assert phase in whats
    assert sum(int(np.load(f)["N"]) == 20 for f in FILES) >= 2


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[16:-4] for f in FILES])
def test_oracle_reproduces_independent_pins(oracle, path):
    pin = np.load(path)
    cs = oracle.default_spec(N=int(pin["N"]), nv=4, tol=1e-9, max_iter=200, k1=float(pin["k1"]), k2=float(pin["k2"]))
    sol, st, it, kkt = oracle.solve(cs, pin["record"])
    assert st in (0, 3) and kkt < 1e-7
    check_against_pin(pin, sol, lambda w: oracle.evaluate(cs, pin["record"], w))
