import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import cmpc_amd  # noqa: E402,F401  (package alias)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def scene():
    from cmpc_amd import workloads as wl
    return wl.scene()


@pytest.fixture(scope="session")
def oracle():
    """ctypes binding of the C oracle (built on demand)."""
    import build as _b
    _b.build_oracle()
    from oracle import oracle_lib as ol
    return ol


def oracle_spec(ol, spec, **over):
    kw = dict(N=spec.N, nv=spec.nv, tol=spec.tol, max_iter=spec.max_iter, k1=spec.k1, k2=spec.k2,
              prox=spec.prox, w_rate=spec.w_rate, relax=spec.relax, delta=spec.delta, acc_tol=spec.acc_tol)
    kw.update(over)
    return ol.default_spec(**kw)


def rel_inf(a, b):
    """Per-instance rel-inf error: max|a-b| / max|b|."""
    a, b = np.atleast_2d(a), np.atleast_2d(b)
    return np.abs(a - b).max(axis=1) / np.abs(b).max(axis=1)


def group_rel_inf(a, b, N, nu):
    """Worst per-variable-group rel-inf error (com, vel, hw, theta, feet, forces, foot velocities),
    each group normalised by its own magnitude (floored so that all-zero groups do not blow up)."""
    a, b = np.atleast_2d(a), np.atleast_2d(b)
    B = a.shape[0]
    Xa, Xb = a[:, :20 * (N + 1)].reshape(B, N + 1, 20), b[:, :20 * (N + 1)].reshape(B, N + 1, 20)
    Ua, Ub = a[:, 20 * (N + 1):].reshape(B, N, nu), b[:, 20 * (N + 1):].reshape(B, N, nu)
    groups = [(Xa[..., 0:3], Xb[..., 0:3], 1e-2), (Xa[..., 3:6], Xb[..., 3:6], 1e-2),
              (Xa[..., 6:9], Xb[..., 6:9], 1e-2), (Xa[..., 9:12], Xb[..., 9:12], 1e-3),
              (Xa[..., 12:20], Xb[..., 12:20], 1e-2), (Ua[..., :nu - 8], Ub[..., :nu - 8], 1.0),
              (Ua[..., nu - 8:], Ub[..., nu - 8:], 1e-1)]
    worst = np.zeros(B)
    for ga, gb, floor in groups:
        d = np.abs(ga - gb).reshape(B, -1).max(axis=1)
        s = np.maximum(np.abs(gb).reshape(B, -1).max(axis=1), floor)
        worst = np.maximum(worst, d / s)
    return worst
