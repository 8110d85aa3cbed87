"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/libcmpc_oracle.so (the C oracle).

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product path.
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class CSpec(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int32), ("nv", ctypes.c_int32),
                ("max_iter", ctypes.c_int32), ("struct_size", ctypes.c_int32),
                ("delta", ctypes.c_double), ("g", ctypes.c_double),
                ("k1", ctypes.c_double), ("k2", ctypes.c_double),
                ("w_rate", ctypes.c_double), ("w_hw", ctypes.c_double),
                ("w_cxy", ctypes.c_double), ("w_cz_const", ctypes.c_double),
                ("w_foot", ctypes.c_double), ("w_force", ctypes.c_double),
                ("cz_max", ctypes.c_double), ("box", ctypes.c_double * 3),
                ("foot_length", ctypes.c_double), ("foot_width", ctypes.c_double),
                ("prox", ctypes.c_double), ("relax", ctypes.c_double),
                ("tol", ctypes.c_double), ("acc_tol", ctypes.c_double),
                ("kernel", ctypes.c_int32), ("reserved", ctypes.c_int32)]   # (kernel choice of the HIP library: unused here)


def build(force=False):
    so = os.path.join(_HERE, "libcmpc_oracle.so")
    src = os.path.join(_HERE, "cmpc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcmpc_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def default_spec(N=20, nv=4, **over):
    s = CSpec()
    lib().cmpc_oracle_default_spec(ctypes.byref(s), N, nv)
    for k, v in over.items():
        if k == "box":
            s.box = (ctypes.c_double * 3)(*v)
        else:
            setattr(s, k, v)
    return s


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def nsol(s):
    return 20 * (s.N + 1) + (6 * s.nv + 8) * s.N


def solve(spec, rec, warm=None, verbose=0):
    rec = np.ascontiguousarray(rec, dtype=np.float64)
    warm = None if warm is None else np.ascontiguousarray(warm, dtype=np.float64)
    out = np.zeros(nsol(spec))
    st, it, kkt = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double()
    rc = lib().cmpc_oracle_solve(ctypes.byref(spec), _p(rec), _p(warm), _p(out), ctypes.byref(st),
                                 ctypes.byref(it), ctypes.byref(kkt), int(verbose))
    assert rc == 0
    return out, st.value, it.value, kkt.value


def solve_batch(spec, recs, warm=None, nthreads=0):
    recs = np.ascontiguousarray(recs, dtype=np.float64)
    B = recs.shape[0]
    warm = None if warm is None else np.ascontiguousarray(warm, dtype=np.float64)
    out = np.zeros((B, nsol(spec)))
    st, it, kkt = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B)
    rc = lib().cmpc_oracle_solve_batch(ctypes.byref(spec), B, _p(recs), _p(warm), _p(out), _p(st), _p(it),
                                       _p(kkt), int(nthreads))
    assert rc == 0
    return out, st, it, kkt


def nstate(s):
    return nsol(s) + (s.N + 1) * ((20 + 2 * s.nv) + 2 * (15 + 10 * s.nv) + 2) + 8


def solve_batch_state(spec, recs, warm=None, state=None, nthreads=0, verbose=0):
    """Closed-loop form: (out, state_out, status, iters, kkt); `state` = the previous tick's state_out (or None)."""
    recs = np.ascontiguousarray(np.atleast_2d(recs), dtype=np.float64)
    B = recs.shape[0]
    warm = None if warm is None else np.ascontiguousarray(warm, dtype=np.float64)
    state = None if state is None else np.ascontiguousarray(state, dtype=np.float64)
    out, state_out = np.zeros((B, nsol(spec))), np.zeros((B, nstate(spec)))
    st, it, kkt = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B)
    rc = lib().cmpc_oracle_solve_batch_state(ctypes.byref(spec), B, _p(recs), _p(warm), _p(state), _p(out), _p(state_out),
                                             _p(st), _p(it), _p(kkt), int(nthreads), int(verbose))
    assert rc == 0
    return out, state_out, st, it, kkt


def evaluate(spec, rec, w, uprox=None):
    """(cost, defects (N,20), ineq ((N+1), ni), act) of the C restatement at a full primal point."""
    N, ni = spec.N, 15 + 10 * spec.nv
    rec = np.ascontiguousarray(rec, dtype=np.float64)
    w = np.ascontiguousarray(w, dtype=np.float64)
    cost = ctypes.c_double()
    defect, ineq, act = np.zeros((N, 20)), np.zeros((N + 1, ni)), np.zeros((N + 1, ni), np.int32)
    up = None if uprox is None else np.ascontiguousarray(uprox, dtype=np.float64)
    lib().cmpc_oracle_eval(ctypes.byref(spec), _p(rec), _p(w), _p(up), ctypes.byref(cost), _p(defect), _p(ineq),
                           _p(act))
    return cost.value, defect, ineq, act


def stage(spec, rec, k, x, u, lamn, zmul, x0n2):
    nu = 6 * spec.nv + 8
    nx = 20 + 2 * spec.nv
    nz, ni = nu + nx, 15 + 10 * spec.nv
    x = np.ascontiguousarray(x, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
    lamn = np.ascontiguousarray(lamn, dtype=np.float64); zmul = np.ascontiguousarray(zmul, dtype=np.float64)
    rec = np.ascontiguousarray(rec, dtype=np.float64)
    xn, G = np.zeros(nx), np.zeros((nx, nz))
    cost = ctypes.c_double()
    hg, H, g, act, Jg = np.zeros(nz), np.zeros((nz, nz)), np.zeros(ni), np.zeros(ni, np.int32), np.zeros((ni, nz))
    lib().cmpc_oracle_stage(ctypes.byref(spec), _p(rec), int(k), _p(x), _p(u), _p(lamn), _p(zmul),
                            ctypes.c_double(x0n2), _p(xn), _p(G), ctypes.byref(cost), _p(hg), _p(H), _p(g),
                            _p(act), _p(Jg))
    return dict(xn=xn, G=G, cost=cost.value, grad=hg, H=H, g=g, act=act, Jg=Jg)
