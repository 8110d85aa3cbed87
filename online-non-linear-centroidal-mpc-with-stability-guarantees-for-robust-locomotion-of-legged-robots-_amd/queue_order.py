"""Host mirror of the launch's queue order (csrc/cmpc_hip.hip: cmpc_order_bucket): the features of a parameter record
that say how many interior-point iterations its solve will take, and the fitted prediction.  The coefficients live in
ONE place, csrc/cmpc_order_fit.h (written by tools/fit_queue_order.py, included by the kernel, parsed here).  The order
never changes a result; this module exists for the tools that replay a launch on the host and for the fit itself."""
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
FIT_HEADER = os.path.join(_HERE, "csrc", "cmpc_order_fit.h")
ORDER_BUCKETS = 64
NAMES = ("one", "switch", "first", "feet", "d", "d2", "hw", "d_switch", "d_feet", "touch_down", "ev", "n_switch", "z_margin",
         "mu", "mass", "ev2", "ev_feet", "inv_mu")


def coefficients(path=FIT_HEADER):
    text = open(path).read()
    n = int(re.search(r"#define\s+CMPC_ORDER_NFEAT\s+(\d+)", text).group(1))
    body = re.search(r"#define\s+CMPC_ORDER_COEF\s+\{([^}]*)\}", text).group(1)
    coef = np.array([float(x) for x in body.split(",")])
    assert coef.shape == (n,) and n == len(NAMES), "csrc/cmpc_order_fit.h and queue_order.NAMES disagree"
    origin = float(re.search(r"#define\s+CMPC_ORDER_BUCKET_ORIGIN\s+([-+0-9.eE]+)", text).group(1))
    return coef, origin


def features(rec, spec):
    """(B, len(NAMES)) design matrix of records (B, nrec): what cmpc_order_bucket computes, term for term."""
    rec = np.atleast_2d(rec)
    B, N = rec.shape[0], spec.N
    fl = np.stack([rec[:, 24 + 19 * np.arange(N) + 17], rec[:, 24 + 19 * np.arange(N) + 18]], -1)
    fl = np.concatenate([fl, rec[:, None, 22:24]], 1)                      # contact flags of the N + 1 nodes
    chg = (np.diff(fl, axis=1) != 0).any(axis=2)
    sw = chg.any(axis=1)
    first = np.where(sw, chg.argmax(axis=1), N)
    nsw = chg.sum(axis=1).astype(float)
    idx = np.arange(B)
    td = (fl[idx, np.minimum(first + 1, N)].sum(axis=1) > fl[idx, np.minimum(first, N)].sum(axis=1)) & sw
    gl0, gr0 = fl[:, 0, 0], fl[:, 0, 1]
    om = np.sqrt(spec.g / spec.cz_max)
    dcm = rec[:, 0:2] + rec[:, 3:5] / om
    both = (gl0 != 0) == (gr0 != 0)
    tgt = np.where(both[:, None], 0.5 * (rec[:, 13:15] + rec[:, 17:19]),
                   np.where((gl0 != 0)[:, None], rec[:, 13:15], rec[:, 17:19]))
    d2 = ((dcm - tgt) ** 2).sum(axis=1)
    d = np.sqrt(d2)
    ev2 = ((rec[:, 3:5] - rec[:, 27:29]) ** 2).sum(axis=1)
    ev = np.sqrt(ev2)
    feet, swf, mu = gl0 + gr0, sw.astype(float), rec[:, 21]
    return np.stack([np.ones(B), swf, first.astype(float), feet, d, d2, np.linalg.norm(rec[:, 6:9], axis=1), d * swf, d * feet,
                     td.astype(float), ev, nsw, spec.cz_max - rec[:, 2], mu, rec[:, 20] / 40.0, ev2, ev * feet, 1.0 / mu], axis=1)


def predicted_iterations(rec, spec, coef=None):
    c = coefficients()[0] if coef is None else np.asarray(coef)
    return features(rec, spec) @ c


def bucket_of(pred, origin=None):
    origin = coefficients()[1] if origin is None else origin
    b = 2.0 * (np.asarray(pred) - origin)
    return np.clip(np.nan_to_num(b, nan=0.0), 0, ORDER_BUCKETS - 1).astype(int)
