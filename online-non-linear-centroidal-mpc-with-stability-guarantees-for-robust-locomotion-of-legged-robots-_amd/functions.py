"""CoM reference generation (host side, offline; output format is the MPC's input contract).

Restates code/functions.py of the reference:

* ``compute_knot``        (reference :11-55)   knot values/times from the footstep plan
* ``quintic_spline``      (reference :129-157) piecewise-quintic coefficients
* ``built_the_reference / _velocity / _acceleration`` (reference :196-248)
* ``references``          (reference :58-124)  dict of nine per-tick arrays
  ``pos_x..acc_z`` consumed by ``centroidal_mpc`` (centroidal_mpc_vertices.py:64-74)

The reference hands the 4n-1 linear continuity equations in 6n unknowns to IPOPT
with a zero objective, starting from p=0.  A regularised Newton step from the
origin on a feasibility problem with linear constraints lands on the minimum-norm
solution, which is what is computed here directly (``numpy.linalg.lstsq``).  CasADi
is not available in the build environment, so this equivalence is documented,
not executed ("parity unpinned").  No plotting.
"""
import numpy as np


def compute_knot(foot_tra, planner):
    ss = planner.plan[2]['ss_duration']
    ds = planner.plan[2]['ds_duration']
    scale = ss + ds

    def feet(t):
        return foot_tra.generate_feet_trajectories_at_time(t)

    def mid(t, axis):
        f = feet(t)
        return (f['lfoot']['pos'][axis] + f['rfoot']['pos'][axis]) / 2

    knot_x = [mid(0, 3), mid(0, 3)]
    knot_y = [mid(0, 4), feet(0)[planner.plan[1]['foot_id']]['pos'][4] * 0.6]
    first_time_knot = int(2 * scale)
    sequence_x = [first_time_knot]
    sequence_y = [first_time_knot]
    first_contact_time = first_time_knot + ss + 1
    for i in range(first_time_knot, len(planner.plan) * scale - 1):
        if (i - first_contact_time) % scale == 0:
            knot_x.append(mid(i, 3))
            sequence_x.append(i)
            nxt = planner.plan[planner.get_step_index_at_time(i) + 1]['foot_id']
            knot_y.append(feet(i)[nxt]['pos'][4] * 0.6)
            sequence_y.append(i + ds - 1)
    return knot_x, knot_y, sequence_x, sequence_y


def quintic_constraints(x):
    """Rows (A, b) of the reference's equality system A p = b  (reference :135-149)."""
    n = len(x)
    rows, rhs = [], []

    def row(entries, b):
        r = np.zeros(6 * n)
        for j, v in entries:
            r[j] += v
        rows.append(r)
        rhs.append(b)

    for i in range(n - 1):                      # segment end-point positions
        row([(6 * i, 1.)], x[i])
        row([(6 * i + j, 1.) for j in range(6)], x[i + 1])
    row([(1, 1.)], 0.)                          # zero initial / final velocity
    row([(6 * (n - 1) + 1, 1.)], 0.)
    for i in range(n - 1):                      # velocity continuity
        row([(6 * i + j, float(j)) for j in range(1, 6)] + [(6 * (i + 1) + 1, -1.)], 0.)
    row([(2, 2.)], 0.)                          # zero initial acceleration
    for i in range(n - 1):                      # acceleration continuity
        row([(6 * i + 2, 2.), (6 * i + 3, 6.), (6 * i + 4, 12.), (6 * i + 5, 20.),
             (6 * (i + 1) + 2, -2.)], 0.)
    return np.array(rows), np.array(rhs)


def quintic_spline(x):
    A, b = quintic_constraints(x)
    p, *_ = np.linalg.lstsq(A, b, rcond=None)   # minimum-norm solution of the consistent system
    return p.reshape(-1, 1)


def _sample(sequence, p_coeff, poly):
    out = []
    tick = 0
    for i, interval in enumerate(sequence):
        a = np.asarray(p_coeff[6 * i:6 * i + 6]).reshape(6)
        span = interval - tick
        for second in range(span):
            out.append(poly(a, second / span, span))
        tick = interval
    return out


def built_the_reference(sequence, p_coeff):
    return _sample(sequence, p_coeff, lambda a, s, T:
                   a[0] + a[1] * s + a[2] * s**2 + a[3] * s**3 + a[4] * s**4 + a[5] * s**5)


def built_the_velocity(sequence, p_coeff):
    # d/dtau, NOT divided by the span (reference :222)
    return _sample(sequence, p_coeff, lambda a, s, T:
                   a[1] + 2 * a[2] * s + 3 * a[3] * s**2 + 4 * a[4] * s**3 + 5 * a[5] * s**4)


def built_the_acceleration(sequence, p_coeff):
    # d2/dtau2 divided by span^2 (reference :243)
    return _sample(sequence, p_coeff, lambda a, s, T:
                   (2 * a[2] + 6 * a[3] * s + 12 * a[4] * s**2 + 20 * a[5] * s**3) / T**2)


def references(foot_tra, planner, SHOW_PLOT=1):
    knot_x, knot_y, seq_x, seq_y = compute_knot(foot_tra, planner)
    co_x = quintic_spline(knot_x)
    co_y = quintic_spline(knot_y)
    ref = {
        'pos_x': [float(v) for v in built_the_reference(seq_x, co_x)],
        'vel_x': [float(v) for v in built_the_velocity(seq_x, co_x)],
        'acc_x': [float(v) for v in built_the_acceleration(seq_x, co_x)],
        'pos_y': [float(v) for v in built_the_reference(seq_y, co_y)],
        'vel_y': [float(v) for v in built_the_velocity(seq_y, co_y)],
        'acc_y': [float(v) for v in built_the_acceleration(seq_y, co_y)],
    }
    n = len(ref['pos_x'])
    ref['pos_z'] = np.full(n, 0.72)             # reference :97-99
    ref['vel_z'] = np.zeros(n)
    ref['acc_z'] = np.zeros(n)
    return ref
