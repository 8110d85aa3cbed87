"""Problem constants and the per-instance parameter record (host side).

``ProblemSpec`` carries the constants the reference hard-codes in
``centroidal_mpc.__init__`` (code/centroidal_mpc_vertices.py:7-60, 228-230,
259-271, 301-351); the record is what the reference pushes into the Opti
parameters every tick (``solve`` front half, :480-600).

Parameter record, fp64, ``nrec = 24 + 19*N`` doubles per instance, instances
contiguous (one wavefront reads its own record with unit stride):

    [0:20]   x0   = [com(3) dcom(3) hw(3) theta_hat(3) yaw_l p_l(3) yaw_r p_r(3)]   (:164-166, :482-509)
    [20]     mass          (reference: params['mass'], baked into the graph, :15)
    [21]     mu            (reference: literal 0.5, :41)
    [22:24]  gamma_l[N], gamma_r[N]                                           (:179-181, :515-534)
    24+19k+[0:9]    com_ref[:,k] = pos(3) vel(3) acc(3) at t+(1+k)*rate       (:567-577)
    24+19k+[9:12]   p_l_ref[:,k]   24+19k+[12:15]  p_r_ref[:,k]               (:581-582)
    24+19k+15/16    yaw_l_ref[k], yaw_r_ref[k]                                (:583-584)
    24+19k+17/18    gamma_l[k], gamma_r[k]
Solution record: X (20 x (N+1), column-major) followed by U (nu x N, column-major),
the layout of ``sol.value(opti_state)`` / ``sol.value(U)`` (:614-617).
"""
from dataclasses import dataclass, field
import ctypes
import numpy as np

NX = 20


@dataclass
class ProblemSpec:
    N: int = 20
    nv: int = 4                   # vertices per foot (reference: 4, :55-60)
    delta: float = 0.01           # world_time_step * mpc_rate (:11)
    g: float = 9.81
    k1: float = 4.0               # :27-31 (payload file: 7, 1)
    k2: float = 0.1
    w_rate: float = 1.0           # :339-341
    w_hw: float = 1000.0          # :312
    w_cxy: float = 1.0            # :313-314
    w_cz_const: float = 2000.0    # :302-305
    w_foot: float = 1000.0        # :316-319
    w_force: float = 10.0         # :320-335
    cz_max: float = 0.76          # :230
    box: tuple = (0.01, 0.005, 0.00005)   # :259-271
    foot_length: float = 0.25     # :51-52
    foot_width: float = 0.13
    prox: float = 1e-4            # build-defined proximal weight on U
    relax: float = 1e-8           # IPOPT bound_relax_factor
    tol: float = 1e-8             # KKT tolerance of the batched solver
    acc_tol: float = 1e-4         # acceptable level (status 3); the reference's IPOPT runs tol = 1e-3 (:128)
    max_iter: int = 100
    kernel: int = 0               # CMPC_KERNEL_* (include/cmpc.h): 0 auto, 1 one wavefront per instance, 2 the pipelined pair
    reserved: int = 0

    @property
    def nu(self):
        return 6 * self.nv + 8

    @property
    def nrec(self):
        return 24 + 19 * self.N

    @property
    def nsol(self):
        return NX * (self.N + 1) + self.nu * self.N

    @property
    def nstate(self):
        """Doubles of the solver state carried between closed-loop ticks (CMPC_NSTATE, include/cmpc.h)."""
        return self.nsol + (self.N + 1) * ((NX + 2 * self.nv) + 2 * (15 + 10 * self.nv) + 2) + 8

    def vertices(self):
        L, W = self.foot_length / 2, self.foot_width / 2
        corners = [[L, W, 0.], [L, -W, 0.], [-L, -W, 0.], [-L, W, 0.]]
        if self.nv == 4:
            return np.array(corners)
        if self.nv == 8:
            return np.array(corners + [[L, 0., 0.], [0., -W, 0.], [-L, 0., 0.], [0., W, 0.]])
        raise ValueError("nv must be 4 or 8")

    @classmethod
    def from_params(cls, params, payload=False, **kw):
        """Constants as ``centroidal_mpc.__init__`` derives them from the params dict."""
        rate = params['mpc_rate']
        if payload:                      # centroidal_mpc_vertices_payload.py:27-31
            k1, k2 = 7.0, 1.0
        else:
            k1, k2 = (5.0, 0.2) if rate == 10 else (4.0, 0.1)
        return cls(N=params['N'], delta=params['world_time_step'] * rate, g=params['g'],
                   k1=k1, k2=k2, w_rate=0.0 if rate == 10 else 1.0, **kw)


def contact_flags(planner, t, N, rate=1):
    """gamma_l, gamma_r over the N+1 horizon nodes (reference :515-534)."""
    gl, gr = np.ones(N + 1), np.ones(N + 1)
    for i in range(N + 1):
        tt = t + i * rate
        if planner.get_phase_at_time(tt) != 'ds':
            if planner.plan[planner.get_step_index_at_time(tt)]['foot_id'] == 'lfoot':
                gr[i] = 0.
            else:
                gl[i] = 0.
    return gl, gr


def current_contacts(planner, pos_ref_l, pos_ref_r, t, first_swing):
    """Foot positions written into x0 (reference :493-509)."""
    if t < 200:
        return pos_ref_l[t], pos_ref_r[t]
    index = planner.get_step_index_at_time(t - 70)
    a = planner.plan[index + (index % 2)]['pos']
    b = planner.plan[index + (index - 1) % 2]['pos']
    return (a, b) if first_swing == 'lfoot' else (b, a)


def build_record(spec, planner, com_ref, t, com, dcom, hw, theta_hat, yaw_l, yaw_r,
                 mass, mu=0.5, first_swing='rfoot', rate=1, contacts_ref=None):
    """One parameter record for tick `t` (front half of ``solve``, reference :482-600)."""
    N = spec.N
    cref = planner.position_contacts_ref if contacts_ref is None else contacts_ref
    pose_l, pose_r = cref['contact_left'], cref['contact_right']
    pl0, pr0 = current_contacts(planner, pose_l[:, 3:6], pose_r[:, 3:6], t, first_swing)
    rec = np.zeros(spec.nrec)
    rec[0:3], rec[3:6], rec[6:9], rec[9:12] = com, dcom, hw, theta_hat
    rec[12], rec[13:16], rec[16], rec[17:20] = yaw_l, pl0, yaw_r, pr0
    rec[20], rec[21] = mass, mu
    gl, gr = contact_flags(planner, t, N, rate)
    rec[22], rec[23] = gl[N], gr[N]
    st = rec[24:].reshape(N, 19)
    keys = ('pos_x', 'pos_y', 'pos_z', 'vel_x', 'vel_y', 'vel_z', 'acc_x', 'acc_y', 'acc_z')
    for i in range(N):
        tt = t + (1 + i) * rate
        for j, k in enumerate(keys):
            st[i, j] = com_ref[k][tt]
        st[i, 9:12] = pose_l[tt, 3:6]
        st[i, 12:15] = pose_r[tt, 3:6]
        st[i, 15], st[i, 16] = pose_l[tt, 2], pose_r[tt, 2]
        st[i, 17], st[i, 18] = gl[i], gr[i]
    return rec


class CSpec(ctypes.Structure):
    """C mirror of ``cmpc_spec`` (include/cmpc.h)."""
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
                ("kernel", ctypes.c_int32), ("reserved", ctypes.c_int32)]


def to_cspec(spec):
    c = CSpec()
    for name, _ in CSpec._fields_:
        if name == "struct_size":
            c.struct_size = ctypes.sizeof(CSpec)
            continue
        if name == "box":
            c.box = (ctypes.c_double * 3)(*spec.box)
        else:
            setattr(c, name, getattr(spec, name))
    return c
