"""Synthetic workloads of BASELINE.json and the vectorised parameter builder.

``Scene`` holds the flat-ground walk of the reference driver (code/simulation.py:24-39,
:97-130): the shipped velocity commands, planner, swing generator and CoM reference, plus
per-tick tables that make the front half of ``centroidal_mpc.solve``
(code/centroidal_mpc_vertices.py:482-600) a pure gather, so a whole batch of parameter
records is built with numpy fancy indexing instead of the reference's Python loops.
"""
import numpy as np

from .footstep_planner_vertices import FootstepPlanner
from .foot_trajectory_generator import FootTrajectoryGenerator
from .functions import references
from .problem import ProblemSpec, current_contacts, contact_flags

HRP4_MASS = 40.05487735          # sum of link masses of code/urdf/hrp4.urdf

# initial sole poses of the DART model, code/Debug/contact_trj_from_centroidal_MPC line 1
LFOOT0 = np.array([0.0, -1.3870180197464853e-16, -4.639952657811354e-18,
                   1.0310923973763693e-17, 0.10163857612916291, -1.3877787807814457e-17])
RFOOT0 = np.array([0.0, -1.3870180197464853e-16, 4.639952657811354e-18,
                   1.0310923973763693e-17, -0.10163857612916291, -1.3877787807814457e-17])

# code/simulation.py:97
VREF = ([(0.15, 0., 0)] * 5 + [(0.15, 0.0, 0)] * 3 + [(0.15, 0.0, 0)] * 3 +
        [(0.13, 0, 0)] * 4 + [(0.1, 0., 0)] * 2 + [(0., 0, 0)] * 3)


def default_params(N=10, mpc_rate=1):
    """The params dict of code/simulation.py:24-44 (mass from the URDF)."""
    p = {'g': 9.81, 'h': 0.72, 'foot_size': 0.1, 'step_height': 0.02, 'world_time_step': 0.01,
         'ss_duration': 70, 'ds_duration': 30, 'first_swing': 'rfoot', 'µ': 0.5, 'N': N, 'dof': 30,
         'mass': HRP4_MASS, 'update_contact': 'YES', 'mpc_rate': mpc_rate}
    p['eta'] = np.sqrt(p['g'] / p['h'])
    return p


class Scene:
    """Planner + references of the shipped walk and per-tick gather tables."""

    def __init__(self, params=None):
        self.params = default_params() if params is None else params
        self.initial = {'lfoot': {'pos': LFOOT0.copy()}, 'rfoot': {'pos': RFOOT0.copy()},
                        'com': {'pos': np.array([0., 0., 0.72]), 'vel': np.zeros(3)},
                        'hw': {'val': np.zeros(3)}}
        self.planner = FootstepPlanner(VREF, LFOOT0, RFOOT0, self.params)
        self.ftg = FootTrajectoryGenerator(self.initial, self.planner, self.params)
        self.com_ref = references(self.ftg, self.planner)
        self.refresh_tables()

    def refresh_tables(self):
        """(Re)build per-tick tables from the planner (call again after a plan write-back)."""
        pl, first = self.planner, self.params['first_swing']
        keys = ('pos_x', 'pos_y', 'pos_z', 'vel_x', 'vel_y', 'vel_z', 'acc_x', 'acc_y', 'acc_z')
        T = min(len(self.com_ref[k]) for k in keys)
        self.T = T
        self.com_tab = np.stack([np.asarray(self.com_ref[k], dtype=np.float64)[:T] for k in keys], axis=1)
        pose_l = pl.position_contacts_ref['contact_left']
        pose_r = pl.position_contacts_ref['contact_right']
        self.pose_l, self.pose_r = pose_l, pose_r
        gl, gr = np.ones(T), np.ones(T)
        cur_l, cur_r = np.zeros((T, 3)), np.zeros((T, 3))
        # schedule tables for per-instance plans (rollout.BatchedRollout): plan entries behind the x0 foot
        # positions (:493-509), phase and step index of every tick, support foot of the step
        self.slot_l, self.slot_r = np.full(T, -1, np.int32), np.full(T, -1, np.int32)
        self.is_ss, self.step_idx = np.zeros(T, bool), np.zeros(T, np.int32)
        self.support_is_l = np.zeros(T, bool)
        for t in range(T):
            idx = pl.get_step_index_at_time(t)
            self.step_idx[t] = idx
            self.support_is_l[t] = pl.plan[idx]['foot_id'] == 'lfoot'
            if pl.get_phase_at_time(t) != 'ds':
                self.is_ss[t] = True
                if pl.plan[idx]['foot_id'] == 'lfoot':
                    gr[t] = 0.
                else:
                    gl[t] = 0.
            cur_l[t], cur_r[t] = current_contacts(pl, pose_l[:, 3:6], pose_r[:, 3:6], t, first)
            if t >= 200:
                index = pl.get_step_index_at_time(t - 70)
                a, b = index + (index % 2), index + (index - 1) % 2
                self.slot_l[t], self.slot_r[t] = (a, b) if first == 'lfoot' else (b, a)
        self.plan_pos = np.stack([np.asarray(p['pos'], dtype=np.float64) for p in pl.plan])
        self.gl_tab, self.gr_tab, self.cur_l, self.cur_r = gl, gr, cur_l, cur_r

    def t_max(self, N, rate=1):
        return self.T - 1 - (N + 1) * rate

    def build_records(self, spec, t, com, dcom, hw, theta_hat, yaw_l, yaw_r, mass, mu, rate=1):
        """Vectorised front half of ``solve`` for B instances.  All inputs have leading dim B."""
        t = np.asarray(t, dtype=np.int64)
        B, N = t.shape[0], spec.N
        rec = np.zeros((B, spec.nrec))
        rec[:, 0:3], rec[:, 3:6], rec[:, 6:9], rec[:, 9:12] = com, dcom, hw, theta_hat
        rec[:, 12], rec[:, 13:16] = yaw_l, self.cur_l[t]
        rec[:, 16], rec[:, 17:20] = yaw_r, self.cur_r[t]
        rec[:, 20], rec[:, 21] = mass, mu
        rec[:, 22], rec[:, 23] = self.gl_tab[t + N * rate], self.gr_tab[t + N * rate]
        st = rec[:, 24:].reshape(B, N, 19)
        tt = t[:, None] + (1 + np.arange(N))[None, :] * rate          # (B, N): ticks t+(1+i)*rate
        tn = t[:, None] + np.arange(N)[None, :] * rate                # contact flags at t+i*rate
        st[:, :, 0:9] = self.com_tab[tt]
        st[:, :, 9:12] = self.pose_l[tt, 3:6]
        st[:, :, 12:15] = self.pose_r[tt, 3:6]
        st[:, :, 15] = self.pose_l[tt, 2]
        st[:, :, 16] = self.pose_r[tt, 2]
        st[:, :, 17] = self.gl_tab[tn]
        st[:, :, 18] = self.gr_tab[tn]
        return rec

    def nominal_state(self, t):
        t = np.asarray(t, dtype=np.int64)
        com = np.stack([self.com_tab[t, 0], self.com_tab[t, 1], np.full(t.shape, 0.72)], axis=1)
        dcom = np.stack([self.com_tab[t, 3], self.com_tab[t, 4], np.zeros(t.shape)], axis=1)
        return com, dcom


_SCENE = None


def scene():
    global _SCENE
    if _SCENE is None:
        _SCENE = Scene()
    return _SCENE


CONFIGS = {
    # name: (seed, N, nv, payload gains, default batch)
    "perturbed": (20250711, 20, 4, False, 256),          # BASELINE config 2
    "payload": (20250712, 20, 4, True, 4096),            # config 3
    "randomized": (20250713, 20, 4, False, 65536),       # config 4
    "long_horizon": (20250714, 40, 8, False, 16384),     # config 5
}


def make_workload(name, B=None, N=None, scale=1.0, rate=1, seed=None):
    """(spec, records (B, nrec) numpy) for one of the BASELINE synthetic configs.
    seed = None: the configuration's own seed; another value draws a fresh batch of the same distribution.
    rate = 10 is the reference's ``mpc_rate == 10`` variant: delta = 0.1 s, k1, k2 = 5, 0.2, no force-rate
    cost (:11, :27-31, :339-341), references sampled every tenth tick (:548-600)."""
    seed0, N0, nv, payload, B0 = CONFIGS[name]
    seed = seed0 if seed is None else seed
    B = B0 if B is None else B
    N = N0 if N is None else N
    sc = scene()
    spec = ProblemSpec.from_params(default_params(N=N, mpc_rate=rate), payload=payload, nv=nv)
    rng = np.random.default_rng(seed)
    t = rng.integers(200, min(1700, sc.t_max(N, rate)) + 1, size=B)
    com, dcom = sc.nominal_state(t)
    com = com + scale * rng.uniform(-0.02, 0.02, size=(B, 3))
    com[:, 2] = np.minimum(com[:, 2], 0.755)
    dcom = dcom + scale * rng.normal(0, 0.05, size=(B, 3))
    hw = scale * rng.normal(0, 0.3, size=(B, 3))
    theta = np.zeros((B, 3))
    mass = np.full(B, HRP4_MASS)
    mu = np.full(B, 0.5)
    if name == "payload":
        theta = rng.uniform(-25, 25, size=(B, 3))
        com[:, 2] -= rng.uniform(0, 0.02, size=B)
    if name == "randomized":
        mass = HRP4_MASS * rng.uniform(0.8, 1.2, size=B)
        mu = rng.uniform(0.3, 0.9, size=B)
        F = rng.uniform(0, 100, size=B)
        ang = rng.uniform(0, 2 * np.pi, size=B)
        arm = rng.uniform(0, 0.4, size=B)
        Fv = np.stack([F * np.cos(ang), F * np.sin(ang), np.zeros(B)], axis=1)
        dcom = dcom + Fv * 0.1 / mass[:, None]
        r = np.stack([np.zeros(B), np.zeros(B), arm], axis=1)
        hw = hw + np.cross(r, Fv) * 0.1
    rec = sc.build_records(spec, t, com, dcom, hw, theta, np.zeros(B), np.zeros(B), mass, mu, rate=rate)
    return spec, rec


def walk_records(spec, ticks, hw=None, theta_hat=None):
    """Config 1 sampled open loop: records of the nominal flat-ground walk at `ticks` (x0 on the CoM
    reference, measured angular momentum `hw` (len(ticks), 3) -- zero if None, which puts late single support
    at the edge of feasibility)."""
    sc = scene()
    t = np.asarray(ticks, dtype=np.int64)
    com, dcom = sc.nominal_state(t)
    B = t.shape[0]
    hw = np.zeros((B, 3)) if hw is None else np.asarray(hw, dtype=np.float64)
    th = np.zeros((B, 3)) if theta_hat is None else np.asarray(theta_hat, dtype=np.float64)
    return sc.build_records(spec, t, com, dcom, hw, th, np.zeros(B), np.zeros(B),
                            np.full(B, HRP4_MASS), np.full(B, 0.5))


WBC_ND, WBC_NC = 30, 12           # dofs and contact-wrench dimensions of the whole-body QP (code/inverse_dynamics.py:30-66)


def wbc_synthetic(B, seed=0, contact="ds", mass=HRP4_MASS, g=9.81):
    """Synthetic instances of the size and structure of the reference's QP for HRP-4 (30 dofs): task Jacobians of the
    shapes of :46-51 with random entries, a positive definite mass matrix with the robot's total mass on the base
    translation, gravity on the base, contact Jacobians [.. foot wrench ..] scaled by the contact flags (:109)."""
    rng = np.random.default_rng(seed)
    Hq = np.zeros((B, WBC_ND, WBC_ND)); Fq = np.zeros((B, WBC_ND)); M = np.zeros((B, WBC_ND, WBC_ND)); h = np.zeros((B, WBC_ND)); Jc = np.zeros((B, WBC_NC, WBC_ND))
    sel = np.zeros(WBC_ND); sel[18:30] = 1.0                         # "redundant dofs" of the joint task
    weights = {'lfoot': 1.0, 'rfoot': 1.0, 'com': 1.0, 'torso': 1.0, 'base': 1.0}
    rows = {'lfoot': 6, 'rfoot': 6, 'com': 3, 'torso': 3, 'base': 3}
    for b in range(B):
        Jt = {k: rng.normal(0, 0.4, size=(r, WBC_ND)) for k, r in rows.items()}
        for k in ('lfoot', 'rfoot'):
            Jt[k][:, :6] += np.eye(6)                             # feet move with the floating base
        Jt['com'][:, 3:6] += np.eye(3)
        Hb = sum(weights[k] * Jt[k].T @ Jt[k] for k in rows) + 0.1 * np.diag(sel)
        acc = {k: rng.normal(0, 1.0, size=r) for k, r in rows.items()}
        Fb = -sum(weights[k] * Jt[k].T @ acc[k] for k in rows) - 0.1 * sel * rng.normal(0, 1.0, size=WBC_ND)
        L = rng.normal(0, 0.15, size=(WBC_ND, WBC_ND))
        Mb = L @ L.T + np.diag(np.concatenate([np.full(3, 2.0), np.full(3, mass), rng.uniform(0.05, 1.0, WBC_ND - 6)]))
        hb = rng.normal(0, 2.0, size=WBC_ND); hb[5] += mass * g      # gravity on the base translation (z)
        cl, cr = contact in ("ds", "lfoot"), contact in ("ds", "rfoot")
        Jcb = np.vstack([cl * Jt['lfoot'], cr * Jt['rfoot']])
        Hq[b], Fq[b], M[b], h[b], Jc[b] = Hb, Fb, Mb, hb, Jcb
    return Hq, Fq, M, h, Jc
