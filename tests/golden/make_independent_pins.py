"""Generates tests/golden/independent_pin_*.npz: (record, solution) pairs for the flat-ground walk computed
WITHOUT the C oracle and without the HIP solver -- by oracle/ipm_dense.py (dense full-space interior point on
the literal torch restatement oracle/nlp_reference.py, autograd derivatives) and, where it finishes, by
scipy.optimize.minimize(method="trust-constr") on the same restatement.  These files are the algorithm-
independent anchor of the solver parity tests (the reference's CasADi/IPOPT stack cannot run here, SURVEY.md
8c).  REGENERATION RULE: only by this script, never from oracle/cmpc_oracle.c or the GPU.

    python tests/golden/make_independent_pins.py [case ...]      (about one minute per N=10 case)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import cmpc_amd  # noqa: E402,F401
from cmpc_amd import workloads as wl  # noqa: E402
from cmpc_amd.problem import ProblemSpec  # noqa: E402
from oracle import nlp_reference as nlp, ipm_dense  # noqa: E402

HW = np.loadtxt(os.path.join(HERE, "measured_hw_cuhw.txt"))

# name: (N, tick, payload gains, what the horizon covers)
CASES = {
    "N10_t120_ds": (10, 120, False, "double support"),
    "N10_t195_liftoff": (10, 195, False, "double support, lift-off inside the horizon"),
    "N10_t210_early_ss": (10, 210, False, "early single support"),
    "N10_t240_mid_ss": (10, 240, False, "mid single support"),
    "N10_t262_late_ss": (10, 262, False, "late single support, touch-down inside the horizon"),
    "N10_t268_late_ss": (10, 268, False, "last ticks of single support"),
    "N10_t285_ds": (10, 285, False, "double support after touch-down"),
    "N10_t292_liftoff": (10, 292, False, "double support, next lift-off inside the horizon"),
    "N10_t210_payload": (10, 210, True, "early single support, payload gains k1, k2 = 7, 1"),
    "N10_t262_payload": (10, 262, True, "late single support + touch-down, payload gains"),
    "N20_t205_early_ss": (20, 205, False, "early single support, N = 20"),
    # (round 5: the dense interior point hovers at a KKT error of 6e-9 ... 1e-8 on this case whatever its tolerance and
    # iteration limit -- 400 iterations at 1e-11 end where 120 at 1e-9 do --, which pins the foot-velocity directions,
    # curvature rho = 1e-4, to 1e-4 only.  Its answer is therefore REFINED: Newton's method on the KKT conditions of the
    # active-set problem at that answer (ipm_dense.refine_active_set), converged to 1e-13.)
    "N20_t255_switch": (20, 255, False, "single support, touch-down inside the horizon, N = 20", dict(refine=True)),
}


# Round 3: the build-defined generalisations of BASELINE configs 4 and 5 (per-instance mass and friction coefficient,
# a push on the initial velocity / angular momentum, 8-vertex contact patches).  name: (N, tick, payload gains, what,
# extras) with extras = mass scale, mu, velocity push (m/s), angular-momentum push (kg m^2/s), vertices per foot.
CASES_X = {
    "N10_c4_m080_mu03_ss": (10, 230, False, "config 4: mass x0.8, mu 0.3, pushed, single support",
                             dict(mass=0.8, mu=0.3, dv=(0.06, -0.04, 0.0), dhw=(0.3, -0.5, 0.1))),
    "N10_c4_m120_mu09_late_ss": (10, 262, False, "config 4: mass x1.2, mu 0.9, pushed, late single support + touch-down",
                                  dict(mass=1.2, mu=0.9, dv=(-0.05, 0.08, 0.0), dhw=(-0.6, 0.4, -0.2))),
    "N10_c4_m090_mu03_ds": (10, 120, False, "config 4: mass x0.9, mu 0.3, pushed, double support",
                             dict(mass=0.9, mu=0.3, dv=(0.10, 0.05, 0.0), dhw=(0.8, 0.8, 0.0))),
    "N10_c4_m110_mu09_liftoff": (10, 292, False, "config 4: mass x1.1, mu 0.9, pushed, lift-off inside the horizon",
                                  dict(mass=1.1, mu=0.9, dv=(0.0, -0.10, 0.0), dhw=(0.5, 0.0, 0.3))),
    "N10_nv8_t120_ds": (10, 120, False, "config 5: 8 vertices per foot, double support", dict(nv=8)),
    "N10_nv8_t240_mid_ss": (10, 240, False, "config 5: 8 vertices per foot, mid single support", dict(nv=8)),
    "N10_nv8_c4_m085_mu04_late_ss": (10, 262, False, "configs 4 + 5: 8 vertices, mass x0.85, mu 0.4, pushed, late single support",
                                      dict(nv=8, mass=0.85, mu=0.4, dv=(0.04, 0.04, 0.0), dhw=(0.2, -0.3, 0.0))),
}
CASES.update(CASES_X)

LINE_SEARCH_CASES = ("N10_t120_ds", "N10_t210_early_ss", "N10_t262_late_ss")


def record(N, t, payload, mass=1.0, mu=0.5, dv=(0.0, 0.0, 0.0), dhw=(0.0, 0.0, 0.0), nv=4):
    spec = ProblemSpec(N=N, nv=nv)
    if payload:
        spec.k1, spec.k2 = 7.0, 1.0
    sc = wl.scene()
    com, dcom = sc.nominal_state(np.array([t]))
    com = com + np.array([[0.004, -0.003, 0.001]])
    dcom = dcom + np.array([[0.01, -0.02, 0.0]]) + np.asarray(dv)[None]
    theta = np.array([[2.0, -3.0, 1.0]]) if payload else np.zeros((1, 3))
    rec = sc.build_records(spec, np.array([t]), com, dcom, HW[t][None] + np.asarray(dhw)[None], theta, np.zeros(1), np.zeros(1),
                           np.full(1, wl.HRP4_MASS * mass), np.full(1, mu))[0]
    return spec, rec


def cold_start(ns, par):
    N, nu = ns.N, ns.nu
    w0 = np.zeros(20 * (N + 1) + nu * N)
    for k in range(N + 1):
        w0[20 * k:20 * k + 20] = par['x0']
    for k in range(N):
        gl, gr = par['gl'][k], par['gr'][k]
        fz = par['mass'] * ns.g / (ns.nv * (gl + gr))
        for j in range(ns.nv):
            w0[20 * (N + 1) + k * nu + 3 * j + 2] = fz * gl
            w0[20 * (N + 1) + k * nu + 3 * (ns.nv + j) + 2] = fz * gr
    return w0


def kkt_check(ns, par, w):
    """Independent first-order check of a primal point: least-squares multipliers on the active set."""
    wt = torch.tensor(w, requires_grad=True)
    f = nlp.cost(ns, par, wt)
    gf = torch.autograd.grad(f, wt)[0].numpy()
    Jc = torch.autograd.functional.jacobian(lambda v: nlp.equalities(ns, par, v), wt.detach(), vectorize=True).numpy()
    Jg = torch.autograd.functional.jacobian(lambda v: nlp.inequalities(ns, par, v), wt.detach(), vectorize=True).numpy()
    g = nlp.inequalities(ns, par, wt.detach()).numpy()
    c = nlp.equalities(ns, par, wt.detach()).numpy()
    return f.item(), np.abs(c).max(), g.max()


def trust_constr(ns, par, w0, maxiter):
    import scipy.optimize as so
    def fun(w):
        wt = torch.tensor(w, requires_grad=True)
        f = nlp.cost(ns, par, wt)
        return f.item(), torch.autograd.grad(f, wt)[0].numpy()
    def hess(w):
        return torch.autograd.functional.hessian(lambda v: nlp.cost(ns, par, v), torch.tensor(w), vectorize=True).numpy()
    ceq = so.NonlinearConstraint(
        lambda w: nlp.equalities(ns, par, torch.tensor(w)).numpy(), 0.0, 0.0,
        jac=lambda w: torch.autograd.functional.jacobian(lambda v: nlp.equalities(ns, par, v), torch.tensor(w), vectorize=True).numpy(),
        hess=lambda w, v: torch.autograd.functional.hessian(lambda q: (torch.tensor(v) * nlp.equalities(ns, par, q)).sum(), torch.tensor(w), vectorize=True).numpy())
    cin = so.NonlinearConstraint(
        lambda w: nlp.inequalities(ns, par, torch.tensor(w)).numpy(), -np.inf, 0.0,
        jac=lambda w: torch.autograd.functional.jacobian(lambda v: nlp.inequalities(ns, par, v), torch.tensor(w), vectorize=True).numpy(),
        hess=lambda w, v: torch.autograd.functional.hessian(lambda q: (torch.tensor(v) * nlp.inequalities(ns, par, q)).sum(), torch.tensor(w), vectorize=True).numpy())
    r = so.minimize(fun, w0, jac=True, hess=hess, constraints=[ceq, cin], method="trust-constr",
                    options=dict(maxiter=maxiter, gtol=1e-9, xtol=1e-12, barrier_tol=1e-10, initial_barrier_parameter=1.0))
    return r


def main():
    torch.set_num_threads(int(os.environ.get("PIN_THREADS", "4")))
    names = sys.argv[1:] or list(CASES)
    for name in names:
        N, t, payload, what = CASES[name][:4]
        extras = dict(CASES[name][4]) if len(CASES[name]) > 4 else {}
        tol = extras.pop("tol", 1e-9)
        max_iter = extras.pop("max_iter", 120)
        refine = extras.pop("refine", False)
        spec, rec = record(N, t, payload, **extras)
        ns = nlp.Spec(N=N, nv=spec.nv, k1=spec.k1, k2=spec.k2)
        par = nlp.unpack_record(ns, rec)
        w0 = cold_start(ns, par)
        out = {"record": rec, "N": N, "nv": spec.nv, "k1": spec.k1, "k2": spec.k2, "tick": t, "what": what}
        t0 = time.time()
        cache = os.path.join(os.environ["PIN_CACHE_DIR"], f"ipm_dense_{name}.npz") if os.environ.get("PIN_CACHE_DIR") else None
        if cache and os.path.exists(cache):                  # (developer convenience: the dense solve of an N = 20 case takes minutes)
            c = np.load(cache)
            r = {k: (c[k] if c[k].ndim else c[k].item()) for k in c.files}
        else:
            r = ipm_dense.solve(ns, par, w0=w0, tol=tol, max_iter=max_iter, linesearch=False,
                                verbose=os.environ.get("PIN_VERBOSE", "0") == "1")
            if cache:
                np.savez(cache, **r)
        out.update(sol_ipm_dense=r["w"], ipm_dense_status=r["status"], ipm_dense_iters=r["iters"], ipm_dense_kkt=r["kkt"])
        print(f"{name}: ipm_dense (full Newton steps) status {r['status']} iters {r['iters']} kkt {r['kkt']:.2e} "
              f"{time.time() - t0:.0f} s", flush=True)
        if refine:
            t0 = time.time()
            rr = ipm_dense.refine_active_set(ns, par, r, verbose=os.environ.get("PIN_VERBOSE", "0") == "1")
            print(f"{name}: active-set Newton refinement kkt {rr['kkt']:.2e} ok {rr['ok']} active rows {int(rr['active'].sum())} "
                  f"|w - ipm_dense| {np.abs(rr['w'] - r['w']).max():.2e} {time.time() - t0:.0f} s", flush=True)
            if rr["ok"]:
                # kkt_scale: the s_d of the scaled KKT error at this optimum (mean multiplier magnitude / 100, at least 1): a
                # solver that meets a scaled tolerance tol has a dual residual of tol * kkt_scale, which along the directions
                # the NLP leaves flat (curvature rho = 1e-4) is a displacement of tol * kkt_scale / rho
                mult = np.concatenate([np.abs(rr["lam"]), np.abs(rr["z"][rr["active"]])])
                out.update(sol_ipm_dense_unrefined=r["w"], ipm_dense_kkt_unrefined=r["kkt"], sol_ipm_dense=rr["w"],
                           ipm_dense_kkt=rr["kkt"], ipm_dense_refined=1, kkt_scale=max(100.0, mult.sum() / mult.size) / 100.0)
                r = dict(r, w=rr["w"], kkt=rr["kkt"])
        if name in LINE_SEARCH_CASES:
            # the same solver with its l1-merit backtracking line search switched on: recorded for the
            # record (it stalls on this problem -- Maratos effect -- which is why neither the C oracle nor
            # the HIP solver has one); not used as a pin
            t0 = time.time()
            r2 = ipm_dense.solve(ns, par, w0=w0, tol=1e-9, max_iter=120, linesearch=True)
            out.update(sol_ipm_dense_ls=r2["w"], ipm_dense_ls_status=r2["status"], ipm_dense_ls_kkt=r2["kkt"])
            print(f"{name}: ipm_dense (l1 line search)   status {r2['status']} iters {r2['iters']} kkt {r2['kkt']:.2e} "
                  f"{time.time() - t0:.0f} s", flush=True)
        if N == 10 and os.environ.get("PIN_TRUST_CONSTR", "1") == "1":
            t0 = time.time()
            # second, unrelated algorithm (scipy's trust-region interior point, Byrd-Hribar-Nocedal): started
            # near the ipm_dense point, it must stay there if that point is a local minimiser
            rng = np.random.default_rng(0)
            start = r["w"] + 1e-3 * rng.standard_normal(r["w"].shape)
            r3 = trust_constr(ns, par, start, maxiter=int(os.environ.get("PIN_TC_ITERS", "400")))
            f3, c3, g3 = kkt_check(ns, par, r3.x)
            out.update(sol_trust_constr=r3.x, trust_constr_status=r3.status, trust_constr_optimality=r3.optimality,
                       trust_constr_constr_violation=r3.constr_violation, trust_constr_iters=r3.nit)
            print(f"{name}: trust-constr status {r3.status} nit {r3.nit} optimality {r3.optimality:.2e} "
                  f"violation {r3.constr_violation:.2e} |x - ipm_dense| {np.abs(r3.x - r['w']).max():.2e} "
                  f"{time.time() - t0:.0f} s", flush=True)
        f, c, g = kkt_check(ns, par, r["w"])
        out.update(objective=f, max_defect=c, max_ineq=g)
        np.savez_compressed(os.path.join(os.environ.get("PIN_OUT_DIR", HERE), f"independent_pin_{name}.npz"), **out)


if __name__ == "__main__":
    main()
