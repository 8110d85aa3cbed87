/*
 * cmpc_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, fp64) of the
 * reference's centroidal-MPC NLP and of the interior-point / Riccati algorithm the
 * HIP solver implements.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path never does.
 *
 * PARITY UNPINNED: the reference solves the NLP with CasADi + IPOPT
 * (code/centroidal_mpc_vertices.py:126-130, :606); neither is available in the build
 * environment and the reference holds no solver golden vectors.  This
 * file is pinned instead by (a) oracle/nlp_reference.py, an independent literal
 * restatement with autograd derivatives, on function values and on optima found by an
 * independent dense solver, and (b) finite differences of its own functions.
 *
 * Reference lines followed (code/centroidal_mpc_vertices.py):
 *   stage_dynamics()   :371-461 (centroidal_dynamic), :185-190 (forward Euler)
 *   stage_cost()       :275-353
 *   stage_ineq()       :193-271 (Lyapunov rows with CoM[:,i+1], dCoM[:,i+1] substituted
 *                       through the linear CoM dynamics; contraction; height; cone; box)
 * Formulation notes: x_0 is eliminated; the force-rate cost (:343-351) is
 * made stage-local by carrying the previous stage's f_z as 2*nv extra states.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/cmpc.h"

#define MAXNV 8
#define MAXNU (6 * MAXNV + 8)
#define MAXNX (CMPC_NX + 2 * MAXNV)
#define MAXNZ (MAXNU + MAXNX)
#define MAXNI (15 + 10 * MAXNV)

/* termination safeguards (mirrored by the HIP solver) */
#define ACC_FACTOR 100.0
#ifndef ACC_ITERS
#define ACC_ITERS 8
#endif
/* no progress at the final barrier value: NOPROG_ITERS iterations without halving the best KKT error
 * seen there.  The run then ends as CMPC_ACCEPTABLE as soon as the error is within spec.acc_tol. */
#ifndef NOPROG_FACTOR              /* "progress" at a barrier value = the best error of that barrier problem fell below this share */
#define NOPROG_FACTOR 0.5
#endif
#ifndef NOPROG_ITERS
#define NOPROG_ITERS 12
#endif
/* pivot acceptance of the stage factorisation (see riccati_backward) */
#define PIV_MIN 1e-8
#define PIV_FRAC 0.1
#define STALL_STEP 1e-7
#define STALL_ITERS 6
/* after the tolerance is first met: POLISH_ITERS more Newton iterations at the final barrier value,
 * so that the returned point is the mu = tol/10 central-path point to ~1e-11 whatever path led there
 * (directions whose only curvature is the proximal term need this to be reproducible). */
#define POLISH_ITERS 1
/* barrier schedule: start value and linear decrease factor (tuned on the synthetic configs: a large
 * start value centres the first iterates; 0.1 -> 100 cut the mean iteration count from 33 to 23) */
#ifndef REG_FIRST_FACTOR         /* escalation of the inertia correction: first correction of a solve, later ones (IPOPT: 100, 8) */
#define REG_FIRST_FACTOR 100.0
#endif
#ifndef REG_NEXT_FACTOR
#define REG_NEXT_FACTOR 8.0
#endif
#ifndef S_FLOOR                  /* floor of the initial slacks of a cold start */
#define S_FLOOR 1e-2
#endif
#ifndef COLD_ROLLOUT             /* 1: cold start rolled out under the initial inputs (see initial_point) */
#define COLD_ROLLOUT 1
#endif
#ifndef MU_INIT                  /* (-D overrides: tuning experiments only, tools/tune_schedule.py) */
#define MU_INIT 100.0
#endif
#ifndef MU_FACTOR
#define MU_FACTOR 0.1
#endif
#ifndef MU_POWER                 /* superlinear decrease mu <- mu^MU_POWER */
#define MU_POWER 1.5
#endif
#ifndef KAPPA_EPS                /* the barrier problem counts as solved at an error of KAPPA_EPS * mu */
#define KAPPA_EPS 10.0
#endif
#ifndef TAU_MIN                  /* fraction to the boundary */
#define TAU_MIN 0.99
#endif
/* Warm start of the interior point method (closed-loop ticks): the iterate that solved the barrier problem at the
 * last barrier value >= MU_WARM is kept as the solver state -- primal, dynamics multipliers, slacks and inequality
 * multipliers, a point ON the central path of this tick's problem -- and the next tick's solve resumes from it at
 * that barrier value instead of restarting at MU_INIT from the (boundary) solution.  See DESIGN.md section 4.
 * Level: measured on the flat-ground walk (N = 10, 1900 ticks, tools/warm_walk.py), mean / median iterations and
 * ticks ending "acceptable": primal warm start only 17.7 / 16 / 5; MU_WARM 1e-2: 12.9 / 11 / 10; 1e-5 (level 3.2e-5):
 * 10.4 / 8 / 24; 1e-7 (level 1.8e-7, the last one before the final barrier value): 10.0 / 7 / 1.  From a level mu the
 * complementarity products of weakly active rows shrink by at most 4x per Newton step (the ds*dz term), so the tail
 * costs log4(mu / 1e-9) iterations whatever the schedule: the lowest interior level wins. */
#ifndef RESUME_RECENTRE_ITERS
#define RESUME_RECENTRE_ITERS 20     /* a resumed solve still at the state's barrier value after this many iterations gives up */
#endif
#ifndef MU_WARM
#define MU_WARM 1e-7
#endif

/* inequality row slots of one stage */
enum { R_LYAP = 0, R_CZ = 1, R_HWC = 2, R_BOX = 3, R_FRIC = 15 };

typedef struct {
  int N, nv, nu, nx, nz, ni;
  const cmpc_spec *sp;
  const double *rec;
  double mass, mu_f;
  double vert[MAXNV][2];
} prob_t;

typedef struct {
  /* iterate */
  double *x, *u, *lam, *s, *z;   /* [N+1][nx], [N][nu], [N+1][nx], [N+1][ni], [N+1][ni] */
  int *act;                      /* [N+1][ni] */
  double *uprox;                 /* [N][nu] */
  /* per-stage linearisation */
  double *G, *b, *H, *h, *g, *Jg; /* [N][nx][nz], [N][nx], [N+1][nz][nz], [N+1][nz], [N+1][ni], [N+1][ni][nz] */
  double *hobj;                  /* objective-only gradient [N+1][nz] */
  /* Riccati storage */
  double *Lam, *Ls, *l, *P, *p;  /* [N][nu][nu], [N][nx][nu], [N][nu], [N+1][nx][nx], [N+1][nx] */
  double *dx, *du, *lamn, *ds, *dz;
  double *M, *m;                 /* scratch nz*nz, nz */
} work_t;

static const double *stage_rec(const prob_t *P, int k) { return P->rec + 24 + 19 * k; }
static double gam(const prob_t *P, int k, int foot) {
  if (k == P->N) return P->rec[22 + foot];
  return stage_rec(P, k)[17 + foot];
}
static double w_cz(const cmpc_spec *sp, int i) {
  double half = sp->w_cz_const / 2;
  return (sp->w_cz_const - half) * exp(-(double)i) + half;
}

static void prob_init(prob_t *P, const cmpc_spec *sp, const double *rec) {
  P->sp = sp; P->rec = rec; P->N = sp->N; P->nv = sp->nv;
  P->nu = 6 * sp->nv + 8; P->nx = CMPC_NX + 2 * sp->nv; P->nz = P->nu + P->nx;
  P->ni = 15 + 10 * sp->nv;
  P->mass = rec[20]; P->mu_f = rec[21];
  double L = sp->foot_length / 2, W = sp->foot_width / 2;
  double c[8][2] = {{L, W}, {L, -W}, {-L, -W}, {-L, W}, {L, 0}, {0, -W}, {-L, 0}, {0, W}};
  for (int j = 0; j < sp->nv; ++j) { P->vert[j][0] = c[j][0]; P->vert[j][1] = c[j][1]; }
}

static void cross3(const double *a, const double *b, double *o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
/* skew(a) b = a x b ; S[r][c] */
static void skew3(const double *a, double S[3][3]) {
  S[0][0] = 0; S[0][1] = -a[2]; S[0][2] = a[1];
  S[1][0] = a[2]; S[1][1] = 0; S[1][2] = -a[0];
  S[2][0] = -a[1]; S[2][1] = a[0]; S[2][2] = 0;
}

/* ------------------------------------------------------------------------------------------
 * Dynamics of stage k: xn = F(x,u) (nx entries incl. carried f_z).  If G != NULL also
 * G = [B A] (nx x nz, row-major, columns: u first then x).  If Hd != NULL adds
 * sum_i lamn[i] * d2F_i to Hd (nz x nz), lamn = multiplier of x_{k+1} = F.
 * reference :371-461, :188-190
 * ---------------------------------------------------------------------------------------- */
static void stage_dynamics(const prob_t *P, int k, const double *x, const double *u, double *xn,
                           double *G, const double *lamn, double *Hd) {
  const cmpc_spec *sp = P->sp;
  const int nv = P->nv, nu = P->nu, nx = P->nx, nz = P->nz;
  const double d = sp->delta, m = P->mass;
  const double *sr = stage_rec(P, k);
  const double gm[2] = {sr[17], sr[18]};
  const double *c = x, *v = x + 3, *hh = x + 6, *th = x + 9;
  const int iy[2] = {12, 16}, ip[2] = {13, 17};
  double tau[3] = {0, 0, 0}, Fs[2][3] = {{0, 0, 0}, {0, 0, 0}};
  if (G) memset(G, 0, sizeof(double) * nx * nz);
#define GA(r, cx) G[(r) * nz + nu + (cx)]
#define GB(r, cu) G[(r) * nz + (cu)]
#define HD(a, bb) Hd[(a) * nz + (bb)]
  if (G) for (int i = 0; i < CMPC_NX; ++i) GA(i, i) = 1.0;
  double pi[3] = {0, 0, 0};
#ifdef CMPC_NO_DYN_CURV
  Hd = NULL;
#endif
  if (Hd) for (int a = 0; a < 3; ++a) pi[a] = d * lamn[6 + a];
  for (int f = 0; f < 2; ++f) {
    const double yaw = x[iy[f]], *pf = x + ip[f];
    const double cs = cos(yaw), sn = sin(yaw);
    for (int j = 0; j < nv; ++j) {
      const double *fj = u + 3 * (f * nv + j);
      const double vx = P->vert[j][0], vy = P->vert[j][1];
      double rv[3] = {cs * vx - sn * vy, sn * vx + cs * vy, 0.0};       /* R v */
      double dv[3] = {-sn * vx - cs * vy, cs * vx - sn * vy, 0.0};      /* R' v */
      double r[3] = {pf[0] + rv[0] - c[0], pf[1] + rv[1] - c[1], pf[2] + rv[2] - c[2]};
      double t[3];
      cross3(r, fj, t);
      for (int a = 0; a < 3; ++a) { tau[a] += gm[f] * t[a]; Fs[f][a] += fj[a]; }
      if (G) {
        double Sf[3][3], Sr[3][3], dt[3];
        skew3(fj, Sf); skew3(r, Sr); cross3(dv, fj, dt);
        for (int a = 0; a < 3; ++a) {
          for (int bq = 0; bq < 3; ++bq) {
            GA(6 + a, bq) += d * gm[f] * Sf[a][bq];            /* d tau / d c  */
            GA(6 + a, ip[f] + bq) -= d * gm[f] * Sf[a][bq];    /* d tau / d p  */
            GB(6 + a, 3 * (f * nv + j) + bq) = d * gm[f] * Sr[a][bq];
          }
          GA(6 + a, iy[f]) += d * gm[f] * dt[a];
          GB(3 + a, 3 * (f * nv + j) + a) = d * gm[f] / m;
        }
      }
      if (Hd) {
        double Sp[3][3], spd[3], ddv[3] = {-rv[0], -rv[1], 0.0}, t2[3];
        skew3(pi, Sp);
        for (int a = 0; a < 3; ++a) spd[a] = Sp[a][0] * dv[0] + Sp[a][1] * dv[1] + Sp[a][2] * dv[2];
        cross3(ddv, fj, t2);
        const int uf = 3 * (f * nv + j);
        for (int a = 0; a < 3; ++a) {
          for (int bq = 0; bq < 3; ++bq) {
            double val = gm[f] * Sp[a][bq];
            HD(uf + a, nu + bq) -= val; HD(nu + bq, uf + a) -= val;                 /* f - c */
            HD(uf + a, nu + ip[f] + bq) += val; HD(nu + ip[f] + bq, uf + a) += val; /* f - p */
          }
          HD(uf + a, nu + iy[f]) += gm[f] * spd[a]; HD(nu + iy[f], uf + a) += gm[f] * spd[a];
        }
        HD(nu + iy[f], nu + iy[f]) += gm[f] * (pi[0] * t2[0] + pi[1] * t2[1] + pi[2] * t2[2]);
      }
    }
  }
  for (int a = 0; a < 3; ++a) {
    double grav = (a == 2) ? -sp->g : 0.0;
    const double *cr = sr;
    xn[a] = c[a] + d * v[a];
    xn[3 + a] = v[a] + d * (grav + (gm[0] * Fs[0][a] + gm[1] * Fs[1][a]) / m);
    xn[6 + a] = hh[a] + d * tau[a];
    xn[9 + a] = th[a] + d / m * (sp->k1 * (c[a] - cr[a]) + v[a] - cr[3 + a]);
    xn[13 + a] = x[13 + a] + d * (1 - gm[0]) * u[6 * nv + a];
    xn[17 + a] = x[17 + a] + d * (1 - gm[1]) * u[6 * nv + 3 + a];
    if (G) {
      GA(a, 3 + a) = d;
      GA(9 + a, a) = d * sp->k1 / m; GA(9 + a, 3 + a) = d / m;
      GB(13 + a, 6 * nv + a) = d * (1 - gm[0]);
      GB(17 + a, 6 * nv + 3 + a) = d * (1 - gm[1]);
    }
  }
  xn[12] = x[12] + d * (1 - gm[0]) * u[6 * nv + 6];
  xn[16] = x[16] + d * (1 - gm[1]) * u[6 * nv + 7];
  if (G) { GB(12, 6 * nv + 6) = d * (1 - gm[0]); GB(16, 6 * nv + 7) = d * (1 - gm[1]); }
  for (int j = 0; j < 2 * nv; ++j) {
    xn[CMPC_NX + j] = u[3 * j + 2];
    if (G) GB(CMPC_NX + j, 3 * j + 2) = 1.0;
  }
#undef GA
#undef GB
#undef HD
}

/* ------------------------------------------------------------------------------------------
 * Cost of stage k on z=(u,x).  Returns value; adds gradient to hv (nz) and Hessian to Hm
 * (nz x nz) when non-NULL.  k == N: terminal (x only; u ignored).  reference :275-353
 * ---------------------------------------------------------------------------------------- */
static double stage_cost(const prob_t *P, int k, const double *x, const double *u, const double *uprox,
                         double *hv, double *Hm) {
  const cmpc_spec *sp = P->sp;
  const int nv = P->nv, nu = P->nu, nz = P->nz, N = P->N;
  double J = 0;
#define HV(i) hv[i]
#define HM(a, bb) Hm[(a) * nz + (bb)]
  if (k >= 1) {                                   /* tracking terms written on X[:,i+1], i = k-1 */
    const double *pr = stage_rec(P, k - 1);
    double wq[3] = {sp->w_cxy, sp->w_cxy, w_cz(sp, k - 1)};
    for (int a = 0; a < 3; ++a) {
      double e = x[a] - pr[a];
      J += wq[a] * e * e;
      if (hv) HV(nu + a) += 2 * wq[a] * e;
      if (Hm) HM(nu + a, nu + a) += 2 * wq[a];
    }
    const int iy[2] = {12, 16}, ip[2] = {13, 17};
    for (int f = 0; f < 2; ++f) {
      double g2 = gam(P, k, f); g2 *= g2;
      double wf = sp->w_foot * g2;
      for (int a = 0; a < 3; ++a) {
        double e = x[ip[f] + a] - pr[9 + 3 * f + a];
        J += wf * e * e;
        if (hv) HV(nu + ip[f] + a) += 2 * wf * e;
        if (Hm) HM(nu + ip[f] + a, nu + ip[f] + a) += 2 * wf;
      }
      double e = x[iy[f]] - pr[15 + f];
      J += wf * e * e;
      if (hv) HV(nu + iy[f]) += 2 * wf * e;
      if (Hm) HM(nu + iy[f], nu + iy[f]) += 2 * wf;
    }
  }
  if (k < N) {
    for (int a = 0; a < 3; ++a) {                 /* 1000*||hw_k||^2 (constant at k = 0) */
      J += sp->w_hw * x[6 + a] * x[6 + a];
      if (k >= 1) {
        if (hv) HV(nu + 6 + a) += 2 * sp->w_hw * x[6 + a];
        if (Hm) HM(nu + 6 + a, nu + 6 + a) += 2 * sp->w_hw;
      }
    }
    for (int f = 0; f < 2; ++f) {
      const double g1 = gam(P, k, f);
      const double a_ = g1 * g1 / nv;             /* f_bar = a_ * sum F (:277-279) */
      const double coef = nv * a_ * a_ - 2 * a_;
      double Fs[3] = {0, 0, 0}, sq = 0;
      for (int j = 0; j < nv; ++j)
        for (int a = 0; a < 3; ++a) { double fv = u[3 * (f * nv + j) + a]; Fs[a] += fv; sq += fv * fv; }
      double ss = Fs[0] * Fs[0] + Fs[1] * Fs[1] + Fs[2] * Fs[2];
      const double wa = sp->w_force * g1, wb = sp->w_force * (1 - g1);
      J += wa * (coef * ss + sq) + wb * sq;
      for (int j = 0; j < nv; ++j)
        for (int a = 0; a < 3; ++a) {
          const int iu = 3 * (f * nv + j) + a;
          if (hv) HV(iu) += wa * (2 * coef * Fs[a] + 2 * u[iu]) + wb * 2 * u[iu];
          if (Hm) {
            HM(iu, iu) += 2 * wa + 2 * wb;
            for (int l = 0; l < nv; ++l) HM(iu, 3 * (f * nv + l) + a) += 2 * wa * coef;
          }
        }
      if (k >= 1) {                               /* force-rate term (:343-351), weight gamma[k-1] */
        const double wr = sp->w_rate * gam(P, k - 1, f);
        for (int j = 0; j < nv; ++j) {
          const int iu = 3 * (f * nv + j) + 2, ix = nu + CMPC_NX + f * nv + j;
          double e = u[iu] - x[CMPC_NX + f * nv + j];
          J += wr * e * e;
          if (hv) { HV(iu) += 2 * wr * e; HV(ix) -= 2 * wr * e; }
          if (Hm) { HM(iu, iu) += 2 * wr; HM(ix, ix) += 2 * wr; HM(iu, ix) -= 2 * wr; HM(ix, iu) -= 2 * wr; }
        }
      }
    }
    for (int i = 0; i < nu; ++i) {                /* proximal term (build-defined) */
      double e = u[i] - (uprox ? uprox[i] : 0.0);
      J += 0.5 * sp->prox * e * e;
      if (hv) HV(i) += sp->prox * e;
      if (Hm) HM(i, i) += sp->prox;
    }
  }
#undef HV
#undef HM
  return J;
}

/* ------------------------------------------------------------------------------------------
 * Inequalities of stage k in the form g <= 0 (already relaxed).  act[i] = 1 for rows that
 * exist.  Jg: ni x nz dense rows (may be NULL).  If Hm != NULL adds sum_i zmul[i]*d2g_i.
 * x0norm2 = ||hw_0||^2 (for the contraction row at k = 1).  reference :193-271
 * ---------------------------------------------------------------------------------------- */
static void stage_ineq(const prob_t *P, int k, const double *x, const double *u, double x0norm2,
                       double *g, int *act, double *Jg, const double *zmul, double *Hm) {
  const cmpc_spec *sp = P->sp;
  const int nv = P->nv, nu = P->nu, nz = P->nz, ni = P->ni, N = P->N;
  const double m = P->mass, rl = sp->relax;
  for (int i = 0; i < ni; ++i) { g[i] = 0; act[i] = 0; }
  if (Jg) memset(Jg, 0, sizeof(double) * ni * nz);
#define JG(i, cidx) Jg[(i) * nz + (cidx)]
#define HM(a, bb) Hm[(a) * nz + (bb)]
  if (k < N) {
    const double *sr = stage_rec(P, k);
    const double gm[2] = {sr[17], sr[18]};
    /* Lyapunov row (:202-220) with CoM[:,k+1] = c + d v, dCoM[:,k+1] = v + d (grav + V) */
    const double d = sp->delta, k1 = sp->k1, k2 = sp->k2;
    double V[3] = {0, 0, 0}, z1[3], z2[3], un[3], gz1[3], gz2[3];
    for (int f = 0; f < 2; ++f)
      for (int j = 0; j < nv; ++j)
        for (int a = 0; a < 3; ++a) V[a] += gm[f] * u[3 * (f * nv + j) + a] / m;
    double val = 0;
    for (int a = 0; a < 3; ++a) {
      double grav = (a == 2) ? -sp->g : 0.0;
      z1[a] = x[a] + d * x[3 + a] - sr[a];
      z2[a] = k1 * z1[a] + x[3 + a] + d * (grav + V[a]) - sr[3 + a];
      un[a] = -(k1 + k2) * z2[a] + k1 * k1 * z1[a] - grav + sr[6 + a] - x[9 + a] / m;
      val += -k1 * z1[a] * z1[a] - k2 * z2[a] * z2[a] + z1[a] * z2[a] + z2[a] * (V[a] - un[a]);
      gz1[a] = -2 * k1 * z1[a] + z2[a] - k1 * k1 * z2[a];
      gz2[a] = -2 * k2 * z2[a] + z1[a] + (V[a] - un[a]) + (k1 + k2) * z2[a];
    }
    g[R_LYAP] = val - rl; act[R_LYAP] = 1;
    if (Jg) {
      for (int a = 0; a < 3; ++a) {
        if (k >= 1) {
          JG(R_LYAP, nu + a) = gz1[a] + k1 * gz2[a];
          JG(R_LYAP, nu + 3 + a) = d * gz1[a] + (k1 * d + 1) * gz2[a];
          JG(R_LYAP, nu + 9 + a) = z2[a] / m;
        }
        double dV = d * gz2[a] + z2[a];
        for (int f = 0; f < 2; ++f)
          for (int j = 0; j < nv; ++j) JG(R_LYAP, 3 * (f * nv + j) + a) = gm[f] / m * dV;
      }
    }
    if (Hm) {
      /* quadratic form in q = (c, v, theta, V): coefficient matrix hq (x) I3 */
      const double a1[4] = {1, d, 0, 0}, a2[4] = {k1, k1 * d + 1, 0, d};
      const double aV[4] = {0, 0, 0, 1}, aT[4] = {0, 0, 1.0 / m, 0};
      double hq[4][4];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
          hq[i][j] = -2 * k1 * a1[i] * a1[j] + 2 * k1 * a2[i] * a2[j] +
                     (1 - k1 * k1) * (a1[i] * a2[j] + a2[i] * a1[j]) +
                     a2[i] * (aV[j] + aT[j]) + (aV[i] + aT[i]) * a2[j];
      const double zz = zmul[R_LYAP];
      const int xo[3] = {0, 3, 9};
      for (int a = 0; a < 3; ++a) {
        if (k >= 1)
          for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) HM(nu + xo[i] + a, nu + xo[j] + a) += zz * hq[i][j];
        for (int f = 0; f < 2; ++f)
          for (int j = 0; j < nv; ++j) {
            const int iu = 3 * (f * nv + j) + a;
            if (k >= 1)
              for (int i = 0; i < 3; ++i) {
                double vv = zz * hq[i][3] * gm[f] / m;
                HM(nu + xo[i] + a, iu) += vv; HM(iu, nu + xo[i] + a) += vv;
              }
            for (int f2 = 0; f2 < 2; ++f2)
              for (int j2 = 0; j2 < nv; ++j2)
                HM(iu, 3 * (f2 * nv + j2) + a) += zz * hq[3][3] * gm[f] * gm[f2] / (m * m);
          }
      }
    }
    /* friction cone + unilateral (:236-254), rows multiplied by gamma */
    for (int f = 0; f < 2; ++f) {
      if (gm[f] == 0.0) continue;
      for (int j = 0; j < nv; ++j) {
        const int iu = 3 * (f * nv + j), r0 = R_FRIC + 5 * (f * nv + j);
        const double fx = u[iu], fy = u[iu + 1], fz = u[iu + 2], mf = P->mu_f, gg = gm[f];
        g[r0 + 0] = gg * (fx - mf * fz) - rl; g[r0 + 1] = gg * (-fx - mf * fz) - rl;
        g[r0 + 2] = gg * (fy - mf * fz) - rl; g[r0 + 3] = gg * (-fy - mf * fz) - rl;
        g[r0 + 4] = gg * (-fz) - rl;
        for (int q = 0; q < 5; ++q) act[r0 + q] = 1;
        if (Jg) {
          JG(r0 + 0, iu) = gg; JG(r0 + 0, iu + 2) = -gg * mf;
          JG(r0 + 1, iu) = -gg; JG(r0 + 1, iu + 2) = -gg * mf;
          JG(r0 + 2, iu + 1) = gg; JG(r0 + 2, iu + 2) = -gg * mf;
          JG(r0 + 3, iu + 1) = -gg; JG(r0 + 3, iu + 2) = -gg * mf;
          JG(r0 + 4, iu + 2) = -gg;
        }
      }
    }
    if (k >= 1) {                                  /* CoM height (:230) */
      g[R_CZ] = x[2] - sp->cz_max - rl; act[R_CZ] = 1;
      if (Jg) JG(R_CZ, nu + 2) = 1.0;
    }
  }
  if (k == 1) {                                    /* ||hw_1||^2 <= ||hw_0||^2 (:223-224) */
    g[R_HWC] = x[6] * x[6] + x[7] * x[7] + x[8] * x[8] - x0norm2 - rl; act[R_HWC] = 1;
    for (int a = 0; a < 3; ++a) {
      if (Jg) JG(R_HWC, nu + 6 + a) = 2 * x[6 + a];
      if (Hm) HM(nu + 6 + a, nu + 6 + a) += 2 * zmul[R_HWC];
    }
  }
  if (k >= 1) {                                    /* contact-location box (:258-271), ref index k-1 */
    const double *pr = stage_rec(P, k - 1);
    const int ip[2] = {13, 17};
    for (int f = 0; f < 2; ++f) {
      const double gg = gam(P, k, f);
      if (gg == 0.0) continue;
      for (int a = 0; a < 3; ++a) {
        const int r0 = R_BOX + 6 * f + 2 * a;
        double dd = (x[ip[f] + a] - pr[9 + 3 * f + a]) * gg;
        g[r0] = dd - sp->box[a] - rl; g[r0 + 1] = -dd - sp->box[a] - rl;
        act[r0] = act[r0 + 1] = 1;
        if (Jg) { JG(r0, nu + ip[f] + a) = gg; JG(r0 + 1, nu + ip[f] + a) = -gg; }
      }
    }
  }
#undef JG
#undef HM
}

/* ---------------------------------- workspace ------------------------------------------- */
static work_t *work_alloc(const prob_t *P) {
  const int N = P->N, nx = P->nx, nu = P->nu, nz = P->nz, ni = P->ni;
  work_t *W = (work_t *)calloc(1, sizeof(work_t));
#define AL(field, n) W->field = (double *)calloc((size_t)(n), sizeof(double))
  AL(x, (N + 1) * nx); AL(u, (N + 1) * nu); AL(lam, (N + 2) * nx); AL(s, (N + 1) * ni); AL(z, (N + 1) * ni);
  AL(uprox, (N + 1) * nu);
  AL(G, (N + 1) * nx * nz); AL(b, (N + 1) * nx); AL(H, (N + 1) * nz * nz); AL(h, (N + 1) * nz);
  AL(hobj, (N + 1) * nz);
  AL(g, (N + 1) * ni); AL(Jg, (size_t)(N + 1) * ni * nz);
  AL(Lam, (N + 1) * nu * nu); AL(Ls, (N + 1) * nx * nu); AL(l, (N + 1) * nu);
  AL(P, (N + 2) * nx * nx); AL(p, (N + 2) * nx);
  AL(dx, (N + 1) * nx); AL(du, (N + 1) * nu); AL(lamn, (N + 2) * nx); AL(ds, (N + 1) * ni); AL(dz, (N + 1) * ni);
  AL(M, nz * nz); AL(m, nz);
#undef AL
  W->act = (int *)calloc((size_t)(N + 1) * ni, sizeof(int));
  return W;
}
static void work_free(work_t *W) {
  double **f[] = {&W->x, &W->u, &W->lam, &W->s, &W->z, &W->uprox, &W->G, &W->b, &W->H, &W->h, &W->hobj,
                  &W->g, &W->Jg, &W->Lam, &W->Ls, &W->l, &W->P, &W->p, &W->dx, &W->du, &W->lamn,
                  &W->ds, &W->dz, &W->M, &W->m};
  for (size_t i = 0; i < sizeof(f) / sizeof(f[0]); ++i) free(*f[i]);
  free(W->act); free(W);
}

/* Riccati backward sweep on the barrier-augmented stage QPs.  Returns 0, or -1 if a pivot of
 * R + B'PB is not positive (wrong inertia). */
/* returns 0, or -(k + 1) when the factorisation of stage k meets a pivot below the threshold (wrong inertia) */
static int riccati_backward(const prob_t *P, work_t *W, double reg) {
  const int N = P->N, nx = P->nx, nu = P->nu, nz = P->nz;
  double *M = W->M, *mv = W->m;
  /* smallest pivot accepted: with a regularisation delta that barely repairs the inertia a pivot can be
   * ~1e-14 and the step along that direction is garbage of size 1e+14 * residual (there is no line search
   * to reject it); asking for PIV_FRAC * delta means delta must clear |lambda_min| by that margin */
  const double piv_min = fmax(PIV_MIN, PIV_FRAC * reg);
  /* terminal */
  {
    const double *H = W->H + (size_t)N * nz * nz, *h = W->h + (size_t)N * nz;
    double *Pn = W->P + (size_t)N * nx * nx, *pn = W->p + (size_t)N * nx;
    for (int i = 0; i < nx; ++i) {
      for (int j = 0; j < nx; ++j) Pn[i * nx + j] = H[(nu + i) * nz + nu + j] + (i == j ? reg : 0.0);
      pn[i] = h[nu + i];
    }
  }
  double *T = (double *)malloc(sizeof(double) * nx * nz);
  double *pv = (double *)malloc(sizeof(double) * nx);
  int rc = 0;
  for (int k = N - 1; k >= 0 && rc == 0; --k) {
    const double *G = W->G + (size_t)k * nx * nz, *b = W->b + (size_t)k * nx;
    const double *H = W->H + (size_t)k * nz * nz, *h = W->h + (size_t)k * nz;
    const double *Pn = W->P + (size_t)(k + 1) * nx * nx, *pn = W->p + (size_t)(k + 1) * nx;
    /* T = Pn G ; pv = pn + Pn b */
    for (int i = 0; i < nx; ++i) {
      for (int j = 0; j < nz; ++j) {
        double a = 0;
        for (int q = 0; q < nx; ++q) a += Pn[i * nx + q] * G[q * nz + j];
        T[i * nz + j] = a;
      }
      double a = pn[i];
      for (int q = 0; q < nx; ++q) a += Pn[i * nx + q] * b[q];
      pv[i] = a;
    }
    for (int i = 0; i < nz; ++i) {
      for (int j = 0; j < nz; ++j) {
        double a = H[i * nz + j] + (i == j ? reg : 0.0);
        for (int q = 0; q < nx; ++q) a += G[q * nz + i] * T[q * nz + j];
        M[i * nz + j] = a;
      }
      double a = h[i];
      for (int q = 0; q < nx; ++q) a += G[q * nz + i] * pv[q];
      mv[i] = a;
    }
    /* Cholesky of the u-block, Ls, Schur complement */
    double *Lm = W->Lam + (size_t)k * nu * nu, *Ls = W->Ls + (size_t)k * nx * nu, *l = W->l + (size_t)k * nu;
    double *Pk = W->P + (size_t)k * nx * nx, *pk = W->p + (size_t)k * nx;
    for (int j = 0; j < nu && rc == 0; ++j) {
      double dsum = M[j * nz + j];
      for (int q = 0; q < j; ++q) dsum -= Lm[j * nu + q] * Lm[j * nu + q];
      if (!(dsum > piv_min)) { rc = -(k + 1); break; }      /* (the failing stage, for the diagnostics) */
      double dj = sqrt(dsum);
      Lm[j * nu + j] = dj;
      for (int i = j + 1; i < nu; ++i) {
        double a = M[i * nz + j];
        for (int q = 0; q < j; ++q) a -= Lm[i * nu + q] * Lm[j * nu + q];
        Lm[i * nu + j] = a / dj;
      }
      for (int i = 0; i < nx; ++i) {
        double a = M[(nu + i) * nz + j];
        for (int q = 0; q < j; ++q) a -= Ls[i * nu + q] * Lm[j * nu + q];
        Ls[i * nu + j] = a / dj;
      }
      double a = mv[j];
      for (int q = 0; q < j; ++q) a -= l[q] * Lm[j * nu + q];
      l[j] = a / dj;
    }
    if (rc) break;
    for (int i = 0; i < nx; ++i) {
      for (int j = 0; j < nx; ++j) {
        double a = M[(nu + i) * nz + nu + j];
        for (int q = 0; q < nu; ++q) a -= Ls[i * nu + q] * Ls[j * nu + q];
        Pk[i * nx + j] = a;
      }
      double a = mv[nu + i];
      for (int q = 0; q < nu; ++q) a -= Ls[i * nu + q] * l[q];
      pk[i] = a;
    }
    for (int i = 0; i < nx; ++i)                    /* symmetrise */
      for (int j = 0; j < i; ++j) { double a = 0.5 * (Pk[i * nx + j] + Pk[j * nx + i]); Pk[i * nx + j] = Pk[j * nx + i] = a; }
  }
  free(T); free(pv);
  return rc;
}

static void riccati_forward(const prob_t *P, work_t *W) {
  const int N = P->N, nx = P->nx, nu = P->nu, nz = P->nz;
  for (int i = 0; i < nx; ++i) W->dx[i] = 0.0;
  for (int k = 0; k < N; ++k) {
    const double *Lm = W->Lam + (size_t)k * nu * nu, *Ls = W->Ls + (size_t)k * nx * nu, *l = W->l + (size_t)k * nu;
    const double *G = W->G + (size_t)k * nx * nz, *b = W->b + (size_t)k * nx;
    double *dx = W->dx + (size_t)k * nx, *du = W->du + (size_t)k * nu, *dxn = W->dx + (size_t)(k + 1) * nx;
    double t[MAXNU];
    for (int j = 0; j < nu; ++j) {
      double a = l[j];
      for (int i = 0; i < nx; ++i) a += Ls[i * nu + j] * dx[i];
      t[j] = -a;
    }
    for (int j = nu - 1; j >= 0; --j) {             /* Lam^T du = t */
      double a = t[j];
      for (int q = j + 1; q < nu; ++q) a -= Lm[q * nu + j] * du[q];
      du[j] = a / Lm[j * nu + j];
    }
    for (int i = 0; i < nx; ++i) {
      double a = b[i];
      for (int j = 0; j < nu; ++j) a += G[i * nz + j] * du[j];
      for (int j = 0; j < nx; ++j) a += G[i * nz + nu + j] * dx[j];
      dxn[i] = a;
    }
    const double *Pn = W->P + (size_t)(k + 1) * nx * nx, *pn = W->p + (size_t)(k + 1) * nx;
    double *ln = W->lamn + (size_t)(k + 1) * nx;
    for (int i = 0; i < nx; ++i) {
      double a = pn[i];
      for (int j = 0; j < nx; ++j) a += Pn[i * nx + j] * dxn[j];
      ln[i] = a;
    }
  }
}

typedef struct { int iters, status; double kkt, mu, reg_last; int n_reg; double waste; } stats_t;

static void initial_point(const prob_t *P, work_t *W, const double *warm, const int *src) {
  const int N = P->N, nx = P->nx, nu = P->nu, nv = P->nv;
  memset(W->x, 0, sizeof(double) * (N + 1) * nx);
  memset(W->u, 0, sizeof(double) * (N + 1) * nu);
  memset(W->uprox, 0, sizeof(double) * (N + 1) * nu);
  if (warm) {
    for (int k = 0; k <= N; ++k) memcpy(W->x + (size_t)k * nx, warm + (size_t)(src ? src[k] : k) * CMPC_NX, sizeof(double) * CMPC_NX);
    for (int k = 0; k < N; ++k) {
      const int ks = src ? (src[k] < N ? src[k] : N - 1) : k;
      memcpy(W->u + (size_t)k * nu, warm + (size_t)CMPC_NX * (N + 1) + (size_t)ks * nu, sizeof(double) * nu);
      memcpy(W->uprox + (size_t)k * nu, W->u + (size_t)k * nu, sizeof(double) * nu);
    }
  } else {
    for (int k = 0; k <= N; ++k) memcpy(W->x + (size_t)k * nx, P->rec, sizeof(double) * CMPC_NX);
    for (int k = 0; k < N; ++k) {
      double gl = gam(P, k, 0), gr = gam(P, k, 1);
      double fz = P->mass * P->sp->g / (nv * (gl + gr));
      for (int j = 0; j < nv; ++j) {
        W->u[(size_t)k * nu + 3 * j + 2] = fz * gl;
        W->u[(size_t)k * nu + 3 * (nv + j) + 2] = fz * gr;
      }
    }
  }
  memcpy(W->x, P->rec, sizeof(double) * CMPC_NX);          /* x_0 is data */
  for (int k = 1; k <= N; ++k)
    for (int j = 0; j < 2 * nv; ++j) W->x[(size_t)k * nx + CMPC_NX + j] = W->u[(size_t)(k - 1) * nu + 3 * j + 2];
#if COLD_ROLLOUT
  if (!warm) {
    /* dynamics-consistent cold start: the states are rolled out from x_0 under the initial inputs, x_{k+1} = F(x_k, u_k),
     * instead of standing still at x_0 (zero dynamics defect at the first iterate) */
    double *xn = (double *)malloc(sizeof(double) * nx);
    for (int k = 0; k < N; ++k) {
      stage_dynamics(P, k, W->x + (size_t)k * nx, W->u + (size_t)k * nu, xn, NULL, NULL, NULL);
      memcpy(W->x + (size_t)(k + 1) * nx, xn, sizeof(double) * nx);
    }
    free(xn);
  }
#endif
}

/* X (20 x (N+1)) then U (nu x N), the reference's layout */
static void write_solution(const prob_t *P, const work_t *W, double *out) {
  const int N = P->N, nx = P->nx, nu = P->nu;
  for (int k = 0; k <= N; ++k) memcpy(out + (size_t)k * CMPC_NX, W->x + (size_t)k * nx, sizeof(double) * CMPC_NX);
  for (int k = 0; k < N; ++k)
    memcpy(out + (size_t)CMPC_NX * (N + 1) + (size_t)k * nu, W->u + (size_t)k * nu, sizeof(double) * nu);
}

/* One interior-point solve.  out: X then U (reference layout). */
static void solve_one_capped(const cmpc_spec *sp, const double *rec, const double *warm, double *out,
                             stats_t *st, int verbose, double *full, const double *state_in, double *state_out, int cap,
                             double kkt_saved_in);
static void solve_one(const cmpc_spec *sp, const double *rec, const double *warm, double *out,
                      stats_t *st, int verbose, double *full, const double *state_in, double *state_out) {
  solve_one_capped(sp, rec, warm, out, st, verbose, full, state_in, state_out, sp->max_iter, INFINITY);
}
/* cap: iteration budget of this attempt (a resumed attempt and the plain one that may follow it share max_iter) */
/* kkt_saved_in: error of the acceptable point a failed resumed attempt left in `out` (INFINITY: none) -- the plain attempt
 * that follows it only replaces that point by a better one, and falls back on it like on a saved point of its own */
static void solve_one_capped(const cmpc_spec *sp, const double *rec, const double *warm, double *out,
                             stats_t *st, int verbose, double *full, const double *state_in, double *state_out, int cap,
                             double kkt_saved_in) {
  prob_t Pb; prob_init(&Pb, sp, rec);
  const prob_t *P = &Pb;
  const int N = P->N, nx = P->nx, nu = P->nu, nz = P->nz, ni = P->ni;
  work_t *W = work_alloc(P);
  /* solver state (CMPC_NSTATE doubles): [XU of the snapshot | lam (N+1) x nx | s (N+1) x ni | z (N+1) x ni | mu, 7 spare] */
  const size_t nsol_ = CMPC_NSOL(N, P->nv), o_lam = nsol_, o_s = o_lam + (size_t)(N + 1) * nx,
               o_z = o_s + (size_t)(N + 1) * ni, o_mu = o_z + (size_t)(N + 1) * ni;
  const int resume = state_in && state_in[o_mu] > 0.0 && isfinite(state_in[o_mu]);
  /* The horizon moves by one stage per tick, but the stage INDEX carries structure of its own (contraction row at
   * node 1, height weight e^{-i}): resuming stage k from the old stage k (unshifted, as the reference's set_initial,
   * :630-631) measured 12.6 iterations per tick on the walk against 15.5 for the shifted state. */
  /* Stage k resumes from the state's stage k -- unless a contact switch has moved: where the flags of stage k no
   * longer match the state's stage k but do match its stage k + 1 (the switch sits one stage earlier than a tick
   * ago), that stage is taken instead. */
  const size_t o_fl = o_mu + 8;
  int src[CMPC_MAX_N + 1];
  for (int k = 0; k <= N; ++k) {
    src[k] = k;
    if (resume) {
      const double gl = gam(P, k, 0), gr = gam(P, k, 1);
      const int same = gl == state_in[o_fl + k] && gr == state_in[o_fl + N + 1 + k];
      const int next = k < N && gl == state_in[o_fl + k + 1] && gr == state_in[o_fl + N + 1 + k + 1];
      if (!same && next) src[k] = k + 1;
    }
  }
  initial_point(P, W, resume ? state_in : warm, resume ? src : NULL);
  if (resume) {                              /* the proximal centre stays the caller's warm_XU (or 0) */
    memset(W->uprox, 0, sizeof(double) * (N + 1) * nu);
    if (warm) for (int k = 0; k < N; ++k)
      memcpy(W->uprox + (size_t)k * nu, warm + (size_t)CMPC_NX * (N + 1) + (size_t)k * nu, sizeof(double) * nu);
  }
  if (state_out) state_out[o_mu] = 0.0;      /* invalid until a snapshot is taken */
  const double x0n2 = rec[6] * rec[6] + rec[7] * rec[7] + rec[8] * rec[8];
  double mu = resume ? state_in[o_mu] : MU_INIT, reg_last = 0.0;
  const double tol = sp->tol;
  /* slacks / multipliers */
  for (int k = 0; k <= N; ++k) {
    stage_ineq(P, k, W->x + (size_t)k * nx, W->u + (size_t)k * nu, x0n2, W->g + (size_t)k * ni,
               W->act + (size_t)k * ni, NULL, NULL, NULL);
    for (int i = 0; i < ni; ++i) {
      const size_t e = (size_t)k * ni + i, es = (size_t)src[k] * ni + i;
      double gi = W->g[e];
      const double s0 = resume ? state_in[o_s + es] : 0.0, z0 = resume ? state_in[o_z + es] : 0.0;
      if (resume && W->act[e] && s0 > 0.0 && z0 > 0.0) {     /* row carried over from the snapshot */
        W->s[e] = s0; W->z[e] = z0;
      } else {                                                  /* cold rule (also: rows a contact switch has just activated) */
        W->s[e] = W->act[e] ? fmax(-gi, fmin(S_FLOOR, sqrt(mu))) : 1.0;
        W->z[e] = W->act[e] ? mu / W->s[e] : 0.0;
      }
    }
  }
  if (resume) for (int k = 0; k <= N; ++k)
    memcpy(W->lam + (size_t)k * nx, state_in + o_lam + (size_t)src[k] * nx, sizeof(double) * nx);
  int snapped = 0;
  st->status = CMPC_MAX_ITER; st->n_reg = 0; st->waste = 0.0;
  int it, n_acc = 0, n_stall = 0, polish = -1, since_best = 0, use_saved = 0;
  double kkt_best = INFINITY, kkt_saved = kkt_saved_in;
  const double acc_tol = fmax(sp->acc_tol, tol);
  /* every iterate the acceptable-level counter n_acc counts is also saved: with acc_tol < ACC_FACTOR * tol the
   * counter could otherwise end the run with nothing in `out` */
  const double save_tol = fmax(acc_tol, ACC_FACTOR * tol);
  double dbg_ap = 0, dbg_ad = 0;
  int dbg_bk = -1, dbg_bi = -1;       /* row that limited the last primal step (verbose trace) */
  double kkt = INFINITY;
  double *xn = (double *)malloc(sizeof(double) * nx);
  for (it = 0; it <= cap; ++it) {
    /* ---- linearise every stage ---- */
    double e_d = 0, e_p = 0, e_c = 0, e_cmu = 0, sum_mult = 0, fobj = 0;
    int n_mult = 0, dbg_k = -1, dbg_j = -1;
    for (int k = 0; k <= N; ++k) {
      double *x = W->x + (size_t)k * nx, *u = W->u + (size_t)k * nu;
      double *H = W->H + (size_t)k * nz * nz, *ho = W->hobj + (size_t)k * nz;
      double *g = W->g + (size_t)k * ni, *Jg = W->Jg + (size_t)k * ni * nz;
      double *s = W->s + (size_t)k * ni, *z = W->z + (size_t)k * ni;
      int *act = W->act + (size_t)k * ni;
      memset(H, 0, sizeof(double) * nz * nz); memset(ho, 0, sizeof(double) * nz);
      fobj += stage_cost(P, k, x, u, W->uprox + (size_t)k * nu, ho, H);
      stage_ineq(P, k, x, u, x0n2, g, act, Jg, z, H);
      if (k < N) {
        stage_dynamics(P, k, x, u, xn, W->G + (size_t)k * nx * nz, W->lam + (size_t)(k + 1) * nx, H);
        for (int i = 0; i < nx; ++i) {
          W->b[(size_t)k * nx + i] = xn[i] - W->x[(size_t)(k + 1) * nx + i];
          e_p = fmax(e_p, fabs(W->b[(size_t)k * nx + i]));
        }
      }
      /* dual residual: grad_obj + Jg' z + G' lam_{k+1} - [0; lam_k] */
      for (int j = 0; j < nz; ++j) {
        double r = ho[j];
        for (int i = 0; i < ni; ++i) if (act[i]) r += Jg[i * nz + j] * z[i];
        if (k < N) for (int q = 0; q < nx; ++q) r += W->G[(size_t)k * nx * nz + q * nz + j] * W->lam[(size_t)(k + 1) * nx + q];
        if (j >= nu) r -= W->lam[(size_t)k * nx + (j - nu)];
        int is_var = (j < nu) ? (k < N) : (k >= 1);
        if (is_var && fabs(r) > e_d) { dbg_k = k; dbg_j = j; }
        if (is_var) e_d = fmax(e_d, fabs(r));
      }
      for (int i = 0; i < ni; ++i) if (act[i]) {
        e_p = fmax(e_p, fabs(g[i] + s[i]));
        e_c = fmax(e_c, fabs(s[i] * z[i])); e_cmu = fmax(e_cmu, fabs(s[i] * z[i] - mu));
        sum_mult += fabs(z[i]); ++n_mult;
      }
      if (k >= 1) for (int i = 0; i < nx; ++i) { sum_mult += fabs(W->lam[(size_t)k * nx + i]); ++n_mult; }
    }
    double sd = fmax(100.0, sum_mult / n_mult) / 100.0;
    kkt = fmax(fmax(e_d / sd, e_p), e_c / sd);
    if (verbose)
      printf("it %3d f=%.8e d=%.2e p=%.2e c=%.2e mu=%.1e reg=%.1e  (last step: ap %.3f ad %.3f, limited by row %d of stage %d; largest dual residual: stage %d column %d)\n", it, fobj, e_d / sd, e_p, e_c / sd, mu, reg_last, dbg_ap, dbg_ad, dbg_bi, dbg_bk, dbg_k, dbg_j);
    if (polish >= 0 && kkt > ACC_FACTOR * tol) {
      /* polishing lost ground (the step at the final barrier value needed an inertia correction): the
       * point that met the tolerance was written to `out` before the polish and is what is returned */
      st->status = CMPC_CONVERGED; kkt = kkt_saved; use_saved = 1; break;
    }
    if (polish < 0) {
      /* best acceptable iterate so far: at the final barrier value the KKT systems are ill conditioned
       * (z/s up to 1e+15) and an iterate within a few percent of the tolerance can be followed by worse
       * ones; whatever ends the run, the best point seen is what is returned */
      if (kkt <= save_tol && kkt < kkt_saved) { write_solution(P, W, out); kkt_saved = kkt; }
      if (kkt <= tol) {
        if (state_out && !snapped && mu >= MU_WARM) {   /* (the tolerance was met from a level >= MU_WARM: same snapshot) */
          write_solution(P, W, state_out);
          memcpy(state_out + o_lam, W->lam, sizeof(double) * (N + 1) * nx);
          memcpy(state_out + o_s, W->s, sizeof(double) * (N + 1) * ni);
          memcpy(state_out + o_z, W->z, sizeof(double) * (N + 1) * ni);
          state_out[o_mu] = mu;
          for (int k = 0; k <= N; ++k) { state_out[o_fl + k] = gam(P, k, 0); state_out[o_fl + N + 1 + k] = gam(P, k, 1); }
          snapped = 1;
        }
        polish = POLISH_ITERS; mu = tol / 10;
      } else {
        /* IPOPT-style acceptable level: ACC_ITERS consecutive iterates within ACC_FACTOR*tol */
        n_acc = (kkt <= ACC_FACTOR * tol) ? n_acc + 1 : 0;
        if (n_acc >= ACC_ITERS) { st->status = CMPC_ACCEPTABLE; kkt = kkt_saved; use_saved = 1; break; }
        {
          /* Progress watch on the error of the current barrier problem (at the final barrier value: the KKT error):
           * NOPROG_ITERS iterations at one barrier value without halving the best error seen there end the run as
           * "acceptable" once a point within acc_tol is in hand.  Round 3 watched the final barrier value only; the
           * instances at the very end of the tail hover one level above it (primal feasible, complementary, dual
           * residual oscillating at 1e-4 under an inertia correction: up to 85 iterations at mu = 1.8e-7 before the
           * cap) and end "acceptable" either way.  The watch restarts whenever the barrier value changes. */
          const double kw = (mu <= tol / 10) ? kkt : fmax(fmax(e_d / sd, e_p), e_cmu / sd);
          if (kw < NOPROG_FACTOR * kkt_best) { kkt_best = kw; since_best = 0; } else ++since_best;
          if (since_best >= NOPROG_ITERS && kkt_saved <= acc_tol) {
            st->status = CMPC_ACCEPTABLE; kkt = kkt_saved; use_saved = 1; break;
          }
        }
      }
    }
    if (polish == 0) {
      /* The polish step is there to land on the mu = tol/10 central-path point whichever path led to the tolerance.  Where
       * it ends ABOVE the tolerance with a larger error than the point that met it (an inertia correction in the polish
       * step, a saddle-type end point: up to 100 * tol passed as "converged" in rounds 1-3 and replaced the better
       * point), that point -- written to `out` before the polish -- is what is returned. */
      st->status = CMPC_CONVERGED;
      if (kkt > tol && kkt > kkt_saved) { kkt = kkt_saved; use_saved = 1; }
      break;
    }
    /* a resumed solve that is still at the state's barrier value after RESUME_RECENTRE_ITERS iterations: the state does
     * not fit this tick's problem (a push, a re-planned contact); give up here instead of crawling to the cap -- the
     * plain solve follows with the rest of the budget (round-3 advisor: one stale state stretched a closed-loop launch
     * to twice the longest cold solve) */
    const int stale = resume && it >= RESUME_RECENTRE_ITERS && polish < 0 && mu == state_in[o_mu];
    if (it == cap || !isfinite(kkt) || n_stall >= STALL_ITERS || stale) {
      if (polish >= 0) st->status = CMPC_CONVERGED;               /* (cap reached inside the polish) */
      else if (kkt_saved <= acc_tol && !stale) { st->status = CMPC_ACCEPTABLE; kkt = kkt_saved; use_saved = 1; }
      else st->status = (it == cap && !stale) ? CMPC_MAX_ITER : CMPC_NUMERICAL;
      break;
    }
    if (polish > 0) --polish;
    else if (!(resume && it == 0)) {
      /* (a resumed solve takes one Newton step at the state's barrier value first, whatever the error there: the
       * state written below is then a central-path point of THIS tick's problem, not a copy of the one read --
       * copies went stale and every other tick paid 18 iterations instead of 7) */
      const double mu_before = mu;
      while (mu > tol / 10 && fmax(fmax(e_d / sd, e_p), e_cmu / sd) < KAPPA_EPS * mu)
        mu = fmax(tol / 10, fmin(MU_FACTOR * mu, (MU_POWER == 1.5) ? mu * sqrt(mu) : pow(mu, MU_POWER)));
      if (mu != mu_before) { kkt_best = INFINITY; since_best = 0; }   /* a new barrier problem: the progress watch restarts */
      if (state_out && !snapped && mu_before >= MU_WARM && mu < MU_WARM) {
        /* this iterate solves the barrier problem at mu_before: the state the next tick resumes from */
        write_solution(P, W, state_out);
        memcpy(state_out + o_lam, W->lam, sizeof(double) * (N + 1) * nx);
        memcpy(state_out + o_s, W->s, sizeof(double) * (N + 1) * ni);
        memcpy(state_out + o_z, W->z, sizeof(double) * (N + 1) * ni);
        state_out[o_mu] = mu_before;
        for (int k = 0; k <= N; ++k) { state_out[o_fl + k] = gam(P, k, 0); state_out[o_fl + N + 1 + k] = gam(P, k, 1); }
        snapped = 1;
      }
    }
    /* ---- barrier-augmented QP data ---- */
    for (int k = 0; k <= N; ++k) {
      double *H = W->H + (size_t)k * nz * nz, *h = W->h + (size_t)k * nz, *ho = W->hobj + (size_t)k * nz;
      double *g = W->g + (size_t)k * ni, *Jg = W->Jg + (size_t)k * ni * nz;
      double *s = W->s + (size_t)k * ni, *z = W->z + (size_t)k * ni;
      int *act = W->act + (size_t)k * ni;
      memcpy(h, ho, sizeof(double) * nz);
      for (int i = 0; i < ni; ++i) if (act[i]) {
        double sig = z[i] / s[i], kap = mu / s[i] + sig * (g[i] + s[i]);
        const double *a = Jg + (size_t)i * nz;
        for (int p = 0; p < nz; ++p) if (a[p] != 0.0) {
          h[p] += a[p] * kap;
          for (int q = 0; q < nz; ++q) H[p * nz + q] += sig * a[p] * a[q];
        }
      }
    }
    /* ---- factorise with inertia correction ---- */
    double reg = 0.0;
    int fail = 0;
    int rb_;
    while ((rb_ = riccati_backward(P, W, reg)) != 0) {
      st->waste += (double)(N - (-rb_ - 1) + 1) / (N + 1);   /* share of a sweep done before the failing stage */
      if (reg == 0.0) reg = (reg_last == 0.0) ? 1e-4 : fmax(1e-20, reg_last / 3);
      else reg *= (reg_last == 0.0) ? REG_FIRST_FACTOR : REG_NEXT_FACTOR;
      ++st->n_reg;
      if (reg > 1e20) { fail = 1; break; }
    }
    if (fail) { st->status = CMPC_NUMERICAL; break; }
    if (reg > 0) reg_last = reg;
    riccati_forward(P, W);
    /* ---- slack / multiplier steps, fraction to the boundary ---- */
    double tau = fmax(TAU_MIN, 1 - mu), ap = 1.0, ad = 1.0;
    for (int k = 0; k <= N; ++k) {
      const double *Jg = W->Jg + (size_t)k * ni * nz, *g = W->g + (size_t)k * ni;
      const double *s = W->s + (size_t)k * ni, *z = W->z + (size_t)k * ni;
      const int *act = W->act + (size_t)k * ni;
      for (int i = 0; i < ni; ++i) {
        double ds = 0, dz = 0;
        if (act[i]) {
          double jd = 0;
          if (k < N) for (int p = 0; p < nu; ++p) jd += Jg[(size_t)i * nz + p] * W->du[(size_t)k * nu + p];
          for (int p = 0; p < nx; ++p) jd += Jg[(size_t)i * nz + nu + p] * W->dx[(size_t)k * nx + p];
          ds = -(g[i] + s[i]) - jd;
          dz = (mu - s[i] * z[i] - z[i] * ds) / s[i];
          if (ds < 0 && -tau * s[i] / ds < ap) { dbg_bk = k; dbg_bi = i; }
          if (ds < 0) ap = fmin(ap, -tau * s[i] / ds);
          if (dz < 0) ad = fmin(ad, -tau * z[i] / dz);
        }
        W->ds[(size_t)k * ni + i] = ds; W->dz[(size_t)k * ni + i] = dz;
      }
    }
    n_stall = (ap < STALL_STEP) ? n_stall + 1 : 0;   /* no room to move: locally infeasible */
    dbg_ap = ap; dbg_ad = ad;
    for (int k = 0; k <= N; ++k) {
      if (k >= 1) for (int i = 0; i < nx; ++i) {
        W->x[(size_t)k * nx + i] += ap * W->dx[(size_t)k * nx + i];
        W->lam[(size_t)k * nx + i] += ap * (W->lamn[(size_t)k * nx + i] - W->lam[(size_t)k * nx + i]);
      }
      if (k < N) for (int i = 0; i < nu; ++i) W->u[(size_t)k * nu + i] += ap * W->du[(size_t)k * nu + i];
      for (int i = 0; i < ni; ++i) if (W->act[(size_t)k * ni + i]) {
        double *s = &W->s[(size_t)k * ni + i], *z = &W->z[(size_t)k * ni + i];
        *s += ap * W->ds[(size_t)k * ni + i];
        *z += ad * W->dz[(size_t)k * ni + i];
        double lo = mu / *s / 1e10, hi = mu / *s * 1e10;
        *z = fmin(fmax(*z, lo), hi);
      }
    }
  }
  free(xn);
  /* A resumed attempt that failed (stale state, collapsed steps, its share of the cap) is followed by a plain one with the
   * rest of the budget.  An acceptable point it saved on the way is NOT given up (round-4 advisor: a tick that used to end
   * "acceptable" could end at the cap): it stays in `out`, the plain attempt starts with its error as the level to beat,
   * and with no budget left for a plain attempt it is the answer. */
  const int failed_resume = resume && (st->status == CMPC_MAX_ITER || st->status == CMPC_NUMERICAL);
#ifdef CMPC_ORACLE_DROP_SAVED_ON_RETRY        /* (the behaviour of rounds 3-5, for tests/tools that show the difference) */
  const int keep_saved = 0;
#else
  const int keep_saved = failed_resume && kkt_saved <= acc_tol;
#endif
  if (keep_saved && !(it < sp->max_iter)) { st->status = CMPC_ACCEPTABLE; kkt = kkt_saved; use_saved = 1; }
  if (!use_saved && !keep_saved) write_solution(P, W, out);
  if (full) {   /* x, lam ((N+1) x nx each), s, z ((N+1) x ni each) */
    memcpy(full, W->x, sizeof(double) * (N + 1) * nx); full += (N + 1) * nx;
    memcpy(full, W->lam, sizeof(double) * (N + 1) * nx); full += (N + 1) * nx;
    memcpy(full, W->s, sizeof(double) * (N + 1) * ni); full += (N + 1) * ni;
    memcpy(full, W->z, sizeof(double) * (N + 1) * ni); full += (N + 1) * ni;
    full[0] = mu; full[1] = reg_last; full[2] = dbg_ap; full[3] = dbg_ad; full[4] = st->n_reg;
    if (N >= 18) for (int kk = 15; kk <= 18; ++kk) memcpy(full + 20000 + (kk - 15) * 1000, W->P + (size_t)kk * nx * nx, sizeof(double) * nx * nx);
    full += 8;
    for (int k = 0; k <= N; ++k) {   /* vectors of the last Newton step */
      double *o = full + (size_t)k * (nz + 3 * nx + 3 * nu);
      for (int e = 0; e < nu; ++e) o[nz + 3 * nx + 2 * nu + e] = W->Lam[(size_t)k * nu * nu + e * nu + e];
      memcpy(o, W->h + (size_t)k * nz, sizeof(double) * nz);
      memcpy(o + nz, W->b + (size_t)k * nx, sizeof(double) * nx);
      memcpy(o + nz + nx, W->l + (size_t)k * nu, sizeof(double) * nu);
      memcpy(o + nz + nx + nu, W->p + (size_t)k * nx, sizeof(double) * nx);
      memcpy(o + nz + 2 * nx + nu, W->du + (size_t)k * nu, sizeof(double) * nu);
      memcpy(o + nz + 2 * nx + 2 * nu, W->dx + (size_t)k * nx, sizeof(double) * nx);
    }
  }
  st->iters = it; st->kkt = kkt; st->mu = mu; st->reg_last = reg_last;
  work_free(W);
  if (resume && (st->status == CMPC_MAX_ITER || st->status == CMPC_NUMERICAL)) {
    /* the resumed solve got nowhere (the state was too far from this tick's problem): start again the plain way;
     * the iterations of both attempts are reported */
    const int spent = st->iters;
    if (spent < sp->max_iter) {                /* both attempts together stay within max_iter (+ 1: `iters` <= max_iter + 1) */
      solve_one_capped(sp, rec, warm, out, st, verbose, full, NULL, state_out, sp->max_iter - spent,
                       keep_saved ? kkt_saved : INFINITY);
      st->iters += spent;
    }
  }
  /* first spare word of the state: what this solve took (the kernel's launch queues its instances by it) */
  if (state_out) state_out[o_mu + 1] = (double)st->iters;
}

/* ------------------------------------ exported API -------------------------------------- */
void cmpc_oracle_default_spec(cmpc_spec *s, int32_t N, int32_t nv) {
  memset(s, 0, sizeof(*s));
  s->struct_size = (int32_t)sizeof(*s);
  s->N = N; s->nv = nv; s->max_iter = 200;
  s->delta = 0.01; s->g = 9.81; s->k1 = 4.0; s->k2 = 0.1; s->w_rate = 1.0;
  s->w_hw = 1000.0; s->w_cxy = 1.0; s->w_cz_const = 2000.0; s->w_foot = 1000.0; s->w_force = 10.0;
  s->cz_max = 0.76; s->box[0] = 0.01; s->box[1] = 0.005; s->box[2] = 0.00005;
  s->foot_length = 0.25; s->foot_width = 0.13; s->prox = 1e-4; s->relax = 1e-8; s->tol = 1e-10;
  s->acc_tol = 1e-4;
}

int cmpc_oracle_solve(const cmpc_spec *sp, const double *rec, const double *warm, double *out,
                      int32_t *status, int32_t *iters, double *kkt, int verbose) {
  if (sp->N < 1 || sp->N > CMPC_MAX_N || (sp->nv != 4 && sp->nv != 8)) return 1;
  stats_t st;
  solve_one(sp, rec, warm, out, &st, verbose, NULL, NULL, NULL);
  if (status) *status = st.status;
  if (iters) *iters = st.iters;
  if (kkt) *kkt = st.kkt;
  return 0;
}

int cmpc_oracle_solve_batch(const cmpc_spec *sp, int32_t B, const double *recs, const double *warm,
                            double *out, int32_t *status, int32_t *iters, double *kkt, int nthreads) {
  if (sp->N < 1 || sp->N > CMPC_MAX_N || (sp->nv != 4 && sp->nv != 8)) return 1;
  const size_t nrec = CMPC_NREC(sp->N), nsol = CMPC_NSOL(sp->N, sp->nv);
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int b = 0; b < B; ++b) {
    stats_t st;
    solve_one(sp, recs + b * nrec, warm ? warm + b * nsol : NULL, out + b * nsol, &st, 0, NULL, NULL, NULL);
    if (status) status[b] = st.status;
    if (iters) iters[b] = st.iters;
    if (kkt) kkt[b] = st.kkt;
  }
  return 0;
}

/* Diagnostic (tools/launch_model.py): as cmpc_oracle_solve_batch, plus the number of factorisation retries (inertia
 * corrections) of every solve -- each costs the kernel up to one more matrix sweep. */
int cmpc_oracle_solve_batch_stats(const cmpc_spec *sp, int32_t B, const double *recs, const double *warm,
                                  double *out, int32_t *status, int32_t *iters, double *kkt, int32_t *n_reg, double *waste, int nthreads) {
  if (sp->N < 1 || sp->N > CMPC_MAX_N || (sp->nv != 4 && sp->nv != 8)) return 1;
  const size_t nrec = CMPC_NREC(sp->N), nsol = CMPC_NSOL(sp->N, sp->nv);
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int b = 0; b < B; ++b) {
    stats_t st;
    solve_one(sp, recs + b * nrec, warm ? warm + b * nsol : NULL, out + b * nsol, &st, 0, NULL, NULL, NULL);
    status[b] = st.status; iters[b] = st.iters; kkt[b] = st.kkt; n_reg[b] = st.n_reg;
    if (waste) waste[b] = st.waste;
  }
  return 0;
}

/* Batch solve with solver states (closed-loop ticks): state_in / state_out [B][CMPC_NSTATE(N, nv)], either may be NULL. */
int cmpc_oracle_solve_batch_state(const cmpc_spec *sp, int32_t B, const double *recs, const double *warm,
                                  const double *state_in, double *out, double *state_out, int32_t *status,
                                  int32_t *iters, double *kkt, int nthreads, int verbose) {
  if (sp->N < 1 || sp->N > CMPC_MAX_N || (sp->nv != 4 && sp->nv != 8)) return 1;
  const size_t nrec = CMPC_NREC(sp->N), nsol = CMPC_NSOL(sp->N, sp->nv), nst = CMPC_NSTATE(sp->N, sp->nv);
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int b = 0; b < B; ++b) {
    stats_t st;
    solve_one(sp, recs + b * nrec, warm ? warm + b * nsol : NULL, out + b * nsol, &st, verbose, NULL,
              state_in ? state_in + b * nst : NULL, state_out ? state_out + b * nst : NULL);
    if (status) status[b] = st.status;
    if (iters) iters[b] = st.iters;
    if (kkt) kkt[b] = st.kkt;
  }
  return 0;
}

/* Diagnostic: solve and also return the full primal-dual iterate (x, lam, s, z). */
int cmpc_oracle_solve_full(const cmpc_spec *sp, const double *rec, const double *warm, double *out, double *full) {
  stats_t st;
  solve_one(sp, rec, warm, out, &st, 0, full, NULL, NULL);
  return st.status;
}

/* Function values on a full primal point w = [X (20 x (N+1)), U (nu x N)] in reference layout:
 * cost, dynamics defects x_{k+1} - F(x_k,u_k) (20 per stage) and all inequality rows
 * ((N+1) x ni, inactive rows = 0 with act = 0).  Used by the tests to compare against
 * oracle/nlp_reference.py. */
int cmpc_oracle_eval(const cmpc_spec *sp, const double *rec, const double *w, const double *uprox,
                     double *cost, double *defect, double *ineq, int32_t *act_out) {
  prob_t Pb; prob_init(&Pb, sp, rec);
  const prob_t *P = &Pb;
  const int N = P->N, nu = P->nu, ni = P->ni, nv = P->nv;
  double x[MAXNX], xp[MAXNX], xn[MAXNX], u[MAXNU], g[MAXNI];
  int act[MAXNI];
  double J = 0;
  const double x0n2 = w[6] * w[6] + w[7] * w[7] + w[8] * w[8];
  for (int k = 0; k <= N; ++k) {
    memset(x, 0, sizeof(x)); memset(u, 0, sizeof(u));
    memcpy(x, w + (size_t)k * CMPC_NX, sizeof(double) * CMPC_NX);
    if (k >= 1) for (int j = 0; j < 2 * nv; ++j) x[CMPC_NX + j] = w[(size_t)CMPC_NX * (N + 1) + (size_t)(k - 1) * nu + 3 * j + 2];
    if (k < N) memcpy(u, w + (size_t)CMPC_NX * (N + 1) + (size_t)k * nu, sizeof(double) * nu);
    J += stage_cost(P, k, x, u, uprox ? uprox + (size_t)k * nu : NULL, NULL, NULL);
    stage_ineq(P, k, x, u, x0n2, g, act, NULL, NULL, NULL);
    for (int i = 0; i < ni; ++i) { ineq[(size_t)k * ni + i] = g[i]; act_out[(size_t)k * ni + i] = act[i]; }
    if (k < N) {
      stage_dynamics(P, k, x, u, xn, NULL, NULL, NULL);
      for (int i = 0; i < CMPC_NX; ++i) defect[(size_t)k * CMPC_NX + i] = w[(size_t)(k + 1) * CMPC_NX + i] - xn[i];
    }
    (void)xp;
  }
  *cost = J;
  return 0;
}

/* Analytic stage derivatives for finite-difference tests: fills G (nx x nz), the cost gradient /
 * Hessian plus multiplier-weighted constraint curvature H (nz x nz), and Jg (ni x nz). */
int cmpc_oracle_stage(const cmpc_spec *sp, const double *rec, int k, const double *x, const double *u,
                      const double *lamn, const double *zmul, double x0n2, double *xn, double *G,
                      double *cost, double *hgrad, double *H, double *g, int32_t *act, double *Jg) {
  prob_t Pb; prob_init(&Pb, sp, rec);
  const prob_t *P = &Pb;
  const int nz = P->nz;
  memset(H, 0, sizeof(double) * nz * nz); memset(hgrad, 0, sizeof(double) * nz);
  *cost = stage_cost(P, k, x, u, NULL, hgrad, H);
  int a2[MAXNI];
  stage_ineq(P, k, x, u, x0n2, g, a2, Jg, zmul, H);
  for (int i = 0; i < P->ni; ++i) act[i] = a2[i];
  if (k < P->N) stage_dynamics(P, k, x, u, xn, G, lamn, H);
  return 0;
}
