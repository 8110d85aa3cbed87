// wbc_qp.hip -- batched whole-body inverse-dynamics QP for gfx950 (C ABI: include/cmpc_wbc.h), SURVEY.md 8f row 4.
//
// The QP of code/inverse_dynamics.py:92-134 (solved there by CasADi's conic interface + OSQP, code/utils.py:40-92) for B
// robots at once.  One wavefront (one 64-thread workgroup) owns one instance:
//   * the statement in x = [qdd (30), tau (30), f_c (12)] is reduced to (qdd, f_c) and the six floating-base rows of
//     the dynamics (tau[6:] is defined by the actuated rows, tau[0:6] appears nowhere: oracle/wbc_qp_oracle.py shows
//     the equivalence and tests/test_wbc_qp.py checks the result against the 72-variable KKT conditions);
//   * primal-dual interior point on the 42-variable problem, the same monotone barrier schedule and
//     fraction-to-the-boundary rule as the centroidal MPC solver;
//   * every Newton step is one L D L' factorisation of the 48 x 48 quasi-definite KKT matrix
//         [ Hq                                 M_b'  ]      in REGISTERS: lane i holds row i of the lower triangle (48
//         [      1e-6 I + A' diag(z/s) A      -Jc_b  ]      doubles, statically indexed: the pivot loops are fully
//         [ M_b        -Jc_b'                   0    ]      unrolled); element (k, j) of the pivot column reaches the
//     other lanes by v_readlane.  Round 3 kept the matrix in LDS (a read-modify-write per element and update, 48.9 KB per
//     instance -> three instances per CU); now LDS holds the problem data and the packed factor only (20.7 KB -> seven).
//     Forward substitution from the registers; the backward one needs the transposed factor and reads it from LDS.
// Failure (status 1 / 2: iteration cap, wrong-inertia pivot, non-finite KKT error) returns zeros in tau, qdd and f_c, as
// the reference's QPSolver.solve does when OSQP fails (code/utils.py:85-92).
// The problem data of an instance (Hq, M_b, Jc_b: 8.4 KB) is staged in LDS once; per iteration nothing touches HBM.
// Instances are independent: workgroups stride over the batch, no inter-workgroup communication.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string>

#include "../../include/cmpc_wbc.h"
#include "cmpc_wave.hpp"

namespace {

constexpr int ND = CMPC_WBC_DOFS, NB = CMPC_WBC_BASE, NC = CMPC_WBC_CONTACT, NI = CMPC_WBC_INEQ;
constexpr int NX = ND + NC;           // 42 primal variables (qdd, f_c)
constexpr int NK = NX + NB;           // 48 rows of the KKT matrix
constexpr double F_REG = 1e-6;        // code/inverse_dynamics.py:105
constexpr double MU0 = 10.0;

// LDS map (doubles); odd row strides: column reads across lanes are conflict free
constexpr int HS = ND + 1;                               // Hq rows
constexpr int oH = 0;                                    // Hq   30 x 31
constexpr int oMB = oH + ND * HS;                        // M_b   6 x 31   (floating-base rows of the mass matrix)
constexpr int oJB = oMB + NB * HS;                       // Jc_b 12 x 7    (Jc[:, 0:6])
constexpr int oF = oJB + NC * 7;                         // Fq (30) then zeros (12)
constexpr int oHB = oF + NX;                             // h_b (6)
constexpr int oX = oHB + NB;                             // x (42)
constexpr int oNU = oX + NX;                             // nu (6)
constexpr int oS = oNU + NB;                             // s (16)
constexpr int oZ = oS + NI;                              // z (16)
constexpr int oSIG = oZ + NI;                            // z / s (16)
constexpr int oW = oSIG + NI;                            // mu / s + sigma * (A x + s) (16)
constexpr int oRHS = oW + NI;                            // right-hand side / solution (48)
constexpr int oL = oRHS + NK;                            // strictly lower part of L, packed: row i at tri(i)
constexpr int oC = oL + NK * (NK + 1) / 2;                // constant part of the trailing 18 x 18 block, lower triangle packed by rows
constexpr int LDS_DOUBLES = oC + (NK - ND) * (NK - ND + 1) / 2 + 1;
__device__ __forceinline__ constexpr int tri(int i) { return i * (i + 1) / 2; }

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }   // one wave per workgroup
// wave-wide reductions: six butterfly steps through v_permlane*_swap / DPP (cmpc_wave.hpp), no LDS round trip
#define WBC_BFLY(op)                                                                                             \
  { double a, b;                                                                                                 \
    cmpc_pair_of<32>(v, a, b); v = op; cmpc_pair_of<16>(v, a, b); v = op; cmpc_pair_of<8>(v, a, b); v = op;      \
    cmpc_pair_of<4>(v, a, b); v = op; cmpc_pair_of<2>(v, a, b); v = op; cmpc_pair_of<1>(v, a, b); v = op; }
__device__ __forceinline__ double wave_max(double v) { WBC_BFLY(fmax(a, b)) return v; }
__device__ __forceinline__ double wave_min(double v) { WBC_BFLY(fmin(a, b)) return v; }
__device__ __forceinline__ double wave_sum(double v) { WBC_BFLY(a + b) return v; }
#undef WBC_BFLY
// 1 / d by two Newton steps on the hardware estimate (full double accuracy up to the last bit or two; the IEEE division is
// a ~35-instruction sequence and there is one per pivot on the critical path).  d is a pivot that passed the sign test.
__device__ __forceinline__ double pivot_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double bcast(double v, int src) {
  union { double d; int i[2]; } u; u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], src);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
  return u.d;
}

// entry (r, c) of the 8 x 6 wrench rows of code/inverse_dynamics.py:116-123; a wrench is [moment(3), force(3)]
__device__ __forceinline__ double wrench_entry(int r, int c, double d, double mu) {
  const int pair = r >> 1;                         // 0: mx, 1: my, 2: fx, 3: fy
  const int lead = (pair < 2) ? pair : pair + 1;   // column of the +-1
  const double sgn = (r & 1) ? -1.0 : 1.0;
  if (c == lead) return sgn;
  if (c == 5) return (pair < 2) ? -d : -mu;
  return 0.0;
}

__global__ void __launch_bounds__(64, 2) wbc_qp_kernel(int B, const double *__restrict__ Hq, const double *__restrict__ Fq,
                                                    const double *__restrict__ M, const double *__restrict__ hvec,
                                                    const double *__restrict__ Jc, double dfoot, double muf, double tol,
                                                    int max_iter, double *__restrict__ tau, double *__restrict__ qdd,
                                                    double *__restrict__ fc, int32_t *__restrict__ status,
                                                    int32_t *__restrict__ iters) {
  __shared__ double L[LDS_DOUBLES];
  const int lane = threadIdx.x;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const double *Hb = Hq + (size_t)b * ND * ND, *Mb = M + (size_t)b * ND * ND, *Jb = Jc + (size_t)b * NC * ND;
    // ---- stage the instance: Hq, the base rows of M, the base columns of Jc, Fq, h_b
    for (int e = lane; e < ND * ND; e += 64) L[oH + (e / ND) * HS + e % ND] = Hb[e];
    for (int e = lane; e < NB * ND; e += 64) L[oMB + (e / ND) * HS + e % ND] = Mb[e];
    for (int e = lane; e < NC * NB; e += 64) L[oJB + (e / NB) * 7 + e % NB] = Jb[(e / NB) * ND + e % NB];
    if (lane < NX) L[oF + lane] = (lane < ND) ? Fq[(size_t)b * ND + lane] : 0.0;
    if (lane < NB) L[oHB + lane] = hvec[(size_t)b * ND + lane];
    // ---- initial point: x = 0, nu = 0, s = 1, z = mu / s
    double mu = MU0;
    if (lane < NX) L[oX + lane] = 0.0;
    if (lane < NB) L[oNU + lane] = 0.0;
    if (lane < NI) { L[oS + lane] = 1.0; L[oZ + lane] = mu; }
    lds_fence();
    // ---- the part of the factorisation that does not change from step to step.  Only the wrench block of the KKT
    // matrix carries barrier terms (columns ND .. NX-1 of rows ND .. NX-1); the first ND pivots run over Hq alone, so the
    // rows' first ND words of L and the Schur complement they leave in columns ND .. NK-1 are computed ONCE per instance:
    // ac[0 .. ND) = row `lane` of L in the qdd columns, ac[ND .. NK) = constant part of the trailing 18 x 18 block.
    // Per Newton step: 153 multiply-adds of the factorisation instead of 1128.
    double ac[NK], dlc = 1.0;
    bool ok0 = true;
    {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const bool isq = ln < ND, isf = ln >= ND && ln < NX, ise = ln >= NX && ln < NK;
      const int rowq = isq ? ln : ND - 1, rowe = ise ? ln - NX : 0;
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        double v = 0.0;
        if (j < ND) {
          const double hv = L[oH + rowq * HS + j], mv = L[oMB + rowe * HS + j];
          v = isq ? hv : (ise ? mv : 0.0);
        } else if (j < NX) {
          const double jv = -L[oJB + (j - ND) * 7 + rowe];
          v = isf ? ((ln == j) ? F_REG : 0.0) : (ise ? jv : 0.0);
        }
        ac[j] = v;
        if (j % 8 == 7) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < ND; ++j) {
        if (j % 6 == 0) asm volatile("" : "+v"(ln));       // (predicates of later pivots are not formed -- and held -- ahead of time)
        const double dj = bcast(ac[j], j);
        ok0 = ok0 && (dj > 0.0);
        dlc = (ln == j) ? dj : dlc;
        const double lij = (ln > j) ? ac[j] * pivot_rcp(dj) : 0.0;
#pragma unroll
        for (int k = j + 1; k < NK; ++k) {
          ac[k] -= lij * bcast(ac[j], k);                    // column j, still unscaled
          // (opaque: the trailing words k >= ND are not read again before the loop is over, and hipcc sinks their thirty
          // updates down to that point -- with the thirty broadcast multipliers of each parked in lanes of a vector
          // register by v_writelane and fetched back: three instructions per word instead of one, 1100 of them per
          // instance, and the vector registers holding them spilled to scratch)
          if (k >= ND) asm volatile("" : "+v"(ac[k]));
          if ((k - j) % 8 == 0) __builtin_amdgcn_sched_barrier(0);
        }
        ac[j] = (ln > j) ? lij : ac[j];
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ln < NK) {                                         // the constant columns of the packed factor
        const int tl = tri(ln);
#pragma unroll
        for (int j = 0; j < ND; ++j)
          if (j < ln) L[oL + tl + j] = ac[j];
      }
      // ... and the constant Schur complement of the trailing block (rows ND .. NK-1, lower triangle): every Newton step
      // starts its 18 x 18 factorisation from it.  Kept in LDS (171 words), not in 36 registers across the Newton loop --
      // with the constant columns above that is the whole `ac` array out of the loop, and the kernel's spills with it.
      if (ln >= ND && ln < NK) {
        const int tc = tri(ln - ND);
#pragma unroll
        for (int t = 0; t < NK - ND; ++t)
          if (t <= ln - ND) L[oC + tc + t] = ac[ND + t];
      }
      lds_fence();
    }
    int st = 1, it = 0;
    for (it = 0; it <= max_iter; ++it) {
      // (the ln id is made opaque in every iteration, and again between its phases: otherwise the ~300 ln-against-
      // constant predicates and clamped addresses of the unrolled loops below are hoisted out of the iteration loop and
      // held -- spilled -- across it)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      // ---- residuals.  lanes 0..41: dual residual of x_i; lanes 42..47: equality rows; lanes 0..15 also: A x + s
      double hx = 0.0, rd = 0.0, rp = 0.0, rg = 0.0, aiz = 0.0;
      if (ln < ND) {
#pragma unroll
        for (int j = 0; j < ND; ++j) hx += L[oH + ln * HS + j] * L[oX + j];
#pragma unroll
        for (int e = 0; e < NB; ++e) rd += L[oMB + e * HS + ln] * L[oNU + e];
      } else if (ln < NX) {
        const int c = ln - ND, foot = c / 6, comp = c % 6;
        hx = F_REG * L[oX + ln];
#pragma unroll
        for (int e = 0; e < NB; ++e) rd -= L[oJB + c * 7 + e] * L[oNU + e];
#pragma unroll
        for (int r = 0; r < 8; ++r) aiz += wrench_entry(r, comp, dfoot, muf) * L[oZ + 8 * foot + r];
      } else if (ln < NK) {
        const int e = ln - NX;
#pragma unroll
        for (int j = 0; j < ND; ++j) rp += L[oMB + e * HS + j] * L[oX + j];
#pragma unroll
        for (int c = 0; c < NC; ++c) rp -= L[oJB + c * 7 + e] * L[oX + ND + c];
        rp += L[oHB + e];                                  // A_e x - b_e with b_e = -h_b
      }
      double gx = 0.0;                                     // (A_i x)_r for lanes 0..15
      if (ln < NI) {
        const int foot = ln / 8, r = ln % 8;
#pragma unroll
        for (int c = 0; c < 6; ++c) gx += wrench_entry(r, c, dfoot, muf) * L[oX + ND + 6 * foot + c];
        rg = gx + L[oS + ln];
      }
      const double grad = (ln < NX) ? hx + L[oF + ln] : 0.0;      // H x + F
      rd = (ln < NX) ? grad + rd + aiz : 0.0;
      const double sl = (ln < NI) ? L[oS + ln] : 1.0, zl = (ln < NI) ? L[oZ + ln] : 0.0;
      const double sum_mult = wave_sum(((ln < NB) ? fabs(L[oNU + ln]) : 0.0) + ((ln < NI) ? fabs(zl) : 0.0));
      const double sd = fmax(100.0, sum_mult / (NB + NI)) / 100.0;
      const double e_d = wave_max(fabs(rd)) / sd, e_p = wave_max(fmax(fabs(rp), fabs(rg)));
      const double e_c = wave_max((ln < NI) ? fabs(sl * zl) : 0.0) / sd;
      const double e_cmu = wave_max((ln < NI) ? fabs(sl * zl - mu) : 0.0) / sd;
      const double kkt = fmax(fmax(e_d, e_p), e_c);
      // (fmax drops NaNs: a NaN residual must be looked for, or a NaN input "converges" with NaN outputs)
      const double nonfinite = wave_max((isfinite(rd) && isfinite(rp) && isfinite(rg) && isfinite(sl * zl)) ? 0.0 : 1.0);
      if (nonfinite != 0.0 || !(kkt < INFINITY)) { st = 2; break; }
      if (kkt <= tol) { st = 0; break; }
      if (it == max_iter) break;
      while (mu > tol / 10 && fmax(fmax(e_d, e_p), e_cmu) < 10 * mu) mu = fmax(tol / 10, fmin(0.1 * mu, mu * sqrt(mu)));
      // ---- barrier weights, right-hand side
      if (ln < NI) {
        const double sg = zl / sl;
        L[oSIG + ln] = sg;
        L[oW + ln] = mu / sl + sg * rg;
      }
      lds_fence();
      double rhs = 0.0;
      if (ln < ND) rhs = -grad;
      else if (ln < NX) {
        const int c = ln - ND, foot = c / 6, comp = c % 6;
        double aw = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) aw += wrench_entry(r, comp, dfoot, muf) * L[oW + 8 * foot + r];
        rhs = -grad - aw;
      } else if (ln < NK) rhs = -rp;
      // ---- the trailing block of this step, row `ln`, columns ND .. NK-1 in registers (words right of the diagonal are
      // never used): the constant Schur complement + A' diag(sigma) A in the wrench block of the row's foot
      asm volatile("" : "+v"(ln));
      if (!ok0) { st = 2; break; }
      constexpr int NT = NK - ND;                          // 18
      const bool isf = ln >= ND && ln < NX;
      const int cf = isf ? ln - ND : 0, foot_l = cf / 6, comp_l = cf % 6;
      double wl[8];                                        // this row's column of the wrench rows
#pragma unroll
      for (int r = 0; r < 8; ++r) wl[r] = wrench_entry(r, comp_l, dfoot, muf);
      double w[NT];
      const double *crow = &L[oC + tri((ln >= ND && ln < NK) ? ln - ND : 0)];   // (other lanes' words are never used)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int j = ND + t;
        double v = crow[t];
        if (j < NX) {
          const int cj = (j - ND) % 6, fj = (j - ND) / 6;
          double acc = 0.0;
#pragma unroll
          for (int r = 0; r < 8; ++r) acc += L[oSIG + 8 * fj + r] * wl[r] * wrench_entry(r, cj, dfoot, muf);
          v += (isf && foot_l == fj && cj <= comp_l) ? acc : 0.0;
        }
        w[t] = v;
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- L D L' of the trailing block in registers (no pivoting: quasi-definite).  Step j: l_ij = a_ij / d_j,
      // a_ik -= l_ij a_kj (j < k <= i); a_kj is lane k's word j.  Lanes <= j run along with l_ij = 0, words right of a
      // lane's diagonal take garbage.
      bool ok = true;
      double dl = dlc;                                     // this lane's pivot
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int j = ND + t;
        const double dj = bcast(w[t], j);
        ok = ok && ((j < NX) ? (dj > 0.0) : (dj < 0.0));
        dl = (ln == j) ? dj : dl;
        const double lij = (ln > j) ? w[t] * pivot_rcp(dj) : 0.0;
#pragma unroll
        for (int u = t + 1; u < NT; ++u) w[u] -= lij * bcast(w[t], ND + u);   // column j, still unscaled
        w[t] = (ln > j) ? lij : w[t];
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!ok) { st = 2; break; }
      // ---- forward substitution L y = rhs out of the registers, D, then the step's columns of the factor to LDS (packed
      // rows; the qdd columns stand there since before the loop) for the backward substitution L' x = y, which reads
      // the factor by columns
      asm volatile("" : "+v"(ln));
      double y = rhs;
      {
        // the constant columns come back from the packed factor in LDS (row `ln`, contiguous: written once per instance),
        // fifteen words at a time -- held in registers across the Newton loop they were 60 of the kernel's 256 and the
        // reason for its spills (round 4: 116 bytes of scratch per lane)
        const double *lrow = &L[oL + tri((ln < NK) ? ln : 0)];
#pragma unroll
        for (int j0 = 0; j0 < ND; j0 += 15) {
          double lc[15];
#pragma unroll
          for (int q = 0; q < 15; ++q) lc[q] = lrow[j0 + q];         // (words at or right of the diagonal: never used)
#pragma unroll
          for (int q = 0; q < 15; ++q) {
            const int j = j0 + q;
            const double yj = bcast(y, j);
            if (ln > j) y -= lc[q] * yj;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int j = ND; j < NK; ++j) {
        const double yj = bcast(y, j);
        if (ln > j) y -= w[j - ND] * yj;
      }
      y *= pivot_rcp(dl);
      if (ln < NK) {
        const int tl = tri(ln);
#pragma unroll
        for (int j = ND; j < NK - 1; ++j)
          if (j < ln) L[oL + tl + j] = w[j - ND];
      }
      lds_fence();
#pragma unroll
      for (int j0 = NK - 12; j0 >= 0; j0 -= 12) {            // column `ln` of L below the diagonal, twelve reads at a time
        double col[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) { const int j = j0 + q; col[q] = (j >= 1) ? L[oL + tri(j) + ((ln < j) ? ln : 0)] : 0.0; }
#pragma unroll
        for (int q = 11; q >= 0; --q) {
          const int j = j0 + q;
          if (j >= 1) {
            const double xj = bcast(y, j);
            if (ln < j) y -= col[q] * xj;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // y: lanes 0..41 dx, lanes 42..47 the new equality multipliers
      if (ln < NK) L[oRHS + ln] = y;
      lds_fence();
      // ---- slack / multiplier directions, fraction to the boundary
      double ds = 0.0, dz = 0.0, ap = 1.0, ad = 1.0;
      if (ln < NI) {
        const int foot = ln / 8, r = ln % 8;
        double adx = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) adx += wrench_entry(r, c, dfoot, muf) * L[oRHS + ND + 6 * foot + c];
        ds = -rg - adx;
        dz = (mu - sl * zl - zl * ds) / sl;
        const double tf = fmax(0.99, 1.0 - mu);
        if (ds < 0.0) ap = fmin(ap, -tf * sl / ds);
        if (dz < 0.0) ad = fmin(ad, -tf * zl / dz);
      }
      ap = wave_min(ap); ad = wave_min(ad);
      if (ln < NX) L[oX + ln] += ap * y;
      else if (ln < NK) L[oNU + ln - NX] += ap * (y - L[oNU + ln - NX]);
      if (ln < NI) { L[oS + ln] = sl + ap * ds; L[oZ + ln] = zl + ad * dz; }
      lds_fence();
    }
    // ---- outputs: qdd, f_c, tau[6:] = M_a qdd + h_a - Jc_a' f_c  (tau[0:6] = 0)
    // (failure: zeros, the reference's behaviour -- never the last iterate, which may be NaN / Inf after a bad pivot)
    const bool good = st == 0;
    if (lane < ND) qdd[(size_t)b * ND + lane] = good ? L[oX + lane] : 0.0;
    if (lane < NC) fc[(size_t)b * NC + lane] = good ? L[oX + ND + lane] : 0.0;
    if (lane < ND) {
      double t = 0.0;
      if (lane >= NB && good) {
        t = hvec[(size_t)b * ND + lane];
        for (int j = 0; j < ND; ++j) t += Mb[lane * ND + j] * L[oX + j];
        for (int c = 0; c < NC; ++c) t -= Jb[c * ND + lane] * L[oX + ND + c];
      }
      tau[(size_t)b * ND + lane] = t;
    }
    if (lane == 0) { status[b] = st; iters[b] = it; }
    lds_fence();
  }
}

thread_local std::string g_wbc_err;

}  // namespace

extern "C" {

int cmpc_wbc_qp_solve_batch(int device, int32_t B, const double *Hq, const double *Fq, const double *M, const double *h,
                            const double *Jc, double half_foot_size, double mu, double tol, int32_t max_iter,
                            double *tau, double *qdd, double *f_c, int32_t *status, int32_t *iters, void *stream) {
  if (B < 0) { g_wbc_err = "cmpc_wbc_qp_solve_batch: negative batch"; return 1; }
  if (B == 0) return 0;
  if (!Hq || !Fq || !M || !h || !Jc || !tau || !qdd || !f_c || !status || !iters) { g_wbc_err = "cmpc_wbc_qp_solve_batch: null buffer"; return 1; }
  if (!(tol > 0) || max_iter < 1 || !(half_foot_size > 0) || !(mu > 0)) { g_wbc_err = "cmpc_wbc_qp_solve_batch: bad argument"; return 1; }
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess) prev = -1;
  if (prev != device && hipSetDevice(device) != hipSuccess) { g_wbc_err = "cmpc_wbc_qp_solve_batch: bad device"; return 1; }
  // CU count per device, queried once (hipGetDeviceProperties is a slow host call; this entry point runs every tick)
  static int cu_cache[64] = {0};
  int cus = (device >= 0 && device < 64) ? cu_cache[device] : 0;
  if (cus <= 0) {
    hipDeviceProp_t prop;
    cus = (hipGetDeviceProperties(&prop, device) == hipSuccess) ? prop.multiProcessorCount : 256;
    if (device >= 0 && device < 64) cu_cache[device] = cus;
  }
  const int per_cu = (int)((160 * 1024) / (sizeof(double) * LDS_DOUBLES + 64));       // LDS-limited residency
  int grid = cus * (per_cu < 1 ? 1 : per_cu);
  if (B < grid) grid = B;
  hipLaunchKernelGGL(wbc_qp_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, B, Hq, Fq, M, h, Jc, half_foot_size, mu,
                     tol, max_iter, tau, qdd, f_c, status, iters);
  const hipError_t e = hipGetLastError();
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  if (e != hipSuccess) { g_wbc_err = std::string("cmpc_wbc_qp_solve_batch: ") + hipGetErrorString(e); return 1; }
  return 0;
}

const char *cmpc_wbc_last_error(void) { return g_wbc_err.c_str(); }

}  // extern "C"
