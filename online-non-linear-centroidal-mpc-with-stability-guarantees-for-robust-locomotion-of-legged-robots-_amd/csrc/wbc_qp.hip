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
//         [ Hq                                 M_b'  ]      in LDS, lane i owns row i of the lower triangle (row stride
//         [      1e-6 I + A' diag(z/s) A      -Jc_b  ]      49: column reads across lanes are conflict free); column j
//         [ M_b        -Jc_b'                   0    ]      is read by every lane at wave-uniform addresses
//     followed by two triangular solves against the factor (v_readlane broadcasts of the running solution).
// Failure (status 1 / 2: iteration cap, wrong-inertia pivot, non-finite KKT error) returns zeros in tau, qdd and f_c, as
// the reference's QPSolver.solve does when OSQP fails (code/utils.py:85-92).
// The problem data of an instance (Hq, M_b, Jc_b: 8.4 KB) is staged in LDS once; per iteration nothing touches HBM.
// Instances are independent: workgroups stride over the batch, no inter-workgroup communication.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string>

#include "../../include/cmpc_wbc.h"

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
constexpr int LS = NK + 1;
constexpr int oL = oRHS + NK;                            // L (48 x 49), D on the diagonal
constexpr int oK0 = oL + NK * LS;                        // constant part of the KKT matrix, lower triangle (48 x 49)
constexpr int LDS_DOUBLES = oK0 + NK * LS;

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }   // one wave per workgroup
__device__ __forceinline__ double wave_max(double v) {
  for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
  for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
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

__global__ void __launch_bounds__(64) wbc_qp_kernel(int B, const double *__restrict__ Hq, const double *__restrict__ Fq,
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
    // constant part of this lane's row of the KKT matrix (lower triangle, columns 0 .. lane; zeros right of it)
    if (lane < NK) {
      for (int j = 0; j < NK; ++j) {
        double v = 0.0;
        if (j <= lane) {
          if (lane < ND) v = L[oH + lane * HS + j];
          else if (lane < NX) v = (j == lane) ? F_REG : 0.0;
          else {
            const int e = lane - NX;
            v = (j < ND) ? L[oMB + e * HS + j] : ((j < NX) ? -L[oJB + (j - ND) * 7 + e] : 0.0);
          }
        }
        L[oK0 + lane * LS + j] = v;
      }
    }
    lds_fence();
    int st = 1, it = 0;
    for (it = 0; it <= max_iter; ++it) {
      // ---- residuals.  lanes 0..41: dual residual of x_i; lanes 42..47: equality rows; lanes 0..15 also: A x + s
      double hx = 0.0, rd = 0.0, rp = 0.0, rg = 0.0, aiz = 0.0;
      if (lane < ND) {
#pragma unroll
        for (int j = 0; j < ND; ++j) hx += L[oH + lane * HS + j] * L[oX + j];
#pragma unroll
        for (int e = 0; e < NB; ++e) rd += L[oMB + e * HS + lane] * L[oNU + e];
      } else if (lane < NX) {
        const int c = lane - ND, foot = c / 6, comp = c % 6;
        hx = F_REG * L[oX + lane];
#pragma unroll
        for (int e = 0; e < NB; ++e) rd -= L[oJB + c * 7 + e] * L[oNU + e];
#pragma unroll
        for (int r = 0; r < 8; ++r) aiz += wrench_entry(r, comp, dfoot, muf) * L[oZ + 8 * foot + r];
      } else if (lane < NK) {
        const int e = lane - NX;
#pragma unroll
        for (int j = 0; j < ND; ++j) rp += L[oMB + e * HS + j] * L[oX + j];
#pragma unroll
        for (int c = 0; c < NC; ++c) rp -= L[oJB + c * 7 + e] * L[oX + ND + c];
        rp += L[oHB + e];                                  // A_e x - b_e with b_e = -h_b
      }
      double gx = 0.0;                                     // (A_i x)_r for lanes 0..15
      if (lane < NI) {
        const int foot = lane / 8, r = lane % 8;
#pragma unroll
        for (int c = 0; c < 6; ++c) gx += wrench_entry(r, c, dfoot, muf) * L[oX + ND + 6 * foot + c];
        rg = gx + L[oS + lane];
      }
      const double grad = (lane < NX) ? hx + L[oF + lane] : 0.0;      // H x + F
      rd = (lane < NX) ? grad + rd + aiz : 0.0;
      const double sl = (lane < NI) ? L[oS + lane] : 1.0, zl = (lane < NI) ? L[oZ + lane] : 0.0;
      const double sum_mult = wave_sum(((lane < NB) ? fabs(L[oNU + lane]) : 0.0) + ((lane < NI) ? fabs(zl) : 0.0));
      const double sd = fmax(100.0, sum_mult / (NB + NI)) / 100.0;
      const double e_d = wave_max(fabs(rd)) / sd, e_p = wave_max(fmax(fabs(rp), fabs(rg)));
      const double e_c = wave_max((lane < NI) ? fabs(sl * zl) : 0.0) / sd;
      const double e_cmu = wave_max((lane < NI) ? fabs(sl * zl - mu) : 0.0) / sd;
      const double kkt = fmax(fmax(e_d, e_p), e_c);
      // (fmax drops NaNs: a NaN residual must be looked for, or a NaN input "converges" with NaN outputs)
      const double nonfinite = wave_max((isfinite(rd) && isfinite(rp) && isfinite(rg) && isfinite(sl * zl)) ? 0.0 : 1.0);
      if (nonfinite != 0.0 || !(kkt < INFINITY)) { st = 2; break; }
      if (kkt <= tol) { st = 0; break; }
      if (it == max_iter) break;
      while (mu > tol / 10 && fmax(fmax(e_d, e_p), e_cmu) < 10 * mu) mu = fmax(tol / 10, fmin(0.1 * mu, mu * sqrt(mu)));
      // ---- barrier weights, right-hand side
      if (lane < NI) {
        const double sg = zl / sl;
        L[oSIG + lane] = sg;
        L[oW + lane] = mu / sl + sg * rg;
      }
      lds_fence();
      double rhs = 0.0;
      if (lane < ND) rhs = -grad;
      else if (lane < NX) {
        const int c = lane - ND, foot = c / 6, comp = c % 6;
        double aw = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) aw += wrench_entry(r, comp, dfoot, muf) * L[oW + 8 * foot + r];
        rhs = -grad - aw;
      } else if (lane < NK) rhs = -rp;
      // ---- the KKT matrix of this step (lower triangle, lane i owns row i): constant part + A' diag(sigma) A in the
      // wrench block of the row's foot
      double *A = &L[oL];
      if (lane < NK) {
        for (int j = 0; j <= lane; ++j) {
          double v = L[oK0 + lane * LS + j];
          if (lane >= ND && lane < NX && j >= ND) {
            const int c = lane - ND, foot = c / 6, comp = c % 6, cj = j - ND - 6 * foot;
            if (cj >= 0 && cj <= comp)
              for (int r = 0; r < 8; ++r)
                v += L[oSIG + 8 * foot + r] * wrench_entry(r, comp, dfoot, muf) * wrench_entry(r, cj, dfoot, muf);
          }
          A[lane * LS + j] = v;
        }
      }
      lds_fence();
      // ---- L D L' in place (no pivoting: quasi-definite).  Step j: l_ij = a_ij / d_j, a_ik -= l_ij a_kj (j < k <= i)
      bool ok = true;
#pragma unroll 1
      for (int j = 0; j < NK; ++j) {
        const double dj = A[j * LS + j];
        ok = ok && ((j < NX) ? (dj > 0.0) : (dj < 0.0));
        const bool mine = lane > j && lane < NK;
        const double lij = mine ? A[lane * LS + j] / dj : 0.0;
#pragma unroll 4
        for (int k = j + 1; k < NK; ++k) {
          const double akj = A[k * LS + j];                // column j, still unscaled
          if (mine && lane >= k) A[lane * LS + k] -= lij * akj;
        }
        lds_fence();
        if (mine) A[lane * LS + j] = lij;
        lds_fence();
      }
      if (!ok) { st = 2; break; }
      // ---- forward substitution L y = rhs, D, backward substitution L' x = y
      double y = rhs;
#pragma unroll 1
      for (int j = 0; j < NK; ++j) {
        const double yj = bcast(y, j);
        if (lane > j && lane < NK) y -= A[lane * LS + j] * yj;
      }
      if (lane < NK) y /= A[lane * LS + lane];
#pragma unroll 1
      for (int j = NK - 1; j >= 0; --j) {
        const double xj = bcast(y, j);
        if (lane < j) y -= A[j * LS + lane] * xj;
      }
      // y: lanes 0..41 dx, lanes 42..47 the new equality multipliers
      if (lane < NK) L[oRHS + lane] = y;
      lds_fence();
      // ---- slack / multiplier directions, fraction to the boundary
      double ds = 0.0, dz = 0.0, ap = 1.0, ad = 1.0;
      if (lane < NI) {
        const int foot = lane / 8, r = lane % 8;
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
      if (lane < NX) L[oX + lane] += ap * y;
      else if (lane < NK) L[oNU + lane - NX] += ap * (y - L[oNU + lane - NX]);
      if (lane < NI) { L[oS + lane] = sl + ap * ds; L[oZ + lane] = zl + ad * dz; }
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
