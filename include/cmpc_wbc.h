/*
 * cmpc_wbc.h -- C ABI of the batched whole-body inverse-dynamics QP (libcmpc_amd.so), SURVEY.md 8f row 4.
 *
 * Replaces, for B robots at once, the QP that the reference assembles and hands to CasADi's conic interface / OSQP
 * once per simulator tick:
 *   code/inverse_dynamics.py:92-105   cost          1/2 qdd' Hq qdd + Fq' qdd + 1/2 1e-6 |f_c|^2  (Hq, Fq: task sums)
 *   code/inverse_dynamics.py:107-111  dynamics      M qdd + h - Jc' f_c = S tau,  S = blockdiag(0_6, I)
 *   code/inverse_dynamics.py:113-129  inequalities  8 CoP / friction rows per foot wrench, d = foot_size / 2, mu
 *   code/inverse_dynamics.py:131-134  qp_solver.set_values(...); solve(); tau = solution[tau_indices]; return tau[6:]
 *   code/utils.py:40-92               QPSolver (Opti('conic') + OSQP)
 * The reference has no native interface here either (CasADi's SWIG layer); this header is what a ctypes stub binds
 * (INTEGRATION.md).  Sizes are the reference's for HRP-4: 30 dofs (6 floating-base), two 6-D contact wrenches.
 * Every pointer is a DEVICE pointer owned by the caller; matrices are row-major, instance-major:
 *   Hq [B][30][30]  Fq [B][30]   task Hessian / gradient in qdd (symmetric positive definite)
 *   M  [B][30][30]  h  [B][30]   mass matrix, Coriolis + gravity forces
 *   Jc [B][12][30]               contact Jacobian, rows already scaled by the contact flags (:109)
 * Outputs: tau [B][30] (tau[0:6] = 0: the statement leaves them free and the reference discards them), qdd [B][30],
 * f_c [B][12], status [B] (0 = KKT error <= tol, 1 = iteration cap, 2 = numerical failure; for 1 and 2 tau, qdd and f_c are
 * ZEROS, the reference's QPSolver.solve on failure, code/utils.py:85-92), iters [B].
 * Asynchronous on `stream` (hipStream_t; NULL = default stream).  Returns 0 on success; message via cmpc_wbc_last_error.
 */
#ifndef CMPC_WBC_H
#define CMPC_WBC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMPC_WBC_DOFS 30
#define CMPC_WBC_BASE 6
#define CMPC_WBC_CONTACT 12
#define CMPC_WBC_INEQ 16

int cmpc_wbc_qp_solve_batch(int device, int32_t B, const double *Hq, const double *Fq, const double *M, const double *h,
                            const double *Jc, double half_foot_size, double mu, double tol, int32_t max_iter,
                            double *tau, double *qdd, double *f_c, int32_t *status, int32_t *iters, void *stream);
const char *cmpc_wbc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* CMPC_WBC_H */
