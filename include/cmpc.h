/*
 * cmpc.h -- C ABI of the MI355X batched centroidal-MPC solver (libcmpc_amd.so).
 *
 * Drop-in boundary for the hot path of the reference:
 *   code/centroidal_mpc_vertices.py:606   sol = self.opt.solve()          (CasADi Opti -> IPOPT)
 *   code/centroidal_mpc_vertices.py:126-353  NLP definition (constants -> cmpc_spec)
 *   code/centroidal_mpc_vertices.py:511-600  opt.set_value(...)           (-> parameter record)
 *   code/centroidal_mpc_vertices.py:614-617,630-631  sol.value / set_initial (-> out_XU / warm_XU)
 * and the same lines of code/centroidal_mpc_vertices_payload.py (gains k1,k2 = 7,1 at :27-31).
 *
 * The reference has no native interface for this path (it reaches IPOPT through
 * CasADi's SWIG layer); this header is what a ctypes stub binds instead
 * (INTEGRATION.md).  Plain pointers and sizes only; every pointer passed to
 * cmpc_solve_batch is a DEVICE pointer owned by the caller
 * (torch.Tensor.data_ptr()).  All functions return 0 on success, non-zero on
 * error (message via cmpc_last_error).  Kernels never throw; per-instance
 * outcome is reported in status[].
 */
#ifndef CMPC_H
#define CMPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMPC_NX 20            /* reference state size (:164-166)                      */
#define CMPC_MAX_N 64         /* horizon limit of this build                          */

/* Problem constants shared by a batch.  Defaults = the literals of the reference. */
typedef struct cmpc_spec {
  int32_t N;                  /* horizon, params['N'] (:10)                           */
  int32_t nv;                 /* contact vertices per foot: 4 (reference, :55-60) or 8 */
  int32_t max_iter;           /* interior-point iteration cap: iters[] <= max_iter + 1, also for a resumed solve that
                                 falls back to the plain one (the two attempts share the budget)          */
  int32_t struct_size;        /* sizeof(cmpc_spec): set by cmpc_default_spec, checked by cmpc_create (a caller built
                                 against an older, shorter layout is refused instead of read past its end)      */
  double delta;               /* world_time_step*mpc_rate (:11)                       */
  double g;                   /* params['g'] (:18)                                    */
  double k1, k2;              /* change-of-coordinates gains (:27-31)                 */
  double w_rate;              /* force-rate weight (:339-341)                         */
  double w_hw;                /* 1000 (:312)                                          */
  double w_cxy;               /* 1 (:313-314)                                         */
  double w_cz_const;          /* 2000 (:302-305): w_z[i] = 1000*exp(-i)+1000          */
  double w_foot;              /* 1000 (:316-319)                                      */
  double w_force;             /* 10 (:320-335)                                        */
  double cz_max;              /* 0.76 (:230)                                          */
  double box[3];              /* 0.01, 0.005, 0.00005 (:259-271)                      */
  double foot_length;         /* 0.25 (:51)                                           */
  double foot_width;          /* 0.13 (:52)                                           */
  double prox;                /* proximal weight on U (build-defined)      */
  double relax;               /* inequality relaxation, IPOPT bound_relax_factor 1e-8 */
  double tol;                 /* scaled KKT tolerance                                 */
  double acc_tol;             /* acceptable level (CMPC_ACCEPTABLE), > 0, default 1e-4: tighter than what the
                                 reference's IPOPT configuration (tol = 1e-3, :128; constr_viol_tol and
                                 compl_inf_tol 1e-4 by default) returns as full success              */
  int32_t kernel;             /* CMPC_KERNEL_*: which solver kernel cmpc_solve_batch launches (create-time choice;
                                 results are bit for bit the same either way)                         */
  int32_t reserved;           /* 0                                                                    */
} cmpc_spec;

/* Solver kernel of a handle (cmpc_spec.kernel).  AUTO: the pipelined pair (two wavefronts per instance) for batches that
 * do not fill the GPU for long, one wavefront per instance otherwise.  SINGLE / PAIR force one of them for every batch
 * size (nv = 4 only has both; nv = 8 always runs its two-wave kernel and accepts AUTO or SINGLE). */
enum { CMPC_KERNEL_AUTO = 0, CMPC_KERNEL_SINGLE = 1, CMPC_KERNEL_PAIR = 2 };

/* Doubles per instance in the parameter / solution records. */
#define CMPC_NREC(N) (24 + 19 * (N))
#define CMPC_NU(nv) (6 * (nv) + 8)
#define CMPC_NSOL(N, nv) (CMPC_NX * ((N) + 1) + CMPC_NU(nv) * (N))
/* Solver state carried from one closed-loop tick to the next (cmpc_solve_batch_state): the central-path point the
 * previous solve passed through at its last barrier value >= 1e-7 -- XU in the layout of out_XU, the dynamics
 * multipliers ((N+1) x (20 + 2 nv)), slacks and inequality multipliers ((N+1) x (15 + 10 nv) each), then the barrier
 * value (0 = no valid state), 7 further words (the first: the iterations the solve that wrote the state took, by which the next launch queues the
 * instance; 6 spare) and the contact flags of the N+1 nodes (left, right).  Opaque to callers:
 * pass last tick's state_out as state_in. */
#define CMPC_NSTATE(N, nv) (CMPC_NSOL(N, nv) + ((N) + 1) * ((CMPC_NX + 2 * (nv)) + 2 * (15 + 10 * (nv)) + 2) + 8)

/* Per-instance outcome. */
enum {
  CMPC_CONVERGED = 0,         /* scaled KKT error <= tol                               */
  CMPC_MAX_ITER = 1,          /* iteration cap reached, error above acc_tol            */
  CMPC_NUMERICAL = 2,         /* no usable point: the step length collapsed for several iterations (the usual
                                 case: a point of local infeasibility), regularisation exhausted, or a non-finite
                                 iterate (e.g. a NaN record from an out-of-range tick)                */
  CMPC_INFEASIBLE = 2,        /* alias kept for callers: same code, the usual meaning of 2            */
  CMPC_ACCEPTABLE = 3         /* stopped short of tol with error <= acc_tol (IPOPT's "Solved To Acceptable
                                 Level", which CasADi's Opti.solve() returns without raising): iteration cap,
                                 or no progress at the final barrier value                            */
};

typedef struct cmpc_handle cmpc_handle;

/* Fill *spec with the reference's constants for horizon N, nv vertices per foot. */
void cmpc_default_spec(cmpc_spec *spec, int32_t N, int32_t nv);

/* Create a solver bound to HIP device `device` (one handle per GPU / per stream). */
int cmpc_create(const cmpc_spec *spec, int device, cmpc_handle **out);
int cmpc_destroy(cmpc_handle *h);

/* Device scratch of a handle for batches of up to B instances: the slabs (bounded by the resident grid of the current
 * HIP device -- its CU count times the workgroups a CU holds; 256 CUs are assumed when no device can be queried --,
 * allocated by cmpc_create) plus the queue-order arrays (8 bytes per instance; (re)allocated by the first
 * cmpc_solve_batch call with a larger B than any before -- that call synchronises the device and must not be made
 * under stream capture; later calls with B up to that size allocate nothing). */
size_t cmpc_workspace_bytes(const cmpc_spec *spec, int32_t B);

/*
 * Solve B independent instances.
 *   params   [B][CMPC_NREC(N)]        parameter records (layout: DESIGN.md / problem.py)
 *   warm_XU  [B][CMPC_NSOL(N,nv)]     previous solution (initial guess and proximal centre); may be NULL
 *   out_XU   [B][CMPC_NSOL(N,nv)]     X (20 x (N+1), column-major) then U (nu x N, column-major)
 *   status   [B]  CMPC_* code;  iters [B] iterations used;  kkt_res [B] final scaled KKT error
 *   stream   hipStream_t (NULL = default stream).  Asynchronous: returns after enqueueing.
 */
int cmpc_solve_batch(cmpc_handle *h, int32_t B, const double *params, const double *warm_XU,
                     double *out_XU, int32_t *status, int32_t *iters, double *kkt_res,
                     void *stream);

/*
 * Closed-loop form of cmpc_solve_batch: the solver state of the previous tick comes in, this tick's goes out.
 *   state_in   [B][CMPC_NSTATE(N,nv)]  last tick's state_out; NULL (or a state whose barrier word is 0) = start as
 *                                      cmpc_solve_batch does
 *   state_out  [B][CMPC_NSTATE(N,nv)]  may be NULL; may NOT overlap state_in (checked: the call fails)
 * The state is the interior point method's own iterate at its last barrier value >= 1e-7 -- a point on the central
 * path of this tick's problem, one level short of the solution -- and the next solve resumes from it at that barrier
 * value instead of restarting at mu = 100 from the boundary solution (10.0 instead of 17.7 iterations per tick on
 * the flat-ground walk).  warm_XU keeps its
 * meaning (previous solution: proximal centre; initial guess only when there is no valid state).  Replaces the
 * reference's opt.set_initial(sol.value(...)) (code/centroidal_mpc_vertices.py:630-631), which IPOPT likewise uses
 * for the primal variables only.  A resumed solve whose state does not fit this tick's problem (still at the state's
 * barrier value after 20 iterations, or ending without a usable point) is followed by the plain solve inside the same
 * call with what is left of max_iter; iters[] reports both attempts.  A point within acc_tol that the resumed attempt
 * had in hand is not given up: the plain solve replaces it only by a better one, and it is returned (CMPC_ACCEPTABLE) when
 * the plain solve finds none or has no iterations left.
 */
int cmpc_solve_batch_state(cmpc_handle *h, int32_t B, const double *params, const double *warm_XU,
                           const double *state_in, double *out_XU, double *state_out, int32_t *status,
                           int32_t *iters, double *kkt_res, void *stream);

/* Average kernel time (ms) of the last cmpc_solve_batch on this handle, measured with
 * HIP events on the launch stream; synchronises that stream. */
int cmpc_last_kernel_ms(cmpc_handle *h, float *ms);
/* Name of the solver kernel the last cmpc_solve_batch on this handle launched (e.g. "cmpc_solve_kernel<4, 1>",
 * "cmpc_solve_pair_kernel<4, 2>"), as a profiler lists it; "" before the first launch.  Owned by the library. */
const char *cmpc_last_kernel_name(cmpc_handle *h);

/*
 * Batched parameter builder = front half of centroidal_mpc.solve (code/centroidal_mpc_vertices.py:482-600)
 * and the planner lookups behind it (code/footstep_planner_vertices.py:82-147) as one gather kernel.
 * The per-tick tables (host arrays, T rows each) are uploaded once:
 *   com_tab  T x 9   pos(3) vel(3) acc(3) of the CoM reference          (:64-74, :567-577)
 *   pose_l/r T x 6   nominal contact poses [ang(3), pos(3)]              (:77-84,  :581-584)
 *   gl/gr    T       contact flags at every tick                         (:515-534)
 *   cur_l/r  T x 3   foot positions written into x0 at tick t            (:493-509)
 */
typedef struct cmpc_tables cmpc_tables;
int cmpc_tables_create(int device, int32_t T, const double *com_tab, const double *pose_l, const double *pose_r,
                       const double *gl, const double *gr, const double *cur_l, const double *cur_r,
                       cmpc_tables **out);
int cmpc_tables_destroy(cmpc_tables *tb);
/*
 * records[b] for b < B from tick t[b] and the measured state
 *   state[b] = [com(3) dcom(3) hw(3) theta_hat(3) yaw_l yaw_r mass mu]   (16 doubles, device)
 * t (int32, device) must satisfy 0 <= t[b] and t[b] + (N+1)*rate < T (checked on the device: an
 * offending record is filled with NaN).  Asynchronous on `stream`.
 */
int cmpc_build_records(const cmpc_tables *tb, int32_t N, int32_t rate, int32_t B, const int32_t *t,
                       const double *state, double *records, void *stream);

/*
 * Per-instance contact plans (closed-loop rollouts in which every robot rewrites its own plan,
 * code/centroidal_mpc_vertices.py:656-675).  The foot positions of x0 come from the plan once t >= 200 (:493-509):
 * slot_l[t] / slot_r[t] (host arrays, T entries, < n_steps) name the plan entry holding the left / right foot at
 * tick t, -1 = take the nominal table (cur_l / cur_r).  cmpc_build_records_planned then reads
 *   plan_pos [B][n_steps][3]   device, the 'pos' of every plan entry of every instance
 * for those words; everything else is cmpc_build_records.  plan_pos == NULL is cmpc_build_records.
 */
int cmpc_tables_set_plan_slots(cmpc_tables *tb, int32_t n_steps, const int32_t *slot_l, const int32_t *slot_r);
int cmpc_build_records_planned(const cmpc_tables *tb, int32_t N, int32_t rate, int32_t B, const int32_t *t,
                               const double *state, const double *plan_pos, double *records, void *stream);

const char *cmpc_last_error(cmpc_handle *h);
const char *cmpc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CMPC_H */
