// cmpc_hip.hip -- gfx950 kernels + the C ABI of include/cmpc.h (libcmpc_amd.so).
//
// Launch shape: one 64-thread workgroup (= one wavefront) per resident problem slot; the grid is
// sized to the LDS-limited residency (CUs x workgroups per CU) and every workgroup pulls instance
// indices from a global ticket counter until the batch is drained, so instances with very
// different iteration counts do not serialise behind a static assignment.  Instances are
// independent: no inter-workgroup communication besides the ticket.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>

#include "cmpc_kernel.hpp"
#include "cmpc_order_fit.h"

// Register budget: 2 waves per SIMD (<= 256 VGPR+AGPR) lets a CU hold 5 single-wave workgroups
// (LDS-limited) instead of 4 (one per SIMD).
#ifndef CMPC_WAVES_PER_SIMD
#define CMPC_WAVES_PER_SIMD 2
#endif

namespace {

// Queue order.  A launch ends when its longest instance does, and the instances drawn last decide how long the drain
// is: with six workgroups per CU a quarter of a B = 8192 launch is drain (tools/tail_study.py), which longest-first with
// perfect knowledge would cut to 7 %.  The iteration count is not known, but part of it is visible in the record: how
// far the capture point of the initial state, c + v / omega, lies from the support (the stance foot, or the middle of
// the feet in double support), whether and how late in the horizon a contact switches, the support phase at node 0,
// the measured angular momentum.  A least-squares fit of the iteration count on these six numbers (2048 instances of
// the domain-randomised workload, cold start; R^2 0.4) is all that is used: the instances are queued by decreasing
// predicted count in ORDER_BUCKETS buckets of half an iteration.  Replayed on measured iteration counts the makespan
// falls from 1.32 to 1.22-1.25 times the balanced bound on that workload and -- without refitting -- from 1.27 to 1.14
// (payload gains), 1.48 to 1.36 (perturbed walk states) and 1.64 to 1.36 (N = 40, eight vertices) on the others.  The
// order never changes a result (instances are independent: the bitwise batch-composition tests); the order inside a
// bucket is whatever the atomics make it.  counters = {ticket, -, -, histogram[ORDER_BUCKETS], cursor[ORDER_BUCKETS]}.
constexpr int ORDER_BUCKETS = 64;
constexpr int ORDER_COUNTERS = 3 + 2 * ORDER_BUCKETS;

// The features of a record that say how long its solve will take (mirrored by cmpc_amd/queue_order.py::features; the
// coefficients: csrc/cmpc_order_fit.h).  Round 4 added the interactions and the velocity / friction terms: out of sample
// over five seeds the replayed makespan falls from 1.35 to 1.31 x the balanced bound (tools/fit_queue_order.py).
__device__ __forceinline__ int cmpc_order_bucket(const double *__restrict__ r, int N, double omega, double cz_max) {
  double gl = r[24 + 17], gr = r[24 + 18];
  const double gl0 = gl, gr0 = gr;
  int first = N, nsw = 0;
  double gain = 0.0;                                                        // 1: the first switch is a touch-down
  for (int k = 1; k <= N; ++k) {
    const double l = (k < N) ? r[24 + 19 * k + 17] : r[22], q = (k < N) ? r[24 + 19 * k + 18] : r[23];
    const bool ch = (l != gl) || (q != gr);
    if (ch && first == N) { first = k - 1; gain = (l + q > gl + gr) ? 1.0 : 0.0; }
    nsw += ch ? 1 : 0;
    gl = l; gr = q;
  }
  const double dx = r[0] + r[3] / omega, dy = r[1] + r[4] / omega;          // capture point of x_0
  const bool both = (gl0 != 0.0) == (gr0 != 0.0);                           // (no foot down: treated like both)
  const double tx = both ? 0.5 * (r[13] + r[17]) : (gl0 != 0.0) ? r[13] : r[17];
  const double ty = both ? 0.5 * (r[14] + r[18]) : (gl0 != 0.0) ? r[14] : r[18];
  const double d2 = (dx - tx) * (dx - tx) + (dy - ty) * (dy - ty), d = sqrt(d2);
  const double hw = sqrt(r[6] * r[6] + r[7] * r[7] + r[8] * r[8]);
  const double evx = r[3] - r[24 + 3], evy = r[4] - r[24 + 4], ev2 = evx * evx + evy * evy, ev = sqrt(ev2);
  const double sw = (first < N) ? 1.0 : 0.0, feet = gl0 + gr0, mu = r[21];
  const double f[CMPC_ORDER_NFEAT] = {1.0, sw, (double)first, feet, d, d2, hw, d * sw, d * feet, gain, ev, (double)nsw,
                                      cz_max - r[2], mu, r[20] / 40.0, ev2, ev * feet, 1.0 / mu};
  const double c[CMPC_ORDER_NFEAT] = CMPC_ORDER_COEF;                        // csrc/cmpc_order_fit.h
  double its = 0.0;
#pragma unroll
  for (int i = 0; i < CMPC_ORDER_NFEAT; ++i) its += c[i] * f[i];
  const double b = 2.0 * (its - CMPC_ORDER_BUCKET_ORIGIN);                   // buckets of half an iteration
  return (b > 0.0) ? ((b < ORDER_BUCKETS - 1) ? (int)b : ORDER_BUCKETS - 1) : 0;   // (a NaN record lands in bucket 0)
}

// A resumed instance (valid solver state) is queued by what its previous solve took, which that solve left in the state's
// first spare word: consecutive ticks of a closed loop take similar numbers of iterations.  One bucket per iteration
// there; the formula's buckets (10 + b / 2 iterations) and these meet around 20 iterations, which is where a mixed
// batch needs them comparable (an instance without a state is a cold solve).
__global__ void __launch_bounds__(256) cmpc_order_score_kernel(int B, int N, double omega, double cz_max, const double *__restrict__ recs,
                                                               const double *__restrict__ state_in, size_t nstate, size_t mu_word,
                                                               int *__restrict__ key, int *__restrict__ counters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  int b = -1;
  if (state_in) {
    const double ms = state_in[(size_t)i * nstate + mu_word], cnt = state_in[(size_t)i * nstate + mu_word + 1];
    if (ms > 0.0 && ms < INFINITY && cnt >= 1.0 && cnt < 1e6) b = (cnt < ORDER_BUCKETS - 1) ? (int)cnt : ORDER_BUCKETS - 1;
  }
  if (b < 0) b = cmpc_order_bucket(recs + (size_t)i * CMPC_NREC(N), N, omega, cz_max);
  key[i] = b;
  atomicAdd(counters + 3 + b, 1);
}

__global__ void __launch_bounds__(256) cmpc_order_scatter_kernel(int B, const int *__restrict__ key, int *__restrict__ order,
                                                                 int *__restrict__ counters) {
  __shared__ int base[ORDER_BUCKETS];                // queue position of a bucket's first instance: the buckets above it
  if (threadIdx.x < ORDER_BUCKETS) {
    int s = 0;
    for (int b = ORDER_BUCKETS - 1; b > (int)threadIdx.x; --b) s += counters[3 + b];
    base[threadIdx.x] = s;
  }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const int b = key[i];
  order[base[b] + atomicAdd(counters + 3 + ORDER_BUCKETS + b, 1)] = i;
}

// One workgroup per instance in flight: a single wave for the 4-vertex solver, NW = WAVES_NV8 waves for the 8-vertex
// one (its stage block has 92 rows: with 128 lanes every row / column has its own lane, cmpc_kernel.hpp).
template <int NV, int NW>
__global__ void __launch_bounds__(64 * NW, (NV == 4 ? CMPC_WAVES_PER_SIMD : 1)) cmpc_solve_kernel(cmpc::KArgs ka, int *ticket,
                                                                                                   const int *__restrict__ order) {
  using D = cmpc::Dims<NV, NW>;
  __shared__ __attribute__((aligned(16))) double lds[D::LDS_DOUBLES];
  __shared__ int next;
  double *slab = ka.scratch + (size_t)blockIdx.x * ka.scratch_stride;
  const size_t nrec = CMPC_NREC(ka.sp.N), nsol = CMPC_NSOL(ka.sp.N, NV), nstate = CMPC_NSTATE(ka.sp.N, NV);
  for (;;) {
    // (single-wave workgroups: the lane id from the execution mask, re-derived per instance, instead of threadIdx.x kept
    // live -- and spilled -- across the whole solve)
    const int tid = (NW == 1) ? (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : (int)threadIdx.x;
    if (tid == 0) next = atomicAdd(ticket, 1);
    __syncthreads();
    const int tk = next;
    __syncthreads();
    if (tk >= ka.B) break;                      // every wave reaches this exit
    // the queue position comes out of LDS in a vector register; the instance index is wave-uniform, and saying so
    // keeps the record / output base addresses in scalar registers
    const int p = __builtin_amdgcn_readfirstlane(order[__builtin_amdgcn_readfirstlane(tk)]);
    cmpc::Solver<NV, NW> s(ka, lds, slab, ka.recs + (size_t)p * nrec);
    s.solve(ka.warm ? ka.warm + (size_t)p * nsol : nullptr, ka.state_in ? ka.state_in + (size_t)p * nstate : nullptr,
            ka.state_out ? ka.state_out + (size_t)p * nstate : nullptr, ka.out + (size_t)p * nsol, ka.status + p,
            ka.iters + p, ka.kkt + p);
  }
}

// The pipelined pair (cmpc::Solver<4, 1, true>): two waves per instance, two LDS images.  For batches that do not keep
// the GPU full for long (cmpc_solve_batch picks it up to 14 instances per CU): same results bit for bit, an instance
// finishes ~1.55x sooner.  WPS = waves per SIMD the build is for: 2 (256 registers, three pairs per CU) or 1.
template <int NV, int WPS>
__global__ void __launch_bounds__(128, WPS) cmpc_solve_pair_kernel(cmpc::KArgs ka, int *ticket,
                                                                                  const int *__restrict__ order) {
  using D = cmpc::Dims<NV, 1, true>;
  __shared__ __attribute__((aligned(16))) double lds[2 * D::LDS_DOUBLES];
  __shared__ int next;
  double *slab = ka.scratch + (size_t)blockIdx.x * ka.scratch_stride;
  const size_t nrec = CMPC_NREC(ka.sp.N), nsol = CMPC_NSOL(ka.sp.N, NV), nstate = CMPC_NSTATE(ka.sp.N, NV);
  for (;;) {
    if (threadIdx.x == 0) next = atomicAdd(ticket, 1);
    __syncthreads();
    const int tk = next;
    __syncthreads();
    if (tk >= ka.B) break;                      // both waves reach this exit
    const int p = __builtin_amdgcn_readfirstlane(order[__builtin_amdgcn_readfirstlane(tk)]);
    cmpc::Solver<NV, 1, true> s(ka, lds, slab, ka.recs + (size_t)p * nrec);
    s.solve(ka.warm ? ka.warm + (size_t)p * nsol : nullptr, ka.state_in ? ka.state_in + (size_t)p * nstate : nullptr,
            ka.state_out ? ka.state_out + (size_t)p * nstate : nullptr, ka.out + (size_t)p * nsol, ka.status + p,
            ka.iters + p, ka.kkt + p);
  }
}

// One workgroup per record (grid-stride over the batch), one lane per pair of output words: each
// lane assembles two consecutive doubles and issues one 16-byte store, so a wavefront writes 1 KiB of
// contiguous record per instruction; reads are gathers from tables that stay L2-resident
// (T x 30 doubles ~ 0.5 MB).  No integer division by the record length.  HBM-bound on the write side.
__device__ __forceinline__ double cmpc_record_word(int e, int t, int N, int rate, const double *__restrict__ st16,
                                                   const double *__restrict__ com_tab, const double *__restrict__ pose_l,
                                                   const double *__restrict__ pose_r, const double *__restrict__ gl,
                                                   const double *__restrict__ gr, const double *__restrict__ cur_l,
                                                   const double *__restrict__ cur_r, const double *__restrict__ plan_b,
                                                   int slot_l, int slot_r) {
  if (e < 12) return st16[e];
  if (e == 12) return st16[12];
  // foot positions of x0 (:493-509): the instance's own plan entry once the plan is consulted (t >= 200),
  // the nominal table before that (slot < 0) or when the batch shares the nominal plan (plan_b == null)
  if (e < 16) return (plan_b && slot_l >= 0) ? plan_b[3 * slot_l + e - 13] : cur_l[(size_t)t * 3 + e - 13];
  if (e == 16) return st16[13];
  if (e < 20) return (plan_b && slot_r >= 0) ? plan_b[3 * slot_r + e - 17] : cur_r[(size_t)t * 3 + e - 17];
  if (e < 22) return st16[14 + e - 20];
  if (e == 22) return gl[t + N * rate];
  if (e == 23) return gr[t + N * rate];
  const int i = (e - 24) / 19, c = (e - 24) - 19 * i;
  const int tt = t + (1 + i) * rate;
  if (c < 9) return com_tab[(size_t)tt * 9 + c];
  if (c < 12) return pose_l[(size_t)tt * 6 + 3 + c - 9];
  if (c < 15) return pose_r[(size_t)tt * 6 + 3 + c - 12];
  if (c == 15) return pose_l[(size_t)tt * 6 + 2];
  if (c == 16) return pose_r[(size_t)tt * 6 + 2];
  if (c == 17) return gl[t + i * rate];
  return gr[t + i * rate];
}

__global__ void __launch_bounds__(256) cmpc_build_records_kernel(
    int T, int N, int rate, int B, int nrec, const int *__restrict__ tick, const double *__restrict__ state,
    const double *__restrict__ com_tab, const double *__restrict__ pose_l, const double *__restrict__ pose_r,
    const double *__restrict__ gl, const double *__restrict__ gr, const double *__restrict__ cur_l,
    const double *__restrict__ cur_r, const double *__restrict__ plan_pos, int n_steps,
    const int *__restrict__ slot_l_tab, const int *__restrict__ slot_r_tab, double *__restrict__ rec) {
  const double nanv = __builtin_nan("");
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const int t = tick[b];
    const bool bad = t < 0 || t + (N + 1) * rate >= T;
    const double *st16 = state + (size_t)b * 16;
    const double *plan_b = plan_pos ? plan_pos + (size_t)b * n_steps * 3 : nullptr;
    const int slot_l = (plan_pos && !bad) ? slot_l_tab[t] : -1, slot_r = (plan_pos && !bad) ? slot_r_tab[t] : -1;
    double *out = rec + (size_t)b * nrec;
    const bool aligned = ((nrec & 1) == 0);             // even record length: every pair is 16-byte aligned
    for (int e = 2 * threadIdx.x; e < nrec; e += 2 * blockDim.x) {
      const double v0 = bad ? nanv : cmpc_record_word(e, t, N, rate, st16, com_tab, pose_l, pose_r, gl, gr, cur_l, cur_r, plan_b, slot_l, slot_r);
      if (e + 1 < nrec) {
        const double v1 = bad ? nanv : cmpc_record_word(e + 1, t, N, rate, st16, com_tab, pose_l, pose_r, gl, gr, cur_l, cur_r, plan_b, slot_l, slot_r);
        if (aligned) *reinterpret_cast<double2 *>(out + e) = make_double2(v0, v1);
        else { out[e] = v0; out[e + 1] = v1; }
      } else out[e] = v0;
    }
  }
}

}  // namespace

struct cmpc_tables {
  int device = 0;
  int T = 0;
  double *com_tab = nullptr, *pose_l = nullptr, *pose_r = nullptr, *gl = nullptr, *gr = nullptr,
         *cur_l = nullptr, *cur_r = nullptr;
  int n_steps = 0;                                  // per-instance plans: entries per plan
  int *slot_l = nullptr, *slot_r = nullptr;         // [T] plan entry holding the left / right foot of x0, -1 = nominal table
};

struct cmpc_handle {
  cmpc_spec spec;
  int device = 0;
  int grid = 0;
  int pair_grid = 0;                                // resident grid of the pipelined pair kernel (nv = 4)
  int pair_max_batch = 0;                           // largest batch that goes to the pair kernel, 0 = never used
  int pair_per_cu = 2;
  int num_cu = 0;
  int slabs = 0;                                    // slabs allocated: every launch grid stays within it
  const char *last_kernel = "";                     // name of the solver kernel the last launch used
  size_t slab_doubles = 0;
  double *scratch = nullptr;
  int *ticket = nullptr;                            // ORDER_COUNTERS words: the ticket and the counters of the queue order
  int *order = nullptr;                             // queue order of the last launch, order_cap entries
  int order_cap = 0;
  long long *prof = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  std::string err;
};

static thread_local std::string g_err;

// Every entry point runs on the handle's device and leaves the caller's current device as it found it.
struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess);
    if (prev == dev) prev = -1;                 // nothing to restore
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

static int fail(cmpc_handle *h, const std::string &msg) {
  if (h) h->err = msg;
  g_err = msg;
  return 1;
}
#define HIP_TRY(h, call)                                                                         \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) return fail(h, std::string(#call) + ": " + hipGetErrorString(e_));     \
  } while (0)

static bool spec_ok(const cmpc_spec *s) {
  return s && s->struct_size == (int32_t)sizeof(cmpc_spec) && s->N >= 1 && s->N <= CMPC_MAX_N &&
         (s->nv == 4 || s->nv == 8) && s->max_iter >= 1 && s->delta > 0 && s->tol > 0 && s->tol < INFINITY &&
         s->acc_tol > 0 && s->acc_tol < INFINITY &&
         (s->kernel == CMPC_KERNEL_AUTO || s->kernel == CMPC_KERNEL_SINGLE || (s->kernel == CMPC_KERNEL_PAIR && s->nv == 4)) &&
         s->reserved == 0;
}
static size_t lds_bytes(int nv) {
  return sizeof(double) * (nv == 4 ? cmpc::Dims<4>::LDS_DOUBLES : cmpc::Dims<8, cmpc::WAVES_NV8>::LDS_DOUBLES) + 16;
}
static size_t slab_doubles(const cmpc_spec *s) {
  return s->nv == 4 ? cmpc::Dims<4>::scratch_doubles(s->N) : cmpc::Dims<8, cmpc::WAVES_NV8>::scratch_doubles(s->N);
}
// Workgroups a CU holds at once: 160 KB of LDS, allocated in 1280-byte granules on this part (measured:
// tools/ubench/lds_residency.hip), and at most two waves per SIMD with the kernel's 256 registers.
static int resident_per_cu(int nv) {
  const size_t granule = 1280, alloc = (lds_bytes(nv) + granule - 1) / granule * granule;
  int n = (int)((160 * 1024) / alloc);
  const int waves = (nv == 8) ? cmpc::WAVES_NV8 : 1;
  if (n * waves > 8) n = 8 / waves;
  return n < 1 ? 1 : n;
}
// Pairs of the pipelined kernel a CU holds (nv = 4): two LDS images per pair (the exchange words stand in the second
// image's unused P region), in 1280-byte granules -> three, six waves per CU; the pair kernel is built for two waves per
// SIMD (256 registers, no scratch).
static int pairs_per_cu() {
  const size_t granule = 1280, alloc = (sizeof(double) * (2 * cmpc::Dims<4, 1, true>::LDS_DOUBLES) + 16 + granule - 1) / granule * granule;
  int n = (int)((160 * 1024) / alloc);
  if (n > 3) n = 3;
  return n < 1 ? 1 : n;
}
// Slabs a handle allocates: one per workgroup of the largest grid any of its kernels is launched with.
static int slab_count(const cmpc_spec *s, int num_cu) {
  const int g = num_cu * resident_per_cu(s->nv), pg = (s->nv == 4) ? num_cu * pairs_per_cu() : 0;
  return g > pg ? g : pg;
}
static int current_device_cus() {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1)
    return 256;                                      // no device to ask (host-only callers): an MI355X
  return prop.multiProcessorCount;
}

extern "C" {

void cmpc_default_spec(cmpc_spec *s, int32_t N, int32_t nv) {
  memset(s, 0, sizeof(*s));
  s->struct_size = (int32_t)sizeof(*s);
  s->N = N; s->nv = nv; s->max_iter = 100;
  s->delta = 0.01; s->g = 9.81; s->k1 = 4.0; s->k2 = 0.1; s->w_rate = 1.0;
  s->w_hw = 1000.0; s->w_cxy = 1.0; s->w_cz_const = 2000.0; s->w_foot = 1000.0; s->w_force = 10.0;
  s->cz_max = 0.76; s->box[0] = 0.01; s->box[1] = 0.005; s->box[2] = 0.00005;
  s->foot_length = 0.25; s->foot_width = 0.13; s->prox = 1e-4; s->relax = 1e-8; s->tol = 1e-8;
  s->acc_tol = 1e-4;
  s->kernel = CMPC_KERNEL_AUTO;
}

size_t cmpc_workspace_bytes(const cmpc_spec *spec, int32_t B) {
  if (!spec_ok(spec)) return 0;
  // the slab count is bounded by the resident grid of the device, not by B (cmpc_create allocates all of them)
  const int grid = slab_count(spec, current_device_cus());
  return (size_t)grid * slab_doubles(spec) * sizeof(double) + ORDER_COUNTERS * sizeof(int) + 2 * (size_t)(B > 0 ? B : 0) * sizeof(int);
}

int cmpc_create(const cmpc_spec *spec, int device, cmpc_handle **out) {
  if (!out) return fail(nullptr, "cmpc_create: null out pointer");
  *out = nullptr;
  if (!spec_ok(spec)) return fail(nullptr, "cmpc_create: invalid spec (struct_size = sizeof(cmpc_spec), N in [1,64], nv in {4,8}, tol > 0, acc_tol > 0, kernel a CMPC_KERNEL_* value)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, "cmpc_create: no HIP device");
  if (device < 0 || device >= ndev) return fail(nullptr, "cmpc_create: bad device index");
  cmpc_handle *h = new cmpc_handle();
  h->spec = *spec; h->device = device;
  hipDeviceProp_t prop;
  DeviceGuard guard(device);
  if (!guard.ok || hipGetDeviceProperties(&prop, device) != hipSuccess) {
    delete h;
    return fail(nullptr, "cmpc_create: cannot query device");
  }
  h->num_cu = prop.multiProcessorCount;
  h->grid = h->num_cu * resident_per_cu(spec->nv);
  if (spec->nv == 4) {
    h->pair_per_cu = pairs_per_cu();
#ifdef CMPC_DEV_KNOBS
    // (round 4's first form of the pair kernel -- two pairs per CU, the build for one wave per SIMD)
    if (const char *e = getenv("CMPC_PAIR_PER_CU")) { if (atoi(e) == 2 && h->pair_per_cu > 2) h->pair_per_cu = 2; }
#endif
    h->pair_grid = h->num_cu * h->pair_per_cu;
    // The pair kernel is the faster one while the queue is short (an instance-iteration takes 0.4 - 0.5 ms in a pair
    // against 0.65 - 1.2 ms in one of two to six single waves of a CU, and a short queue is mostly its longest instance).
    // Measured crossover on 256 CUs with seven one-wave workgroups against three pairs per CU (round 5): between 3072 and
    // 4096 instances on configs 3 and 4 (profiles/r05_kernel_crossover.txt; round 4, six against three: 6144 ... 7168).  The
    // caller can fix the choice when the handle is created (cmpc_spec.kernel): the results are the same bit for bit.
    h->pair_max_batch = (h->pair_per_cu >= 3 ? 14 : 8) * h->num_cu;
    if (spec->kernel == CMPC_KERNEL_SINGLE) h->pair_max_batch = 0;
    if (spec->kernel == CMPC_KERNEL_PAIR) h->pair_max_batch = 1 << 30;
#ifdef CMPC_DEV_KNOBS
    if (const char *e = getenv("CMPC_PAIR")) h->pair_max_batch = (atoi(e) == 0) ? 0 : 1 << 30;
#endif
  }
  h->slabs = slab_count(spec, h->num_cu);              // (both kernels' grids: either may be launched on this handle)
#ifdef CMPC_DEV_KNOBS
  // Developer build only (tools/, never the shipped library): fewer resident workgroups per CU for occupancy studies
  if (const char *e = getenv("CMPC_WG_PER_CU")) {
    const int n = atoi(e);
    if (n >= 1 && n < resident_per_cu(spec->nv)) h->grid = h->num_cu * n;
  }
#endif
  h->slab_doubles = slab_doubles(spec);
#ifdef CMPC_PROFILE
  if (hipMalloc(&h->prof, 28 * sizeof(long long)) == hipSuccess) (void)hipMemset(h->prof, 0, 28 * sizeof(long long));
#endif
  if (hipMalloc(&h->scratch, (size_t)h->slabs * h->slab_doubles * sizeof(double)) != hipSuccess ||
      hipMalloc(&h->ticket, ORDER_COUNTERS * sizeof(int)) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess ||
      hipEventCreate(&h->ev1) != hipSuccess) {
    cmpc_destroy(h);
    return fail(nullptr, "cmpc_create: device allocation failed");
  }
  *out = h;
  return 0;
}

int cmpc_destroy(cmpc_handle *h) {
  if (!h) return 0;
  DeviceGuard guard(h->device);
  if (h->scratch) (void)hipFree(h->scratch);
  if (h->ticket) (void)hipFree(h->ticket);
  if (h->order) (void)hipFree(h->order);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  delete h;
  return 0;
}

int cmpc_solve_batch(cmpc_handle *h, int32_t B, const double *params, const double *warm_XU, double *out_XU,
                     int32_t *status, int32_t *iters, double *kkt_res, void *stream) {
  return cmpc_solve_batch_state(h, B, params, warm_XU, nullptr, out_XU, nullptr, status, iters, kkt_res, stream);
}

int cmpc_solve_batch_state(cmpc_handle *h, int32_t B, const double *params, const double *warm_XU, const double *state_in,
                           double *out_XU, double *state_out, int32_t *status, int32_t *iters, double *kkt_res,
                           void *stream) {
  if (!h) return fail(nullptr, "cmpc_solve_batch: null handle");
  if (B < 0) return fail(h, "cmpc_solve_batch: negative batch");
  if (B == 0) return 0;
  if (!params || !out_XU || !status || !iters || !kkt_res) return fail(h, "cmpc_solve_batch: null buffer");
  if (state_in && state_out) {
    // the kernel invalidates state_out's barrier word before it reads state_in's, and the order kernel reads state_in
    // in the same launch: overlapping ranges would resume from a half-overwritten state without any error
    const size_t bytes = (size_t)B * CMPC_NSTATE(h->spec.N, h->spec.nv) * sizeof(double);
    const uintptr_t a = (uintptr_t)state_in, b = (uintptr_t)state_out;
    if (a < b + bytes && b < a + bytes) return fail(h, "cmpc_solve_batch_state: state_in and state_out overlap");
  }
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard guard(h->device);
  if (!guard.ok) return fail(h, "cmpc_solve_batch: cannot select the handle's device");
  cmpc::KArgs ka;
  ka.sp = h->spec; ka.B = B; ka.recs = params; ka.warm = warm_XU; ka.out = out_XU;
  ka.state_in = state_in; ka.state_out = state_out;
  ka.status = status; ka.iters = iters; ka.kkt = kkt_res;
  ka.scratch = h->scratch; ka.scratch_stride = h->slab_doubles;
  ka.prof = h->prof;
  cmpc::fill_levels(ka);
  const int grid = B < h->grid ? B : h->grid;
  if (B > h->order_cap) {                       // grows rarely; hipFree / hipMalloc synchronise the device
    if (h->order) (void)hipFree(h->order);
    h->order = nullptr; h->order_cap = 0;
    HIP_TRY(h, hipMalloc(&h->order, 2 * (size_t)B * sizeof(int)));          // queue order, then the bucket keys
    h->order_cap = B;
  }
  HIP_TRY(h, hipMemsetAsync(h->ticket, 0, ORDER_COUNTERS * sizeof(int), st));
  HIP_TRY(h, hipEventRecord(h->ev0, st));
  const double omega = sqrt(h->spec.g / h->spec.cz_max);                   // natural frequency of the pendulum at the height limit
  const size_t nstate = CMPC_NSTATE(h->spec.N, h->spec.nv), mu_word = nstate - 8 - 2 * (size_t)(h->spec.N + 1);
  hipLaunchKernelGGL(cmpc_order_score_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, h->spec.N, omega, h->spec.cz_max, params, state_in, nstate,
                     mu_word, h->order + B, h->ticket);
  hipLaunchKernelGGL(cmpc_order_scatter_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, h->order + B, h->order, h->ticket);
  const bool pair = h->spec.nv == 4 && B <= h->pair_max_batch;   // the batch does not fill the GPU for long: two waves per instance
  const int launch_grid = pair ? (B < h->pair_grid ? B : h->pair_grid) : grid;
  if (launch_grid > h->slabs) return fail(h, "cmpc_solve_batch: launch grid exceeds the slabs of the handle");
  if (pair) {
    const dim3 pg(launch_grid);
#ifdef CMPC_DEV_KNOBS
    if (h->pair_per_cu < 3) { hipLaunchKernelGGL((cmpc_solve_pair_kernel<4, 1>), pg, dim3(128), 0, st, ka, h->ticket, h->order); h->last_kernel = "cmpc_solve_pair_kernel<4, 1>"; }
    else
#endif
    { hipLaunchKernelGGL((cmpc_solve_pair_kernel<4, 2>), pg, dim3(128), 0, st, ka, h->ticket, h->order); h->last_kernel = "cmpc_solve_pair_kernel<4, 2>"; }
  } else if (h->spec.nv == 4) {
    hipLaunchKernelGGL((cmpc_solve_kernel<4, 1>), dim3(grid), dim3(64), 0, st, ka, h->ticket, h->order);
    h->last_kernel = "cmpc_solve_kernel<4, 1>";
  } else {
    hipLaunchKernelGGL((cmpc_solve_kernel<8, cmpc::WAVES_NV8>), dim3(grid), dim3(64 * cmpc::WAVES_NV8), 0, st, ka, h->ticket, h->order);
    h->last_kernel = "cmpc_solve_kernel<8, 2>";
  }
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipEventRecord(h->ev1, st));
  h->timed = true;
  return 0;
}

int cmpc_last_kernel_ms(cmpc_handle *h, float *ms) {
  if (!h || !ms) return fail(h, "cmpc_last_kernel_ms: null argument");
  if (!h->timed) return fail(h, "cmpc_last_kernel_ms: no launch recorded");
  HIP_TRY(h, hipEventSynchronize(h->ev1));
  HIP_TRY(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
  return 0;
}

int cmpc_tables_create(int device, int32_t T, const double *com_tab, const double *pose_l, const double *pose_r,
                       const double *gl, const double *gr, const double *cur_l, const double *cur_r,
                       cmpc_tables **out) {
  if (!out) return fail(nullptr, "cmpc_tables_create: null out pointer");
  *out = nullptr;
  if (T <= 0 || !com_tab || !pose_l || !pose_r || !gl || !gr || !cur_l || !cur_r)
    return fail(nullptr, "cmpc_tables_create: bad argument");
  DeviceGuard guard(device);
  if (!guard.ok) return fail(nullptr, "cmpc_tables_create: bad device");
  cmpc_tables *tb = new cmpc_tables();
  tb->device = device; tb->T = T;
  struct { double **dst; const double *src; size_t n; } items[] = {
      {&tb->com_tab, com_tab, (size_t)T * 9}, {&tb->pose_l, pose_l, (size_t)T * 6}, {&tb->pose_r, pose_r, (size_t)T * 6},
      {&tb->gl, gl, (size_t)T}, {&tb->gr, gr, (size_t)T}, {&tb->cur_l, cur_l, (size_t)T * 3}, {&tb->cur_r, cur_r, (size_t)T * 3}};
  for (auto &it : items) {
    if (hipMalloc(it.dst, it.n * sizeof(double)) != hipSuccess ||
        hipMemcpy(*it.dst, it.src, it.n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
      cmpc_tables_destroy(tb);
      return fail(nullptr, "cmpc_tables_create: device allocation / upload failed");
    }
  }
  *out = tb;
  return 0;
}

int cmpc_tables_destroy(cmpc_tables *tb) {
  if (!tb) return 0;
  DeviceGuard guard(tb->device);
  double *ptrs[] = {tb->com_tab, tb->pose_l, tb->pose_r, tb->gl, tb->gr, tb->cur_l, tb->cur_r};
  for (double *p : ptrs) if (p) (void)hipFree(p);
  if (tb->slot_l) (void)hipFree(tb->slot_l);
  if (tb->slot_r) (void)hipFree(tb->slot_r);
  delete tb;
  return 0;
}

int cmpc_tables_set_plan_slots(cmpc_tables *tb, int32_t n_steps, const int32_t *slot_l, const int32_t *slot_r) {
  if (!tb || n_steps < 1 || !slot_l || !slot_r) return fail(nullptr, "cmpc_tables_set_plan_slots: bad argument");
  for (int t = 0; t < tb->T; ++t)
    if (slot_l[t] >= n_steps || slot_r[t] >= n_steps) return fail(nullptr, "cmpc_tables_set_plan_slots: slot out of range");
  DeviceGuard guard(tb->device);
  if (!guard.ok) return fail(nullptr, "cmpc_tables_set_plan_slots: bad device");
  const size_t bytes = (size_t)tb->T * sizeof(int);
  if ((!tb->slot_l && hipMalloc(&tb->slot_l, bytes) != hipSuccess) || (!tb->slot_r && hipMalloc(&tb->slot_r, bytes) != hipSuccess) ||
      hipMemcpy(tb->slot_l, slot_l, bytes, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(tb->slot_r, slot_r, bytes, hipMemcpyHostToDevice) != hipSuccess)
    return fail(nullptr, "cmpc_tables_set_plan_slots: device allocation / upload failed");
  tb->n_steps = n_steps;
  return 0;
}

int cmpc_build_records(const cmpc_tables *tb, int32_t N, int32_t rate, int32_t B, const int32_t *t,
                       const double *state, double *records, void *stream) {
  return cmpc_build_records_planned(tb, N, rate, B, t, state, nullptr, records, stream);
}

int cmpc_build_records_planned(const cmpc_tables *tb, int32_t N, int32_t rate, int32_t B, const int32_t *t,
                               const double *state, const double *plan_pos, double *records, void *stream) {
  if (!tb) return fail(nullptr, "cmpc_build_records: null tables");
  if (plan_pos && (!tb->slot_l || tb->n_steps < 1)) return fail(nullptr, "cmpc_build_records_planned: call cmpc_tables_set_plan_slots first");
  if (N < 1 || N > CMPC_MAX_N || rate < 1 || B < 0) return fail(nullptr, "cmpc_build_records: bad argument");
  if (B == 0) return 0;
  if (!t || !state || !records) return fail(nullptr, "cmpc_build_records: null buffer");
  DeviceGuard guard(tb->device);
  if (!guard.ok) return fail(nullptr, "cmpc_build_records: bad device");
  const int nrec = CMPC_NREC(N);
  const int blocks = B < 256 * 32 ? B : 256 * 32;      // grid-stride beyond 32 workgroups per CU
  hipLaunchKernelGGL(cmpc_build_records_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tb->T, N, rate,
                     B, nrec, t, state, tb->com_tab, tb->pose_l, tb->pose_r, tb->gl, tb->gr, tb->cur_l, tb->cur_r,
                     plan_pos, tb->n_steps, tb->slot_l, tb->slot_r, records);
  if (hipGetLastError() != hipSuccess) return fail(nullptr, "cmpc_build_records: launch failed");
  return 0;
}

const char *cmpc_last_kernel_name(cmpc_handle *h) { return h ? h->last_kernel : ""; }
const char *cmpc_last_error(cmpc_handle *h) { return h ? h->err.c_str() : g_err.c_str(); }
const char *cmpc_version(void) { return "cmpc_amd 0.5 (gfx950)"; }

#ifdef CMPC_DEV_KNOBS
/* developer build only: the first slab of the handle (the slab of workgroup 0), for dumps of a B = 1 launch */
int cmpc_debug_slab(cmpc_handle *h, double **ptr, size_t *doubles) {
  if (!h || !ptr || !doubles) return 1;
  *ptr = h->scratch; *doubles = h->slab_doubles;
  return 0;
}
#endif

#ifdef CMPC_PROFILE
/* diagnostic build only: read and reset the phase cycle sums */
int cmpc_profile_read(cmpc_handle *h, long long *out8) {
  if (!h || !h->prof) return 1;
  if (hipMemcpy(out8, h->prof, 28 * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  (void)hipMemset(h->prof, 0, 28 * sizeof(long long));
  return 0;
}
#endif

}  // extern "C"
