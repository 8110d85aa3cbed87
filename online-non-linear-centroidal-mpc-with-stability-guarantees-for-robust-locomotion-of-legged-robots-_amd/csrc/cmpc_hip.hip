// cmpc_hip.hip -- gfx950 kernels + the C ABI of include/cmpc.h (libcmpc_amd.so).
//
// Launch shape: one 64-thread workgroup (= one wavefront) per resident problem slot; the grid is
// sized to the LDS-limited residency (CUs x workgroups per CU) and every workgroup pulls instance
// indices from a global ticket counter until the batch is drained, so instances with very
// different iteration counts do not serialise behind a static assignment.  Instances are
// independent: no inter-workgroup communication besides the ticket.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <string>

#include "cmpc_kernel.hpp"

namespace {

template <int NV>
__global__ void __launch_bounds__(64) cmpc_solve_kernel(cmpc::KArgs ka, int *ticket) {
  using D = cmpc::Dims<NV>;
  __shared__ double lds[D::LDS_DOUBLES];
  __shared__ int next;
  double *slab = ka.scratch + (size_t)blockIdx.x * ka.scratch_stride;
  const size_t nrec = CMPC_NREC(ka.sp.N), nsol = CMPC_NSOL(ka.sp.N, NV);
  for (;;) {
    if (threadIdx.x == 0) next = atomicAdd(ticket, 1);
    __syncthreads();
    const int p = next;
    __syncthreads();
    if (p >= ka.B) break;                       // every wave reaches this exit
    cmpc::Solver<NV> s(ka, lds, slab, ka.recs + (size_t)p * nrec);
    s.solve(ka.warm ? ka.warm + (size_t)p * nsol : nullptr, ka.out + (size_t)p * nsol, ka.status + p,
            ka.iters + p, ka.kkt + p, p == 0);
  }
}

}  // namespace

struct cmpc_handle {
  cmpc_spec spec;
  int device = 0;
  int grid = 0;
  int num_cu = 0;
  size_t slab_doubles = 0;
  double *scratch = nullptr;
  int *ticket = nullptr;
  long long *prof = nullptr;
  double *dbg = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  std::string err;
};

static thread_local std::string g_err;

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
  return s && s->N >= 1 && s->N <= CMPC_MAX_N && (s->nv == 4 || s->nv == 8) && s->max_iter >= 1 &&
         s->delta > 0 && s->tol > 0;
}
static size_t lds_bytes(int nv) {
  return sizeof(double) * (nv == 4 ? cmpc::Dims<4>::LDS_DOUBLES : cmpc::Dims<8>::LDS_DOUBLES) + 16;
}
static size_t slab_doubles(const cmpc_spec *s) {
  return s->nv == 4 ? cmpc::Dims<4>::scratch_doubles(s->N) : cmpc::Dims<8>::scratch_doubles(s->N);
}
static int resident_per_cu(int nv) {
  int n = (int)((160 * 1024) / lds_bytes(nv));
  return n < 1 ? 1 : (n > 8 ? 8 : n);
}

extern "C" {

void cmpc_default_spec(cmpc_spec *s, int32_t N, int32_t nv) {
  memset(s, 0, sizeof(*s));
  s->N = N; s->nv = nv; s->max_iter = 100;
  s->delta = 0.01; s->g = 9.81; s->k1 = 4.0; s->k2 = 0.1; s->w_rate = 1.0;
  s->w_hw = 1000.0; s->w_cxy = 1.0; s->w_cz_const = 2000.0; s->w_foot = 1000.0; s->w_force = 10.0;
  s->cz_max = 0.76; s->box[0] = 0.01; s->box[1] = 0.005; s->box[2] = 0.00005;
  s->foot_length = 0.25; s->foot_width = 0.13; s->prox = 1e-4; s->relax = 1e-8; s->tol = 1e-8;
}

size_t cmpc_workspace_bytes(const cmpc_spec *spec, int32_t B) {
  if (!spec_ok(spec)) return 0;
  // the slab count is bounded by the resident grid, not by B
  int grid = 256 * resident_per_cu(spec->nv);
  if (B > 0 && B < grid) grid = B;
  return (size_t)grid * slab_doubles(spec) * sizeof(double) + sizeof(int);
}

int cmpc_create(const cmpc_spec *spec, int device, cmpc_handle **out) {
  if (!out) return fail(nullptr, "cmpc_create: null out pointer");
  *out = nullptr;
  if (!spec_ok(spec)) return fail(nullptr, "cmpc_create: invalid spec (N in [1,64], nv in {4,8})");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, "cmpc_create: no HIP device");
  if (device < 0 || device >= ndev) return fail(nullptr, "cmpc_create: bad device index");
  cmpc_handle *h = new cmpc_handle();
  h->spec = *spec; h->device = device;
  hipDeviceProp_t prop;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
    delete h;
    return fail(nullptr, "cmpc_create: cannot query device");
  }
  h->num_cu = prop.multiProcessorCount;
  h->grid = h->num_cu * resident_per_cu(spec->nv);
  h->slab_doubles = slab_doubles(spec);
#ifdef CMPC_PROFILE
  (void)hipMalloc(&h->dbg, 65536 * sizeof(double));
  if (hipMalloc(&h->prof, 16 * sizeof(long long)) == hipSuccess) (void)hipMemset(h->prof, 0, 16 * sizeof(long long));
#endif
  if (hipMalloc(&h->scratch, (size_t)h->grid * h->slab_doubles * sizeof(double)) != hipSuccess ||
      hipMalloc(&h->ticket, sizeof(int)) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess ||
      hipEventCreate(&h->ev1) != hipSuccess) {
    cmpc_destroy(h);
    return fail(nullptr, "cmpc_create: device allocation failed");
  }
  *out = h;
  return 0;
}

int cmpc_destroy(cmpc_handle *h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  if (h->scratch) (void)hipFree(h->scratch);
  if (h->ticket) (void)hipFree(h->ticket);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  delete h;
  return 0;
}

int cmpc_solve_batch(cmpc_handle *h, int32_t B, const double *params, const double *warm_XU, double *out_XU,
                     int32_t *status, int32_t *iters, double *kkt_res, void *stream) {
  if (!h) return fail(nullptr, "cmpc_solve_batch: null handle");
  if (B < 0) return fail(h, "cmpc_solve_batch: negative batch");
  if (B == 0) return 0;
  if (!params || !out_XU || !status || !iters || !kkt_res) return fail(h, "cmpc_solve_batch: null buffer");
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(h, hipSetDevice(h->device));
  cmpc::KArgs ka;
  ka.sp = h->spec; ka.B = B; ka.recs = params; ka.warm = warm_XU; ka.out = out_XU;
  ka.status = status; ka.iters = iters; ka.kkt = kkt_res;
  ka.scratch = h->scratch; ka.scratch_stride = h->slab_doubles;
  ka.prof = h->prof;
  ka.dbg = h->dbg;
  const int grid = B < h->grid ? B : h->grid;
  HIP_TRY(h, hipMemsetAsync(h->ticket, 0, sizeof(int), st));
  HIP_TRY(h, hipEventRecord(h->ev0, st));
  if (h->spec.nv == 4)
    hipLaunchKernelGGL(cmpc_solve_kernel<4>, dim3(grid), dim3(64), 0, st, ka, h->ticket);
  else
    hipLaunchKernelGGL(cmpc_solve_kernel<8>, dim3(grid), dim3(64), 0, st, ka, h->ticket);
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

const char *cmpc_last_error(cmpc_handle *h) { return h ? h->err.c_str() : g_err.c_str(); }
const char *cmpc_version(void) { return "cmpc_amd 0.1 (gfx950)"; }

#ifdef CMPC_PROFILE
/* diagnostic build only: copy out instance 0's full iterate (x, lam, s, z) */
int cmpc_debug_read(cmpc_handle *h, double *out, int n) {
  if (!h || !h->dbg || n > 65536) return 1;
  return hipMemcpy(out, h->dbg, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess;
}
/* diagnostic build only: read and reset the phase cycle sums */
int cmpc_profile_read(cmpc_handle *h, long long *out8) {
  if (!h || !h->prof) return 1;
  if (hipMemcpy(out8, h->prof, 16 * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  (void)hipMemset(h->prof, 0, 16 * sizeof(long long));
  return 0;
}
#endif

}  // extern "C"
