// cmpc_emu.cpp -- TEST HARNESS ONLY.  Runs the *device* source of the solver
// (csrc/cmpc_kernel.hpp) on the CPU: the 64 lanes of a wavefront are 64 OS threads (128 for the
// two-wave workgroup of Solver<NV, 2>), LDS is a heap buffer they share, CMPC_SYNC() is a barrier of
// the whole workgroup and the wave-level primitives (broadcast, butterfly, MFMA) synchronise one wave only.  It lets the CPU test tier
// (-m "not gpu") exercise the kernel logic against the oracle, and makes the kernel debuggable
// with gdb / sanitizers.  It is never loaded by the product package: the product path is the HIP
// build of the same header and fails loudly without it.
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <atomic>
#include <string.h>
#include <thread>
#include <vector>

#define CMPC_HOST_EMU 1
static thread_local int emu_lane_id = 0;
// Barrier of the 64 lane threads.  A futex-based pthread barrier costs a wake-up per thread and the solver crosses
// tens of thousands of them per instance; 64 runnable threads on a handful of cores get through a yielding
// sense-reversing barrier several times faster.
struct EmuBarrier { std::atomic<int> count{0}, gen{0}; int width = 64; };
static EmuBarrier emu_barrier;              // the workgroup
static EmuBarrier emu_wave_barrier[2];      // one wave each
static inline int emu_barrier_wait(EmuBarrier *b) {
  const int g = b->gen.load(std::memory_order_acquire);
  if (b->count.fetch_add(1, std::memory_order_acq_rel) == b->width - 1) {
    b->count.store(0, std::memory_order_relaxed);
    b->gen.store(g + 1, std::memory_order_release);
  } else {
    while (b->gen.load(std::memory_order_acquire) == g) sched_yield();
  }
  return 0;
}
#define CMPC_DEV inline
#define CMPC_DEVN
#define CMPC_LANE (emu_lane_id)
#define CMPC_SYNC() emu_barrier_wait(&emu_barrier)
#define CMPC_SYNC_GLOBAL() emu_barrier_wait(&emu_barrier)
// broadcast of lane `src`'s value: through a shared slot, two barriers
#define CMPC_WAVE_ID() (emu_lane_id >> 6)
static double emu_bcast_slot[2];
static inline double emu_bcast(double v, int src) {       // src: lane of the caller's own wave
  const int w = emu_lane_id >> 6;
  if ((emu_lane_id & 63) == src) emu_bcast_slot[w] = v;
  emu_barrier_wait(&emu_wave_barrier[w]);
  double r = emu_bcast_slot[w];
  emu_barrier_wait(&emu_wave_barrier[w]);
  return r;
}
#define CMPC_BCAST(v, src) emu_bcast((v), (src))
static double emu_xor_slots[128];
static inline double emu_xor(double v, int m) {
  const int w = emu_lane_id >> 6;
  emu_xor_slots[emu_lane_id] = v;
  emu_barrier_wait(&emu_wave_barrier[w]);
  double r = emu_xor_slots[emu_lane_id ^ m];
  emu_barrier_wait(&emu_wave_barrier[w]);
  return r;
}
// butterfly step: this lane's value and its partner's (the device gets them without an LDS round trip, see cmpc_kernel.hpp)
#define CMPC_PAIR_OF(M, v, a, b) do { (a) = (v); (b) = emu_xor((v), (M)); } while (0)
#define CMPC_SCHED_FENCE() do { } while (0)
// emulated v_mfma_f64_16x16x4: every lane publishes its A / B element, then gathers its 4 results
struct cmpc_v4d { double v[4]; double &operator[](int i) { return v[i]; } const double &operator[](int i) const { return v[i]; } };
static double emu_mfma_a[128], emu_mfma_b[128];
static inline cmpc_v4d emu_mfma(double a, double b, cmpc_v4d c) {
  const int w = emu_lane_id >> 6, wl = emu_lane_id & 63;
  const double *ma = emu_mfma_a + 64 * w, *mb = emu_mfma_b + 64 * w;
  emu_mfma_a[emu_lane_id] = a; emu_mfma_b[emu_lane_id] = b;
  emu_barrier_wait(&emu_wave_barrier[w]);
  const int col = wl & 15, rq = wl >> 4;
  for (int r = 0; r < 4; ++r) {
    const int row = rq + 4 * r;
    double acc = c[r];
    for (int k = 0; k < 4; ++k) acc += ma[row + 16 * k] * mb[col + 16 * k];
    c[r] = acc;
  }
  emu_barrier_wait(&emu_wave_barrier[w]);
  return c;
}
#define CMPC_MFMA_F64(a, b, c) emu_mfma((a), (b), (c))
#define CMPC_OPAQUE(x) do { } while (0)

#include "../../online-non-linear-centroidal-mpc-with-stability-guarantees-for-robust-locomotion-of-legged-robots-_amd/csrc/cmpc_kernel.hpp"

template <int NV, int NW, bool PIPE = false>
static void run_batch(const cmpc::KArgs &ka, double *lds) {
  const cmpc_spec &sp = ka.sp;
  const size_t nrec = CMPC_NREC(sp.N), nsol = CMPC_NSOL(sp.N, NV), nstate = CMPC_NSTATE(sp.N, NV);
  for (int p = 0; p < ka.B; ++p) {
    std::vector<std::thread> th;
    for (int l = 0; l < 64 * (PIPE ? 2 : NW); ++l)
      th.emplace_back([&, l]() {
        emu_lane_id = l;
        cmpc::Solver<NV, NW, PIPE> s(ka, lds, ka.scratch, ka.recs + p * nrec);
        s.solve(ka.warm ? ka.warm + p * nsol : nullptr, ka.state_in ? ka.state_in + p * nstate : nullptr,
                ka.state_out ? ka.state_out + p * nstate : nullptr, ka.out + p * nsol, ka.status + p, ka.iters + p, ka.kkt + p);
      });
    for (auto &t : th) t.join();
  }
}

extern "C" int cmpc_emu_solve_batch_state(const cmpc_spec *sp, int32_t B, const double *recs, const double *warm,
                                          const double *state_in, double *out, double *state_out, int32_t *status,
                                          int32_t *iters, double *kkt);
extern "C" int cmpc_emu_solve_batch(const cmpc_spec *sp, int32_t B, const double *recs, const double *warm,
                                    double *out, int32_t *status, int32_t *iters, double *kkt) {
  return cmpc_emu_solve_batch_state(sp, B, recs, warm, nullptr, out, nullptr, status, iters, kkt);
}
extern "C" int cmpc_emu_solve_batch_state(const cmpc_spec *sp, int32_t B, const double *recs, const double *warm,
                                          const double *state_in, double *out, double *state_out, int32_t *status,
                                          int32_t *iters, double *kkt) {
  if (sp->N < 1 || sp->N > CMPC_MAX_N || (sp->nv != 4 && sp->nv != 8)) return 1;
  cmpc::KArgs ka;
  ka.sp = *sp; ka.B = B; ka.recs = recs; ka.warm = warm; ka.out = out;
  ka.state_in = state_in; ka.state_out = state_out;
  ka.status = status; ka.iters = iters; ka.kkt = kkt; ka.prof = nullptr;
  cmpc::fill_levels(ka);
  // (the 8-vertex solver is a two-wave workgroup: round 5's G'PG keeps one column of the stage block per lane, which one
  // wave of 64 lanes does not have for its 92 columns -- the one-wave form of rounds 2-4 is gone)
  const int nw8 = cmpc::WAVES_NV8;
  static_assert(cmpc::WAVES_NV8 == 2, "two waves");
  const size_t nd = (sp->nv == 4) ? cmpc::Dims<4>::scratch_doubles(sp->N) : cmpc::Dims<8, 2>::scratch_doubles(sp->N);
  // CMPC_EMU_PAIR=1 runs the 4-vertex solver as the pipelined pair of waves (two LDS images + the exchange words)
  const bool pair = sp->nv == 4 && getenv("CMPC_EMU_PAIR") && atoi(getenv("CMPC_EMU_PAIR")) == 1;
  const size_t nl = pair ? 2 * cmpc::Dims<4, 1, true>::LDS_DOUBLES : (sp->nv == 4) ? cmpc::Dims<4>::LDS_DOUBLES
                    : cmpc::Dims<8, 2>::LDS_DOUBLES;
  const double fill = getenv("CMPC_EMU_FILL") ? atof(getenv("CMPC_EMU_FILL")) : 0.0;
  std::vector<double> scratch(nd, fill), lds(nl, fill);
  ka.scratch = scratch.data(); ka.scratch_stride = nd;
  emu_barrier.count.store(0); emu_barrier.gen.store(0);
  emu_barrier.width = ((sp->nv == 8 && nw8 == 2) || pair) ? 128 : 64;
  for (auto &b : emu_wave_barrier) { b.count.store(0); b.gen.store(0); b.width = 64; }
  if (pair) run_batch<4, 1, true>(ka, lds.data());
  else if (sp->nv == 4) run_batch<4, 1>(ka, lds.data());
  else run_batch<8, 2>(ka, lds.data());
  return 0;
}

extern "C" int cmpc_emu_lds_bytes(int nv) {
  return (int)(sizeof(double) * ((nv == 4) ? cmpc::Dims<4>::LDS_DOUBLES : cmpc::Dims<8, 2>::LDS_DOUBLES));
}
