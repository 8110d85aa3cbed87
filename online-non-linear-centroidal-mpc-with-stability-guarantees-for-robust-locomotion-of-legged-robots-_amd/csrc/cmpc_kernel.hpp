// cmpc_kernel.hpp -- device code of the batched centroidal-MPC solver (gfx950 / CDNA4).
//
// One wavefront (one 64-thread workgroup) owns one problem instance at a time and walks the
// horizon serially; the 64 lanes share every stage's small dense blocks through LDS:
//   * lane i owns row i of the stage KKT block  M = H_k + [B A]' P_{k+1} [B A]   (packed lower
//     triangle in LDS: triangular row offsets are a permutation mod 32, so column sweeps of
//     64-bit words are bank-conflict free),
//   * the factorisation  M_uu = L L',  Ls = M_xu L^-T,  P_k = M_xx - Ls Ls'  runs in LDS,
//   * factors / gains of all stages spill to a per-workgroup global scratch slab that is written
//     and re-read with unit stride (lane = fastest index).
// Algorithm (identical to oracle/cmpc_oracle.c, which restates the reference NLP
// code/centroidal_mpc_vertices.py:126-353, :371-461): primal-dual interior point with a
// monotone barrier schedule, exact Lagrangian Hessian, Riccati recursion with inertia-correcting
// regularisation, fraction-to-the-boundary steps.
//
// The same source is compiled (a) by hipcc for the product library and (b) by g++ under
// -DCMPC_HOST_EMU for the CPU test harness (tests/emu), where the 64 lanes are OS threads and
// CMPC_SYNC() is a real barrier: every cross-lane LDS hand-off therefore carries an explicit
// barrier.
#pragma once
#include "../../include/cmpc.h"
#include <math.h>
#include <utility>
#ifdef CMPC_HOST_EMU
#include <cstdio>
#include <cstdlib>
#endif

#ifndef CMPC_HOST_EMU
#include <hip/hip_runtime.h>
#define CMPC_DEV __device__ __forceinline__
#define CMPC_LANE ((int)threadIdx.x)
// One wavefront per workgroup: LDS operations of a wave execute in issue order, so an LDS hand-off
// between lanes needs no s_barrier -- only that the compiler neither caches nor reorders LDS
// accesses across it.  CMPC_SYNC() is that (plus lgkmcnt(0)); it deliberately does NOT wait for
// outstanding global stores (a __syncthreads() would add vmcnt(0): ~2 us after every spill).
// CMPC_SYNC_GLOBAL() is the full fence, used where lanes exchange data through the global slab.
#define CMPC_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define CMPC_SYNC_GLOBAL() __syncthreads()
// Two waves per instance (Solver<NV, 2>): every LDS hand-off is a workgroup barrier.  The source carries a CMPC_SYNC at
// every cross-lane hand-off (the host emulation runs the lanes as threads), so the same code is correct for 128 lanes.
#define CMPC_SYNC_WG() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define CMPC_WAVE_ID() ((int)__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6))
// value held by lane `src` (wave-uniform index) broadcast to every lane, no LDS round trip
static __device__ __forceinline__ double cmpc_bcast(double v, int src) {
  union { double d; int i[2]; } u; u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], src);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
  return u.d;
}
#define CMPC_BCAST(v, src) cmpc_bcast((v), (src))
// D(16x16) += A(16x4) B(4x16) on the matrix cores, fp64.  Lane l supplies A[l&15][l>>4] and
// B[l>>4][l&15]; it receives D[(l>>4) + 4r][l&15] in component r (gfx950 f64 layout).
typedef double cmpc_v4d __attribute__((ext_vector_type(4)));
#define CMPC_MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#include "cmpc_wave.hpp"   // cmpc_pair_of<M>: butterfly steps without LDS round trips
#define CMPC_PAIR_OF(M, v, a, b) cmpc_pair_of<M>((v), (a), (b))
// makes a per-lane value opaque to the optimiser: stops loop-invariant code motion from hoisting the
// hundreds of lane-derived index computations out of the stage / iteration loops (they were kept live
// across the whole solve and spilled)
#define CMPC_OPAQUE(x) asm volatile("" : "+v"(x))
#define CMPC_OPAQUE_D(x) asm volatile("" : "+v"(x))
// a wave-uniform integer the compiler cannot prove uniform: said so, it lives in a scalar register
#define CMPC_UNIFORM_INT(x) __builtin_amdgcn_readfirstlane(x)
// a wave-uniform number of the problem spec (a kernel argument, in scalar registers), opaque at every use: what the
// optimiser can derive from the spec alone (2 w, w / 2, 1 - k1^2, ... -- some twenty products) it computes at kernel entry
// and keeps in VECTOR registers through the whole solve (there is no scalar fp64 arithmetic), forty-five of them
static __device__ __forceinline__ double cmpc_fresh(double a) { asm volatile("" : "+s"(a)); return a; }
#define CMPC_FRESH_D(x) cmpc_fresh(x)
#ifdef CMPC_FRESH_SPEC
#define SPD(field) cmpc_fresh(sp.field)
#else
#define SPD(field) (sp.field)
#endif
// a double every lane holds the same value of, said so: it lives in a scalar register pair (spilled, that is two
// v_readlane; a vector register spilled is a scratch load whose wait also drains every global load in flight)
static __device__ __forceinline__ double cmpc_uniform_d(double v) {
  union { double d; int i[2]; } u; u.d = v;
  int lo, hi;   // (spelled out: the builtin is folded away where the optimiser can see that the value is uniform -- and the value stays where the vector ALU put it)
  // gfx950 wants a wait state between a vector-ALU write of a register and a v_readfirstlane of it, and two between the
  // scalar result and a vector-ALU read of it; the compiler pads the instructions it issues itself, not the text of an asm
  asm("s_nop 0\n\tv_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3\n\ts_nop 1" : "=s"(lo), "=s"(hi) : "v"(u.i[0]), "v"(u.i[1]));
  u.i[0] = lo; u.i[1] = hi;
  return u.d;
}
#ifdef CMPC_NO_UNIFORM_D
#define CMPC_UNIFORM_D(x) (x)
#else
#define CMPC_UNIFORM_D(x) cmpc_uniform_d(x)
#endif
// nothing is scheduled across this point (keeps a batch of loads, its wait and its arithmetic together: left to itself
// the scheduler parks the loaded words and spills them)
#define CMPC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// the lane id again from the execution mask (two instructions): where it is re-derived the old value need not stay
// live -- or be spilled -- across the code in front (one wave per workgroup: lane id = thread id)
#define CMPC_RELANE(x) do { (x) = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) + lane_base; } while (0)
// Pipelined pair of waves (Solver<NV, 1, true>): each wave works through its own phase as a single-wave team, so inside
// a phase a hand-off is wave-local (CMPC_SYNC_WAVE = CMPC_SYNC, CMPC_FENCE_WAVE for data that went through the slab); the
// two waves meet at workgroup barriers only (CMPC_SYNC_WG: LDS hand-off, CMPC_SYNC_GLOBAL: global memory as well).
#define CMPC_SYNC_WAVE(w) CMPC_SYNC()
#define CMPC_FENCE_WAVE(w) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#endif
#ifdef CMPC_HOST_EMU
#define CMPC_SYNC_WAVE(w) emu_barrier_wait(&emu_wave_barrier[(w)])
#define CMPC_FENCE_WAVE(w) emu_barrier_wait(&emu_wave_barrier[(w)])
#define CMPC_SYNC_WG() emu_barrier_wait(&emu_barrier)
#endif
#ifndef CMPC_RELANE
#define CMPC_RELANE(x) do { } while (0)
#define SPD(field) (sp.field)
#define CMPC_FRESH_D(x) (x)
#define CMPC_UNIFORM_D(x) (x)
#endif
// x * y + z in ONE rounding, spelled out.  Everywhere else the multiply-adds are formed by the compiler's contraction of
// a * b + c, whose choices depend on the code around (which product is hoisted out of a masked block, which sum is
// restarted at a neighbouring add): fine for code both solver kernels share verbatim, not for G'PG, which the single wave
// and the pair reach in different orders and must still form bit for bit alike.  (The host emulation is built with
// -mfma -ffp-contract=off: the same fused operations, nothing else fused.)
#define CMPC_FMA(x, y, z) __builtin_fma((x), (y), (z))
#ifndef CMPC_OPAQUE_D
#define CMPC_OPAQUE_D(x) do { } while (0)
#endif
#ifndef CMPC_UNIFORM_INT
#define CMPC_UNIFORM_INT(x) (x)
#endif

// Optional phase timers (diagnostic build only, -DCMPC_PROFILE): cycles per phase summed over the
// launch, written to a buffer nothing else reads.
#if defined(CMPC_PROFILE) && !defined(CMPC_HOST_EMU)
#define CMPC_TICK(slot) do { long long now_ = clock64(); tprof[slot] += now_ - tlast; tlast = now_; } while (0)
#define CMPC_TICK_RESET() do { tlast = clock64(); } while (0)
#else
#define CMPC_TICK(slot) do { } while (0)
#define CMPC_TICK_RESET() do { } while (0)
#endif

namespace cmpc {

struct KArgs {
  cmpc_spec sp;
  int B;
  const double *recs;   // [B][nrec]
  const double *warm;   // [B][nsol] or null
  const double *state_in;   // [B][nstate] or null: solver state of the previous tick (CMPC_NSTATE, include/cmpc.h)
  double *state_out;        // [B][nstate] or null
  double *out;          // [B][nsol]
  int32_t *status;
  int32_t *iters;
  double *kkt;
  double *scratch;      // [grid][scratch_stride]
  size_t scratch_stride;
  long long *prof;      // [8] phase cycle sums (CMPC_PROFILE builds), else null
  // levels of the outer loop derived from sp.tol (cmpc::fill_levels, on the host): as kernel arguments they sit in scalar
  // registers; computed in the kernel they are vector-ALU results hoisted to kernel entry -- and spilled
  double tol_acc, tol_tenth;
};

enum { R_LYAP = 0, R_CZ = 1, R_HWC = 2, R_BOX = 3, R_FRIC = 15 };

// termination safeguards (same constants as the oracle)
constexpr double ACC_FACTOR = 100.0;
constexpr int ACC_ITERS = 8;
inline void fill_levels(KArgs &ka) { ka.tol_acc = ACC_FACTOR * ka.sp.tol; ka.tol_tenth = ka.sp.tol / 10; }
constexpr double STALL_STEP = 1e-7;
constexpr int STALL_ITERS = 6;
// no progress at a barrier value: NOPROG_ITERS iterations without halving the best error of that barrier problem
// (the KKT error at the final value); the run then ends as CMPC_ACCEPTABLE once a point within spec.acc_tol is in hand
// (see the oracle)
constexpr int NOPROG_ITERS = 12;
// smallest pivot accepted by the stage factorisation: max(PIV_MIN, PIV_FRAC * delta) (see the oracle)
constexpr double PIV_MIN = 1e-8;
constexpr double PIV_FRAC = 0.1;
// Newton iterations at the final barrier value after the tolerance is first met (see the oracle)
constexpr int POLISH_ITERS = 1;
// barrier schedule (see the oracle)
constexpr double MU_INIT = 100.0;
constexpr double MU_FACTOR = 0.1;
// warm start of the interior point method: the solver state is the iterate at the last barrier value >= MU_WARM
// (see the oracle for the choice of the level)
constexpr double MU_WARM = 1e-7;
// cold start: states rolled out from x_0 under the initial inputs (see the oracle)
constexpr bool COLD_ROLLOUT = true;
// a resumed solve that has not left the state's barrier value after this many iterations gives up (the state was too far
// from this tick's problem) and the plain solve follows, with what is left of the iteration budget (see the oracle)
constexpr int RESUME_RECENTRE_ITERS = 20;
// waves per instance of the 8-vertex solver (Solver<8, WAVES_NV8>; the 4-vertex one is a single wave)
constexpr int WAVES_NV8 = 2;

#ifndef CMPC_NO_DEVICE_CODE
#include "cmpc_lds_asm.hpp"
#endif

#ifndef CMPC_NO_DEVICE_CODE
// v[0..CNT) = p[0..CNT) (LDS), in batches of independent reads
template <int CNT> CMPC_DEV void lds_read_row(double (&v)[CNT], const double *p) {
  static_assert(CNT == 28 || CNT == 32 || CNT == 36 || CNT == 56, "supported row lengths");
  if constexpr (CNT == 28) lds_read_strided28<1>(v, p);
  else if constexpr (CNT == 32) {
    double a[16], b[16];
    lds_read_strided16<1>(a, p); lds_read_strided16<1>(b, p + 16);
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = a[i]; v[16 + i] = b[i]; }
  } else if constexpr (CNT == 36) {
    double a[18], b[18];
    lds_read_strided18<1>(a, p); lds_read_strided18<1>(b, p + 18);
#pragma unroll
    for (int i = 0; i < 18; ++i) { v[i] = a[i]; v[18 + i] = b[i]; }
  } else {
    double a[28], b[28];
    lds_read_strided28<1>(a, p); lds_read_strided28<1>(b, p + 28);
#pragma unroll
    for (int i = 0; i < 28; ++i) { v[i] = a[i]; v[28 + i] = b[i]; }
  }
}
#endif

// Sizes of the stage block of a solver (NV contact vertices per foot, NW waves per instance).
template <int NV, int NW> struct Sizes {
  static_assert(NW == 1 || NW == 2, "one or two waves per instance");
  static constexpr int WS = 64 * NW;          // lanes of the workgroup that owns an instance
  static constexpr int NF = 2 * NV;           // contact vertices
  static constexpr int NU = 6 * NV + 8;
  static constexpr int NXA = CMPC_NX + 2 * NV;
  static constexpr int NZ = NU + NXA;
  static constexpr int NI = 15 + 10 * NV;
  static constexpr int NTRI = NZ * (NZ + 1) / 2;
  static constexpr int NH = (NZ + WS - 1) / WS;   // rows / columns of the stage block owned by one lane
  static constexpr int PS = NXA + 1;          // odd row strides: conflict-free column access
  static constexpr int LS = NU + 1;
  static constexpr int NPT = NXA * (NXA + 1) / 2;
};

// ---- LDS map (doubles), 8-vertex solver (one or two waves): M, the full symmetric P, the stage vectors.  Round 5: G'PG out of
// registers here too (Solver::gt_phase, Hessian rows first as in the pipelined pair): the staging tile of T = P [B A]
// (36 x 47 doubles in two column halves) is gone; its place holds the 80 words of the factorisation's hand-off table.
template <int NV, int NW, bool PIPE> struct LdsMap : Sizes<NV, NW> {
  using S = Sizes<NV, NW>;
  static constexpr bool GT = true, GT_FIRST = false;
  static constexpr int oM = 0;
  static constexpr int oP = oM + S::NTRI + (S::NTRI & 1);
  static constexpr bool P_PACKED = false;
  static constexpr int P_DOUBLES = S::NXA * S::PS + ((S::NXA * S::PS) & 1);
  static constexpr int DUMPN = S::WS;                          // write-only slots for masked-off stores (lane & (DUMPN - 1))
  static constexpr int MISCN = 64;                             // scalars of the geometry / inequality phases (0 .. 49 used)
  static constexpr int oXN1 = oP + P_DOUBLES;                  // v0 = p_{k+1} + P_{k+1} b
  static constexpr int oSR = oXN1 + S::NXA;   // stage record k (19), k-1 (19), header (24)
  static constexpr int oSRP = oSR + 20;
  static constexpr int oHDR = oSRP + 20;
  static constexpr int oH0 = oHDR + 24;       // gradient parts h = h0 + mu*h1 of the stage (NZ each)
  static constexpr int oH1 = oH0 + S::NZ;
  static constexpr int oPC1 = oH1 + S::NZ;    // mu-coefficient of the cost-to-go gradient (NXA)
  static constexpr int oRED = oPC1 + S::NXA;  // 4 scratch slots
  static constexpr int oCOLD = oRED + 4;      // outer-loop state that is touched once per iteration (8 scalars): kept here
                                              // instead of in registers, where it was spilled to scratch
  static constexpr int oDUMP = oCOLD + 8;
  // ---- evaluation vectors: dead from the end of the gradient (the P b product for BV / PC) to the next stage's load
  static constexpr int oXK = oDUMP + DUMPN + ((oDUMP + DUMPN) & 1);
  static constexpr int oUK = oXK + S::NXA;
  static constexpr int oLAMK = oUK + S::NU;
  static constexpr int oLAMN = oLAMK + S::NXA;
  static constexpr int oUPX = oLAMN + S::NXA;
  static constexpr int oSK = oUPX + S::NU;
  static constexpr int oZK = oSK + S::NI;
  static constexpr int oGK = oZK + S::NI;
  static constexpr int oW0 = oGK + S::NI;     // sigma = z/s
  static constexpr int oW1 = oW0 + S::NI;     // sigma*(g+s)
  static constexpr int oW2 = oW1 + S::NI;     // 1/s
  static constexpr int oVDV = oW2 + S::NI;    // R' v_j (NF x 3)
  static constexpr int oMISC = oVDV + 3 * S::NF;
  static constexpr int oAL = oMISC + MISCN;   // Lyapunov gradient (NZ); second temporary of the backward vectors
  static constexpr int oGH = oAL + S::NZ;     // rows 6..8 of [B A] without identity (3 x NZ)
  static constexpr int oBV = oGH + 3 * S::NZ;
  static constexpr int oPC = oBV + S::NXA;
  static constexpr int oEND = oPC + S::NXA;
  // overlays: r_j (NF x 3) lives from the first to the third step of the geometry, in the words the inequality rows
  // then fill with the Lyapunov gradient; the two NZ-word temporaries of the backward vectors take the slacks' and the
  // Lyapunov gradient's place
  static constexpr int oVR = oAL;
  static constexpr int oTV = oSK;
  static constexpr int oTV2 = oAL;
  static_assert(3 * S::NF <= S::NZ && S::NZ <= 2 * S::NI, "overlays fit");
  static constexpr int oUB = oEND + (oEND & 1);   // in-block multiplier table of the factorisation (64 words) + the second
  static constexpr int LDS_DOUBLES = oUB + 80;    // wave's panel hand-off (11 words)
  static_assert(oUB % 2 == 0, "16-byte reads of the in-block table");
};

// ---- LDS map (doubles), one-wave 4-vertex solver (round 5).  T = P [B A] is held one column per lane in REGISTERS and
// M (+)= [B A]' T is formed column-per-lane from it (Solver::gt_phase), so there is no staging tile; P is a packed
// triangle.  The single wave forms G'PG BEFORE the Hessian rows (GT_FIRST): P_{k+1} is dead after the T columns are in
// registers, i.e. from early in a stage until its factorisation writes P_k, and M is dead from the factor store of stage
// k + 1 until stage k's G'T is written.  What is born and dies inside those windows is overlaid on the two regions:
//   in M (dead: load ... T phase):   GH, BV, r_j        (geometry -> column lists / P b / T phase)
//   in P (dead: T phase ... factor): AL, GK, W0, W1, W2 (inequality rows -> gradient)
// 2866 doubles = 22 928 bytes: SEVEN workgroups per CU (1280-byte granules: <= 23 040).  The pipelined pair (PIPE) keeps
// the Hessian rows first -- its evaluating wave runs a stage ahead of the cost-to-go -- and nothing of it is overlaid
// (two images, three pairs per CU as before).
template <bool PIPE> struct LdsMap<4, 1, PIPE> : Sizes<4, 1> {
  using S = Sizes<4, 1>;
#ifdef CMPC_NO_GT_FIRST                          // (diagnostic build: the single wave in the pair's order, Hessian rows first)
  static constexpr bool GT = true, GT_FIRST = false;
#else
  static constexpr bool GT = true, GT_FIRST = !PIPE;
#endif
  static constexpr int oM = 0;
  static constexpr int oP = oM + S::NTRI + (S::NTRI & 1);
  static constexpr bool P_PACKED = true;
  static constexpr int P_DOUBLES = S::NPT + (S::NPT & 1);
  static constexpr int DUMPN = 16;
  static constexpr int MISCN = 52;
  // live across a whole stage
  static constexpr int oXN1 = oP + P_DOUBLES;
  static constexpr int oSR = oXN1 + S::NXA;
  static constexpr int oSRP = oSR + 20;
  static constexpr int oHDR = oSRP + 20;
  static constexpr int oH0 = oHDR + 24;
  static constexpr int oH1 = oH0 + S::NZ;
  static constexpr int oPC1 = oH1 + S::NZ;
  static constexpr int oPC = oPC1 + S::NXA;
  static constexpr int oRED = oPC + S::NXA;
  static constexpr int oCOLD = oRED + 4;
  static constexpr int oDUMP = oCOLD + 8;
  // from the stage's loads to its gradient
  static constexpr int oXK = oDUMP + DUMPN + ((oDUMP + DUMPN) & 1);
  static constexpr int oUK = oXK + S::NXA;
  static constexpr int oLAMK = oUK + S::NU;
  static constexpr int oLAMN = oLAMK + S::NXA;
  static constexpr int oUPX = oLAMN + S::NXA;
  static constexpr int oSK = oUPX + S::NU;
  static constexpr int oZK = oSK + S::NI;
  static constexpr int oVDV = oZK + S::NI;
  static constexpr int oMISC = oVDV + 3 * S::NF;
  static constexpr int oLATE = oMISC + MISCN;                  // end of what every variant keeps of its own
  // inequality rows -> gradient: AL, GK, W0, W1, W2
  static constexpr int LATE_DOUBLES = S::NZ + 4 * S::NI;
  static constexpr int oAL = GT_FIRST ? oP : oLATE;
  static constexpr int oGK = oAL + S::NZ;
  static constexpr int oW0 = oGK + S::NI;
  static constexpr int oW1 = oW0 + S::NI;
  static constexpr int oW2 = oW1 + S::NI;
  static_assert(!GT_FIRST || LATE_DOUBLES <= P_DOUBLES, "the late evaluation vectors fit the dead P region");
  // geometry -> T phase: GH, BV, r_j
  static constexpr int oGH = GT_FIRST ? oM : oLATE + LATE_DOUBLES;
  static constexpr int oBV = oGH + 3 * S::NZ;
  static constexpr int oVR = GT_FIRST ? oBV + S::NXA : oAL;   // (PIPE: in the words the inequality rows then fill, as before)
  static constexpr int oEND = GT_FIRST ? oLATE : oBV + S::NXA;
  // temporaries of the backward vectors (NZ words each), in-block multiplier table of the factorisation (64 words):
  // all in evaluation vectors that are dead by then
  static constexpr int oTV = oSK;
  static constexpr int oTV2 = oXK;
  static constexpr int oUB = oSK;
  static_assert(oTV2 + S::NZ <= oLAMK && oTV + S::NZ <= oVDV && oUB + 64 <= oVDV && oUB % 2 == 0, "temporaries fit");
  static_assert(3 * S::NF <= S::NZ, "r_j fits");
  static constexpr int LDS_DOUBLES = oEND + (oEND & 1);
};

template <int NV, int NW = 1, bool PIPE = false> struct Dims : LdsMap<NV, NW, PIPE> {
  using S = Sizes<NV, NW>;
  using Lm = LdsMap<NV, NW, PIPE>;
  static constexpr int WS = S::WS, NF = S::NF, NU = S::NU, NXA = S::NXA, NZ = S::NZ, NI = S::NI, NTRI = S::NTRI, NH = S::NH,
                       PS = S::PS, LS = S::LS, NPT = S::NPT;
  static constexpr bool COMPACT = (NV == 4 && NW == 1);
  // ---- global scratch map per stage (doubles) ----
  // The factorised stage block as it stands in LDS, a packed lower triangle of NZ rows, copied word for word:
  // rows 0..NU-1 hold Lambda, row NU+c holds [Ls row c | P_k row c up to the diagonal].  (Round 2 wrote Lambda as a
  // zero-filled NU x NU square and Ls | P_k as NXA full 64-word rows: 2816 words per stage against 1830.)
  static constexpr int gM = 0;
  static constexpr bool W_MERGE = (NZ <= WS);            // forward sweep: lanes >= NU take the P_k columns
  static constexpr int gAL = ((NTRI + 7) / 8) * 8;
  static constexpr int gGH = gAL + NZ;
  static constexpr int GHS = WS * NH;      // row stride of the three dense dynamics rows in the slab
  static constexpr int gB = gGH + 3 * GHS;
  static constexpr int gPB = gB + NXA;
  static constexpr int gPV = gPB + NXA;
  static constexpr int gL = gPV + NXA;
  static constexpr int gG = gL + NU;
  static constexpr int gL1 = gG + NI;         // l = gL + mu*gL1,  p = gPV + mu*gPV1
  static constexpr int gPV1 = gL1 + NU;
  static constexpr int STAGE = ((gPV1 + NXA + 7) / 8) * 8;
  // solver state of one instance (CMPC_NSTATE): [XU | lam (N+1) x NXA | s (N+1) x NI | z (N+1) x NI | mu, 7 spare |
  // contact flags of the N+1 nodes, left then right]
  static constexpr int state_lam(int N) { return CMPC_NSOL(N, NV); }
  static constexpr int state_s(int N) { return state_lam(N) + (N + 1) * NXA; }
  static constexpr int state_z(int N) { return state_s(N) + (N + 1) * NI; }
  static constexpr int state_mu(int N) { return state_z(N) + (N + 1) * NI; }
  static constexpr int state_fl(int N) { return state_mu(N) + 8; }
  static_assert(state_fl(20) + 2 * 21 == CMPC_NSTATE(20, NV), "layout of include/cmpc.h");
  // iterate arrays follow the (N+1) stage blocks
  static size_t scratch_doubles(int N) {
    size_t n = (size_t)(N + 1) * STAGE;
    n += (size_t)(N + 1) * NXA * 4;   // x, lam, dx, lamn
    n += (size_t)(N + 1) * NU * 3;    // u, du, uprox
    n += (size_t)(N + 1) * NI * 4;    // s, z, ds, dz
    return (n + 7) / 8 * 8;
  }
};

#ifndef CMPC_NO_DEVICE_CODE

CMPC_DEV int tri(int i) { return i * (i + 1) / 2; }

// Global array addressed as (wave-uniform base) + (32-bit element index).  Written this way hipcc selects the
// SGPR-base form of the global memory instructions (global_load_dwordx2 v, v_offset, s[base:base+1]); with 64-bit
// index arithmetic every access went through a 64-bit per-lane address in a VGPR pair, a dozen of which were
// hoisted to kernel entry, kept live through the whole solve and spilled.  All arrays are far below 4 GB.
typedef double cmpc_v2d __attribute__((vector_size(16)));
struct GArr {
  double *p;
  CMPC_DEV double &operator[](unsigned i) const { return *(double *)((char *)p + (size_t)(i * 8u)); }
  CMPC_DEV cmpc_v2d &pair(unsigned i) const { return *(cmpc_v2d *)((char *)p + (size_t)(i * 16u)); }   // 16-byte aligned pairs
};

// PIPE (one-wave solver only): the workgroup is a PAIR of waves on one instance.  Wave 1 evaluates stage k - 1 (loads,
// geometry, inequality rows, Hessian rows, gradient: everything that does not need the cost-to-go of stage k) into one of
// two LDS images while wave 0 runs the Riccati step of stage k (G'PG, factorisation, backward vectors, factor store) out
// of the other; the cost-to-go P, its gradient and the cold scalars live in image 0 only.  Same arithmetic in the same
// order as the single wave: results are bit for bit those of Solver<NV, 1>.  For batches that do not fill the GPU (the
// reference's own use is ONE instance per tick, code/simulation.py:203-204): an instance finishes ~1.4x sooner.
template <int NV, int NW = 1, bool PIPE = false> struct Solver {
  using D = Dims<NV, NW, PIPE>;
  // PIPE: words behind the two LDS images.  [0..5] error measures, [6] ap, [7] ad, [8..9] factorisation verdict of the
  // sweep step in hand (by step parity), [16 ..): du_k of the forward sweep by stage parity (read by the slack wave)
  static constexpr int XCH_DU = 16, XCH_DOUBLES = XCH_DU + 2 * D::NU;
  // They stand in the P region of image 1, which nothing else uses (the cost-to-go lives in image 0 only): the pair's
  // allocation is exactly two images, 42 granules of 1280 bytes -> three pairs per CU.
  static_assert(!PIPE || XCH_DOUBLES <= D::P_DOUBLES, "the exchange words fit in the unused P region of the second image");
  static constexpr int XCH_AT = D::LDS_DOUBLES + D::oP;
  static_assert(!PIPE || NW == 1, "the pipelined pair runs the one-wave solver");
  // PIPE: while wave 0 runs riccati_stage(k - 1) in the other image (G'PG out of registers, the factorisation with its
  // in-block table), wave 1's pair_vectors(k) reads that image's BV and writes its XN1: the table must not touch them.
  static_assert(!PIPE || (D::oUB + 64 <= D::oBV && D::oUB >= D::oXN1 + D::NXA && !D::GT_FIRST),
                "pair_vectors reads BV / writes XN1 of the image the Riccati wave is working in");
  static constexpr int NF = D::NF, NU = D::NU, NXA = D::NXA, NZ = D::NZ, NI = D::NI, NH = D::NH, WS = D::WS;
  static_assert(NW == 1 || NU <= 64, "the input rows (pivot chains, substitutions) live in the first wave");

  const KArgs &ka;
  const cmpc_spec &sp;
  double *lds;             // LDS image the wave works on (PIPE: image k & 1 of the stage in hand)
  double *ldsR;            // image 0: home of P, p, the cold scalars (PIPE: the Riccati wave's; otherwise = lds)
  int lane_base = 0;       // first lane id of this wave's team (two-wave solver: 64 * wave)
  GArr gs;                 // this workgroup's scratch slab
  GArr rec;                // this instance's parameter record (read only)
  int N, lane;               // lane: 0 .. WS-1 over the workgroup (two waves: 64 * wave + lane of the wave)
  int wv = 0;                // wave of the workgroup (0 when NW = 1), wave-uniform
  // global iterate arrays
  GArr gx, glam, gdx, glamn, gu, gdu, gupx, gsl, gz, gds, gdz;
  // per-lane column list of [B A]: rows / coefficients (id, h0, h1, h2, sp1, sp2)
  int lr[NH][6];
  double lg[NH][6];
  double piv_min = PIV_MIN;  // pivot acceptance threshold of the current sweep
  GArr st_in{nullptr};       // solver state resumed from (null: cold rule for slacks / multipliers)
  long long tprof[28] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;

  CMPC_DEV Solver(const KArgs &a, double *l, double *g, const double *r)
      : ka(a), sp(a.sp), lds(l), ldsR(l), gs{g}, rec{const_cast<double *>(r)}, N(a.sp.N), lane(CMPC_LANE) {
    if constexpr (NW > 1 || PIPE) wv = CMPC_WAVE_ID();
    if constexpr (NW > 1) lane_base = 64 * wv;
    if constexpr (PIPE) lane &= 63;
    // (one wave per workgroup: the lane id comes from the execution mask, so that threadIdx.x need not stay live -- or be
    // spilled -- across the instance loop)
    if constexpr (NW == 1 && !PIPE) CMPC_RELANE(lane);
    double *p = g + (size_t)(N + 1) * D::STAGE;
    gx = GArr{p}; p += (size_t)(N + 1) * NXA;
    glam = GArr{p}; p += (size_t)(N + 1) * NXA;
    gdx = GArr{p}; p += (size_t)(N + 1) * NXA;
    glamn = GArr{p}; p += (size_t)(N + 1) * NXA;
    gu = GArr{p}; p += (size_t)(N + 1) * NU;
    gdu = GArr{p}; p += (size_t)(N + 1) * NU;
    gupx = GArr{p}; p += (size_t)(N + 1) * NU;
    gsl = GArr{p}; p += (size_t)(N + 1) * NI;
    gz = GArr{p}; p += (size_t)(N + 1) * NI;
    gds = GArr{p}; p += (size_t)(N + 1) * NI;
    gdz = GArr{p};
  }
  CMPC_DEV GArr stage(int k) const { return GArr{gs.p + (size_t)k * D::STAGE}; }
  CMPC_DEV double &L(int o) const { return lds[o]; }
  CMPC_DEV double &R(int o) const { return ldsR[o]; }         // Riccati-owned words: P, PC, PC1, COLD
  CMPC_DEV void image(int k) { if constexpr (PIPE) lds = ldsR + (k & 1) * D::LDS_DOUBLES; }
  // LDS hand-off between lanes: one wave needs only its own LDS traffic drained, two waves a workgroup barrier
  CMPC_DEV void sync() const {
    if constexpr (PIPE) { CMPC_SYNC_WAVE(wv); }
    else {
#ifdef CMPC_HOST_EMU
      CMPC_SYNC();
#else
      if constexpr (NW == 1) CMPC_SYNC(); else CMPC_SYNC_WG();
#endif
    }
  }
  // full fence for data exchanged through the slab inside a phase (PIPE: the phase belongs to one wave)
  CMPC_DEV void gsync() const {
    if constexpr (PIPE) { CMPC_FENCE_WAVE(wv); } else { CMPC_SYNC_GLOBAL(); }
  }
  // PIPE: the two waves of the pair meet (global memory included)
  CMPC_DEV void pair_sync() const { if constexpr (PIPE) { CMPC_SYNC_GLOBAL(); } }
  CMPC_DEV bool first_wave() const { return NW == 1 || wv == 0; }

  // contact flag gamma_f at node k and k-1 from the staged records
  CMPC_DEV double gam_k(int k, int f) const { return (k == N) ? L(D::oHDR + 22 + f) : L(D::oSR + 17 + f); }
  CMPC_DEV double gam_km1(int f) const { return L(D::oSRP + 17 + f); }
  CMPC_DEV void vert_local(int j, double &vx, double &vy) const {
    const double Lh = SPD(foot_length) * 0.5, Wh = SPD(foot_width) * 0.5;
    const double cx[8] = {Lh, Lh, -Lh, -Lh, Lh, 0.0, -Lh, 0.0};
    const double cy[8] = {Wh, -Wh, -Wh, Wh, 0.0, -Wh, 0.0, Wh};
    vx = cx[j]; vy = cy[j];
  }

  // ---------------------------------------------------------------------------------------
  // Stage loads: x_k, u_k, x_{k+1}, lam_k, lam_{k+1}, s_k, z_k, uprox_k and the record rows.
  // ---------------------------------------------------------------------------------------
  // Stage iterates and record rows to LDS.  Every global load is issued unconditionally (clamped
  // indices) before the first LDS write: loads inside lane- or stage-conditional blocks compile to
  // one exposed HBM round trip each (load, s_waitcnt vmcnt(0), ds_write), a dozen per stage.
  CMPC_DEV void load_stage(int k) {
    static_assert(NXA <= 64 && NU <= 64, "one lane per state / input component");
    constexpr int NIH = (NI + WS - 1) / WS;
    const bool in = k < N;
    const int kn = in ? k + 1 : k, ku = in ? k : N - 1, kp = (k >= 1) ? k - 1 : 0;
    const int ix = (lane < NXA) ? lane : 0, iu = (lane < NU) ? lane : 0, ir = (lane < 19) ? lane : 0;
    const double x0 = gx[k * NXA + ix], x1 = gx[kn * NXA + ix];
    const double u0 = gu[ku * NU + iu], up = gupx[ku * NU + iu];
    const double r0 = rec[24 + 19 * ku + ir], r1 = rec[24 + 19 * kp + ir], hd = rec[(lane < 24) ? lane : 0];
    double sv[NIH], zv[NIH];
    const double l0 = glam[k * NXA + ix], l1 = glam[kn * NXA + ix];
    {
#pragma unroll
      for (int h = 0; h < NIH; ++h) {
        const int r = lane + WS * h, rc = (r < NI) ? r : 0;
        sv[h] = gsl[k * NI + rc]; zv[h] = gz[k * NI + rc];
      }
    }
    if (lane < NXA) {
      L(D::oXK + lane) = x0; L(D::oXN1 + lane) = in ? x1 : 0.0;
      L(D::oLAMK + lane) = l0; L(D::oLAMN + lane) = in ? l1 : 0.0;
    }
    if (lane < NU) { L(D::oUK + lane) = in ? u0 : 0.0; L(D::oUPX + lane) = in ? up : 0.0; }
    {
#pragma unroll
      for (int h = 0; h < NIH; ++h) {
        const int r = lane + WS * h;
        if (r < NI) { L(D::oSK + r) = sv[h]; L(D::oZK + r) = zv[h]; }
      }
    }
    if (lane < 19) { L(D::oSR + lane) = in ? r0 : 0.0; L(D::oSRP + lane) = (k >= 1) ? r1 : 0.0; }
    if (lane < 24) L(D::oHDR + lane) = hd;
    sync();
  }

  // ---------------------------------------------------------------------------------------
  // Per-vertex geometry, sums, dynamics value b = F(x,u) - x_{k+1}, GH rows.  (k < N)
  // reference centroidal_dynamic :371-461
  // MISC: [0..5] Fs[f][a], [6..8] tau, [9..11] pi, [12..20] SF = sum_f gamma_f sum_j skew(f_j),
  //       [21..26] SFf[f][...] unused, [30..31] psi-psi curvature per foot,
  //       [32..34] z1, [35..37] z2, [38..40] un, [41..43] gz1, [44..46] gz2, [47..49] V, [50] lyap val
  // ---------------------------------------------------------------------------------------
  CMPC_DEV void stage_geometry(int k) {
    const double d = SPD(delta), m = L(D::oHDR + 20);
    // (A) one lane per contact vertex: rotated offsets, lever arm, and the vertex's terms of every
    // per-foot sum (torque r x f, yaw term (R'v) x f, yaw-yaw curvature) -- products staged in the
    // H0/H1 block (written later in the sweep), summed in (B) by one lane per result.
    double *tq = &L(D::oH0), *yw = &L(D::oH0 + 3 * NF), *qq = &L(D::oH0 + 6 * NF);
    static_assert(7 * NF <= 2 * NZ, "vertex products fit the H0/H1 block");
    if (lane < NF) {
      const int f = lane / NV, j = lane % NV;
      const double yaw = L(D::oXK + 12 + 4 * f);
      const double px = L(D::oXK + 13 + 4 * f), py = L(D::oXK + 14 + 4 * f), pz = L(D::oXK + 15 + 4 * f);
      const double cx = L(D::oXK + 0), cy = L(D::oXK + 1), cz = L(D::oXK + 2);
      const double fx = L(D::oUK + 3 * lane), fy = L(D::oUK + 3 * lane + 1), fz = L(D::oUK + 3 * lane + 2);
      const double p0 = d * L(D::oLAMN + 6), p1 = d * L(D::oLAMN + 7), p2 = d * L(D::oLAMN + 8);   // pi
      const double cs = cos(yaw), sn = sin(yaw);
      double vx, vy; vert_local(j, vx, vy);
      const double rvx = cs * vx - sn * vy, rvy = sn * vx + cs * vy;
      const double dvx = -sn * vx - cs * vy, dvy = cs * vx - sn * vy;
      const double rx = px + rvx - cx, ry = py + rvy - cy, rz = pz - cz;
      L(D::oVDV + 3 * lane + 0) = dvx; L(D::oVDV + 3 * lane + 1) = dvy; L(D::oVDV + 3 * lane + 2) = 0.0;
      L(D::oVR + 3 * lane + 0) = rx; L(D::oVR + 3 * lane + 1) = ry; L(D::oVR + 3 * lane + 2) = rz;
      tq[3 * lane + 0] = ry * fz - rz * fy; tq[3 * lane + 1] = rz * fx - rx * fz; tq[3 * lane + 2] = rx * fy - ry * fx;
      yw[3 * lane + 0] = dvy * fz; yw[3 * lane + 1] = -dvx * fz; yw[3 * lane + 2] = dvx * fy - dvy * fx;
      qq[lane] = p0 * (-rvy * fz) + p1 * (rvx * fz) + p2 * (-rvx * fy + rvy * fx);   // pi . ((R''v) x f), R''v = -Rv
      if (lane < 3) L(D::oMISC + 9 + lane) = (lane == 0) ? p0 : (lane == 1) ? p1 : p2;
    }
    sync();
    // (B) sums over the vertices, one lane per result, every lane the same NF independent reads:
    //   lanes 0..5   Fs[f][a]  = sum_j f_j[a]                         -> MISC 0..5
    //   lanes 6..8   tau[a]    = sum_f gamma_f sum_j (r_j x f_j)[a]    -> MISC 6..8
    //   lanes 9..14  Y[f][a]   = sum_j ((R'v_j) x f_j)[a]              -> MISC 21..26
    //   lanes 15,16  Q[f]      = gamma_f sum_j pi.((R''v_j) x f_j)     -> MISC 30, 31
    if (lane < 17) {
      const double gam0 = L(D::oSR + 17), gam1 = L(D::oSR + 18);
      // role by integer selects only (fsel: -1 both feet, else the foot summed)
      const int so = (lane < 6) ? D::oUK + lane % 3 : (lane < 9) ? D::oH0 + (lane - 6)
                   : (lane < 15) ? D::oH0 + 3 * NF + (lane - 9) % 3 : D::oH0 + 6 * NF;
      const int stride = (lane < 15) ? 3 : 1;
      const int fsel = (lane < 6) ? lane / 3 : (lane < 9) ? -1 : (lane < 15) ? (lane - 9) / 3 : lane - 15;
      const bool weighted = (lane >= 6 && lane < 9) || lane >= 15;
      const int dst = (lane < 9) ? lane : (lane < 15) ? 12 + lane : 15 + lane;
      const double w0 = weighted ? gam0 : 1.0, w1 = weighted ? gam1 : 1.0;
      double acc = 0.0;
#pragma unroll
      for (int v = 0; v < NF; ++v) {
        const int f = v / NV;
        const double val = L(so + v * stride);
        acc += (fsel < 0 || fsel == f) ? (f ? w1 : w0) * val : 0.0;
      }
      L(D::oMISC + dst) = acc;
    }
    sync();
    // GH[a][col]: rows 6..8 of [B A] minus identity.  One batch of unconditional reads at clamped indices, then the column
    // type picks its expression (as nested branches over the column type: four divergent paths with an exposed LDS round
    // trip each; same expressions, the skipped terms of the per-foot sums are exact zeros: same bits).
    for (int col = lane; col < NZ; col += WS) {
      const bool is_f = col < 6 * NV, is_x = col >= NU;
      const int s = is_x ? col - NU : 0;
      const bool is_cp = is_x && (s < 3 || (s >= 13 && s < 16) || (s >= 17 && s < 20)), is_yaw = is_x && (s == 12 || s == 16);
      const int v = is_f ? col / 3 : 0, a = is_f ? col % 3 : (is_cp ? ((s < 3) ? s : (s - 13) % 4) : 0), f = v / NV;
      const int fy = (s == 16) ? 1 : 0;
      const double gm0 = L(D::oSR + 17), gm1 = L(D::oSR + 18);
      const double rx = L(D::oVR + 3 * v), ry = L(D::oVR + 3 * v + 1), rz = L(D::oVR + 3 * v + 2);
      const double M0 = L(D::oMISC + 0), M1 = L(D::oMISC + 1), M2 = L(D::oMISC + 2), M3 = L(D::oMISC + 3), M4 = L(D::oMISC + 4),
                   M5 = L(D::oMISC + 5);
      const double Y0 = L(D::oMISC + 21 + 3 * fy), Y1 = L(D::oMISC + 22 + 3 * fy), Y2 = L(D::oMISC + 23 + 3 * fy);
      // column a of skew(.) = (.) x e_a, written branch-free
      const double e0 = (a == 0) ? 1.0 : 0.0, e1 = (a == 1) ? 1.0 : 0.0, e2 = (a == 2) ? 1.0 : 0.0;
      // force component: d*gamma*skew(r)[.][a]
      const double gF = d * (f ? gm1 : gm0);
      const double f0 = gF * (ry * e2 - rz * e1), f1 = gF * (rz * e0 - rx * e2), f2 = gF * (rx * e1 - ry * e0);
      // c_a: +d*sum_f gamma_f sum_j skew(f_j)[.][a];  p_{f,a}: -d*gamma_f sum_j skew(f_j)[.][a]
      const bool use0 = !(s >= 13 && (s >= 17)), use1 = !(s >= 13 && !(s >= 17));
      double F0 = 0, F1 = 0, F2 = 0;
      F0 += use0 ? gm0 * M0 : 0.0; F1 += use0 ? gm0 * M1 : 0.0; F2 += use0 ? gm0 * M2 : 0.0;
      F0 += use1 ? gm1 * M3 : 0.0; F1 += use1 ? gm1 * M4 : 0.0; F2 += use1 ? gm1 * M5 : 0.0;
      const double sg = (s < 3) ? d : -d;
      const double c0 = sg * (F1 * e2 - F2 * e1), c1 = sg * (F2 * e0 - F0 * e2), c2 = sg * (F0 * e1 - F1 * e0);
      // yaw: d*gamma_f sum_j (R'v_j) x f_j
      const double gY = d * (fy ? gm1 : gm0);
      const double y0 = gY * Y0, y1 = gY * Y1, y2 = gY * Y2;
      const double g0 = is_f ? f0 : is_cp ? c0 : is_yaw ? y0 : 0.0;
      const double g1 = is_f ? f1 : is_cp ? c1 : is_yaw ? y1 : 0.0;
      const double g2 = is_f ? f2 : is_cp ? c2 : is_yaw ? y2 : 0.0;
      L(D::oGH + col) = g0; L(D::oGH + NZ + col) = g1; L(D::oGH + 2 * NZ + col) = g2;
    }
    // b = F(x,u) - x_{k+1}: one batch of unconditional reads at clamped indices, then the row picks its expression (as an
    // if / else-if chain over the row: nine divergent paths, an exposed LDS round trip each; same expressions, same bits)
    {
      const int q = (lane < NXA) ? lane : 0;
      const double *x = &L(D::oXK), *u = &L(D::oUK), *sr = &L(D::oSR);
      const int a = (q < 3) ? q : (q < 6) ? q - 3 : (q >= 9 && q < 12) ? q - 9 : 0;     // axis of the CoM rows
      const int ja = (q >= CMPC_NX) ? 3 * (q - CMPC_NX) + 2 : (q == 12) ? 6 * NV + 6 : (q >= 13 && q < 16) ? 6 * NV + q - 13
                   : (q == 16) ? 6 * NV + 7 : (q >= 17 && q < 20) ? 6 * NV + 3 + q - 17 : 0;
      const double own = x[(q < CMPC_NX) ? q : 0], xa = x[a], x3a = x[3 + a], sra = sr[a], sr3a = sr[3 + a];
      const double g17 = sr[17], g18 = sr[18], ua = u[ja];
      const double m_a = L(D::oMISC + a), m_3a = L(D::oMISC + 3 + a), m_tau = L(D::oMISC + 6 + ((q >= 6 && q < 9) ? q - 6 : 0));
      const double xnext = L(D::oXN1 + q);
      const double gsel = (q >= 16) ? g18 : g17;
      const double xn = (q < 3) ? own + d * x3a
                      : (q < 6) ? own + d * (((a == 2) ? -SPD(g) : 0.0) + (g17 * m_a + g18 * m_3a) / m)
                      : (q < 9) ? own + d * m_tau
                      : (q < 12) ? own + d / m * (SPD(k1) * (xa - sra) + x3a - sr3a)
                      : (q < CMPC_NX) ? own + d * (1 - gsel) * ua : ua;
      if (lane < NXA) L(D::oBV + q) = xn - xnext;
    }
    sync();
  }

  // Column list of [B A] for this lane's column (lane < NZ): id, 3 h-rows, 2 specials.
  CMPC_DEV void build_list(const double *gh, double gl, double gr, double m) {
#pragma unroll
    for (int h = 0; h < NH; ++h) column_list(lane + WS * h, lr[h], lg[h], gh, gl, gr, m);
  }
  CMPC_DEV void column_list(const int col, int *r, double *g, const double *gh, double gl, double gr, double m) const {
    const double d = SPD(delta);
    {
#pragma unroll
      for (int n = 0; n < 6; ++n) { r[n] = 0; g[n] = 0.0; }
      if (col >= NZ) return;
      r[1] = 6; r[2] = 7; r[3] = 8;
      g[1] = gh[col]; g[2] = gh[NZ + col]; g[3] = gh[2 * NZ + col];
      if (col < 6 * NV) {
        const int v = col / 3, a = col % 3, f = v / NV;
        r[4] = 3 + a; g[4] = d * (f ? gr : gl) / m;
        if (a == 2) { r[5] = 20 + v; g[5] = 1.0; }
      } else if (col < 6 * NV + 6) {
        const int f = (col - 6 * NV) / 3, a = (col - 6 * NV) % 3;
        r[4] = 13 + 4 * f + a; g[4] = d * (1 - (f ? gr : gl));
      } else if (col < NU) {
        const int f = col - 6 * NV - 6;
        r[4] = 12 + 4 * f; g[4] = d * (1 - (f ? gr : gl));
      } else {
        const int s = col - NU;
        if (s < CMPC_NX) { r[0] = s; g[0] = 1.0; }
        if (s < 3) { r[4] = 9 + s; g[4] = d * SPD(k1) / m; }
        else if (s < 6) { r[4] = s - 3; g[4] = d; r[5] = 9 + s - 3; g[5] = d / m; }
      }
    }
  }

  // ---------------------------------------------------------------------------------------
  // Inequality rows g (<= 0), Lyapunov gradient AL, activity.  reference :193-271
  // ---------------------------------------------------------------------------------------
  CMPC_DEV void stage_ineq(int k, double x0n2) {
    const double m = L(D::oHDR + 20), muf = L(D::oHDR + 21), rl = SPD(relax);
    const double d = SPD(delta), k1 = SPD(k1), k2 = SPD(k2);
    if (k < N && lane < 3) {
      const int a = lane;
      const double *x = &L(D::oXK), *sr = &L(D::oSR);
      const double grav = (a == 2) ? -SPD(g) : 0.0;
      const double V = (sr[17] * L(D::oMISC + a) + sr[18] * L(D::oMISC + 3 + a)) / m;
      const double z1 = x[a] + d * x[3 + a] - sr[a];
      const double z2 = k1 * z1 + x[3 + a] + d * (grav + V) - sr[3 + a];
      const double un = -(k1 + k2) * z2 + k1 * k1 * z1 - grav + sr[6 + a] - x[9 + a] / m;
      L(D::oMISC + 32 + a) = z1; L(D::oMISC + 35 + a) = z2; L(D::oMISC + 38 + a) = un;
      L(D::oMISC + 41 + a) = -2 * k1 * z1 + z2 - k1 * k1 * z2;
      L(D::oMISC + 44 + a) = -2 * k2 * z2 + z1 + (V - un) + (k1 + k2) * z2;
      L(D::oMISC + 47 + a) = V;
      L(D::oRED + a) = -k1 * z1 * z1 - k2 * z2 * z2 + z1 * z2 + z2 * (V - un);
    }
    sync();
    // Lyapunov gradient over z = (u, x): every column is c1 * MISC[i1] + c2 * MISC[i2]; the same two
    // reads for every lane instead of one divergent path per column type
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int col = lane + WS * h, cc = (col < NZ) ? col : 0;
      const bool is_f = cc < 6 * NV, is_x = cc >= NU;
      const int a = cc % 3, f = (cc / 3) / NV, sx = is_x ? cc - NU : 0;
      const int grp = (sx < 3) ? 0 : (sx < 6) ? 1 : ((sx >= 9 && sx < 12) ? 2 : 3);   // c, v, theta, none
      const int ax = (grp == 0) ? sx : (grp == 1) ? sx - 3 : (grp == 2) ? sx - 9 : 0;
      const int i1 = is_f ? 44 + a : ((grp == 2) ? 35 + ax : 41 + ax), i2 = is_f ? 35 + a : 44 + ax;
      const double m1 = L(D::oMISC + i1), m2 = L(D::oMISC + i2), gam = L(D::oSR + 17 + (is_f ? f : 0));
      const double c1 = is_f ? gam * d / m : ((grp == 0) ? 1.0 : (grp == 1) ? d : (grp == 2) ? 1.0 / m : 0.0);
      const double c2 = is_f ? gam / m : ((grp == 0) ? k1 : (grp == 1) ? k1 * d + 1 : 0.0);
      const bool on = (k < N) && (is_f || (is_x && k >= 1 && grp < 3));
      if (col < NZ) L(D::oAL + col) = on ? c1 * m1 + c2 * m2 : 0.0;
    }
    // Inequality rows: one batch of clamped reads per lane, the row type picks its combination
    {
      constexpr int NIH = (NI + WS - 1) / WS;
      const double x2 = L(D::oXK + 2), x6 = L(D::oXK + 6), x7 = L(D::oXK + 7), x8 = L(D::oXK + 8);
      const double h6 = L(D::oHDR + 6), h7 = L(D::oHDR + 7), h8 = L(D::oHDR + 8);   // hw_0 (record header = x_0)
      const double hw0n2 = h6 * h6 + h7 * h7 + h8 * h8;
      (void)x0n2;
      const double red = L(D::oRED) + L(D::oRED + 1) + L(D::oRED + 2);
#pragma unroll
      for (int h = 0; h < NIH; ++h) {
        const int r = lane + WS * h, rc = (r < NI) ? r : 0;
        const bool is_box = rc >= R_BOX && rc < R_FRIC, is_fr = rc >= R_FRIC;
        const int qb = is_box ? rc - R_BOX : 0, fb = qb / 6, ab = (qb % 6) / 2;
        const int qf = is_fr ? rc - R_FRIC : 0, v = qf / 5, t = qf % 5;
        const int f = is_box ? fb : v / NV;
        const double xa = L(D::oXK + 13 + 4 * fb + ab), pa = L(D::oSRP + 9 + 3 * fb + ab);
        const double gs = L(D::oSR + 17 + f), gN = L(D::oHDR + 22 + f);
        const double fx = L(D::oUK + 3 * v), fy = L(D::oUK + 3 * v + 1), fz = L(D::oUK + 3 * v + 2);
        const double gk = (k == N) ? gN : gs;                            // contact flag at node k
        const double bx = (ab == 0) ? SPD(box[0]) : (ab == 1) ? SPD(box[1]) : SPD(box[2]);
        const double g_box = ((qb & 1) ? -1.0 : 1.0) * ((xa - pa) * gk) - bx - rl;
        const double e = (t == 0) ? fx - muf * fz : (t == 1) ? -fx - muf * fz : (t == 2) ? fy - muf * fz
                       : (t == 3) ? -fy - muf * fz : -fz;
        const double g_fr = gs * e - rl;
        const double g_sel = (rc == R_LYAP) ? red - rl : (rc == R_CZ) ? x2 - SPD(cz_max) - rl
                           : (rc == R_HWC) ? x6 * x6 + x7 * x7 + x8 * x8 - hw0n2 - rl : (is_box ? g_box : g_fr);
        const bool act = (rc == R_LYAP) ? (k < N) : (rc == R_CZ) ? (k >= 1 && k < N) : (rc == R_HWC) ? (k == 1)
                       : (is_box ? (k >= 1 && gk != 0.0) : (k < N && gs != 0.0));
        if (r < NI) {
          L(D::oGK + r) = act ? g_sel : 0.0;
          L(D::oW2 + r) = act ? 1.0 : 0.0;     // activity flag (overwritten with 1/s later)
        }
      }
    }
    sync();
  }

  // (Jg' w)[col] for row weights w (zero on inactive rows).
  // (Jg' w)[col] for the three weight vectors of a stage at once (multipliers z, sigma*(g+s), 1/s):
  // the same handful of LDS addresses in each, so one round trip serves all three.
  CMPC_DEV void jgt3(int k, int col, double (&out)[3]) const {
    // Every word any column type needs is read unconditionally at clamped indices (one batch, one wait) and the column
    // type only zeroes the coefficients of what it does not use: as an if / else-if chain over the column type this was
    // four divergent paths with an exposed LDS round trip each (round 4; the sums pick up exact zeros, same bits).
    const double muf = L(D::oHDR + 21);
    const double *w0 = &L(D::oZK), *w1 = &L(D::oW1), *w2 = &L(D::oW2);
    const double alc = L(D::oAL + col);
    double v0 = alc * w0[R_LYAP], v1 = alc * w1[R_LYAP], v2 = alc * w2[R_LYAP];
    const bool is_f = col < 6 * NV, is_x = col >= NU;
    const int vtx = is_f ? col / 3 : 0, a = is_f ? col % 3 : 0, f = vtx / NV;
    const int s = is_x ? col - NU : 0;
    const bool s_cz = is_x && s == 2, s_hw = is_x && s >= 6 && s < 9;
    const bool s_pos = is_x && ((s >= 13 && s < 16) || (s >= 17 && s < 20));
    const int fp = (s >= 17) ? 1 : 0, ap = s_pos ? (s - 13) % 4 : 0;
    const int b = R_FRIC + 5 * vtx, bb = R_BOX + 6 * fp + 2 * ap;
    // all fifteen words of the vertex's friction rows, the height / contraction / box words of the state columns
    const double p0 = w0[b], p1 = w0[b + 1], p2 = w0[b + 2], p3 = w0[b + 3], p4 = w0[b + 4];
    const double q0 = w1[b], q1 = w1[b + 1], q2 = w1[b + 2], q3 = w1[b + 3], q4 = w1[b + 4];
    const double r0 = w2[b], r1 = w2[b + 1], r2 = w2[b + 2], r3 = w2[b + 3], r4 = w2[b + 4];
    const double c0 = w0[R_CZ], c1 = w1[R_CZ], c2 = w2[R_CZ], h0 = w0[R_HWC], h1 = w1[R_HWC], h2 = w2[R_HWC];
    const double xs = L(D::oXK + (s_hw ? s : 6));
    const double bx0 = w0[bb], bx0n = w0[bb + 1], bx1 = w1[bb], bx1n = w1[bb + 1], bx2 = w2[bb], bx2n = w2[bb + 1];
    const double gs = L(D::oSR + 17 + f), gp = gam_k(k, fp);
    if (is_f) {
      const double gg = (k < N) ? gs : 0.0;
      const double ex = (a == 0) ? gg : 0.0, ey = (a == 1) ? gg : 0.0, ez = (a == 2) ? gg : 0.0;
      v0 += ex * (p0 - p1) + ey * (p2 - p3) - ez * (muf * ((p0 + p1) + (p2 + p3)) + p4);
      v1 += ex * (q0 - q1) + ey * (q2 - q3) - ez * (muf * ((q0 + q1) + (q2 + q3)) + q4);
      v2 += ex * (r0 - r1) + ey * (r2 - r3) - ez * (muf * ((r0 + r1) + (r2 + r3)) + r4);
    } else if (s_cz) { v0 += c0; v1 += c1; v2 += c2; }
    else if (s_hw) { const double xx = 2.0 * xs; v0 += xx * h0; v1 += xx * h1; v2 += xx * h2; }
    else if (s_pos) {
      const double gg = (k >= 1) ? gp : 0.0;
      v0 += gg * (bx0 - bx0n); v1 += gg * (bx1 - bx1n); v2 += gg * (bx2 - bx2n);
    }
    out[0] = v0; out[1] = v1; out[2] = v2;
  }

  // Objective gradient entry for column col.  reference :275-353
  // Every lane issues the same handful of LDS reads at clamped per-lane indices and combines them with
  // per-lane coefficients: the column types used to be separate divergent branches, each with its own
  // exposed LDS round trips (and the exp() of the height weight inside one of them).
  CMPC_DEV double cost_grad(int k, int col, double wz) const {
    const bool stage = k < N, is_u = col < NU, is_f = col < 6 * NV;
    const int s = is_u ? 0 : col - NU;                                   // state index
    const int vtx = is_f ? col / 3 : ((s >= CMPC_NX) ? s - CMPC_NX : 0), a = is_f ? col % 3 : 0, f = vtx / NV;
    const bool is_fp = !is_u && s >= CMPC_NX;                             // carried f_z state
    const bool is_yaw = !is_u && (s == 12 || s == 16), is_pos = !is_u && ((s >= 13 && s < 16) || (s >= 17 && s < 20));
    const int ff = is_u ? f : (is_fp ? f : ((s >= 16) ? 1 : 0));          // foot of the column
    // indices of the words this column needs
    const int iu = is_u ? col : 3 * vtx + 2;                              // u word (own input, or f_z of the vertex)
    const int ix = is_u ? CMPC_NX + vtx : s;                              // x word (carried f_z, or own state)
    const int ip = (s < 3) ? s : (is_yaw ? 15 + ff : (is_pos ? 9 + 3 * ff + (s - 13) % 4 : 0));
    const double u_ = L(D::oUK + iu), upx = L(D::oUPX + (is_u ? col : 0)), x_ = L(D::oXK + ix), pr = L(D::oSRP + ip);
    const double fsum = L(D::oMISC + 3 * f + a);
    const double g1 = L(D::oSR + 17 + ff), gm1 = L(D::oSRP + 17 + ff), gN = L(D::oHDR + 22 + ff);
    const double gk = (k == N) ? gN : g1;                                 // contact flag at node k
    double v = 0.0;
    if (is_u) {
      if (stage) {
        v = SPD(prox) * (u_ - upx);
        if (is_f) {
          const double a_ = g1 * g1 / NV, coef = NV * a_ * a_ - 2 * a_;
          const double wa = SPD(w_force) * g1, wb = SPD(w_force) * (1 - g1);
          v += wa * (2 * coef * fsum + 2 * u_) + wb * 2 * u_;
          if (a == 2 && k >= 1) v += 2 * SPD(w_rate) * gm1 * (u_ - x_);
        }
      }
    } else if (k >= 1) {
      const double wpos = (s < 3) ? 2 * ((s == 2) ? wz : SPD(w_cxy)) : ((is_yaw || is_pos) ? 2 * SPD(w_foot) * gk * gk : 0.0);
      v = wpos * (x_ - pr);
      if (s >= 6 && s < 9) v = stage ? 2 * SPD(w_hw) * x_ : 0.0;
      if (is_fp) v = stage ? -2 * SPD(w_rate) * gm1 * (u_ - x_) : 0.0;
    }
    return v;
  }

  // ---------------------------------------------------------------------------------------
  // Row `lane` of the Lagrangian Hessian + barrier terms into the packed lower triangle M.
  // ---------------------------------------------------------------------------------------
  CMPC_DEV void build_H(int k, double reg, double wz) {
#ifdef CMPC_DEBUG_SKIP_H                       // (diagnostic build: G'PG alone in M)
    if constexpr (!D::GT_FIRST) { for (int e = lane; e < D::NTRI; e += WS) L(D::oM + e) = 0.0; }
    sync();
    return;
#endif
#pragma unroll 1
    for (int h_ = 0; h_ < NH; ++h_) build_H_row(k, reg, wz, lane + WS * h_);
  }
  // One pass over the columns, the same instruction stream for every row: the column type (force
  // axis / foot, velocity, state group) is wave-uniform, everything that depends on the row is a
  // per-lane coefficient computed up front, and entries right of the diagonal are redirected to the
  // lane's dump slot instead of being branched around.  No LDS read-modify-write, no zero fill -- except for the single
  // wave of the 4-vertex solver (D::GT_FIRST), where M already holds [B A]' P [B A] and the row ADDS its non-zero
  // columns to it (the old values of a column group come in one batch; the structurally zero columns are not touched).
  CMPC_DEV void build_H_row(int k, double reg, double wz, const int irow) {
    constexpr bool RMW = D::GT_FIRST;
    const bool live = irow < NZ;
    const int i = live ? irow : 0;              // idle lanes shadow row 0 and write only to the dump slot
    const int wlim = live ? i : 0;              // columns j < wlim are written
    double *row = &L(D::oM + tri(i)), *dump = &L(D::oDUMP + (lane & (D::DUMPN - 1)));
    const double m = L(D::oHDR + 20), muf = L(D::oHDR + 21);
    const double *x = &L(D::oXK);
    const double *sig = &L(D::oW0);
    const double *al = &L(D::oAL);
    const bool stage = k < N;
    const double sigL = sig[R_LYAP], zL = stage ? L(D::oZK + R_LYAP) : 0.0;
    const double d = SPD(delta), k1 = SPD(k1);
    const double gam[2] = {L(D::oSR + 17), L(D::oSR + 18)};
    // Lyapunov quadratic-form coefficients hq (4x4 over c, v, theta, V)
    const double a1[4] = {1, d, 0, 0}, a2[4] = {k1, k1 * d + 1, 0, d};
    const double aS[4] = {0, 0, 1.0 / m, 1};
    auto hq = [&](int p, int q) {
      return -2 * k1 * a1[p] * a1[q] + 2 * k1 * a2[p] * a2[q] + (1 - k1 * k1) * (a1[p] * a2[q] + a2[p] * a1[q]) +
             a2[p] * aS[q] + aS[p] * a2[q];
    };
    // ---- row role ----
    const bool is_force = i < 6 * NV, is_state = i >= NU;
    const int s = i - NU;                       // state index (state rows)
    const int vtx_i = is_force ? i / 3 : 0, a_i = i % 3, f_i = vtx_i / NV;
    // Lyapunov role: 0 c, 1 v, 2 theta, 3 V (force), -1 none.  Single-value selects on purpose: hipcc has
    // miscompiled if / else-if chains that assign several variables per branch (compiler note).
#ifdef CMPC_REPRO_IFELSE
    // Round-1 form (several assignments per branch), kept compilable as the reduced repro of the round-1 suspicion
    // "hipcc miscompiles this chain in the 8-vertex kernel": built with -DCMPC_REPRO_IFELSE and run through the GPU
    // parity tests in round 3 (tools/repro_ifelse.sh, profiles/r03_repro_ifelse.txt): see DESIGN.md section 5.
    int ti = -1, ai = 0; double sci = 0.0;
    if (stage) {
      if (is_force) { ti = 3; ai = a_i; sci = gam[f_i] / m; }
      else if (is_state && k >= 1) {
        if (s < 3) { ti = 0; ai = s; sci = 1.0; } else if (s < 6) { ti = 1; ai = s - 3; sci = 1.0; }
        else if (s >= 9 && s < 12) { ti = 2; ai = s - 9; sci = 1.0; }
      }
    }
#else
    const int ts = (s < 3) ? 0 : (s < 6) ? 1 : ((s >= 9 && s < 12) ? 2 : -1);
    const int ti = !stage ? -1 : (is_force ? 3 : ((is_state && k >= 1) ? ts : -1));
    const int ai = is_force ? a_i : ((s < 3) ? s : (s < 6) ? s - 3 : s - 9);
    const double sci = is_force ? gam[f_i] / m : 1.0;
#endif
    const int tic = (ti >= 0) ? ti : 0;
    const double al_i = al[i];
    const double sA = (ti >= 0) ? sigL * al_i : 0.0;
    const double hV = (ti >= 0) ? zL * hq(tic, 3) * sci / m : 0.0;       // against a force column of the same axis
    double zq[3];                                                        // against a c / v / theta column of the same axis
#pragma unroll
    for (int t = 0; t < 3; ++t) zq[t] = (ti >= 0) ? zL * hq(tic, t) * sci : 0.0;
    // dynamics curvature  pi . d2 tau  of this row against a force column (foot f, axis a, vertex offsets dvx, dvy):
    //   gm[f] * (cc0[a] + ccx[a] * dvx + ccy[a] * dvy)
    const double p0 = L(D::oMISC + 9), p1 = L(D::oMISC + 10), p2 = L(D::oMISC + 11);
    const double Spm[3][3] = {{0.0, -p2, p1}, {p2, 0.0, -p0}, {-p1, p0, 0.0}};   // skew(pi)
    const int ctype = (!stage || !is_state) ? 0 : (s < 3) ? 1 : (((s >= 13 && s < 16) || (s >= 17 && s < 20)) ? 2 : ((s == 12 || s == 16) ? 3 : 0));
    const int cfoot = (s >= 16) ? 1 : 0, caxis = (s < 3) ? s : ((s >= 13) ? (s - 13) % 4 : 0);
    const int cax = (caxis < 3) ? caxis : 0;
    double cc0[3], ccx[3], ccy[3], gm[2];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double spc = (cax == 0) ? Spm[a][0] : (cax == 1) ? Spm[a][1] : Spm[a][2];
      cc0[a] = (ctype == 1) ? -spc : (ctype == 2) ? spc : 0.0;
      ccx[a] = (ctype == 3) ? Spm[a][0] : 0.0;
      ccy[a] = (ctype == 3) ? Spm[a][1] : 0.0;
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) gm[f] = (ctype == 1 || (ctype >= 2 && f == cfoot)) ? gam[f] : 0.0;
    // force rows: mean-force coupling (same foot, same axis) and the friction block of the row's vertex
    // Every LDS word any row type needs is read here, unconditionally and at clamped indices, and the row type only
    // picks its combination: written as nested if / else-if over the row type this block and the diagonal at the end
    // were one exposed LDS round trip per branch (8 % of the kernel for a handful of flops; phase timers, round 3).
    const double *sr5 = sig + R_FRIC + 5 * vtx_i;                     // (vtx_i = 0 for rows that are not forces)
    const double s50 = sr5[0], s51 = sr5[1], s52 = sr5[2], s53 = sr5[3], s54 = sr5[4];
    const double gkm0 = gam_km1(0), gkm1 = gam_km1(1);
    const bool frow = stage && is_force;
    const double g1 = f_i ? gam[1] : gam[0];
    const double a_ = g1 * g1 / NV, coef = NV * a_ * a_ - 2 * a_;
    const double wa = SPD(w_force) * g1, wb = SPD(w_force) * (1 - g1);
    const double g2 = g1 * g1;
    const double mean_c = frow ? 2 * wa * coef : 0.0;
    const double d_rate = (a_i == 2 && k >= 1) ? 2 * SPD(w_rate) * (f_i ? gkm1 : gkm0) : 0.0;
    const double d_fric = (a_i == 0) ? g2 * (s50 + s51) : (a_i == 1) ? g2 * (s52 + s53)
                        : g2 * (muf * muf * (s50 + s51 + s52 + s53) + s54);
    double diag = reg;                          // (terms added one by one, in the order of the branchy form: adding 0.0
    diag += frow ? 2 * wa + 2 * wb + mean_c : 0.0;   //  keeps the bits, and with them the round-2 fixtures of the
    diag += frow ? d_rate : 0.0;                //  ill-conditioned end game)
    diag += frow ? d_fric : 0.0;
    const double fr0 = (frow && a_i == 2) ? -g2 * muf * (s50 - s51) : 0.0;
    const double fr1 = (frow && a_i == 2) ? -g2 * muf * (s52 - s53) : 0.0;
    const int j_fr0 = (stage && is_force && a_i == 2) ? 3 * vtx_i : -1;            // columns of fr0 / fr1
    // carried-force rows: rate coupling with the f_z column of the same vertex
    const bool is_fp = is_state && s >= CMPC_NX;
    const int f_fp = is_fp ? (s - CMPC_NX) / NV : 0;
    const double wr_fp = (is_fp && k >= 1 && stage) ? SPD(w_rate) * (f_fp ? gkm1 : gkm0) : 0.0;
    const int j_fp = (is_fp && k >= 1 && stage) ? 3 * (s - CMPC_NX) + 2 : -1;
    // words of the diagonal (state rows)
    const int sc = is_state ? s : 0;
    const bool s_c = is_state && s < 3, s_hw = is_state && s >= 6 && s < 9, s_yaw = is_state && (s == 12 || s == 16);
    const bool s_pos = is_state && ((s >= 13 && s < 16) || (s >= 17 && s < 20));
    const int f_ft = (s >= 16) ? 1 : 0, bidx = R_BOX + 6 * f_ft + 2 * (s_pos ? (s - 13) % 4 : 0);
    const double sCZ = sig[R_CZ], zHW = L(D::oZK + R_HWC), sHW = sig[R_HWC], xs = x[sc];
    const double gk0 = gam_k(k, 0), gk1 = gam_k(k, 1), sb0 = sig[bidx], sb1 = sig[bidx + 1];
    // constant part per (foot, axis) of a force column
    double cst[2][3];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int a = 0; a < 3; ++a)
        cst[f][a] = ((a == ai) ? hV * gam[f] : 0.0) + ((is_force && f == f_i && a == a_i) ? mean_c : 0.0) + gm[f] * cc0[a];
    CMPC_TICK(9);
    // ---- force columns, one foot at a time (the batch of both feet held 8 NV doubles more than the register file
    // has to spare at this point: the only spills of the kernel came from here) ----
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      constexpr int FC = 3 * NV;                // force columns of one foot
      double alf[(NV == 4) ? 14 : 28], dvv[(NV == 4) ? 14 : 28];   // Lyapunov gradient and R'v_j of the foot's columns, one batch each
      if constexpr (NV == 4) lds_read_pair14(alf, dvv, al + f * FC, &L(D::oVDV + f * FC));     // (one wait for both)
      else { lds_read_strided28<1>(alf, al + f * FC); lds_read_strided28<1>(dvv, &L(D::oVDV + f * FC)); }
      const double cf0 = cst[f][0], cf1 = cst[f][1], cf2 = cst[f][2];
      const double gmf = gm[f];
      double oldf[14];
      if constexpr (RMW) { static_assert(!RMW || FC <= 14, "one batch"); lds_read_strided14<1>(oldf, row + FC * f); }
#pragma unroll
      for (int vv = 0; vv < NV; ++vv) {
        const double dvx = dvv[3 * vv], dvy = dvv[3 * vv + 1];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const int j = FC * f + 3 * vv + a;
          double val = sA * alf[3 * vv + a] + ((a == 0) ? cf0 : (a == 1) ? cf1 : cf2) + gmf * (ccx[a] * dvx + ccy[a] * dvy);
          if (a < 2) val += (j == j_fr0 + a) ? ((a == 0) ? fr0 : fr1) : 0.0;
          if (a == 2) val += (j == j_fp) ? -2 * wr_fp : 0.0;
          // (opaque: the row's own entry is rounded before the old value is added, as it is where the Hessian rows are
          // written first and G'PG added to them -- the pair; a product fused into this sum would differ in the last bit)
          if constexpr (RMW) { CMPC_OPAQUE_D(val); val += oldf[(3 * vv + a) % 14]; }
          *((j < wlim) ? row + j : dump) = val;
        }
      }
    }
    CMPC_TICK(15);
    // ---- foot velocity columns: proximal term only ----
    if constexpr (!RMW) {
#pragma unroll
      for (int j = 6 * NV; j < NU; ++j) *((j < wlim) ? row + j : dump) = 0.0;
    }
    // ---- state columns ----
    const bool hwc_row = (k == 1) && is_state && s >= 6 && s < 9;
    const double hwc_c = hwc_row ? 4 * sig[R_HWC] * x[(s >= 6 && s < 9) ? s : 6] : 0.0;
    double als[14];                             // al / x of the 12 leading state columns (c, v, hw, theta)
    double olds[14];
    if constexpr (RMW) lds_read_pair14(als, olds, al + NU, row + NU);      // (one wait for both)
    else lds_read_strided14<1>(als, al + NU);
#pragma unroll
    for (int sj = 0; sj < 12; ++sj) {
      const int j = NU + sj;
      const int tj = (sj < 3) ? 0 : (sj < 6) ? 1 : (sj >= 9) ? 2 : -1, aj = sj % 3;
      double val = 0.0;
      if (tj >= 0) {
        val = sA * als[sj] + ((aj == ai) ? zq[tj] : 0.0);
      } else val = hwc_c * x[sj];
      if constexpr (RMW) { CMPC_OPAQUE_D(val); val += olds[sj]; }
      *((j < wlim) ? row + j : dump) = val;
    }
    if constexpr (!RMW) {
#pragma unroll 4
      for (int j = NU + 12; j < NZ; ++j) *((j < wlim) ? row + j : dump) = 0.0;
    }
    CMPC_TICK(23);
    // ---- diagonal ----
    // Lyapunov part: the (i, i) entry of the rank-1 term and of the multiplier-weighted constant Hessian
    const double q30 = L(D::oMISC + 30 + cfoot);
    if (ti >= 0) diag += sA * al_i + (is_force ? hV * g1 : ((ti == 0) ? zq[0] : (ti == 1) ? zq[1] : zq[2]));
    const bool kp = k >= 1;
    const double gk = f_ft ? gk1 : gk0;
    diag += (kp && s_c) ? 2 * ((s == 2) ? wz : SPD(w_cxy)) : 0.0;
    diag += (kp && s_c && s == 2) ? sCZ : 0.0;
    diag += (kp && s_hw && stage) ? 2 * SPD(w_hw) : 0.0;
    diag += (kp && s_hw && k == 1) ? 2 * zHW + 4 * sHW * xs * xs : 0.0;
    diag += (kp && s_yaw) ? 2 * SPD(w_foot) * gk * gk : 0.0;
    diag += (kp && s_pos) ? 2 * SPD(w_foot) * gk * gk + gk * gk * (sb0 + sb1) : 0.0;
    diag += (kp && is_fp && stage) ? 2 * wr_fp : 0.0;
    diag += (ctype == 3) ? q30 : 0.0;
    diag = is_state ? diag : (stage ? diag + SPD(prox) : reg + 1.0);   // no inputs at the terminal node
    if constexpr (RMW) { CMPC_OPAQUE_D(diag); diag += row[i]; }
    if (live) row[i] = diag;
  }

  // ---------------------------------------------------------------------------------------
  // G'PG of the one-wave 4-vertex solver (round 5): no staging tile, no gathered reads of T.
  //   T phase.  Lane j forms column j of T = P_{k+1} [B A] in NXA registers: its column list (<= 6 entries) times the
  //   list's columns of the packed triangle of P.  Word (q, c) of the triangle sits at tri(max(q, c)) + min(q, c) =
  //   max(tri(c), c + tri(q - 1)) + q: one v_add and one v_max per word for the per-lane part, the `+ q` in the
  //   instruction's immediate (rounds 3-4: both forms computed and selected, four instructions per word).
  //   M phase.  (G'T)[i][j] = sum_n g_i[n] T[r_i[n]][j]: with column j of T in lane j's registers, row i of the product is
  //   the same handful of multiply-adds in every lane -- the rows r_i[n] are compile-time constants of the column
  //   structure of [B A] (the loop over i is unrolled), the coefficients g_i[n] are wave-uniform: the three dense rows'
  //   entries gh[.][i] are broadcast from lane i's own list by v_readlane (scalar operands of the multiply-adds), the
  //   others are five scalars of the stage.  Structural zeros cost nothing.  Row i of the packed M is then one coalesced write (RMW = false: the single wave forms G'PG before the
  //   Hessian rows, which add themselves to it) or read-modify-write (the pair: Hessian rows first) by lanes j <= i.
  //   Rounds 1-4 staged T in LDS in three column parts and accumulated M row-owner from gathered T rows: ~600 LDS and
  //   ~1700 vector-ALU instructions per stage, against ~350 and ~800 here.
  // ---------------------------------------------------------------------------------------
  template <int I> static constexpr bool gt_dense() {               // column I of [B A] has entries in the three dense rows
    constexpr int s = I - NU;
    return I < 6 * NV || (I >= NU && (s < 3 || s == 12 || s == 16 || (s >= 13 && s < 16) || (s >= 17 && s < 20)));
  }
  template <int I> CMPC_DEV double gt_row(const double (&tc)[D::NXA], const double (&cf)[2], const double (&cv)[2],
                                          double ck, double cd, double cdm) const {
    // same entries, same order as column_list(I): identity, the three dense rows, the two specials
    double v = 0.0;
    constexpr bool is_f = I < 6 * NV, is_vel = I >= 6 * NV && I < 6 * NV + 6, is_om = I >= 6 * NV + 6 && I < NU;
    constexpr int s = I - NU;
    if constexpr (I >= NU && s < CMPC_NX) v = tc[s];
    if constexpr (gt_dense<I>()) {
      // (lane I holds the column's dense entries in its list: two v_readlane per word, no LDS.  A force column and a CoM /
      // foot-position column hold a cross product with a unit vector, d gamma (r x e_a) resp. -+d (F x e_a): the entry of
      // row 6 + a is an exact zero -- stage_geometry forms it as c (x 0 - y 0) -- and is left out; the yaw columns are dense)
      // (two waves per instance: lane I may sit in the other wave -- the entries come as uniform reads of GH)
      constexpr int ax = is_f ? I % 3 : (s < 3) ? s : (s == 12 || s == 16) ? -1 : (s - 13) % 4;
      if constexpr (ax != 0) { const double g0 = (NW == 1) ? CMPC_BCAST(lg[0][1], I) : L(D::oGH + I); v = CMPC_FMA(g0, tc[6], v); }
      if constexpr (ax != 1) { const double g1 = (NW == 1) ? CMPC_BCAST(lg[0][2], I) : L(D::oGH + NZ + I); v = CMPC_FMA(g1, tc[7], v); }
      if constexpr (ax != 2) { const double g2 = (NW == 1) ? CMPC_BCAST(lg[0][3], I) : L(D::oGH + 2 * NZ + I); v = CMPC_FMA(g2, tc[8], v); }
    }
    if constexpr (is_f) {
      constexpr int vtx = I / 3, a = I % 3, f = vtx / NV;
      v = CMPC_FMA(cf[f], tc[3 + a], v);
      if constexpr (a == 2) v = v + tc[CMPC_NX + vtx];
    } else if constexpr (is_vel) {
      constexpr int f = (I - 6 * NV) / 3, a = (I - 6 * NV) % 3;
      v = CMPC_FMA(cv[f], tc[13 + 4 * f + a], v);
    } else if constexpr (is_om) {
      constexpr int f = I - 6 * NV - 6;
      v = CMPC_FMA(cv[f], tc[12 + 4 * f], v);
    } else if constexpr (s < 3) {
      v = CMPC_FMA(ck, tc[9 + s], v);
    } else if constexpr (s < 6) {
      v = CMPC_FMA(cd, tc[s - 3], v); v = CMPC_FMA(cdm, tc[9 + s - 3], v);
    }
    return v;
  }
  template <int I0, int... R> CMPC_DEV void gt_vals(std::integer_sequence<int, R...>, double (&val)[sizeof...(R)],
                                                    const double (&tc)[D::NXA], const double (&cf)[2], const double (&cv)[2],
                                                    double ck, double cd, double cdm) const {
    ((val[R] = gt_row<I0 + R>(tc, cf, cv, ck, cd, cdm)), ...);
  }
  static constexpr int GT_ROWS = 12;           // rows of M per batch (RMW: their old values come in one batch of LDS reads)
  static constexpr int GT_FULL = NZ / GT_ROWS, GT_TAIL = NZ % GT_ROWS;      // 60 = 5 x 12;  92 = 7 x 12 + 8
  static_assert(GT_TAIL == 0 || GT_TAIL == 8, "row batches of twelve and one of eight");
  template <bool RMW, int... G> CMPC_DEV void gt_groups(std::integer_sequence<int, G...>, const double (&tc)[D::NXA],
                                                        const double (&cf)[2], const double (&cv)[2], double ck, double cd, double cdm) {
    (gt_rows<RMW, GT_ROWS * G, GT_ROWS>(tc, cf, cv, ck, cd, cdm), ...);
    if constexpr (GT_TAIL != 0) gt_rows<RMW, GT_ROWS * GT_FULL, GT_TAIL>(tc, cf, cv, ck, cd, cdm);
  }
  template <bool RMW, int I0, int SUB, int... Q> CMPC_DEV void gt_subs(std::integer_sequence<int, Q...>, double *val, const double *old,
                                                                       const double (&tc)[D::NXA], const double (&cf)[2],
                                                                       const double (&cv)[2], double ck, double cd, double cdm) {
    (gt_sub<RMW, I0 + SUB * Q, SUB>(val + SUB * Q, old + SUB * Q, tc, cf, cv, ck, cd, cdm), ...);
  }
  template <bool RMW, int I0, int SUB> CMPC_DEV void gt_sub(double *val, const double *old, const double (&tc)[D::NXA], const double (&cf)[2],
                                                            const double (&cv)[2], double ck, double cd, double cdm) {
    double *M = &L(D::oM);
    gt_vals<I0>(std::make_integer_sequence<int, SUB>{}, *reinterpret_cast<double (*)[SUB]>(val), tc, cf, cv, ck, cd, cdm);
    // (opaque: the four rows' chains are formed here, interleaved, instead of each being sunk into its own masked block
    // behind a branch; RMW: also rounded before the add, as where it is stored first -- see build_H_row.  Lanes right of
    // the diagonal write to their dump slot: an address select instead of an exec-mask round trip per row)
#pragma unroll
    for (int r = 0; r < SUB; ++r) CMPC_OPAQUE_D(val[r]);
    double *dump = &L(D::oDUMP + (lane & (D::DUMPN - 1)));
#pragma unroll
    for (int r = 0; r < SUB; ++r) *((lane <= I0 + r) ? &M[tri(I0 + r) + lane] : dump) = RMW ? old[r] + val[r] : val[r];
    CMPC_SCHED_FENCE();
  }
  template <bool RMW, int I0, int CNT> CMPC_DEV void gt_rows(const double (&tc)[D::NXA], const double (&cf)[2], const double (&cv)[2],
                                                             double ck, double cd, double cdm) {
    // (two waves: the second wave's lanes are columns >= 64, which rows above 64 do not have)
    if constexpr (NW == 2 && I0 + CNT <= 64) { if (wv != 0) return; }
    double val[CNT], old[CNT];
    if constexpr (RMW) {                       // M[tri(I0 + r) + lane] (right of the diagonal: a harmless word)
      if constexpr (CNT == 12) lds_read_tri12<I0>(old, &L(D::oM) + lane); else lds_read_tri8<I0>(old, &L(D::oM) + lane);
    }
    // (four rows at a time: the broadcast coefficients of a row are six scalar registers, and the scheduler would otherwise
    // fetch those of the whole batch up front -- the scalar file overflows into vector registers, and those into scratch)
    constexpr int SUB = 4;
    static_assert(CNT % SUB == 0, "whole sub-batches");
    gt_subs<RMW, I0, SUB>(std::make_integer_sequence<int, CNT / SUB>{}, val, old, tc, cf, cv, ck, cd, cdm);
  }
  template <bool RMW> CMPC_DEV void gt_phase(double gl, double gr, double m) {
    static_assert(D::GT && NH == 1 && (NXA == 28 || NXA == 36), "one column of the stage block per lane");
    double tc[NXA];
#pragma unroll
    for (int q = 0; q < NXA; ++q) tc[q] = 0.0;
    if constexpr (D::P_PACKED) {
      const double *Pp = &R(D::oP);
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        const double g = lg[0][n];
        const int c = lr[0][n];
        // word (q, c) of the packed triangle = P + max(tri(c), c + tri(q - 1)) + q
        const cmpc_lds_word wt = cmpc_lds_word_at(Pp, tri(c)), wc = cmpc_lds_word_at(Pp, c);
        if (n >= 1 && n <= 3) {
          // the three dense rows: the same column 6, 7, 8 of P in every lane -- uniform addresses, all of them immediates,
          // the whole column under one wait
          double v[28];
          if (n == 1) lds_read_pcol_all<6>(v, Pp); else if (n == 2) lds_read_pcol_all<7>(v, Pp); else lds_read_pcol_all<8>(v, Pp);
#pragma unroll
          for (int q = 0; q < 28; ++q) { tc[q] = CMPC_FMA(g, v[q], tc[q]); CMPC_OPAQUE_D(tc[q]); }
          CMPC_SCHED_FENCE();
          continue;
        }
#pragma unroll
        for (int q0 = 0; q0 < NXA; q0 += 14) {
          double v[14];
          {
            cmpc_lds_word a[14];
#pragma unroll
            for (int q = 0; q < 14; ++q) {
              const cmpc_lds_word alt = wc + CMPC_LDS_WORDS(tri(q0 + q - 1));
              a[q] = (wt > alt) ? wt : alt;
            }
            if (q0 == 0) lds_read_gather_off14<0>(v, a); else lds_read_gather_off14<14>(v, a);
          }
#pragma unroll
          for (int q = 0; q < 14; ++q) { tc[q0 + q] = CMPC_FMA(g, v[q], tc[q0 + q]); CMPC_OPAQUE_D(tc[q0 + q]); }   // (opaque: the multiply-adds stay here
          CMPC_SCHED_FENCE();                  //  instead of being sunk to the M phase with the loaded words kept -- and spilled -- until then)
        }
      }
    } else {
      // full symmetric P (8-vertex solver): column c is a strided read, eighteen words at a time
      static_assert(D::P_PACKED || NXA == 36, "two batches of eighteen");
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        const double g = lg[0][n];
        const double *pc = &R(D::oP + lr[0][n]);
#pragma unroll
        for (int q0 = 0; q0 < NXA; q0 += 18) {
          double v[18];
          lds_read_strided18<D::PS>(v, pc + q0 * D::PS);
#pragma unroll
          for (int q = 0; q < 18; ++q) { tc[q0 + q] = CMPC_FMA(g, v[q], tc[q0 + q]); CMPC_OPAQUE_D(tc[q0 + q]); }
          CMPC_SCHED_FENCE();
        }
      }
    }
    if constexpr (D::GT_FIRST) sync();         // every lane has read its words of P: the region is free for the late vectors
    CMPC_TICK(10);
    const double d = SPD(delta);
    const double cf[2] = {d * gl / m, d * gr / m}, cv[2] = {d * (1 - gl), d * (1 - gr)};
    const double ck = d * SPD(k1) / m, cd = d, cdm = d / m;
    gt_groups<RMW>(std::make_integer_sequence<int, GT_FULL>{}, tc, cf, cv, ck, cd, cdm);
    sync();
    CMPC_TICK(14);
  }

  // Cholesky of the input block, Ls, Schur complement.  Returns false on a non-positive pivot.
  // Left-looking in panels of 4 columns: the four dot products share the loads of the lane's own row
  // and are independent; the 4x4 panel itself is factorised in registers with readlane broadcasts.
  // One 16-column block of the right-looking blocked Cholesky: (1) the block's columns are factorised
  // for all rows below (left-looking inside the block, 4-column panels in registers with readlane
  // broadcasts), (2) the trailing matrix is updated on the matrix cores,
  //     M[i][j] -= sum_{q in block} L[i][q] L[j][q]   for rows/cols beyond the block,
  // 16x16 tiles of v_mfma_f64_16x16x4; the product is L21 L21', so the B operand of column block cb is
  // the A operand of row block cb.  After the last block the state-state corner of M is
  // P_k = M_xx - Ls Ls' (the Schur complement) without a separate pass.
  // sqrt(p) and 1/sqrt(p) of a pivot by coupled Newton iterations on the hardware estimate: the
  // library sqrt followed by a division is a ~40-instruction dependent chain (special-case scaling
  // and fix-ups), and there are NU of them per stage on the critical path.  p > PIV_MIN here.
  CMPC_DEV static void pivot_sqrt(double p, double &s, double &inv) {
#ifdef CMPC_HOST_EMU
    s = sqrt(p); inv = 1.0 / s;
#else
    const double y = __builtin_amdgcn_rsq(p);
    double g = p * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    const double dd = __builtin_fma(-g, g, p);
    s = __builtin_fma(dd, h, g);
    inv = h + h;
#endif
  }
  template <int C0> CMPC_DEV bool chol_block(double *M, bool &ok, bool trailing) {
    constexpr int W = (NU - C0 < 16) ? NU - C0 : 16;
    static_assert(W % 4 == 0, "block width is a multiple of the 4-column panel");
    // (1) The block's W columns of every row at or below the block live in registers while the block is
    // factorised: pivots and multipliers travel by readlane (the panel rows are lanes C0..C0+W-1 of the
    // first row set), so there is no LDS traffic and no barrier inside the block.
    double blk[NH][W];
    double *dump = &L(D::oDUMP + (lane & (D::DUMPN - 1)));
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int rowi = lane + WS * h;
      const bool own = rowi >= C0 && rowi < NZ;
      const double *ri = M + tri(own ? rowi : C0) + C0;
#pragma unroll
      for (int c = 0; c < W; ++c) blk[h][c] = ri[(own && C0 + c <= rowi) ? c : 0];   // right of the diagonal: a harmless in-row word
    }
    CMPC_TICK(19);
#pragma unroll
    for (int p = 0; p < W / 4; ++p) {
      const int J = C0 + 4 * p;
      double t10, t20, t21, t30, t31, t32, i0, i1, i2, i3;
      if (first_wave()) {                      // the panel itself (first row set): pivots and multipliers by readlane
        double a0 = blk[0][4 * p], a1 = blk[0][4 * p + 1], a2 = blk[0][4 * p + 2], a3 = blk[0][4 * p + 3];
        const double p0 = CMPC_BCAST(a0, J);
        ok = ok && (p0 > piv_min);
        double s0; pivot_sqrt(p0, s0, i0);
        const double l0 = (lane == J) ? s0 : a0 * i0;
        t10 = CMPC_BCAST(l0, J + 1);
        a1 -= l0 * t10;
        const double p1 = CMPC_BCAST(a1, J + 1);
        ok = ok && (p1 > piv_min);
        double s1; pivot_sqrt(p1, s1, i1);
        const double l1 = (lane == J + 1) ? s1 : a1 * i1;
        t20 = CMPC_BCAST(l0, J + 2); t21 = CMPC_BCAST(l1, J + 2);
        a2 -= l0 * t20 + l1 * t21;
        const double p2 = CMPC_BCAST(a2, J + 2);
        ok = ok && (p2 > piv_min);
        double s2; pivot_sqrt(p2, s2, i2);
        const double l2 = (lane == J + 2) ? s2 : a2 * i2;
        t30 = CMPC_BCAST(l0, J + 3); t31 = CMPC_BCAST(l1, J + 3); t32 = CMPC_BCAST(l2, J + 3);
        a3 -= l0 * t30 + l1 * t31 + l2 * t32;
        const double p3 = CMPC_BCAST(a3, J + 3);
        ok = ok && (p3 > piv_min);
        double s3; pivot_sqrt(p3, s3, i3);
        const double l3 = (lane == J + 3) ? s3 : a3 * i3;
        blk[0][4 * p] = l0; blk[0][4 * p + 1] = l1; blk[0][4 * p + 2] = l2; blk[0][4 * p + 3] = l3;
      }
      if constexpr (NW > 1) {
        // the second wave's rows are "further rows": the first wave hands it the panel's reciprocals and
        // multipliers (and the pivot verdict) through eleven LDS words behind the in-block table of the T tile
        double *pk = &L(D::oUB + 64);
        if (lane == 0) {
          pk[0] = ok ? 1.0 : 0.0; pk[1] = i0; pk[2] = i1; pk[3] = i2; pk[4] = i3;
          pk[5] = t10; pk[6] = t20; pk[7] = t21; pk[8] = t30; pk[9] = t31; pk[10] = t32;
        }
        sync();
        if (wv != 0) {
          ok = pk[0] != 0.0; i0 = pk[1]; i1 = pk[2]; i2 = pk[3]; i3 = pk[4];
          t10 = pk[5]; t20 = pk[6]; t21 = pk[7]; t30 = pk[8]; t31 = pk[9]; t32 = pk[10];
          const double l0 = blk[0][4 * p] * i0;
          const double l1 = (blk[0][4 * p + 1] - l0 * t10) * i1;
          const double l2 = (blk[0][4 * p + 2] - (l0 * t20 + l1 * t21)) * i2;
          const double l3 = (blk[0][4 * p + 3] - (l0 * t30 + l1 * t31 + l2 * t32)) * i3;
          blk[0][4 * p] = l0; blk[0][4 * p + 1] = l1; blk[0][4 * p + 2] = l2; blk[0][4 * p + 3] = l3;
        }
      }
#pragma unroll
      for (int h = 1; h < NH; ++h) {           // further row sets reuse the broadcast multipliers
        const double l0 = blk[h][4 * p] * i0;
        const double l1 = (blk[h][4 * p + 1] - l0 * t10) * i1;
        const double l2 = (blk[h][4 * p + 2] - (l0 * t20 + l1 * t21)) * i2;
        const double l3 = (blk[h][4 * p + 3] - (l0 * t30 + l1 * t31 + l2 * t32)) * i3;
        blk[h][4 * p] = l0; blk[h][4 * p + 1] = l1; blk[h][4 * p + 2] = l2; blk[h][4 * p + 3] = l3;
      }
      if (!ok) return false;                   // pivots are uniform over the instance's lanes
      // right-looking inside the block: the remaining block columns of every row.  The multipliers of
      // the block's later rows are handed over through a small LDS table (the T tile is dead here) and
      // read back at wave-uniform addresses: one ds_read per pair of doubles on the LDS port instead of
      // four v_readlane on the vector ALU, which is what this phase is bound by.
      if (4 * p + 4 < W) {
        double *ub = &L(D::oUB);
        {
          const int rr = lane - C0;            // row inside the block
          const bool src = rr >= 4 * p + 4 && rr < W;
          double *w = src ? ub + 4 * rr : dump;
          w[0] = blk[0][4 * p];
          *(src ? w + 1 : dump) = blk[0][4 * p + 1];
          *(src ? w + 2 : dump) = blk[0][4 * p + 2];
          *(src ? w + 3 : dump) = blk[0][4 * p + 3];
        }
        sync();
#pragma unroll
        for (int cc = 4 * p + 4; cc < W; ++cc) {
          const double u0 = ub[4 * cc], u1 = ub[4 * cc + 1], u2 = ub[4 * cc + 2], u3 = ub[4 * cc + 3];
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            double xv = blk[h][cc];
            xv = __builtin_fma(-blk[h][4 * p], u0, xv); xv = __builtin_fma(-blk[h][4 * p + 1], u1, xv);
            xv = __builtin_fma(-blk[h][4 * p + 2], u2, xv); xv = __builtin_fma(-blk[h][4 * p + 3], u3, xv);
            blk[h][cc] = xv;
          }
        }
      }
    }
    CMPC_TICK(20);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int rowi = lane + WS * h;
      const bool own = rowi >= C0 && rowi < NZ;
      double *wi = M + tri(own ? rowi : C0) + C0;
#pragma unroll
      for (int c = 0; c < W; ++c) *((own && C0 + c <= rowi) ? wi + c : dump) = blk[h][c];
    }
    sync();
    CMPC_TICK(21);
    if (!trailing) return true;
    constexpr int R0 = C0 + W;                  // first trailing row / column
    constexpr int NBR = (NZ - R0 + 15) / 16;
    const int r16 = lane & 15, kq = (lane & 63) >> 4;
    // two waves: the 16 x 16 tiles of the trailing triangle alternate between them
    auto mine = [&](int rb, int cb) -> bool { return NW == 1 || ((rb * (rb + 1) / 2 + cb) & 1) == wv; };
    cmpc_v4d acc[NBR][NBR];
#pragma unroll
    for (int rb = 0; rb < NBR; ++rb)
#pragma unroll
      for (int cb = 0; cb <= rb; ++cb) acc[rb][cb] = cmpc_v4d{0.0, 0.0, 0.0, 0.0};
    const double *rowp[NBR];
    bool valid[NBR];
#pragma unroll
    for (int rb = 0; rb < NBR; ++rb) {
      const int row = R0 + 16 * rb + r16;
      valid[rb] = row < NZ;
      rowp[rb] = M + tri(valid[rb] ? row : R0) + C0 + kq;
    }
#pragma unroll
    for (int ks = 0; ks < W / 4; ++ks) {
      double a[NBR];
#pragma unroll
      for (int rb = 0; rb < NBR; ++rb) { const double v = rowp[rb][4 * ks]; a[rb] = valid[rb] ? v : 0.0; }
#pragma unroll
      for (int rb = 0; rb < NBR; ++rb)
#pragma unroll
        for (int cb = 0; cb <= rb; ++cb)
          if (mine(rb, cb)) acc[rb][cb] = CMPC_MFMA_F64(a[rb], a[cb], acc[rb][cb]);
    }
    CMPC_TICK(22);
    double *pm[NBR][NBR][4];
    double old[NBR][NBR][4];
#pragma unroll
    for (int rb = 0; rb < NBR; ++rb)
#pragma unroll
      for (int cb = 0; cb <= rb; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = R0 + 16 * rb + kq + 4 * r, j = R0 + 16 * cb + r16;   // D row / column of this component
          pm[rb][cb][r] = (i < NZ && j <= i && mine(rb, cb)) ? M + tri(i) + j : dump;
          old[rb][cb][r] = *pm[rb][cb][r];
        }
    if constexpr (R0 == NU) {
      // last block: the trailing matrix is the Schur complement P_k = M_xx - Ls Ls'; it goes straight to
      // the full symmetric P array (both triangles) that the next stage reads, not back into M
#pragma unroll
      for (int rb = 0; rb < NBR; ++rb)
#pragma unroll
        for (int cb = 0; cb <= rb; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * rb + kq + 4 * r, j = 16 * cb + r16;
            const bool in = (i < NXA && j <= i && mine(rb, cb));
            const double v = old[rb][cb][r] - acc[rb][cb][r];
            if constexpr (D::P_PACKED) {
              *(in ? &R(D::oP + tri(i) + j) : dump) = v;
            } else {
              *(in ? &R(D::oP + i * D::PS + j) : dump) = v;
              *(in ? &R(D::oP + j * D::PS + i) : dump) = v;
            }
            *pm[rb][cb][r] = v;                 // and into the packed image the factor store copies out
          }
    } else {
#pragma unroll
      for (int rb = 0; rb < NBR; ++rb)
#pragma unroll
        for (int cb = 0; cb <= rb; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) *pm[rb][cb][r] = old[rb][cb][r] - acc[rb][cb][r];
    }
    sync();
    return true;
  }

  // Cholesky of the input block, Ls and the cost-to-go P_k.  Returns false on a non-positive pivot.
  CMPC_DEV bool factor_stage(int k) {
    double *M = &L(D::oM);
    static_assert(NU % 4 == 0, "panel width 4");
    static_assert(NU <= 64, "panel rows must live in the first row set");
    bool ok = true;
    // the last block's trailing update only produces P_k, which stage 0 does not need (x_0 is data)
    if (!chol_block<0>(M, ok, true)) return false;
    if constexpr (NU > 16) { if (!chol_block<16>(M, ok, (NU > 32) || k > 0)) return false; }
    if constexpr (NU > 32) { if (!chol_block<32>(M, ok, (NU > 48) || k > 0)) return false; }
    if constexpr (NU > 48) { if (!chol_block<48>(M, ok, k > 0)) return false; }
    CMPC_TICK(8);
    return true;
  }

  // Factor blocks of stage k to the global slab: the packed triangle M (Lambda, Ls, P_k: see Dims::gM) goes out
  // as it stands, 16 bytes per lane and instruction, no index arithmetic.
  CMPC_DEV void store_factors(int k) {
    const GArr st = stage(k);
    static_assert(D::NTRI % 2 == 0 && D::oM % 2 == 0 && D::STAGE % 2 == 0, "whole 16-byte pairs");
    constexpr int NPAIR = D::NTRI / 2, CH = 16;            // pairs; passes per batch of loads
    const cmpc_v2d *src = reinterpret_cast<const cmpc_v2d *>(&L(D::oM));
#pragma unroll 1
    for (int q0 = 0; q0 * WS < NPAIR; q0 += CH) {
      cmpc_v2d v[CH];
#pragma unroll
      for (int q = 0; q < CH; ++q) {             // all LDS reads first; the tail is clamped (duplicate stores of the last pair)
        const int e = lane + WS * (q0 + q);
        v[q] = src[(e < NPAIR) ? e : NPAIR - 1];
      }
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int e = lane + WS * (q0 + q);
        if (WS * (q0 + q) < NPAIR) st.pair(D::gM / 2 + ((e < NPAIR) ? e : NPAIR - 1)) = v[q];
      }
    }
  }

  // Backward vector recursion of stage k, run inside the matrix sweep while L and Ls are still in
  // LDS (packed rows of M).  The barrier value of the step is only chosen after the whole sweep, but the
  // recursion is linear in the gradient h = h(mu_sweep) + (mu - mu_sweep) * h1, so both parts are propagated:
  //   l = l0 + (mu - mu_sweep) * l1,   p = p0 + (mu - mu_sweep) * p1.
  // The split is taken AROUND the sweep's barrier value on purpose: near a solution grad f + Jg' (mu/s) is a
  // difference of O(|z|) terms that cancels to O(KKT error); propagating "h without the barrier term" and
  // "the barrier term" separately and adding them afterwards leaves a noise floor of eps * |z| * cond on the
  // step (seen as a dual residual that stops at ~1e-6 or grows).  With this split the correction vanishes
  // whenever the barrier value is unchanged, which is every iteration of the end game.
  // Entry: XN1 = p0_{k+1} + P_{k+1} b,  PC1 = p1_{k+1},  H0/H1 = gradient parts of this stage.
  CMPC_DEV void backward_vectors(int k) {
    const GArr st = stage(k);
    const double *M = &L(D::oM);
    // m = h + [B A]'(.)   (scratch: TV for the mu^0 part, AL for the mu^1 part; both dead here)
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int col = lane + WS * h;
      if (col >= NZ) continue;
      double a0 = L(D::oH0 + col), a1 = L(D::oH1 + col);
      if constexpr (D::GT_FIRST) {
        // (the single wave added [B A]' (p0 + P b) and [B A]' p1 when it had the lists in registers: eval_stage)
      } else if constexpr (D::GT) {
        a0 += list_dot(h, &L(D::oXN1)); a1 += list_dot(h, &R(D::oPC1));
      } else {
#pragma unroll
        for (int n = 0; n < 6; ++n) {
          a0 += lg[h][n] * L(D::oXN1 + lr[h][n]);
          a1 += lg[h][n] * R(D::oPC1 + lr[h][n]);
        }
      }
      L(D::oTV + col) = a0; L(D::oTV2 + col) = a1;
    }
    sync();
    CMPC_TICK(2);
    static_assert(NU % 2 == 0 && (NU / 2 == 16 || NU / 2 == 28), "half rows match the LDS batch-read helpers");
    constexpr int HB = NU / 2;
    double lf0, lf1;                           // l of this lane's row (lanes < NU)
    if (first_wave()) {                        // l = L^-1 m_u for both right-hand sides at once
      const int li = (lane < NU) ? lane : NU - 1;
      double m0 = L(D::oTV + li), m1 = L(D::oTV2 + li);
      const double dinv = 1.0 / M[tri(li) + li];
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {         // own row of Lambda, half a row of registers at a time
        double rr[HB];                         // (words right of the diagonal are read but never used)
        if constexpr (HB == 16) lds_read_strided16<1>(rr, M + tri(li) + hb * HB);
        else lds_read_strided28<1>(rr, M + tri(li) + hb * HB);
#pragma unroll
        for (int jj = 0; jj < HB; ++jj) {
          const int j = hb * HB + jj;
          const double l0j = CMPC_BCAST(m0 * dinv, j), l1j = CMPC_BCAST(m1 * dinv, j);
          if (lane > j) { m0 -= rr[jj] * l0j; m1 -= rr[jj] * l1j; }
        }
      }
      lf0 = m0 * dinv; lf1 = m1 * dinv;
      if (lane < NU) {
        st[D::gL + lane] = lf0; st[D::gL1 + lane] = lf1;
        L(D::oTV + lane) = lf0; L(D::oTV2 + lane) = lf1;   // handed to the next step at wave-uniform LDS addresses
      }
    }
    sync();
    CMPC_TICK(3);
    if (first_wave()) {                        // p = m_x - Ls l
      const int lx = (lane < NXA) ? lane : 0;
      const double *l0 = &L(D::oTV), *l1 = &L(D::oTV2);
      double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        double ls[HB];
        if constexpr (HB == 16) lds_read_strided16<1>(ls, M + tri(NU + lx) + hb * HB);
        else lds_read_strided28<1>(ls, M + tri(NU + lx) + hb * HB);
        if constexpr (HB == 16) {
#pragma unroll
          for (int q = 0; q < HB; q += 2) {
            const int j = hb * HB + q;
            a0 += ls[q] * l0[j]; a1 += ls[q + 1] * l0[j + 1];
            b0 += ls[q] * l1[j]; b1 += ls[q + 1] * l1[j + 1];
          }
        } else {
          double v0[HB], v1[HB];
          lds_read_strided28<1>(v0, l0 + hb * HB); lds_read_strided28<1>(v1, l1 + hb * HB);
#pragma unroll
          for (int q = 0; q < HB; q += 2) {
            a0 += ls[q] * v0[q]; a1 += ls[q + 1] * v0[q + 1];
            b0 += ls[q] * v1[q]; b1 += ls[q + 1] * v1[q + 1];
          }
        }
      }
      if (lane < NXA) {
        const double p0 = L(D::oTV + NU + lane) - (a0 + a1), p1 = L(D::oTV2 + NU + lane) - (b0 + b1);
        R(D::oPC + lane) = p0; R(D::oPC1 + lane) = p1;
        st[D::gPV + lane] = p0; st[D::gPV1 + lane] = p1;
      }
    }
    sync();
  }

  struct Err { double e_d, e_p, e_c, e_cmu, sum_mult; int n_mult; };

  // One stage of the matrix sweep in two halves.  eval_stage: everything that does not need the cost-to-go of stage k + 1
  // (loads, geometry, inequality rows, barrier weights, Hessian rows, gradient; KKT error measures).  riccati_stage: the
  // part that does (P b, G'PG, factorisation, backward vectors, factor store); false on wrong inertia.  A single wave runs
  // them back to back; the pipelined pair runs eval_stage(k - 1) beside riccati_stage(k) on two LDS images.
  // P b for the vector sweep: XN1 = p0_{k+1} + P_{k+1} b_k (lanes < NXA), b_k to the slab
  CMPC_DEV void form_Pb(const GArr st) {
    if (lane < NXA) {
      const double *bv = &L(D::oBV);
      double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
      if constexpr (D::P_PACKED) {
        // row `lane` of the packed triangle = its column `lane`: word (q, lane) at max(tri(lane), lane + tri(q - 1)) + q
        // (gt_phase), two batches of fourteen
        static_assert(!D::P_PACKED || NXA == 28, "two batches of fourteen");
        double pr[28];
        {
          const cmpc_lds_word wt = cmpc_lds_word_at(&R(D::oP), tri(lane)), wc = cmpc_lds_word_at(&R(D::oP), lane);
#pragma unroll
          for (int q0 = 0; q0 < NXA; q0 += 14) {
            cmpc_lds_word pa[14];
            double v[14];
#pragma unroll
            for (int q = 0; q < 14; ++q) {
              const cmpc_lds_word alt = wc + CMPC_LDS_WORDS(tri(q0 + q - 1));
              pa[q] = (wt > alt) ? wt : alt;
            }
            if (q0 == 0) lds_read_gather_off14<0>(v, pa); else lds_read_gather_off14<14>(v, pa);
#pragma unroll
            for (int q = 0; q < 14; ++q) pr[q0 + q] = v[q];
          }
        }
#pragma unroll
        for (int q = 0; q < NXA; q += 4) {
          b0 += pr[q] * bv[q]; b1 += pr[q + 1] * bv[q + 1]; b2 += pr[q + 2] * bv[q + 2]; b3 += pr[q + 3] * bv[q + 3];
        }
      } else {
        const double *pr = &R(D::oP + lane * D::PS);
#pragma unroll
        for (int q = 0; q < NXA; q += 4) {
          b0 += pr[q] * bv[q]; b1 += pr[q + 1] * bv[q + 1]; b2 += pr[q + 2] * bv[q + 2]; b3 += pr[q + 3] * bv[q + 3];
        }
      }
      const double a = (b0 + b1) + (b2 + b3);
      L(D::oXN1 + lane) = R(D::oPC + lane) + a;       // v0 = p0_{k+1} + P_{k+1} b
      st[D::gB + lane] = L(D::oBV + lane);
    }
  }
  // ([B A]' v)[col] for this lane's column: the list's entries summed from zero, in list order
  CMPC_DEV double list_dot(int h, const double *v) const {
    double a = 0.0;
#pragma unroll
    for (int n = 0; n < 6; ++n) a = CMPC_FMA(lg[h][n], v[lr[h][n]], a);
    // (opaque: where the caller adds this to something in the same basic block, hipcc's aggressive FMA fusion restarts
    // the chain at that something -- fadd(x, fma(.., fmul)) -> fma(.., fma(.., x)) -- and the single wave, which parks the
    // sum in LDS first, would differ from the pair in the last bit)
    CMPC_OPAQUE_D(a);
    return a;
  }

  CMPC_DEV void eval_stage(int k, double mu, double reg, double wz, double x0n2, Err &er, bool init) {
    CMPC_RELANE(lane); CMPC_OPAQUE(lane);
    load_stage(k);
    CMPC_TICK(24);
    const GArr st = stage(k);
    if (k < N) {
      stage_geometry(k);
    } else {
      for (int c = lane; c < 3 * NZ; c += WS) L(D::oGH + c) = 0.0;
      if (lane < NXA) L(D::oBV + lane) = 0.0;
      sync();
    }
    CMPC_TICK(11);
    double gl_dot[NH];                         // ([B A]' lam_{k+1})[col] of the dual residual
    if constexpr (D::GT_FIRST) {
      // The single wave forms G'PG here, before anything else of the stage needs LDS: GH, BV (in the dead M region) and
      // P_{k+1} are consumed, after which the inequality rows' vectors take P's place and M holds [B A]' P [B A].  What the
      // rest of the stage needs of the column lists is formed now, while they are in registers: [B A]' applied to
      // lam_{k+1} (dual residual), to p0_{k+1} + P b and to p1_{k+1} (backward vectors; parked in H0 / H1, which the
      // gradient later adds itself to).
      if (k < N && lane < NXA) er.e_p = fmax(er.e_p, fabs(L(D::oBV + lane)));
      build_list(&L(D::oGH), (k < N) ? L(D::oSR + 17) : 0.0, (k < N) ? L(D::oSR + 18) : 0.0, L(D::oHDR + 20));
      if (lane < NZ) { st[D::gGH + lane] = lg[0][1]; st[D::gGH + D::GHS + lane] = lg[0][2]; st[D::gGH + 2 * D::GHS + lane] = lg[0][3]; }
      if (k < N) {
        form_Pb(st);
        sync();
        CMPC_TICK(13);
        gl_dot[0] = list_dot(0, &L(D::oLAMN));
        if (lane < NZ) { L(D::oH0 + lane) = list_dot(0, &L(D::oXN1)); L(D::oH1 + lane) = list_dot(0, &R(D::oPC1)); }
        gt_phase<false>(L(D::oSR + 17), L(D::oSR + 18), L(D::oHDR + 20));
      } else {
        gl_dot[0] = 0.0;
        for (int e = lane; e < D::NTRI; e += WS) L(D::oM + e) = 0.0;    // terminal node: no cost-to-go; the Hessian rows add to zero
        sync();
      }
      CMPC_RELANE(lane); CMPC_OPAQUE(lane);
    }
    stage_ineq(k, x0n2);
    CMPC_TICK(12);
    // barrier weights (W2 holds the activity flag on entry); on the very first sweep the slacks and
    // multipliers are created here: s = max(-g, 1e-2), z = mu / s
    for (int r = lane; r < NI; r += WS) {
      const bool act = L(D::oW2 + r) != 0.0;
      const double g = L(D::oGK + r);
      double s = L(D::oSK + r), z = L(D::oZK + r);
      if (init) {
        // rows carried over from the solver state; the cold rule for the rest (also: rows a contact switch has
        // just activated)
        // (initial_point left the state's slacks and multipliers in the iterate arrays, zeros on a cold start)
        const bool carry = act && s > 0.0 && z > 0.0;
        s = carry ? s : (act ? fmax(-g, fmin(1e-2, sqrt(mu))) : 1.0);
        z = carry ? z : (act ? mu / s : 0.0);
        L(D::oZK + r) = z;
        gsl[k * NI + r] = s; gz[k * NI + r] = z;
      }
      if (act) {
        const double sg = z / s;
        L(D::oW0 + r) = sg; L(D::oW1 + r) = sg * (g + s); L(D::oW2 + r) = 1.0 / s;
        er.e_p = fmax(er.e_p, fabs(g + s));
        er.e_c = fmax(er.e_c, fabs(s * z)); er.e_cmu = fmax(er.e_cmu, fabs(s * z - mu));
        er.sum_mult += fabs(z); er.n_mult += 1;
      } else {
        L(D::oW0 + r) = 0.0; L(D::oW1 + r) = 0.0; L(D::oW2 + r) = 0.0; L(D::oZK + r) = 0.0;
      }
    }
    if constexpr (!D::GT_FIRST) { if (k < N && lane < NXA) er.e_p = fmax(er.e_p, fabs(L(D::oBV + lane))); }
    if (k >= 1 && lane < NXA) { er.sum_mult += fabs(L(D::oLAMK + lane)); er.n_mult += 1; }
    sync();
    CMPC_TICK(25);
    if constexpr (!D::GT_FIRST) {
      // Hessian rows first: the column lists of [B A] (18 registers per lane) are not live across them
      CMPC_RELANE(lane); CMPC_OPAQUE(lane);
      build_H(k, reg, wz);
      CMPC_TICK(1);
      CMPC_RELANE(lane); CMPC_OPAQUE(lane);
      build_list(&L(D::oGH), (k < N) ? L(D::oSR + 17) : 0.0, (k < N) ? L(D::oSR + 18) : 0.0, L(D::oHDR + 20));
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int col = lane + WS * h;
      if (col >= NZ) continue;
      const double ho = cost_grad(k, col, wz);
      double jw[3];
      jgt3(k, col, jw);
      double r = ho + jw[0];
      if constexpr (D::GT) {                   // (the list's part summed on its own: the single wave formed it before G'PG)
        if constexpr (!D::GT_FIRST) gl_dot[h] = (k < N) ? list_dot(h, &L(D::oLAMN)) : 0.0;
        r += gl_dot[h];
      } else {
        if (k < N) for (int n = 0; n < 6; ++n) r += lg[h][n] * L(D::oLAMN + lr[h][n]);
      }
      if (col >= NU) r -= L(D::oLAMK + col - NU);
      const bool is_var = (col < NU) ? (k < N) : (k >= 1);
      if (is_var) er.e_d = fmax(er.e_d, fabs(r));
      double h0 = ho + jw[1] + mu * jw[2];     // gradient at the sweep's barrier value (see backward_vectors)
      double h1 = jw[2];
      CMPC_OPAQUE_D(h0); CMPC_OPAQUE_D(h1);    // (rounded here: see list_dot)
      if constexpr (D::GT_FIRST) {             // m = h + [B A]' (.) of the backward vectors: the list's part is waiting there
        if (k < N) { L(D::oH0 + col) = h0 + L(D::oH0 + col); L(D::oH1 + col) = h1 + L(D::oH1 + col); }
        else { L(D::oH0 + col) = h0; L(D::oH1 + col) = h1; }
      } else {
        L(D::oH0 + col) = h0; L(D::oH1 + col) = h1;
      }
      st[D::gAL + col] = L(D::oAL + col);
    }
    CMPC_TICK(26);
    if constexpr (!D::GT_FIRST) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int c = lane + WS * h;
        if (c < NZ) { st[D::gGH + c] = L(D::oGH + c); st[D::gGH + D::GHS + c] = L(D::oGH + NZ + c); st[D::gGH + 2 * D::GHS + c] = L(D::oGH + 2 * NZ + c); }
      }
    }
    for (int r = lane; r < NI; r += WS) st[D::gG + r] = L(D::oGK + r);
    CMPC_TICK(0);
    if constexpr (D::GT_FIRST) {
      // the Hessian rows add themselves to [B A]' P [B A] (read-modify-write of the row's non-zero columns)
      CMPC_RELANE(lane); CMPC_OPAQUE(lane);
      build_H(k, reg, wz);
      CMPC_TICK(1);
    }
    sync();
  }
  CMPC_DEV bool riccati_stage(int k) {
    const GArr st = stage(k);
    if constexpr (PIPE) {                      // (the evaluating wave built its own copy of the column lists)
      CMPC_RELANE(lane); CMPC_OPAQUE(lane);
      build_list(&L(D::oGH), (k < N) ? L(D::oSR + 17) : 0.0, (k < N) ? L(D::oSR + 18) : 0.0, L(D::oHDR + 20));
    }
    if (k < N) {
      if constexpr (!D::GT_FIRST) {
        // Pb = P_{k+1} b  (needed by the vector sweep), then M += G'PG  (PIPE: P b is the other wave's, pair_vectors)
        if constexpr (!PIPE) form_Pb(st);
        sync();
        CMPC_TICK(13);
        CMPC_RELANE(lane); CMPC_OPAQUE(lane);
        gt_phase<true>(L(D::oSR + 17), L(D::oSR + 18), L(D::oHDR + 20));
        CMPC_TICK(14);
      }
      CMPC_RELANE(lane); CMPC_OPAQUE(lane);
#ifdef CMPC_DEBUG_PREFACTOR                    // (diagnostic build, tools/slab_diff.py: stage 0's block as it stands before the factorisation)
      if (k == 0) { store_factors(k); sync(); return true; }
#endif
      if (!factor_stage(k)) return false;
      CMPC_TICK(8);
      if constexpr (!PIPE) backward_vectors(k);    // (PIPE: the vector recursion follows one step behind, on the other wave)
      CMPC_TICK(5);
    } else {
      if (lane < NXA) {                      // terminal cost-to-go gradient p_N = h_N (x part), split in mu
        const double p0 = L(D::oH0 + NU + lane), p1 = L(D::oH1 + NU + lane);
        R(D::oPC + lane) = p0; R(D::oPC1 + lane) = p1;
        st[D::gPV + lane] = p0; st[D::gPV1 + lane] = p1;
      }
      for (int e = lane; e < NXA * NXA; e += WS) {
        const int i = e / NXA, c = e % NXA;
        const int hi = (i > c) ? i : c, lo = (i > c) ? c : i;
        if constexpr (D::P_PACKED) { if (c <= i) R(D::oP + tri(i) + c) = L(D::oM + tri(NU + i) + NU + c); }
        else R(D::oP + i * D::PS + c) = L(D::oM + tri(NU + hi) + NU + lo);
      }
      sync();
    }
    CMPC_RELANE(lane); CMPC_OPAQUE(lane);
    if constexpr (!PIPE) store_factors(k);     // (PIPE: the other wave copies the image out, after its vector recursion)
    sync();
    CMPC_TICK(4);
    return true;
  }

  // PIPE, wave 1: the backward vector recursion of stage k < N and the factor store, one step behind the factorisation.  The matrix recursion
  // (P_k) does not depend on it, so the Riccati wave goes straight on to stage k - 1; L, Ls (packed M), the gradient
  // parts and P_{k+1} b (left in XN1 by the Riccati wave) stand in stage k's image until this wave evaluates stage
  // k - 2 into it, which it does right after this.  Same expressions as the single wave's riccati_stage.
  CMPC_DEV void pair_vectors(int k) {
    CMPC_RELANE(lane); CMPC_OPAQUE(lane);
    if (k < N) {
      build_list(&L(D::oGH), L(D::oSR + 17), L(D::oSR + 18), L(D::oHDR + 20));
      if (lane < NXA) L(D::oXN1 + lane) = R(D::oPC + lane) + L(D::oXN1 + lane);   // v0 = p0_{k+1} + P_{k+1} b
      sync();
      backward_vectors(k);
    }
    store_factors(k);                          // the factor image of stage k (terminal node: P_N) to the slab
    if (k >= 1) {
      // P_k b_{k-1} for the stage the Riccati wave is working on: P_k from this image's packed M (rows NU + i hold
      // [Ls row i | P_k row i up to the diagonal]; the same words the single wave reads from its P copy), b from the other
      // image, where stage k - 1 was evaluated; the product is left in that image's XN1 for this wave's next step.
      static_assert(!PIPE || (D::P_PACKED && NXA == 28), "two batches of fourteen");
      double *other = ldsR + ((k - 1) & 1) * D::LDS_DOUBLES;
      if (lane < NXA) {
        const double *bv = other + D::oBV;
        double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
        double pr[28];
        {
          // word (q, lane) of P_k inside the packed image: row NU + max(q, lane), column NU + min(q, lane), i.e.
          // max(f(lane) + q, f(q) + lane) with f(i) = tri(NU + i) + NU -- the max-address form of gt_phase, two batches
          const cmpc_lds_word wt = cmpc_lds_word_at(&L(D::oM), tri(NU + lane) + NU), wc = cmpc_lds_word_at(&L(D::oM), lane);
#pragma unroll
          for (int q0 = 0; q0 < NXA; q0 += 14) {
            cmpc_lds_word pa[14];
            double v[14];
#pragma unroll
            for (int q = 0; q < 14; ++q) {
              const cmpc_lds_word alt = wc + CMPC_LDS_WORDS(tri(NU + q0 + q) + NU - (q0 + q));
              pa[q] = (wt > alt) ? wt : alt;
            }
            if (q0 == 0) lds_read_gather_off14<0>(v, pa); else lds_read_gather_off14<14>(v, pa);
#pragma unroll
            for (int q = 0; q < 14; ++q) pr[q0 + q] = v[q];
          }
        }
#pragma unroll
        for (int q = 0; q < NXA; q += 4) {
          b0 += pr[q] * bv[q]; b1 += pr[q + 1] * bv[q + 1]; b2 += pr[q + 2] * bv[q + 2]; b3 += pr[q + 3] * bv[q + 3];
        }
        other[D::oXN1 + lane] = (b0 + b1) + (b2 + b3);
        stage(k - 1)[D::gB + lane] = bv[lane];
      }
    }
    sync();
  }

  // ---------------------------------------------------------------------------------------
  // Matrix sweep: evaluate + factorise every stage backwards.  Returns false on wrong inertia.
  // Accumulates the KKT error measures (per lane; reduced by the caller).
  // ---------------------------------------------------------------------------------------
  CMPC_DEV bool matrix_sweep(double mu, double reg, double x0n2, Err &er, bool init) {
    er.e_d = er.e_p = er.e_c = er.e_cmu = er.sum_mult = 0.0; er.n_mult = 0;
    piv_min = fmax(PIV_MIN, PIV_FRAC * reg);
    // height weight of node k, w_z[k-1] = (w/2) e^{-(k-1)} + w/2 (reference :301-305): one exp per sweep, then
    // e^{-(k-1)} by repeated multiplication as k runs down (the library exp is ~1.5 KB of code per use)
    const double e1 = 2.718281828459045235360287;
    double ez = CMPC_UNIFORM_D(exp(-(double)(N - 1)));   // (formed at kernel entry -- it depends on N alone --, kept in scalar registers;
    CMPC_OPAQUE_D(ez);                                   //  its products with the weight are formed here, not carried from there)
    if constexpr (!PIPE) {
      for (int k = N; k >= 0; --k) {
        const double wz = SPD(w_cz_const) * 0.5 * ez + SPD(w_cz_const) * 0.5;
        ez *= e1;
        eval_stage(k, mu, reg, wz, x0n2, er, init);
        if (!riccati_stage(k)) return false;
      }
      return true;
    } else {
      // step j: wave 0 takes stage N - j + 1 (P b, G'PG, factorisation, factor store) out of image (N - j + 1) & 1; wave 1
      // first runs the vector recursion of stage N - j + 2, whose factors wave 0 left in image (N - j) & 1 the step
      // before, then evaluates stage N - j into that image.  One workgroup barrier per step.  A failed factorisation is
      // published in the exchange words behind the two images and seen by both waves after the barrier of its step.
      // (the verdict of step j sits in word 8 + (j & 1): wave 0 may be a step ahead of wave 1's read of the last one)
      double *xch = ldsR + XCH_AT;
      if (lane == 0 && wv == 0) { xch[8] = 0.0; xch[9] = 0.0; }
      CMPC_SYNC_WG();
      for (int j = 0; j <= N + 2; ++j) {
        if (wv == 1) {
          const int kv = N - j + 2, k = N - j;
          if (kv >= 0 && kv <= N) { image(kv); pair_vectors(kv); }
          const double wz = SPD(w_cz_const) * 0.5 * ez + SPD(w_cz_const) * 0.5;
          ez *= e1;
          if (k >= 0) { image(k); eval_stage(k, mu, reg, wz, x0n2, er, init); }
        } else if (j >= 1 && j <= N + 1) {
          const int k = N - j + 1;
          image(k);
          if (!riccati_stage(k) && lane == 0) xch[8 + (j & 1)] = 1.0;
        }
        CMPC_SYNC_WG();
        const bool failed = xch[8 + (j & 1)] != 0.0;
        if (failed) { image(0); CMPC_SYNC_WG(); return false; }
      }
      image(0);
      pair_sync();                             // the vector recursion's l, p (global) reach the forward sweep's wave
      return true;
    }
  }

  // reductions over the instance's lanes: six butterfly steps inside a wave, no LDS; with two waves the
  // wave results meet in two LDS words (oCOLD + 4 / + 5) between two workgroup barriers
  enum { OP_MAX = 0, OP_MIN = 1, OP_SUM = 2 };
  CMPC_DEV double across_waves(double v, int op) {
    if constexpr (NW == 1) return v;
    if ((lane & 63) == 0) R(D::oCOLD + 4 + wv) = v;
    sync();
    const double a = R(D::oCOLD + 4), b = R(D::oCOLD + 5);
    sync();
    return (op == OP_MAX) ? fmax(a, b) : (op == OP_MIN) ? fmin(a, b) : a + b;
  }
  // one butterfly step (the six run from 32 down: see CMPC_PAIR_OF)
  template <int M> CMPC_DEV static double bfly_sum(double v) { double a, b; CMPC_PAIR_OF(M, v, a, b); return a + b; }
  template <int M> CMPC_DEV static double bfly_max(double v) { double a, b; CMPC_PAIR_OF(M, v, a, b); return fmax(a, b); }
  template <int M> CMPC_DEV static double bfly_min(double v) { double a, b; CMPC_PAIR_OF(M, v, a, b); return fmin(a, b); }
  CMPC_DEV static double wave_sum(double v) {
    v = bfly_sum<32>(v); v = bfly_sum<16>(v); v = bfly_sum<8>(v); v = bfly_sum<4>(v); v = bfly_sum<2>(v); return bfly_sum<1>(v);
  }
  CMPC_DEV double red_max(double v) {
    v = bfly_max<32>(v); v = bfly_max<16>(v); v = bfly_max<8>(v); v = bfly_max<4>(v); v = bfly_max<2>(v); v = bfly_max<1>(v);
    return across_waves(v, OP_MAX);
  }
  CMPC_DEV double red_min(double v) {
    v = bfly_min<32>(v); v = bfly_min<16>(v); v = bfly_min<8>(v); v = bfly_min<4>(v); v = bfly_min<2>(v); v = bfly_min<1>(v);
    return across_waves(v, OP_MIN);
  }
  CMPC_DEV double red_sum(double v) { return across_waves(wave_sum(v), OP_SUM); }

  // ---------------------------------------------------------------------------------------
  // Forward sweep: du_k, dx_{k+1}, lam_k.  The factors are read from the slab straight into
  // registers, one batch of independent loads per stage (one HBM round trip): lane j < NU holds
  // column j of Ls and of Lambda, and lane NU + r (or lane r when the stage block is wider than the
  // wave) holds column r of P_k, so  Ls' dx  and  P dx  are the same instruction stream.  The three
  // dense rows of [B A] (angular momentum) are held one column per lane and reduced by butterflies.
  // ---------------------------------------------------------------------------------------
  CMPC_DEV void vector_sweeps(double mu, double dmu, double &ap, double &ad) {
    if constexpr (PIPE) {
      // The pair splits the sweep: wave 0 keeps the serial chain (du_k, dx_{k+1}, lam_k), wave 1 forms the slack and
      // multiplier directions of stage k and the step bounds from du_k (handed over by stage parity) and dx_k (wave 0's
      // ping-pong buffer, rewritten two stages later).  One workgroup barrier per stage; same expressions on the same
      // operands as the single wave.
      image(0);
      if (wv == 1) { slack_sweep(mu, ap, ad); return; }
    }
    // (what the whole sweep carries and every lane has the same value of: in scalar registers)
    mu = CMPC_UNIFORM_D(mu); dmu = CMPC_UNIFORM_D(dmu);
    const double m = CMPC_UNIFORM_D(rec[20]), muf = CMPC_UNIFORM_D(rec[21]);
    constexpr int NIH = (NI + WS - 1) / WS;
    const double tau = CMPC_UNIFORM_D(fmax(0.99, 1 - mu));
    const double d_m = CMPC_UNIFORM_D(SPD(delta) / m);
    double lap = 1.0, lad = 1.0;
    double *xdu = ldsR + XCH_AT + XCH_DU;
    constexpr bool MERGE = D::W_MERGE;
    const bool isA = lane < NU;
    const int lb = MERGE ? lane - NU : lane;
    const bool isB = lb >= 0 && lb < NXA;
    gsync();                                  // the slab was written with another lane mapping
    int cur = D::oXK, nxt = D::oXN1;          // dx_k / dx_{k+1} ping-pong
    if (lane < NXA) { L(cur + lane) = 0.0; gdx[lane] = 0.0; }
    sync();
    // Every global word of stage k is in registers when the stage starts: the sweep is one serial chain over the stages
    // and a stage's loads (one HBM round trip, ~2 us) stood exposed at its top.  They are issued a stage ahead instead, each
    // group as soon as the registers it lands in are free: Ls | P_k after the two products, Lambda's column and the five
    // words the top of a stage needs after the back substitution (no second set of registers, and no copy at the end of
    // the stage -- a copy is a use, and a use waits for the load).  What a stage needs after its chain is loaded at its top.
    struct Early { double l0v, l1v, dg, pv0, pv1; };           // needed at the top of the stage: a stage ahead
    struct Late { double gh[3][NH], al[NH], sv[NIH], zv[NIH], gv[NIH], hw0, hw1, hw2, bq; };   // needed after the chain
    double wa[NXA], wb[MERGE ? 1 : NXA], lam[NU];
    auto load_w = [&](int k) {
      // Ls[c][lane] (lanes < NU) and P_k[c][r] out of the packed image: P_k[c][r] sits in row NU + max(c, r)
      const GArr st = stage(k);
      if constexpr (MERGE) {
        const int lc = (lane < NZ) ? lane : 0;           // lane = NU + r for the P role
        const int rb = tri(lc) + NU;                    // start of the state part of row NU + r
#pragma unroll
        for (int c = 0; c < NXA; ++c) wa[c] = st[D::gM + ((lc < NU || NU + c >= lc) ? tri(NU + c) + lc : rb + c)];
        wb[0] = 0.0;
      } else {
        const int la = (lane < NU) ? lane : 0;
        const int lbc = (lane < NXA) ? lane : 0, rb = tri(NU + lbc) + NU;
#pragma unroll
        for (int c = 0; c < NXA; ++c) wa[c] = st[D::gM + tri(NU + c) + la];
#pragma unroll
        for (int c = 0; c < NXA; ++c) wb[c] = st[D::gM + ((c >= lbc) ? tri(NU + c) + NU + lbc : rb + c)];
      }
    };
    auto load_lam = [&](int k) {
      // column la of Lambda: rows j >= la (the words read for j < la belong to other rows and are never used)
      const GArr st = stage(k);
      const int la = (lane < NU) ? lane : 0;
#pragma unroll
      for (int j = 0; j < NU; ++j) lam[j] = st[D::gM + tri(j) + la];
    };
    auto load_early = [&](int k, Early &w) {
      const GArr st = stage(k);
      const int lb_ = MERGE ? lane - NU : lane;
      const int la = (lane < NU) ? lane : 0, lbc = (lb_ >= 0 && lb_ < NXA) ? lb_ : 0;
      w.l0v = st[D::gL + la]; w.l1v = st[D::gL1 + la]; w.dg = st[D::gM + tri(la) + la];
      w.pv0 = st[D::gPV + lbc]; w.pv1 = st[D::gPV1 + lbc];
    };
    auto load_late = [&](int k, Late &w) {
      const GArr st = stage(k);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int h = 0; h < NH; ++h) w.gh[r][h] = st[D::gGH + r * D::GHS + lane + WS * h];
      // slack / multiplier directions of the stage are formed here too (one pass over the stages less)
      if constexpr (!PIPE) {
#pragma unroll
        for (int h = 0; h < NH; ++h) { const int c = lane + WS * h; w.al[h] = st[D::gAL + ((c < NZ) ? c : 0)]; }
#pragma unroll
        for (int h = 0; h < NIH; ++h) {
          const int r = lane + WS * h, rc = (r < NI) ? r : 0;
          w.sv[h] = gsl[k * NI + rc]; w.zv[h] = gz[k * NI + rc]; w.gv[h] = st[D::gG + rc];
        }
        w.hw0 = gx[k * NXA + 6]; w.hw1 = gx[k * NXA + 7]; w.hw2 = gx[k * NXA + 8];
      } else {
#pragma unroll
        for (int h = 0; h < NH; ++h) w.al[h] = 0.0;
#pragma unroll
        for (int h = 0; h < NIH; ++h) { w.sv[h] = 0.0; w.zv[h] = 0.0; w.gv[h] = 0.0; }
        w.hw0 = 0.0; w.hw1 = 0.0; w.hw2 = 0.0;
      }
      w.bq = st[D::gB + ((lane < NXA) ? lane : 0)];
    };
#ifndef CMPC_FWD_AHEAD_MASK
#define CMPC_FWD_AHEAD_MASK 7
#endif
    constexpr bool AHEAD = ((CMPC_FWD_AHEAD_MASK) >> (PIPE ? 1 : (NW == 1 ? 0 : 2))) & 1;
    Early sm;
    // (in the order the loop issues them, and kept so: the waits inside the loop are computed for the worse of the two ways
    // into it, and vmcnt counts in issue order -- with Ls | P_0 issued last here, every stage waited for Lambda's column
    // before its first product)
    if constexpr (AHEAD) { load_w(0); CMPC_SCHED_FENCE(); load_early(0, sm); CMPC_SCHED_FENCE(); load_lam(0); CMPC_SCHED_FENCE(); }
    else { sm.l0v = 0.0; sm.l1v = 0.0; sm.dg = 0.0; sm.pv0 = 0.0; sm.pv1 = 0.0; }
    for (int k = 0; k <= N; ++k) {
      CMPC_RELANE(lane); CMPC_OPAQUE(lane);
      const bool hasA = k < N, hasB = k >= 1;
      const int kn = hasA ? k + 1 : k;         // the node whose factors are loaded during this one (terminal node: its own again)
      // contact flags of the stage (terminal node: header words 22, 23)
      const double gl = rec[(k < N) ? 24 + 19 * k + 17 : 22], gr = rec[(k < N) ? 24 + 19 * k + 18 : 23];
      Late lt;
      load_late(k, lt);
      if constexpr (!AHEAD) { load_w(k); load_early(k, sm); load_lam(k); }
      const double (&gh)[3][NH] = lt.gh;
      const double (&al)[NH] = lt.al, (&sv)[NIH] = lt.sv, (&zv)[NIH] = lt.zv, (&gv)[NIH] = lt.gv;
      const double hw0 = lt.hw0, hw1 = lt.hw1, hw2 = lt.hw2, bq = lt.bq;
      const double l0v = sm.l0v, l1v = sm.l1v, dg = sm.dg, pv0 = sm.pv0, pv1 = sm.pv1;
      CMPC_TICK(16);
      // ---- Ls' dx (lanes < NU) and P dx (the other role)
      double accA, accB;
      {
        const double *dxv = &L(cur);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        static_assert(NXA % 4 == 0, "unroll by 4");
#pragma unroll
        for (int c = 0; c < NXA; c += 4) {
          a0 += wa[c] * dxv[c]; a1 += wa[c + 1] * dxv[c + 1]; a2 += wa[c + 2] * dxv[c + 2]; a3 += wa[c + 3] * dxv[c + 3];
          if (c % 8 == 4) CMPC_SCHED_FENCE();  // (dx eight words at a time: read in one batch it takes 56 registers beside the
        }                                      //  120 of the factors)
        accA = (a0 + a1) + (a2 + a3);
        if constexpr (!MERGE) {
          double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
#pragma unroll
          for (int c = 0; c < NXA; c += 4) {
            b0 += wb[c] * dxv[c]; b1 += wb[c + 1] * dxv[c + 1]; b2 += wb[c + 2] * dxv[c + 2]; b3 += wb[c + 3] * dxv[c + 3];
          }
          accB = (b0 + b1) + (b2 + b3);
        } else accB = accA;
      }
      CMPC_OPAQUE_D(accA); CMPC_OPAQUE_D(accB);   // (formed HERE: left alone the two sums sink to their use, below the loads)
      if constexpr (AHEAD) load_w(kn);         // (the registers of Ls | P_k are free)
      if (hasB && isB) glamn[k * NXA + lb] = pv0 + dmu * pv1 + accB;
      auto slack_dirs = [&](double ldot) {     // ds, dz and the fraction-to-the-boundary bounds of stage k
#pragma unroll
        for (int h = 0; h < NIH; ++h) {
          const int r = lane + WS * h;
          if (r < NI) {
            const double sr_ = sv[h], zr = zv[h], gr_ = gv[h];
            double ds = 0.0, dz = 0.0;
            if (zr != 0.0) {                   // active rows carry z > 0
              ds = -(gr_ + sr_) - jg_dot(r, ldot, &L(cur), &L(D::oUK), hw0, hw1, hw2, gl, gr, muf);
              dz = (mu - sr_ * zr - zr * ds) / sr_;
              if (ds < 0) lap = fmin(lap, -tau * sr_ / ds);
              if (dz < 0) lad = fmin(lad, -tau * zr / dz);
            }
            gds[k * NI + r] = ds; gdz[k * NI + r] = dz;
          }
        }
      };
      double duv = 0.0;
      if (hasA && first_wave()) {              // L' du = -(l + Ls' dx), multipliers by readlane (the terminal node has no inputs)
        double treg = isA ? -(l0v + dmu * l1v + accA) : 0.0;
        const double dinv = 1.0 / dg;
#pragma unroll
        for (int j = NU - 1; j >= 0; --j) {
          const double dj = CMPC_BCAST(treg * dinv, j);
          if (lane < j) treg -= lam[j] * dj;
        }
        duv = treg * dinv;
        if (isA) { gdu[k * NU + lane] = duv; L(D::oUK + lane) = duv; }
        if constexpr (PIPE) { if (isA) xdu[(k & 1) * NU + lane] = duv; }
      }
      // Every word of `sm` and Lambda's column have had their last use.  EVERY way round the loop passes these loads, the
      // terminal node's too (whose "break" below runs through the loop's latch as far as the compiler's wait-count analysis
      // can tell: with the loads behind a branch, Ls | P_k looked freshly issued at the top of a stage, and the wait for it
      // drained Lambda's column -- vmcnt counts in issue order).
      if constexpr (AHEAD) { load_early(kn, sm); load_lam(kn); }
      if (!hasA) {                             // terminal node: Lyapunov row inactive
        if constexpr (PIPE) { CMPC_SYNC_WG(); break; }      // (dx_N is in LDS: the slack wave's last stage)
        double part = 0.0;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const int c = lane + WS * h;
          if (c >= NU && c < NZ) part += al[h] * L(cur + c - NU);
        }
        slack_dirs(red_sum(part));
        sync();                           // the caller reuses the stage vectors
        break;
      }
      if constexpr (PIPE) CMPC_SYNC_WG();      // du_k and dx_k stand in LDS; the slack wave is done with stage k - 1
      // dense rows: s_r = sum_c GH[r][c] z_c, z = (du, dx), one column per lane
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;   // s3: Lyapunov gradient . (du, dx)
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int c = lane + WS * h;
        const bool in = c < NZ;                // the row padding in the slab is never written
        const double z = (c < NU) ? duv : (in ? L(cur + c - NU) : 0.0);
        s0 += in ? gh[0][h] * z : 0.0; s1 += in ? gh[1][h] * z : 0.0; s2 += in ? gh[2][h] * z : 0.0;
        if constexpr (!PIPE) s3 += in ? al[h] * z : 0.0;
      }
      // the four butterflies step together (same sums, same order as four reductions one after the other)
#define CMPC_STEP4(M) do { s0 = bfly_sum<M>(s0); s1 = bfly_sum<M>(s1); s2 = bfly_sum<M>(s2); if constexpr (!PIPE) s3 = bfly_sum<M>(s3); } while (0)
      CMPC_STEP4(32); CMPC_STEP4(16); CMPC_STEP4(8); CMPC_STEP4(4); CMPC_STEP4(2); CMPC_STEP4(1);
#undef CMPC_STEP4
      if constexpr (NW != 1) {                 // the four sums cross the waves in one exchange
        if ((lane & 63) == 0) {
          double *w = (wv == 0) ? &L(D::oRED) : &R(D::oCOLD + 4);
          w[0] = s0; w[1] = s1; w[2] = s2; w[3] = s3;
        }
        sync();
        s0 = L(D::oRED) + R(D::oCOLD + 4); s1 = L(D::oRED + 1) + R(D::oCOLD + 5);
        s2 = L(D::oRED + 2) + R(D::oCOLD + 6); s3 = L(D::oRED + 3) + R(D::oCOLD + 7);
      }
      sync();
      if constexpr (!PIPE) slack_dirs(s3);
      CMPC_TICK(17);
      // dx+ = b + [B A] (du, dx)
      {
        // every word any row of [B A] needs, read unconditionally at clamped indices (one batch of LDS reads, one wait),
        // then the row picks its expression: as an if / else-if chain over the row this was nine divergent paths with an
        // exposed LDS round trip each (round 4; same expressions, same bits)
        const int q = (lane < NXA) ? lane : 0;
        const double d = SPD(delta);
        const double *dx = &L(cur), *du = &L(D::oUK);
        const int ia = (q < 3) ? 3 + q : ((q >= 9 && q < 12) ? q - 9 : 0), ib = (q >= 9 && q < 12) ? q - 6 : 0;
        const int ja = (q >= CMPC_NX) ? 3 * (q - CMPC_NX) + 2 : (q == 12) ? 6 * NV + 6 : (q >= 13 && q < 16) ? 6 * NV + q - 13
                     : (q == 16) ? 6 * NV + 7 : (q >= 17 && q < 20) ? 6 * NV + 3 + q - 17 : 0;
        const int ax = (q >= 3 && q < 6) ? q - 3 : 0;
        const double own = dx[(q < CMPC_NX) ? q : 0], xa = dx[ia], xb = dx[ib], ua = du[ja];
        double fs = 0.0;
#pragma unroll
        for (int v = 0; v < NF; ++v) fs += ((v < NV) ? gl : gr) * du[3 * v + ax];
        const double gsel = (q >= 16) ? gr : gl;
        const double ssel = (q == 6) ? s0 : (q == 7) ? s1 : s2;
        const double inc = (q < 3) ? own + d * xa : (q < 6) ? own + d_m * fs : (q < 9) ? own + ssel
                         : (q < 12) ? own + d_m * (SPD(k1) * xa + xb) : (q < CMPC_NX) ? own + d * (1 - gsel) * ua : ua;
        const double a = bq + inc;
        if (lane < NXA) {
          gdx[(k + 1) * NXA + q] = a;
          L(nxt + q) = a;
        }
      }
      sync();
      { const int t = cur; cur = nxt; nxt = t; }
      CMPC_TICK(18);
    }
    if constexpr (PIPE) {
      CMPC_SYNC_WG();                          // the slack wave has reduced the step bounds
      const double *xch = ldsR + XCH_AT;
      ap = xch[6]; ad = xch[7];
    } else {
      ap = red_min(lap); ad = red_min(lad);
    }
  }

  // PIPE, wave 1: slack / multiplier directions and the fraction-to-the-boundary bounds, stage by stage behind the
  // chain wave (see vector_sweeps).  The stage's own data (slacks, multipliers, row values, Lyapunov gradient) are
  // loaded before the barrier that releases du_k.
  CMPC_DEV void slack_sweep(double mu, double &ap, double &ad) {
    mu = CMPC_UNIFORM_D(mu);
    const double muf = CMPC_UNIFORM_D(rec[21]);
    constexpr int NIH = (NI + WS - 1) / WS;
    const double tau = CMPC_UNIFORM_D(fmax(0.99, 1 - mu));
    double lap = 1.0, lad = 1.0;
    double *xch = ldsR + XCH_AT;
    gsync();                                  // this wave's stores of the evaluation and of the last step
    int cur = D::oXK, nxt = D::oXN1;          // the chain wave's dx_k / dx_{k+1} ping-pong
    for (int k = 0; k <= N; ++k) {
      const GArr st = stage(k);
      const bool hasA = k < N;
      const double gl = rec[(k < N) ? 24 + 19 * k + 17 : 22], gr = rec[(k < N) ? 24 + 19 * k + 18 : 23];
      double al[NH], sv[NIH], zv[NIH], gv[NIH];
#pragma unroll
      for (int h = 0; h < NH; ++h) { const int c = lane + WS * h; al[h] = st[D::gAL + ((c < NZ) ? c : 0)]; }
#pragma unroll
      for (int h = 0; h < NIH; ++h) {
        const int r = lane + WS * h, rc = (r < NI) ? r : 0;
        sv[h] = gsl[k * NI + rc]; zv[h] = gz[k * NI + rc]; gv[h] = st[D::gG + rc];
      }
      const double hw0 = gx[k * NXA + 6], hw1 = gx[k * NXA + 7], hw2 = gx[k * NXA + 8];
      CMPC_SYNC_WG();                          // du_k (this stage's parity slot) and dx_k are in LDS
      const double *du = xch + XCH_DU + (k & 1) * NU;
      double ldot = 0.0;                       // Lyapunov gradient . (du, dx); terminal node: . dx
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int c = lane + WS * h;
        const bool in = c < NZ;
        if (hasA) {
          const double z = (c < NU) ? du[(c < NU) ? c : 0] : (in ? L(cur + c - NU) : 0.0);
          ldot += in ? al[h] * z : 0.0;
        } else if (c >= NU && c < NZ) ldot += al[h] * L(cur + c - NU);
      }
      ldot = wave_sum(ldot);
#pragma unroll
      for (int h = 0; h < NIH; ++h) {
        const int r = lane + WS * h;
        if (r < NI) {
          const double sr_ = sv[h], zr = zv[h], gr_ = gv[h];
          double ds = 0.0, dz = 0.0;
          if (zr != 0.0) {
            ds = -(gr_ + sr_) - jg_dot(r, ldot, &L(cur), du, hw0, hw1, hw2, gl, gr, muf);
            dz = (mu - sr_ * zr - zr * ds) / sr_;
            if (ds < 0) lap = fmin(lap, -tau * sr_ / ds);
            if (dz < 0) lad = fmin(lad, -tau * zr / dz);
          }
          gds[k * NI + r] = ds; gdz[k * NI + r] = dz;
        }
      }
      { const int t = cur; cur = nxt; nxt = t; }
    }
    ap = red_min(lap); ad = red_min(lad);
    if (lane == 0) { xch[6] = ap; xch[7] = ad; }
    CMPC_SYNC_WG();
  }

  // (Jg d)[r] of one stage: dx, du in LDS, hw = x_k[6..8], g0/g1 = the stage's contact flags.  Every word any row type
  // needs is read unconditionally at clamped indices -- one batch of LDS reads, one wait -- and the row type picks its
  // combination: written as a chain of early returns this was nine divergent paths, each with its own exposed LDS round
  // trip, in the middle of the forward sweep's serial part (round 4; same expressions, same bits).
  CMPC_DEV double jg_dot(int r, double lyap_dot, const double *dx, const double *du, double hw0, double hw1, double hw2,
                         double g0, double g1, double muf) const {
    const bool is_box = r >= R_BOX && r < R_FRIC, is_fr = r >= R_FRIC;
    const int qb = is_box ? r - R_BOX : 0, fb = qb / 6, ab = (qb % 6) / 2;
    const int qf = is_fr ? r - R_FRIC : 0, v = qf / 5, t = qf % 5;
    const double d2 = dx[2], d6 = dx[6], d7 = dx[7], d8 = dx[8], dbx = dx[13 + 4 * fb + ab];
    const double fx = du[3 * v], fy = du[3 * v + 1], fz = du[3 * v + 2];
    const int sgn = (qb & 1) ? -1 : 1;
    const double v_box = sgn * (fb ? g1 : g0) * dbx;
    const double gg = (v / NV) ? g1 : g0;
    const double v_fr = (t == 0) ? gg * (fx - muf * fz) : (t == 1) ? gg * (-fx - muf * fz) : (t == 2) ? gg * (fy - muf * fz)
                      : (t == 3) ? gg * (-fy - muf * fz) : -gg * fz;
    const double v_hw = 2.0 * (hw0 * d6 + hw1 * d7 + hw2 * d8);
    return (r == R_LYAP) ? lyap_dot : (r == R_CZ) ? d2 : (r == R_HWC) ? v_hw : (is_box ? v_box : v_fr);
  }

  CMPC_DEV void apply_step(double mu, double ap, double ad) {
    gsync();                                  // directions were written with a per-stage lane mapping
    // elementwise updates, four independent load groups in flight per pass
    constexpr int UF = 4;
    const bool do_xu = !PIPE || wv == 0, do_sz = !PIPE || wv == 1;   // the pair shares the arrays out
    if (do_xu)
    for (int e0 = NXA; e0 < (N + 1) * NXA; e0 += WS * UF) {
      double a[UF], b[UF], c[UF], d[UF];
#pragma unroll
      for (int q = 0; q < UF; ++q) {
        const int e = e0 + lane + WS * q, ec = (e < (N + 1) * NXA) ? e : NXA;
        a[q] = gx[ec]; b[q] = gdx[ec]; c[q] = glam[ec]; d[q] = glamn[ec];
      }
#pragma unroll
      for (int q = 0; q < UF; ++q) {
        const int e = e0 + lane + WS * q;
        if (e < (N + 1) * NXA) { gx[e] = a[q] + ap * b[q]; glam[e] = c[q] + ap * (d[q] - c[q]); }
      }
    }
    if (do_xu)
    for (int e0 = 0; e0 < N * NU; e0 += WS * UF) {
      double a[UF], b[UF];
#pragma unroll
      for (int q = 0; q < UF; ++q) {
        const int e = e0 + lane + WS * q, ec = (e < N * NU) ? e : 0;
        a[q] = gu[ec]; b[q] = gdu[ec];
      }
#pragma unroll
      for (int q = 0; q < UF; ++q) {
        const int e = e0 + lane + WS * q;
        if (e < N * NU) gu[e] = a[q] + ap * b[q];
      }
    }
    if (do_sz)
    for (int e0 = 0; e0 < (N + 1) * NI; e0 += WS * UF) {
      double zq[UF], sq[UF], dsq[UF], dzq[UF];
#pragma unroll
      for (int q = 0; q < UF; ++q) {
        const int e = e0 + lane + WS * q, ec = (e < (N + 1) * NI) ? e : 0;
        zq[q] = gz[ec]; sq[q] = gsl[ec]; dsq[q] = gds[ec]; dzq[q] = gdz[ec];
      }
#pragma unroll
      for (int q = 0; q < UF; ++q) {
        const int e = e0 + lane + WS * q;
        if (e < (N + 1) * NI && zq[q] != 0.0) {
          const double s = sq[q] + ap * dsq[q];
          const double z = zq[q] + ad * dzq[q];
          const double lo = mu / s / 1e10, hi = mu / s * 1e10;
          gsl[e] = s; gz[e] = fmin(fmax(z, lo), hi);
        }
      }
    }
    gsync();
  }

  // Initial point: warm start or hover forces; x_0 from the record; carried f_z states.
  // warm_: primal start (the previous solution, or the XU block of the solver state when resuming); prox_: centre
  // of the proximal term (the previous solution; null = 0); resume: dynamics multipliers from the solver state
  CMPC_DEV void initial_point(const double *warm_, const double *prox_, bool resume) {
    const GArr warm{const_cast<double *>(warm_)}, prox{const_cast<double *>(prox_)};
    const bool has_warm = warm_ != nullptr, has_prox = prox_ != nullptr;
    const double m = rec[20];
    // Source stage of every stage (see the oracle): stage k resumes from the state's stage k, or from its stage k + 1
    // where a contact switch has moved one stage closer since the state was written.  Table in LDS (M is free here).
    double *srct = &L(D::oM);
    for (int k = lane; k <= N; k += WS) {
      int src = k;
      if (resume) {
        const double gl = rec[(k < N) ? 24 + 19 * k + 17 : 22], gr = rec[(k < N) ? 24 + 19 * k + 18 : 23];
        const int kn = (k < N) ? k + 1 : N;
        const bool same = gl == st_in[D::state_fl(N) + k] && gr == st_in[D::state_fl(N) + N + 1 + k];
        const bool next = k < N && gl == st_in[D::state_fl(N) + kn] && gr == st_in[D::state_fl(N) + N + 1 + kn];
        if (!same && next) src = k + 1;
      }
      srct[k] = (double)src;
    }
    sync();
    for (int e = lane; e < (N + 1) * NXA; e += WS) {
      const int k = e / NXA, i = e % NXA, ks = (int)srct[k];
      double v = 0.0;
      if (i < CMPC_NX) v = (has_warm && k >= 1) ? warm[ks * CMPC_NX + i] : rec[i];
      gx[e] = v; glam[e] = resume ? st_in[D::state_lam(N) + ks * NXA + i] : 0.0;
    }
    for (int e = lane; e < N * NU; e += WS) {
      const int k = e / NU, i = e % NU, ks = ((int)srct[k] < N) ? (int)srct[k] : N - 1;
      double v = 0.0, up = has_prox ? prox[CMPC_NX * (N + 1) + e] : 0.0;
      if (has_warm) { v = warm[CMPC_NX * (N + 1) + ks * NU + i]; }
      else if (i < 6 * NV && (i % 3) == 2) {
        const double gl = rec[24 + 19 * k + 17], gr = rec[24 + 19 * k + 18];
        v = m * SPD(g) / (NV * (gl + gr)) * (((i / 3) < NV) ? gl : gr);
      }
      gu[e] = v; gupx[e] = up;
    }
    // slacks / multipliers of the solver state (zeros = none: the first sweep creates them by the cold rule)
    for (int e = lane; e < (N + 1) * NI; e += WS) {
      const int k = e / NI, i = e % NI, ks = (int)srct[k];
      gsl[e] = resume ? st_in[D::state_s(N) + ks * NI + i] : 0.0;
      gz[e] = resume ? st_in[D::state_z(N) + ks * NI + i] : 0.0;
    }
    gsync();
    for (int e = lane; e < N * NF; e += WS) {
      const int k = e / NF + 1, j = e % NF;
      gx[k * NXA + CMPC_NX + j] = gu[(k - 1) * NU + 3 * j + 2];
    }
    gsync();
    if (COLD_ROLLOUT && !has_warm) {
      // dynamics-consistent cold start (see the oracle): x_{k+1} = F(x_k, u_k) under the initial inputs.  The stage loader
      // and the geometry phase do the evaluation: with x_{k+1} zeroed in LDS the defect b = F(x, u) - x_{k+1} IS F(x, u).
      for (int k = 0; k < N; ++k) {
        load_stage(k);
        if (lane < NXA) L(D::oXN1 + lane) = 0.0;
        sync();
        stage_geometry(k);
        if (lane < NXA) gx[(k + 1) * NXA + lane] = L(D::oBV + lane);
        gsync();
      }
    }
  }

  // X (20 x (N+1)) then U (nu x N), the reference's layout.  Reads the iterate arrays as apply_step /
  // initial_point left them (both end with a full fence).
  CMPC_DEV void write_solution(double *out_) {
    const GArr out{out_};
    for (int e = lane; e < (N + 1) * CMPC_NX; e += WS) out[e] = gx[(e / CMPC_NX) * NXA + (e % CMPC_NX)];
    for (int e = lane; e < N * NU; e += WS) out[CMPC_NX * (N + 1) + e] = gu[e];
  }

  // Solver state for the next tick (CMPC_NSTATE, include/cmpc.h): the current iterate, labelled with its barrier value.
  CMPC_DEV void write_state(double *state_out, double mu_level) {
    const GArr so{state_out};
    gsync();                                    // (first iteration: the slacks were written by the sweep's lane mapping)
    write_solution(state_out);
    for (int e = lane; e < (N + 1) * NXA; e += WS) so[D::state_lam(N) + e] = glam[e];
    for (int e = lane; e < (N + 1) * NI; e += WS) { so[D::state_s(N) + e] = gsl[e]; so[D::state_z(N) + e] = gz[e]; }
    for (int k = lane; k <= N; k += WS) {
      so[D::state_fl(N) + k] = rec[(k < N) ? 24 + 19 * k + 17 : 22];
      so[D::state_fl(N) + N + 1 + k] = rec[(k < N) ? 24 + 19 * k + 18 : 23];
    }
    if (lane == 0) so[D::state_mu(N)] = mu_level;
    sync();                                // (host emulation: every lane has read the flag the caller sets next)
  }

  // ---------------------------------------------------------------------------------------
  CMPC_DEV void solve(const double *warm, const double *state_in, double *state_out, double *out, int32_t *status,
                      int32_t *iters, double *kkt_out) {
    // (the tolerance where it is used, from its scalar register: as a local it was a vector register carried through the
    // whole solve -- and the one spilled)
    auto tol_ = [&]() { return CMPC_FRESH_D(sp.tol); };
    // (what the outer loop compares with, every lane the same value: scalar registers)
    const double tol_acc = ka.tol_acc, tol_10 = ka.tol_tenth;   // ACC_FACTOR * tol, tol / 10
    const double x0n2 = 0.0;                    // |hw_0|^2 is read from the record header in LDS where it is used
    // closed-loop ticks: resume from the previous tick's central-path point (see the oracle, MU_WARM)
    bool resume = false;
    if (state_in) {
      const double ms = GArr{const_cast<double *>(state_in)}[D::state_mu(N)];
      resume = ms > 0.0 && ms < INFINITY;
      if (resume) st_in = GArr{const_cast<double *>(state_in)};
    }
    int st = CMPC_MAX_ITER, it = 0, spent = 0;
    // iteration budget of an attempt: sp.max_iter - spent (both attempts together stay within max_iter)
    double kkt = INFINITY;
    // at most two attempts: a resumed solve that gets nowhere (the state was too far from this tick's problem) is
    // followed by the plain one with the rest of the budget; the iterations of both are reported (see the oracle)
    for (;;) {
    if (state_out && lane == 0) GArr{state_out}[D::state_mu(N)] = 0.0;     // invalid until a snapshot is taken
    double mu = resume ? st_in[D::state_mu(N)] : MU_INIT;
    kkt = INFINITY; st = CMPC_MAX_ITER;
    // The counters of the outer loop (acceptable-level run, progress watch, polish steps left, collapsed steps).  One-wave
    // workgroups keep them in four words of LDS (oCOLD + 4 .. 7, which only the two-wave solver's reductions use): in
    // registers they are live across the whole sweep and were what the kernel spilled.  Read into locals with the other
    // cold state -- every lane reads before any lane writes --, written back before the vector sweeps.
    constexpr bool CNT_LDS = (NW == 1);
    int n_acc = 0, n_stall = 0, polish = -1, since_best = 0;
    // ("the answer is the saved iterate, already written" rides in the verdict -- bit SAVED of `st` -- until the loop is left:
    // as a flag of its own it was one more vector register carried across the whole iteration, and the one spilled)
    constexpr int SAVED = 256;
    // cold scalars of the outer loop live in LDS (every lane reads the same word; written by every lane with the
    // same value, fenced by the phases in between)
    double &reg_last = R(D::oCOLD + 0), &kkt_best = R(D::oCOLD + 1), &kkt_saved = R(D::oCOLD + 2);
    double &snapped = R(D::oCOLD + 3);          // 1 once the solver state of this solve has been written
    {
      double inf = INFINITY, zero = 0.0;        // (materialised here: hoisted out of the instance loop they were spilled)
      CMPC_OPAQUE_D(inf); CMPC_OPAQUE_D(zero);
      // (second attempt: the acceptable point the failed resumed attempt left in `out` stays the level to beat -- see the end
      // of the attempt and the oracle; every lane, and both waves of a pair, read before any of them writes)
      const double ks0 = kkt_saved;
      if constexpr (PIPE) { CMPC_SYNC_WG(); } else sync();
      double at0 = sp.acc_tol;
      CMPC_OPAQUE_D(at0);
      const bool carry = spent > 0 && ks0 <= fmax(at0, tol_());
      reg_last = zero; kkt_best = inf; snapped = zero;
      if (!carry) kkt_saved = inf;
      if constexpr (CNT_LDS) { R(D::oCOLD + 4) = zero; R(D::oCOLD + 5) = zero; R(D::oCOLD + 6) = zero - 1.0; R(D::oCOLD + 7) = zero; }
    }
    // acceptable level; every iterate the acceptable-level counter counts is also saved (see the oracle).  Formed
    // where they are used (two instructions) instead of being kept live across the solve.
    // (the copy of the kernel argument is made opaque at every use: hoisted to kernel entry it sat in a vector register
    // pair through the whole solve and was spilled)
    auto acc_raw = [&]() { double a = sp.acc_tol; CMPC_OPAQUE_D(a); return a; };
    auto acc_tol = [&]() { return fmax(acc_raw(), tol_()); };
    auto save_tol = [&]() { return fmax(fmax(acc_raw(), tol_()), tol_acc); };
    // (the pair: phases that walk the horizon serially belong to wave 0; the other wave waits at the next barrier)
    if (!PIPE || wv == 0) initial_point(resume ? state_in : warm, warm, resume);
    pair_sync();
    CMPC_TICK_RESET();
    for (it = 0; it <= sp.max_iter - spent; ++it) {
      double reg = 0.0;
      Err er;
      bool fail = false;
      const double mu_sweep = mu;               // barrier value the sweep's gradients are formed at
      const double rl = reg_last;               // (cold state: read once, ahead of the fences of the sweep)
      while (!matrix_sweep(mu, reg, x0n2, er, it == 0)) {
        sync();
        if (reg == 0.0) reg = (rl == 0.0) ? 1e-4 : fmax(1e-20, rl / 3);
        else reg *= (rl == 0.0) ? 100.0 : 8.0;
        if (reg > 1e20) { fail = true; break; }
      }
      if (fail) { st = CMPC_NUMERICAL; break; }
#ifdef CMPC_DEBUG_FIRST_SWEEP                  // (diagnostic build, tools/slab_diff.py: leave the slab as the first matrix sweep wrote it)
      if (it == 0) break;
#endif
      double e_d = red_max(er.e_d), e_p = red_max(er.e_p), e_c = red_max(er.e_c), e_cmu = red_max(er.e_cmu);
      double sm = red_sum(er.sum_mult), nm = red_sum((double)er.n_mult);
      if constexpr (PIPE) {
        // the error measures were gathered by the evaluating wave: handed to both through the exchange words; the same
        // barrier orders the slab writes of the sweep (both waves') before the vector sweep of wave 0
        double *xch = ldsR + XCH_AT;
        if (wv == 1 && lane == 0) { xch[0] = e_d; xch[1] = e_p; xch[2] = e_c; xch[3] = e_cmu; xch[4] = sm; xch[5] = nm; }
        pair_sync();
        e_d = xch[0]; e_p = xch[1]; e_c = xch[2]; e_cmu = xch[3]; sm = xch[4]; nm = xch[5];
      }
      const double sd = fmax(100.0, sm / fmax(nm, 1.0)) / 100.0;
      kkt = fmax(fmax(e_d / sd, e_p), e_c / sd);
#ifdef CMPC_HOST_EMU
      if (lane == 0 && getenv("CMPC_EMU_TRACE"))
        printf("it %3d d=%.2e p=%.2e c=%.2e mu=%.1e reg=%.1e\n", it, e_d / sd, e_p, e_c / sd, mu, reg);
#endif
      const double ebar = fmax(fmax(e_d / sd, e_p), e_cmu / sd);   // error of the barrier problem at mu
      double ks = kkt_saved, kb = kkt_best;     // cold state: every lane reads before any lane writes
      const bool unsnapped = snapped == 0.0;
      if constexpr (CNT_LDS) {
        n_acc = (int)R(D::oCOLD + 4); since_best = (int)R(D::oCOLD + 5); polish = (int)R(D::oCOLD + 6); n_stall = (int)R(D::oCOLD + 7);
      }
      if constexpr (PIPE) { CMPC_SYNC_WG(); } else sync();
      if (polish >= 0 && kkt > tol_acc) {
        // polishing lost ground (the step at the final barrier value needed an inertia correction): the point
        // that met the tolerance was written to `out` before the polish and is what is returned
        st = CMPC_CONVERGED | SAVED; kkt = ks; break;
      }
      if (polish < 0) {
        // best acceptable iterate so far (see the oracle): whatever ends the run, it is what is returned
        if (kkt <= save_tol() && kkt < ks) { if (!PIPE || wv == 0) write_solution(out); ks = kkt; kkt_saved = kkt; }
        if (kkt <= tol_()) {
          // (the tolerance was met from a level >= MU_WARM: same snapshot)
          if (state_out && mu >= MU_WARM && unsnapped) { if (!PIPE || wv == 0) write_state(state_out, mu); snapped = 1.0; }
          polish = POLISH_ITERS; mu = tol_10;
        } else {
          n_acc = (kkt <= tol_acc) ? n_acc + 1 : 0;
          if (n_acc >= ACC_ITERS) { st = CMPC_ACCEPTABLE | SAVED; kkt = ks; break; }
          {
            // progress watch on the error of the current barrier problem (final barrier value: the KKT error); it
            // restarts whenever the barrier value changes (see the oracle)
            const double kw = (mu <= tol_10) ? kkt : ebar;
            if (kw < 0.5 * kb) { kb = kw; kkt_best = kw; since_best = 0; } else ++since_best;
            if (since_best >= NOPROG_ITERS && ks <= acc_tol()) {
              st = CMPC_ACCEPTABLE | SAVED; kkt = ks; break;
            }
          }
        }
      }
      if (polish == 0) {
        // (a polish step that ends ABOVE the tolerance, with a larger error than the point that met it: that point,
        // written out before the polish, is what is returned -- see the oracle)
        st = CMPC_CONVERGED;
        if (kkt > tol_() && kkt > ks) { kkt = ks; st = CMPC_CONVERGED | SAVED; }
        break;
      }
      // a resumed solve still at the state's barrier value: the state does not fit this tick's problem
      const bool stale = resume && it >= RESUME_RECENTRE_ITERS && polish < 0 && mu == st_in[D::state_mu(N)];
      int it_cap = sp.max_iter - spent;         // (formed here: hoisted to the loop header it sat in a register across the sweep -- spilled)
      CMPC_OPAQUE(it_cap);
      const bool at_cap = it == it_cap;
      if (at_cap || !(kkt < INFINITY) || n_stall >= STALL_ITERS || stale) {
        if (polish >= 0) st = CMPC_CONVERGED;                   // (cap reached inside the polish)
        else if (ks <= acc_tol() && !stale) { st = CMPC_ACCEPTABLE | SAVED; kkt = ks; }
        else st = (at_cap && !stale) ? CMPC_MAX_ITER : CMPC_NUMERICAL;
        break;
      }
      if (reg > 0) reg_last = reg;
      if (polish > 0) --polish;
      else if (!(resume && it == 0)) {          // (a resumed solve re-centres at the state's barrier value first: see the oracle)
        const double mu_before = mu;
        while (mu > tol_10 && ebar < 10 * mu)
          mu = fmax(tol_10, fmin(MU_FACTOR * mu, mu * sqrt(mu)));
        if (mu != mu_before) {                   // a new barrier problem: the progress watch restarts
          double inf = INFINITY;
          CMPC_OPAQUE_D(inf);
          kkt_best = inf; since_best = 0;
        }
        // this iterate solves the barrier problem at mu_before: the state the next tick resumes from
        if (state_out && mu_before >= MU_WARM && mu < MU_WARM && unsnapped) { if (!PIPE || wv == 0) write_state(state_out, mu_before); snapped = 1.0; }
      }
      if constexpr (CNT_LDS) { R(D::oCOLD + 4) = (double)n_acc; R(D::oCOLD + 5) = (double)since_best; R(D::oCOLD + 6) = (double)polish; }
      double ap, ad;
      // (n_stall: read, barrier, written -- no lane may see the new count where it expects the old one)
      auto count_stall = [&]() {
        if constexpr (CNT_LDS) {
          const int ns = (int)R(D::oCOLD + 7);
          if constexpr (PIPE) { CMPC_SYNC_WG(); } else sync();
          R(D::oCOLD + 7) = (ap < STALL_STEP) ? (double)(ns + 1) : 0.0;
        } else n_stall = (ap < STALL_STEP) ? n_stall + 1 : 0;
      };
      if constexpr (PIPE) {
        vector_sweeps(mu, mu - mu_sweep, ap, ad);   // wave 0: du, dx, lam; wave 1: ds, dz, step bounds
        count_stall();
        apply_step(mu, ap, ad);                     // wave 0: x, lam, u; wave 1: s, z
        pair_sync();                                // the new iterate (global) reaches the other wave
      } else {
        vector_sweeps(mu, mu - mu_sweep, ap, ad);
        CMPC_TICK(6);
        count_stall();
        apply_step(mu, ap, ad);
        CMPC_TICK(7);
      }
    }
    bool use_saved = (st & SAVED) != 0;
    st &= SAVED - 1;
    // A resumed attempt that failed is followed by a plain one with the rest of the budget; an acceptable point it saved on
    // the way is not given up: it stays in `out`, its error is the level the plain attempt has to beat, and with no budget
    // left for a plain attempt it is the answer (see the oracle).
    const double ks_end = kkt_saved;
    const bool keep = resume && (st == CMPC_MAX_ITER || st == CMPC_NUMERICAL) && ks_end <= acc_tol();
    if (keep && !(it < sp.max_iter)) { st = CMPC_ACCEPTABLE; kkt = ks_end; use_saved = true; }
    if (!use_saved && !keep && (!PIPE || wv == 0)) write_solution(out);
    // (the verdict is the same in every lane; said so, the attempt loop is a uniform loop and what it carries -- the
    // iterations spent -- lives in a scalar register instead of a spilled vector one)
    const int again = CMPC_UNIFORM_INT((int)(resume && (st == CMPC_MAX_ITER || st == CMPC_NUMERICAL) && it < sp.max_iter));
    if (!again) break;                          // done, or nothing left of the budget
    spent = CMPC_UNIFORM_INT(spent + it); resume = false;
    if constexpr (PIPE) pair_sync(); else gsync();
    }
    if (lane == 0 && (!PIPE || wv == 0)) {
      *status = st; *iters = it + spent; *kkt_out = kkt;
      // (first spare word of the state: what this solve took -- the next launch queues its instances by it)
      if (state_out) GArr{state_out}[D::state_mu(N) + 1] = (double)(it + spent);
    }
#if defined(CMPC_PROFILE) && !defined(CMPC_HOST_EMU)
    if (lane == 0 && ka.prof)
      for (int i = 0; i < 28; ++i) atomicAdd((unsigned long long *)&ka.prof[i], (unsigned long long)tprof[i]);
#endif
    if constexpr (PIPE) pair_sync(); else sync();
  }
};

#endif  // CMPC_NO_DEVICE_CODE
}  // namespace cmpc
