// cmpc_wave.hpp -- cross-lane exchange for wave-wide reductions on gfx950 (device code only; shared by the solver kernels
// and the whole-body QP kernel).
#pragma once
#include <hip/hip_runtime.h>

// Butterfly step of a wave-wide reduction: CMPC_PAIR_OF(M, v, a, b) leaves in {a, b} this lane's v and the v of lane ^ M
// (as a SET: which of the two is which differs by lane, so it serves commutative combinations only -- a + b, fmax, fmin --
// and those are then bit for bit the results of an exchange through ds_bpermute).  No LDS round trip: lanes 32 and 16
// apart through gfx950's v_permlane32_swap / v_permlane16_swap (both operands = v: the two results are the lower and
// the upper partner in every lane), 8 and 4 apart by a row rotate (DPP), 2 and 1 apart by a quad permute (DPP).  The
// rotate by 4 reaches lane ^ 4 or lane ^ 4 ^ 8: the steps must run from 32 down, so that lanes 8 apart already agree.
// Measured on an idle CU: 864 -> 184 cycles per six-step reduction of a double.
template <int CTRL> static __device__ __forceinline__ double cmpc_dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int M> static __device__ __forceinline__ void cmpc_pair_of(double v, double &a, double &b) {
  static_assert(M == 32 || M == 16 || M == 8 || M == 4 || M == 2 || M == 1, "butterfly distance");
  if constexpr (M == 32) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto p = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), q = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    a = __hiloint2double((int)q[0], (int)p[0]); b = __hiloint2double((int)q[1], (int)p[1]);
  } else if constexpr (M == 16) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto p = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), q = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    a = __hiloint2double((int)q[0], (int)p[0]); b = __hiloint2double((int)q[1], (int)p[1]);
  } else {
    a = v;                                   // row_ror:8, row_ror:4, quad_perm:[2,3,0,1], quad_perm:[1,0,3,2]
    b = (M == 8) ? cmpc_dpp_mov<0x128>(v) : (M == 4) ? cmpc_dpp_mov<0x124>(v) : (M == 2) ? cmpc_dpp_mov<0x4e>(v) : cmpc_dpp_mov<0xb1>(v);
  }
}
