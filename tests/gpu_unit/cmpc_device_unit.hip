// cmpc_device_unit.hip -- TEST HARNESS ONLY (GPU tier).  The host emulation of the device source (tests/emu) replaces
// three hand-tuned pieces by plain C++: the Newton pivot square root on v_rsq_f64, the inline-asm LDS batch-read
// helpers, and the fp64 MFMA tile with its lane layout.  These kernels exercise exactly those pieces on the GPU,
// in isolation, against values the host computes (tests/test_gpu_device_units.py).
#include <hip/hip_runtime.h>
#include "../../online-non-linear-centroidal-mpc-with-stability-guarantees-for-robust-locomotion-of-legged-robots-_amd/csrc/cmpc_kernel.hpp"

namespace {

__global__ void k_pivot_sqrt(const double *p, double *s, double *inv, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cmpc::Solver<4>::pivot_sqrt(p[i], s[i], inv[i]);
}

// LDS filled with f(word) = word + 0.25; every helper reads at a lane-dependent base and the kernel counts mismatches.
__global__ void __launch_bounds__(64) k_lds_helpers(int *bad) {
  __shared__ double lds[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = i + 0.25;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int nb = 0;
  const double *base = lds + 3 * lane + 7;
  { double v[8]; cmpc::lds_read_strided8<29>(v, base); for (int i = 0; i < 8; ++i) nb += v[i] != (3 * lane + 7 + 29 * i) + 0.25; }
  { double v[14]; cmpc::lds_read_strided14<1>(v, base); for (int i = 0; i < 14; ++i) nb += v[i] != (3 * lane + 7 + i) + 0.25; }
  { double v[14]; cmpc::lds_read_strided14<29>(v, base); for (int i = 0; i < 14; ++i) nb += v[i] != (3 * lane + 7 + 29 * i) + 0.25; }
  { double v[16]; cmpc::lds_read_strided16<1>(v, base); for (int i = 0; i < 16; ++i) nb += v[i] != (3 * lane + 7 + i) + 0.25; }
  { double v[18]; cmpc::lds_read_strided18<37>(v, base); for (int i = 0; i < 18; ++i) nb += v[i] != (3 * lane + 7 + 37 * i) + 0.25; }
  { double v[28]; cmpc::lds_read_strided28<29>(v, base); for (int i = 0; i < 28; ++i) nb += v[i] != (3 * lane + 7 + 29 * i) + 0.25; }
  { double v[28]; cmpc::lds_read_strided28<1>(v, base); for (int i = 0; i < 28; ++i) nb += v[i] != (3 * lane + 7 + i) + 0.25; }
  { double a[10], b[10]; cmpc::lds_read_pair10(a, b, base, base + 500);
    for (int i = 0; i < 10; ++i) nb += (a[i] != (3 * lane + 7 + i) + 0.25) + (b[i] != (3 * lane + 507 + i) + 0.25); }
  { double a[14], b[14]; cmpc::lds_read_pair14(a, b, base, base + 900);
    for (int i = 0; i < 14; ++i) nb += (a[i] != (3 * lane + 7 + i) + 0.25) + (b[i] != (3 * lane + 907 + i) + 0.25); }
  { double v[10]; cmpc::lds_read_strided10<29>(v, base); for (int i = 0; i < 10; ++i) nb += v[i] != (3 * lane + 7 + 29 * i) + 0.25; }
  { double v[10]; cmpc::cmpc_lds_word p[10];
    for (int i = 0; i < 10; ++i) p[i] = cmpc::cmpc_lds_word_at(lds, (7 * lane + 131 * i * i + 5 * i) & 4095);   // ten unrelated words per lane
    cmpc::lds_read_gather10(v, p);
    for (int i = 0; i < 10; ++i) nb += v[i] != ((7 * lane + 131 * i * i + 5 * i) & 4095) + 0.25; }
  { double v[56]; cmpc::lds_read_row<56>(v, base); for (int i = 0; i < 56; ++i) nb += v[i] != (3 * lane + 7 + i) + 0.25; }
  atomicAdd(bad, nb);
}

// D (16x16) = sum over 16 columns q of A[i][q] * Bm[j][q], operands fed as the blocked Cholesky's trailing update
// feeds them (lane l supplies row l & 15, column 4*ks + (l >> 4); receives D[(l >> 4) + 4 r][l & 15] in component r).
__global__ void __launch_bounds__(64) k_mfma_tile(const double *A, const double *Bm, double *Dout) {
  const int lane = threadIdx.x, r16 = lane & 15, kq = lane >> 4;
  cmpc_v4d acc = {0.0, 0.0, 0.0, 0.0};
  for (int ks = 0; ks < 4; ++ks) acc = CMPC_MFMA_F64(A[r16 * 16 + 4 * ks + kq], Bm[r16 * 16 + 4 * ks + kq], acc);
  for (int r = 0; r < 4; ++r) Dout[(kq + 4 * r) * 16 + r16] = acc[r];
}

}  // namespace

extern "C" {
int unit_pivot_sqrt(const double *p, double *s, double *inv, int n) {
  hipLaunchKernelGGL(k_pivot_sqrt, dim3((n + 255) / 256), dim3(256), 0, 0, p, s, inv, n);
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
int unit_lds_helpers(int *bad) {
  hipLaunchKernelGGL(k_lds_helpers, dim3(8), dim3(64), 0, 0, bad);
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
int unit_mfma_tile(const double *A, const double *B, double *D) {
  hipLaunchKernelGGL(k_mfma_tile, dim3(1), dim3(64), 0, 0, A, B, D);
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
}
