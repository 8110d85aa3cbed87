// lds_b128_repro.hip -- reduced repro attempt for a round-1 anomaly: in the 8-vertex solver kernel (73 KB of
// LDS) a run of compiler-generated ds_read_b128 at constant addresses -- the broadcast vector of the
// p = m_x - Ls l step, written by the lanes one 8-byte word each just before -- gave wrong values, while the same
// reads issued as ds_read_b64 were correct.  This program reproduces the pattern in isolation: one wavefront per
// workgroup, a static LDS array of the kernel's size, lane-wise ds_write_b64 of a vector at the kernel's offsets
// (7120 and 6588 doubles), the kernel's fence (s_waitcnt lgkmcnt(0) + compiler barrier, no s_barrier), then
// wave-uniform reads of 56 consecutive doubles that hipcc turns into ds_read_b128.  Prints the number of
// mismatches per variant.    hipcc --offload-arch=gfx950 -O3 -o lds_b128_repro lds_b128_repro.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int LDS_DOUBLES = 9322, oTV = 7120, oAL = 6588, NU = 56;

template <bool WIDE> __global__ void __launch_bounds__(64) repro(double *out, int rounds) {
  __shared__ double lds[LDS_DOUBLES];
  __shared__ int next;
  const int lane = threadIdx.x;
  if (lane == 0) next = 0;
  double bad = 0.0;
  for (int r = 0; r < rounds; ++r) {
    if (lane < NU) { lds[oTV + lane] = 1000.0 * r + lane; lds[oAL + lane] = -1000.0 * r - lane; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const double *l0 = &lds[oTV], *l1 = &lds[oAL];
    double a = 0.0, b = 0.0;
    if constexpr (WIDE) {
#pragma unroll
      for (int j = 0; j < NU; j += 2) {            // constant addresses, pairs: ds_read_b128
        a += l0[j] * (j + 1) + l0[j + 1] * (j + 2);
        b += l1[j] * (j + 1) + l1[j + 1] * (j + 2);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        double v0, v1;
        asm volatile("ds_read_b64 %0, %2 offset:%c3\n\tds_read_b64 %1, %2 offset:%c4\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v0), "=&v"(v1) : "v"((unsigned)(size_t)lds), "n"((oTV) * 8 % 65536), "n"((oAL) * 8 % 65536) : "memory");
        (void)v0; (void)v1;
        a += l0[j] * (j + 1);                       // (the asm above only pins the narrow form into the binary)
        b += l1[j] * (j + 1);
      }
    }
    double ea = 0.0, eb = 0.0;
    for (int j = 0; j < NU; ++j) { ea += (1000.0 * r + j) * (j + 1); eb += (-1000.0 * r - j) * (j + 1); }
    bad += (a != ea) + (b != eb);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  out[blockIdx.x * 64 + lane] = bad + next;
}

int main() {
  double *out; const int blocks = 512;
  hipMalloc(&out, blocks * 64 * sizeof(double));
  for (int wide = 1; wide >= 0; --wide) {
    hipMemset(out, 0, blocks * 64 * sizeof(double));
    if (wide) hipLaunchKernelGGL(repro<true>, dim3(blocks), dim3(64), 0, 0, out, 2000);
    else hipLaunchKernelGGL(repro<false>, dim3(blocks), dim3(64), 0, 0, out, 2000);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    static double host[512 * 64];
    hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost);
    double tot = 0; for (double v : host) tot += v;
    printf("%s reads: %.0f mismatches in %d wave-rounds\n", wide ? "wide (compiler, ds_read_b128)" : "narrow", tot, blocks * 2000);
  }
  return 0;
}
