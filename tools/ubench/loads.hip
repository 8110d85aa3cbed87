// Microbenchmark: cost of a batch of NL independent 8-byte (or 16-byte) global loads per lane, in a
// launch shaped like the solver (one wave per workgroup, each wave walking its own slab).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int NL, int VEC, bool STORE> __global__ __launch_bounds__(64) void k(const double *buf, double *wbuf, size_t slab, int stages, int reps,
                                                   long long *cyc, double *sink, int lds_pad, int alu) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const double *base = buf + (size_t)blockIdx.x * slab;
  double *wbase = wbuf + (size_t)blockIdx.x * slab;
  double acc = 0.0;
  long long t = 0;
  for (int r = 0; r < reps; ++r)
    for (int s = 0; s < stages; ++s) {
      const double *st = base + (size_t)s * (slab / stages);
      double *wst = wbase + (size_t)s * (slab / stages);
      long long t0 = clock64();
      if constexpr (VEC == 1) {
        double v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) v[i] = st[i * 64 + lane];
#pragma unroll
        for (int i = 0; i < NL; ++i) acc += v[i];
      } else {
        double2 v[NL / 2];
#pragma unroll
        for (int i = 0; i < NL / 2; ++i) v[i] = ((const double2 *)st)[i * 64 + lane];
#pragma unroll
        for (int i = 0; i < NL / 2; ++i) acc += v[i].x + v[i].y;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t += clock64() - t0;
      if constexpr (STORE) {
#pragma unroll
        for (int i = 0; i < NL; ++i) wst[i * 64 + lane] = acc + i;
      }
      // some ALU work between batches (dependent chain ~ 2k cycles)
      for (int q = 0; q < alu; ++q) acc = acc * 1.0000001 + 1e-9;
    }
  if (lane == 0) atomicAdd((unsigned long long *)cyc, (unsigned long long)t);
  if (acc == 1.2345) sink[0] = acc;
  if (lds_pad < 0) lds[lane] = acc;
}
template <int NL, int VEC, bool STORE> void run(int grid, size_t lds_bytes, const char *name, int alu = 256) {
  const int stages = 21, reps = (alu > 1000) ? 2 : 8;
  const size_t slab = 21 * 3296 + 9000;   // doubles, like the solver's slab
  double *buf, *wbuf, *sink; long long *cyc;
  hipMalloc(&buf, slab * grid * 8); hipMalloc(&wbuf, slab * grid * 8); hipMalloc(&sink, 8); hipMalloc(&cyc, 8);
  hipMemset(buf, 0, slab * grid * 8); hipMemset(cyc, 0, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NL, VEC, STORE><<<grid, 64, lds_bytes>>>(buf, wbuf, slab, stages, 1, cyc, sink, 0, alu);
  hipMemset(cyc, 0, 8);
  hipEventRecord(e0);
  k<NL, VEC, STORE><<<grid, 64, lds_bytes>>>(buf, wbuf, slab, stages, reps, cyc, sink, 0, alu);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double per = (double)c / ((double)grid * stages * reps);
  double gb = (double)grid * stages * reps * NL * 64 * 8 / 1e9;
  printf("%-28s grid %5d NL %3d vec %d store %d: %8.0f cycles per batch (%6.1f per load instr)  kernel %.2f ms  load BW %.0f GB/s\n", name, grid, NL, VEC, (int)STORE, per,
         per / (NL / VEC), ms, gb / (ms * 1e-3));
  hipFree(buf); hipFree(wbuf); hipFree(sink); hipFree(cyc);
}
int main() {
  const size_t lds = 31000;   // 5 workgroups per CU
  run<8, 1, false>(1280, lds, "5wg/CU alu12000", 12000);
  run<16, 1, false>(1280, lds, "5wg/CU alu12000", 12000);
  run<32, 1, false>(1280, lds, "5wg/CU alu12000", 12000);
  run<64, 1, false>(1280, lds, "5wg/CU alu12000", 12000);
  run<64, 2, false>(1280, lds, "5wg/CU x2 alu12000", 12000);
  run<32, 1, true>(1280, lds, "5wg/CU +st alu12000", 12000);
  run<64, 1, true>(1280, lds, "5wg/CU +st alu12000", 12000);
  run<8, 1, false>(1280, lds, "5wg/CU");
  run<16, 1, false>(1280, lds, "5wg/CU");
  run<32, 1, false>(1280, lds, "5wg/CU");
  run<64, 1, false>(1280, lds, "5wg/CU");
  run<32, 2, false>(1280, lds, "5wg/CU x2");
  run<64, 2, false>(1280, lds, "5wg/CU x2");
  run<32, 1, true>(1280, lds, "5wg/CU +stores");
  run<64, 1, true>(1280, lds, "5wg/CU +stores");
  run<32, 1, false>(256, lds, "1wg/CU");
  run<64, 1, false>(256, lds, "1wg/CU");
  run<32, 1, false>(2560, 15000, "10wg/CU");
  run<64, 1, false>(2560, 15000, "10wg/CU");
  return 0;
}
