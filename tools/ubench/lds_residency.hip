// Developer microbenchmark: how many single-wave workgroups with S bytes of LDS does a CU of this part hold at once?
// Grid = 256 CUs x n workgroups, each spinning for a fixed time: the launch takes one spin if all are resident together,
// two if not.  usage: ./lds_residency  (prints the threshold for n = 5, 6, 7, 8)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) spin(long long cycles, int *sink) {
  extern __shared__ double lds[];
  lds[threadIdx.x] = threadIdx.x;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) { }
  if (lds[threadIdx.x] < 0) *sink = 1;
}
int main() {
  int *sink; hipMalloc(&sink, 4);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const long long spin_ticks = 200000;   // 2 ms at the 100 MHz wall clock
  hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int n = 5; n <= 8; ++n) {
    int best = 0;
    for (int bytes = 163840 / n + 1024; bytes >= 163840 / n - 2048; bytes -= 16) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      hipLaunchKernelGGL(spin, dim3(cus * n), dim3(64), bytes, 0, spin_ticks, sink);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < 3.0f) { best = bytes; break; }
    }
    printf("%d workgroups per CU resident together up to %d bytes of LDS each (160 KB / %d = %d)\n", n, best, n, 163840 / n);
  }
  return 0;
}
