// hbm_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the solver's access width.
// MI355X_MICROARCH.md (HBM section): the counters are exact only for 16 B/lane streaming accesses (FETCH_SIZE
// reports 1/2 there); "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern".  The solver's slab traffic is 8 B/lane (global_load/store_dwordx2, lane = fastest index), so this
// program streams a known number of bytes with exactly that pattern, once reading and once writing, far past
// the 256 MiB Infinity Cache.  tools/profile_round.sh runs it under the same --pmc passes and divides.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_calib hbm_calib.hip ; ./hbm_calib [GiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void calib_read8(const double *__restrict__ src, size_t n, double *__restrict__ sink) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
  if (acc == 1.2345e300) *sink = acc;            // never true: keeps the loads
}
__global__ void calib_write8(double *__restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (double)i;
}

int main(int argc, char **argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 4.0;
  const size_t n = (size_t)(gib * 1024.0 * 1024.0 * 1024.0 / 8.0);
  double *buf = nullptr, *sink = nullptr;
  if (hipMalloc(&buf, n * 8) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 0, n * 8);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(calib_write8, dim3(256 * 16), dim3(256), 0, 0, buf, n);
  hipLaunchKernelGGL(calib_read8, dim3(256 * 16), dim3(256), 0, 0, buf, n, sink);
  hipDeviceSynchronize();
  printf("{\"bytes\": %zu}\n", n * 8);
  hipFree(buf); hipFree(sink);
  return 0;
}
