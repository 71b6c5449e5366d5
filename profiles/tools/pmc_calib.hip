// Calibration for the gfx950 FETCH_SIZE / WRITE_SIZE counters in THIS kernel family's access
// pattern (one dword per lane, 128-B team rows), as MI355X_MICROARCH.md §HBM prescribes for
// widths other than 16 B/lane: a copy of a known byte count, profiled with the same --pmc passes.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_copy_dword(const float *in, float *out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] + 1.0f;
}
int main() {
  const size_t n = 256u << 20;  // 1 GiB read + 1 GiB written: far beyond L2 + Infinity Cache
  float *a, *b;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
  hipMemset(a, 0, n * 4);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(calib_copy_dword, dim3((n + 63) / 64), dim3(64), 0, 0, a, b, n);
  hipDeviceSynchronize();
  printf("calib bytes_read=%zu bytes_written=%zu per launch\n", n * 4, n * 4);
  return 0;
}
