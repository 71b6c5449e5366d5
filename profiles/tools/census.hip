// Where do the workgroups of the step launch land? Same launch shape as trex_step_kernel (64 threads, 10 KB LDS,
// 128 VGPRs -> 4 waves per SIMD): every workgroup records HW_ID / XCC_ID and spins long enough for the whole
// grid to be resident. Prints how workgroup ids map to (XCC, SE, CU, SIMD). Speed only - nothing may depend on it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(64, 4) void census(unsigned *out, int spin) {
  __shared__ float lds[2500];
  float acc[96];
  for (int i = 0; i < 96; i++) acc[i] = threadIdx.x * 0.5f + i;
  lds[threadIdx.x] = 0.f;
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < spin) {
    for (int i = 0; i < 96; i++) acc[i] = acc[i] * 1.0001f + 0.5f;
  }
  float s = 0.f;
  for (int i = 0; i < 96; i++) s += acc[i];
  lds[threadIdx.x + 64] = s;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc & 0xf; }
  if (s == 12345.f) out[0] = (unsigned)lds[100];
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 4096;
  unsigned *d; hipMalloc(&d, n * 8);
  hipLaunchKernelGGL(census, dim3(n), dim3(64), 0, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(2 * n);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> bysimd;
  for (int b = 0; b < n; b++) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1];
    const unsigned wave = hw & 15, simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    if (b < 40) printf("wg %4d: xcc %u se %u sh %u cu %2u simd %u wave %u\n", b, xcc, se, sh, cu, simd, wave);
    bysimd[(xcc << 16) | (se << 12) | (sh << 10) | (cu << 4) | simd].push_back(b);
  }
  printf("%zu distinct SIMDs used by %d workgroups\n", bysimd.size(), n);
  int shown = 0;
  std::map<int, int> hist;
  for (auto &kv : bysimd) {
    hist[(int)kv.second.size()]++;
    if (shown++ < 12) { printf("simd %06x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
  }
  for (auto &kv : hist) printf("%d SIMDs hold %d workgroups\n", kv.second, kv.first);
  return 0;
}
