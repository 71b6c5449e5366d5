// What a launch of the trainer kernels' shape costs before any work: 256 workgroups with a large dynamic LDS allocation, a
// barrier and one store each. hipcc -O3 --offload-arch=gfx950 -o launch_floor_bench launch_floor_bench.hip && ./launch_floor_bench
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void touch(float *out) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = lds[1];
}
int main() {
  float *out; (void)hipMalloc(&out, 4096);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  struct { int gx, gy, bs, lds; } cfg[] = {{128, 2, 128, 120832}, {128, 2, 256, 146432}, {128, 2, 128, 32768}, {128, 2, 128, 0}, {64, 1, 256, 120832}, {83, 1, 512, 0}, {21, 1, 256, 0}};
  for (auto c : cfg) {
    (void)hipFuncSetAttribute((const void *)touch, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(touch, dim3(c.gx, c.gy), dim3(c.bs), c.lds, 0, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int N = 2000;
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(touch, dim3(c.gx, c.gy), dim3(c.bs), c.lds, 0, out);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("grid (%d, %d) x %d threads, %d B of LDS: %.2f us per back-to-back launch\n", c.gx, c.gy, c.bs, c.lds, 1e3 * ms / N);
  }
  return 0;
}
