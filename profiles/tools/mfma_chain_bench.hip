// What the pieces of learn_grad_kernel's chain cost on one wave (s_memtime ticks): a chain of dependent f32 MFMAs, tanh_fast,
// expf, the 32-lane DPP sums. hipcc -O3 --offload-arch=gfx950 -o mfma_chain_bench mfma_chain_bench.hip && ./mfma_chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float tanh_fast(float x) {
  const float ax = fabsf(x), x2 = x * x;
  const float e = __builtin_amdgcn_exp2f(ax * 2.885390081777927f);
  const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  const float p = x * (1.0f + x2 * (-0.3333333333f + x2 * (0.1333333333f + x2 * (-0.05396825397f))));
  return ax < 0.1f ? p : copysignf(t, x);
}
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
__device__ __forceinline__ float half_sum(float v) {
  v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
#define T() (__builtin_amdgcn_sched_barrier(0), __builtin_amdgcn_s_memtime())
__global__ void bench(float *out, unsigned long long *t, const float *in) {
  __shared__ float L[64 * 33];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 33; i += 64) L[i] = in[i & 63];
  __syncthreads();
  f32x16 acc, acc2;
  for (int r = 0; r < 16; r++) { acc[r] = in[r]; acc2[r] = in[16 + r]; }
  float a[32], b[32];
  for (int s = 0; s < 32; s++) { a[s] = in[s] + lane; b[s] = in[32 + s] - lane; }
  unsigned long long t0 = T();
#pragma unroll
  for (int s = 0; s < 32; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
  asm volatile("" : "+v"(acc));
  unsigned long long t1 = T();
#pragma unroll
  for (int s = 0; s < 16; s++) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], a[s], acc2, 0, 0, 0); }
  asm volatile("" : "+v"(acc), "+v"(acc2));
  unsigned long long t2 = T();
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = tanh_fast(acc[r]);
  asm volatile("" : "+v"(acc));
  unsigned long long t3 = T();
#pragma unroll
  for (int r = 0; r < 16; r++) acc2[r] = expf(acc2[r] * 1e-3f);
  asm volatile("" : "+v"(acc2));
  unsigned long long t4 = T();
#pragma unroll
  for (int r = 0; r < 16; r++) { acc[r] = half_sum(acc[r]); acc2[r] = half_sum(acc2[r]); }
  asm volatile("" : "+v"(acc), "+v"(acc2));
  unsigned long long t5 = T();
  // 32 LDS operand reads, then a chain that uses them
#pragma unroll
  for (int s = 0; s < 32; s++) a[s] = L[(s * 2 + (lane >> 5)) * 33 + (lane & 31)];
  asm volatile("" ::: "memory");
#pragma unroll
  for (int s = 0; s < 32; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
  asm volatile("" : "+v"(acc));
  unsigned long long t6 = T();
#pragma unroll
  for (int r = 0; r < 16; r++) L[(r * 2 + (lane >> 5)) * 33 + (lane & 31)] = acc[r];
  __syncthreads();
  unsigned long long t7 = T();
  float sum = 0.f;
  for (int r = 0; r < 16; r++) sum += acc[r] + acc2[r];
  out[lane] = sum + L[lane];
  if (lane == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t4 - t3; t[4] = t5 - t4; t[5] = t6 - t5; t[6] = t7 - t6; }
}
int main() {
  float *out, *in; unsigned long long *t;
  hipMalloc(&out, 256); hipMalloc(&in, 256); hipMalloc(&t, 64);
  float h[64]; for (int i = 0; i < 64; i++) h[i] = 0.01f * i;
  hipMemcpy(in, h, 256, hipMemcpyHostToDevice);
  unsigned long long ht[8];
  for (int rep = 0; rep < 3; rep++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(bench, dim3(1), dim3(64), 0, 0, out, t, in); hipEventRecord(e1);
    hipDeviceSynchronize();
    hipMemcpy(ht, t, 56, hipMemcpyDeviceToHost);
    printf("32 chained MFMA 32x32x2 f32: %llu | 2 x 16 interleaved: %llu | 16 tanh_fast: %llu | 16 expf: %llu | 32 half_sum: %llu | 32 LDS reads + 32 MFMA: %llu | 16 LDS writes + barrier: %llu ticks\n",
           ht[0], ht[1], ht[2], ht[3], ht[4], ht[5], ht[6]);
  }
  return 0;
}
