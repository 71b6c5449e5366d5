// What one instruction of the Gauss-Seidel rows costs a SIMD when nothing else limits it: 64 INDEPENDENT copies of one
// instruction in a loop, 1 / 2 / 4 / 8 waves per SIMD -> s_memtime cycles per instruction per wave and per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o issue_cost_bench issue_cost_bench.hip ; run: ./issue_cost_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, long long *cyc, int iters) {
  const int tid = threadIdx.x;
  float a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3, b = 1.0001f, c = 0.5f;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, pb = {b, c};
  int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  __shared__ float buf[64];
  buf[tid] = tid;
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) float *)buf;
  float l0, l1, l2, l3;
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    if (KIND == 0)        // v_fmac_f32, four independent accumulators
      asm volatile(REP8(REP8("v_fmac_f32_e32 %0, %4, %5\n\tv_fmac_f32_e32 %1, %4, %5\n\tv_fmac_f32_e32 %2, %4, %5\n\tv_fmac_f32_e32 %3, %4, %5\n\t"))
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    else if (KIND == 1)   // v_med3_f32
      asm volatile(REP8(REP8("v_med3_f32 %0, %0, %4, %5\n\tv_med3_f32 %1, %1, %4, %5\n\tv_med3_f32 %2, %2, %4, %5\n\tv_med3_f32 %3, %3, %4, %5\n\t"))
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    else if (KIND == 2)   // v_readlane_b32 into four scalars
      asm volatile(REP8(REP8("v_readlane_b32 %0, %4, 3\n\tv_readlane_b32 %1, %5, 7\n\tv_readlane_b32 %2, %6, 11\n\tv_readlane_b32 %3, %7, 19\n\t"))
                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    else if (KIND == 3)   // v_writelane_b32
      asm volatile(REP8(REP8("v_writelane_b32 %0, %4, 3\n\tv_writelane_b32 %1, %4, 7\n\tv_writelane_b32 %2, %4, 11\n\tv_writelane_b32 %3, %4, 19\n\t"))
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s0));
    else if (KIND == 4)   // v_pk_fma_f32
      asm volatile(REP8(REP8("v_pk_fma_f32 %0, %2, %2, %0\n\tv_pk_fma_f32 %1, %2, %2, %1\n\tv_pk_fma_f32 %0, %2, %2, %0\n\tv_pk_fma_f32 %1, %2, %2, %1\n\t"))
                   : "+v"(p0), "+v"(p1) : "v"(pb));
    else if (KIND == 5)   // s_nop 1
      asm volatile(REP8(REP8("s_nop 1\n\ts_nop 1\n\ts_nop 1\n\ts_nop 1\n\t")));
    else if (KIND == 6)   // ds_read_b128 at one address (broadcast), waited for every 16
      asm volatile(REP8(REP8("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:16\n\tds_read_b32 %2, %4 offset:32\n\tds_read_b32 %3, %4 offset:48\n\t") "s_waitcnt lgkmcnt(0)\n\t")
                   : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3) : "v"(addr));
    else if (KIND == 7)   // the row: v_med3, v_writelane, v_readlane, s_nop 1, v_fmac - 51 rows with NO dependency between rows
      asm volatile(REP8(REP8("v_med3_f32 %1, %0, %4, %5\n\tv_writelane_b32 %2, %3, 5\n\tv_readlane_b32 %3, %1, 9\n\ts_nop 1\n\tv_fmac_f32_e32 %6, %3, %4\n\t"))
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+s"(s0) : "v"(b), "v"(c), "v"(a3));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + tid] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + (float)(s0 + s1 + s2 + s3) + ((KIND == 6) ? l0 + l1 + l2 + l3 : 0.f);
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name, int per_iter, float *out, long long *cyc, double scale) {
  printf("%-44s", name);
  const int iters = 200;
  for (int wps : {1, 2, 4, 8}) {
    const int nb = 1024 * wps;
    hipLaunchKernelGGL(k<KIND>, dim3(nb), dim3(64), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<KIND>, dim3(nb), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(nb);
    hipMemcpy(h.data(), cyc, nb * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    const double per = s / nb / iters / per_iter * scale;
    printf("  %d/SIMD: %5.2f (%5.2f per SIMD)", wps, per, per / wps);
  }
  printf("\n");
}

int main() {
  float *out; long long *cyc;
  hipMalloc(&out, 8192 * 64 * sizeof(float));
  hipMalloc(&cyc, 8192 * sizeof(long long));
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const double scale = 1.0;     // (s_memtime counts shader-clock cycles here: a wave's lifetime in the step kernel times 1 / 2.4 GHz is its share of the launch)
  printf("shader clock %.0f MHz; cycles per instruction and wave (per SIMD), 256 (320 for the row) instructions per loop trip\n", prop.clockRate / 1000.0);
  run<0>("v_fmac_f32", 256, out, cyc, scale);
  run<1>("v_med3_f32", 256, out, cyc, scale);
  run<2>("v_readlane_b32", 256, out, cyc, scale);
  run<3>("v_writelane_b32", 256, out, cyc, scale);
  run<4>("v_pk_fma_f32", 256, out, cyc, scale);
  run<5>("s_nop 1", 256, out, cyc, scale);
  run<6>("ds_read_b32, one address (+ 1 wait per 256)", 256, out, cyc, scale);
  run<7>("the 5-slot row, rows independent (per slot)", 320, out, cyc, scale);
  return 0;
}
