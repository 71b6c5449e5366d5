// Micro-benchmark of one projected-Gauss-Seidel ROW VISIT in Delassus form (the inner loop of trex_step_kernel):
// cycles per row for a wave alone on its SIMD and with 2 / 4 / 8 waves per SIMD, for several codings of
//   nl = clamp(lam + y); d = nl - lam; lam[row lane] = nl; y += B[row] * broadcast(d)
// Build: hipcc -O3 --offload-arch=gfx950 -o row_bench row_bench.hip ; run: ./row_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float rl(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

constexpr int NR = 32;   // rows per sweep (static lanes 1..32)

template <int VARIANT>
__global__ __launch_bounds__(64) void k(float *out, long long *cyc, int sweeps, float hi) {
  const int tid = threadIdx.x;
  float B[NR];
#pragma unroll
  for (int j = 0; j < NR; j++) B[j] = (tid == j + 1) ? -1.f : 1e-3f * (float)((tid * 7 + j * 13) % 11 - 5);
  float y = 0.01f * (tid + 1), lam = 0.f, blo = -hi, bhi = hi;
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < sweeps; it++) {
    int vs = tid;
    asm volatile("" : "+v"(vs));
    unsigned long long one = 1ull;
    asm volatile("" : "+s"(one));
#pragma unroll
    for (int j = 0; j < NR; j++) {
      const int L = j + 1;
      if (VARIANT == 0) {          // v_cmp + v_cndmask (round-1 kernel)
        const float nl = __builtin_amdgcn_fmed3f(lam + y, -hi, hi);
        const float d = nl - lam;
        const float sd = rl(d, L);
        if (vs == L) lam = nl;
        y = __builtin_fmaf(B[j], sd, y);
      } else if (VARIANT == 1) {   // s_lshl mask + v_cndmask
        const float nl = __builtin_amdgcn_fmed3f(lam + y, -hi, hi);
        const float d = nl - lam;
        const float sd = rl(d, L);
        lam = __builtin_amdgcn_inverse_ballot_w64(one << L) ? nl : lam;
        y = __builtin_fmaf(B[j], sd, y);
      } else if (VARIANT == 2) {   // exec-masked move
        const float nl = __builtin_amdgcn_fmed3f(lam + y, -hi, hi);
        const float d = nl - lam;
        const float sd = rl(d, L);
        asm volatile("s_mov_b64 exec, %2\n\tv_mov_b32 %0, %1\n\ts_mov_b64 exec, -1" : "+v"(lam) : "v"(nl), "s"(one << L));
        y = __builtin_fmaf(B[j], sd, y);
      } else if (VARIANT == 3) {   // unclamped fast row: lam[lane] += y; y += B * y[lane]
        const float sd = rl(y, L);
        asm volatile("s_mov_b64 exec, %2\n\tv_add_f32 %0, %0, %1\n\ts_mov_b64 exec, -1" : "+v"(lam) : "v"(y), "s"(one << L));
        y = __builtin_fmaf(B[j], sd, y);
      } else if (VARIANT == 4) {   // unclamped fast row, add + cndmask
        const float sd = rl(y, L);
        lam = __builtin_amdgcn_inverse_ballot_w64(one << L) ? lam + y : lam;
        y = __builtin_fmaf(B[j], sd, y);
      } else if (VARIANT == 6) {   // short chain: d = med3(y, lo - lam, hi - lam) with the shifted bounds kept per lane
        const float d = __builtin_amdgcn_fmed3f(y, blo, bhi);
        const float sd = rl(d, L);
        y = __builtin_fmaf(B[j], sd, y);
        const float dm = (vs == L) ? d : 0.f;
        lam += dm; blo -= dm; bhi -= dm;
      } else if (VARIANT == 7) {   // short chain, bounds formed from lam before the visit
        const float d = __builtin_amdgcn_fmed3f(y, -hi - lam, hi - lam);
        const float sd = rl(d, L);
        y = __builtin_fmaf(B[j], sd, y);
        lam += (vs == L) ? d : 0.f;
      } else if (VARIANT == 5) {   // unclamped, lam not tracked (lower bound: readlane + fma)
        const float sd = rl(y, L);
        y = __builtin_fmaf(B[j], sd, y);
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + tid] = y + lam + blo + bhi;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char *name, float *out, long long *cyc, int sweeps) {
  printf("%-44s", name);
  for (int wps : {1, 2, 4, 8}) {
    const int nb = 1024 * wps;
    hipLaunchKernelGGL(k<V>, dim3(nb), dim3(64), 0, 0, out, cyc, sweeps, 1e30f);
    hipLaunchKernelGGL(k<V>, dim3(nb), dim3(64), 0, 0, out, cyc, sweeps, 1e30f);
    hipDeviceSynchronize();
    std::vector<long long> h(nb);
    hipMemcpy(h.data(), cyc, nb * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("  %d/SIMD: %6.1f cyc/row/wave (%5.1f per SIMD)", wps, s / nb / sweeps / NR, s / nb / sweeps / NR / wps);
  }
  printf("\n");
}

int main() {
  float *out; long long *cyc;
  hipMalloc(&out, 8192 * 64 * sizeof(float));
  hipMalloc(&cyc, 8192 * sizeof(long long));
  const int sweeps = 2000;
  run<0>("0 v_cmp + v_cndmask (round 1)", out, cyc, sweeps);
  run<1>("1 s_lshl mask + v_cndmask", out, cyc, sweeps);
  run<2>("2 exec-masked v_mov", out, cyc, sweeps);
  run<3>("3 unclamped: exec-masked v_add", out, cyc, sweeps);
  run<4>("4 unclamped: v_add + v_cndmask", out, cyc, sweeps);
  run<5>("5 unclamped, lam untracked (bound)", out, cyc, sweeps);
  run<6>("6 short chain, shifted bounds kept", out, cyc, sweeps);
  run<7>("7 short chain, bounds from lam", out, cyc, sweeps);
  return 0;
}
