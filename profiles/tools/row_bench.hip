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
  int dv = 0, sprev = 0, scur = 0;
  __shared__ float cap[NR * 64];
  const int lds_addr = (int)(size_t)(void *)cap * 0 + 4 * tid;   // (offset inside the workgroup's LDS)
  float bsub = 1e-3f * (float)(tid % 5 - 2);
  float scratchv = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  // 16: the kernel's row with only the LOW HALF of the wave enabled (does a wave64 VALU instruction skip a half whose EXEC
  // is 0?) - the compiler sets EXEC for the branch; the loop inside it is uniform
  if (VARIANT != 16 || tid < 32)
#pragma unroll 1
  for (int it = 0; it < sweeps; it++) {
    int vs = tid;
    asm volatile("" : "+v"(vs));
    unsigned long long one = 1ull;
    asm volatile("" : "+s"(one));
    if (VARIANT == 15) { asm volatile("s_waitcnt lgkmcnt(0)"); dv ^= __float_as_int(cap[65 * (tid & 31)]); }   // the diagonal, once per sweep
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
      } else if (VARIANT == 17) {
        // the short row of the speculative motor block: no clamp - writelane (capture of the row before), readlane of y, s_nop 1, fmac (4 slots)
        asm volatile("v_writelane_b32 %1, %2, %4\n\tv_readlane_b32 %2, %0, %4\n\ts_nop 1\n\tv_fmac_f32_e32 %0, %2, %3"
                     : "+v"(y), "+v"(dv), "+s"(sprev) : "v"(B[j]), "n"(1 + (j % 63)));
      } else if (VARIANT == 18) {
        // the same with s_nop 0 in the place of the capture
        asm volatile("s_nop 0\n\tv_readlane_b32 %2, %0, %4\n\ts_nop 1\n\tv_fmac_f32_e32 %0, %2, %3"
                     : "+v"(y), "+v"(dv), "+s"(sprev) : "v"(B[j]), "n"(1 + (j % 63)));
      } else if ((VARIANT >= 8 && VARIANT <= 12) || VARIANT == 16) {
        // hand-placed rows of the round-2 kernel: bounds shifted by the impulse, d = med3(y, blo, bhi); the impulse
        // changes are captured in `dv` (lane L) by v_writelane and committed after the sweep
        //  8: the kernel's row        med3, writelane, readlane, s_nop 1, fmac                 (5 slots)
        //  9: without the capture     med3, s_nop 0,   readlane, s_nop 1, fmac                 (5 slots, one VALU less)
        // 10: two s_nop 0 for s_nop 1 med3, writelane, readlane, s_nop 0, s_nop 0, fmac         (6 slots)
        // 11: VALU fillers            med3, writelane, readlane, v_mov, v_mov, fmac             (6 slots, two VALU more)
        // 12: the 7-slot row of before: add, med3, sub, writelane, readlane, s_nop 1, fmac
        float d_;
        int s_;
        if (VARIANT == 8 || VARIANT == 16)
          asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_writelane_b32 %1, %3, %7\n\tv_readlane_b32 %3, %2, %7\n\ts_nop 1\n\tv_fmac_f32_e32 %0, %3, %6"
                       : "+v"(y), "+v"(dv), "=&v"(d_), "+s"(sprev) : "v"(blo), "v"(bhi), "v"(B[j]), "n"(1 + (j % 63)));
        else if (VARIANT == 9)
          asm volatile("v_med3_f32 %2, %0, %4, %5\n\ts_nop 0\n\tv_readlane_b32 %3, %2, %7\n\ts_nop 1\n\tv_fmac_f32_e32 %0, %3, %6"
                       : "+v"(y), "+v"(dv), "=&v"(d_), "+s"(sprev) : "v"(blo), "v"(bhi), "v"(B[j]), "n"(1 + (j % 63)));
        else if (VARIANT == 10)
          asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_writelane_b32 %1, %3, %7\n\tv_readlane_b32 %3, %2, %7\n\ts_nop 0\n\ts_nop 0\n\tv_fmac_f32_e32 %0, %3, %6"
                       : "+v"(y), "+v"(dv), "=&v"(d_), "+s"(sprev) : "v"(blo), "v"(bhi), "v"(B[j]), "n"(1 + (j % 63)));
        else if (VARIANT == 11)
          asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_writelane_b32 %1, %3, %7\n\tv_readlane_b32 %3, %2, %7\n\tv_mov_b32 %8, %4\n\tv_mov_b32 %8, %5\n\tv_fmac_f32_e32 %0, %3, %6"
                       : "+v"(y), "+v"(dv), "=&v"(d_), "+s"(sprev) : "v"(blo), "v"(bhi), "v"(B[j]), "n"(1 + (j % 63)), "v"(scratchv));
        else
          asm volatile("v_add_f32_e32 %2, %8, %0\n\tv_med3_f32 %2, %2, %4, %5\n\tv_sub_f32_e32 %2, %2, %8\n\tv_writelane_b32 %1, %3, %7\n\tv_readlane_b32 %3, %2, %7\n\ts_nop 1\n\tv_fmac_f32_e32 %0, %3, %6"
                       : "+v"(y), "+v"(dv), "=&v"(d_), "+s"(sprev) : "v"(blo), "v"(bhi), "v"(B[j]), "n"(1 + (j % 63)), "v"(lam));
        (void)s_;
      } else if (VARIANT == 15) {
        // capture through LDS: every lane stores its d at [row][lane] (the diagonal is read back once per sweep),
        // the store is the wait state between v_med3 and v_readlane - an LDS instruction instead of a VALU one
        float d_;
        asm volatile("v_med3_f32 %1, %0, %3, %4\n\t"
                     "ds_write_b32 %6, %1 offset:%8\n\t"
                     "v_readlane_b32 %2, %1, %7\n\t"
                     "s_nop 1\n\t"
                     "v_fmac_f32_e32 %0, %2, %5"
                     : "+v"(y), "=&v"(d_), "+s"(sprev) : "v"(blo), "v"(bhi), "v"(B[j]), "v"(lds_addr), "n"(1 + (j % 63)), "n"(256 * j));
      } else if (VARIANT == 13 || VARIANT == 14) {
        // neighbour fast path: the next row's lane gets this row's change by DPP (wave_shr:1, VGPR to VGPR), the other
        // lanes by the SGPR broadcast ONE ROW LATER (software-pipelined: no s_nop, the readlane round trip is off the
        // critical path). 13: with the v_writelane capture, 14: without
        float d_;
        if (VARIANT == 13)
          asm volatile("v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_fmac_f32_e32 %0, %3, %7\n\t"
                       "v_readlane_b32 %4, %2, %9\n\t"
                       "v_fmac_f32_dpp %0, %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                       "v_writelane_b32 %1, %3, %9"
                       : "+v"(y), "+v"(dv), "=&v"(d_), "+s"(sprev), "+s"(scur) : "v"(blo), "v"(bhi), "v"(B[j]), "v"(bsub), "n"(1 + (j % 63)));
        else
          asm volatile("v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_fmac_f32_e32 %0, %3, %7\n\t"
                       "v_readlane_b32 %4, %2, %9\n\t"
                       "v_fmac_f32_dpp %0, %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf"
                       : "+v"(y), "+v"(dv), "=&v"(d_), "+s"(sprev), "+s"(scur) : "v"(blo), "v"(bhi), "v"(B[j]), "v"(bsub), "n"(1 + (j % 63)));
        { const int t_ = sprev; sprev = scur; scur = t_; }
      } else if (VARIANT == 5) {   // unclamped, lam not tracked (lower bound: readlane + fma)
        const float sd = rl(y, L);
        y = __builtin_fmaf(B[j], sd, y);
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + tid] = y + lam + blo + bhi + __int_as_float(dv) + scratchv;
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
  run<12>("12 asm, 7 slots (add med3 sub wl rl nop1 fmac)", out, cyc, sweeps);
  run<8>("8 asm, 5 slots (med3 wl rl nop1 fmac)", out, cyc, sweeps);
  run<17>("17 asm, 4 slots, no clamp (wl rl nop1 fmac)", out, cyc, sweeps);
  run<18>("18 asm, 4 slots, no clamp, no capture (nop0 rl nop1 fmac)", out, cyc, sweeps);
  run<16>("16 the same (8) with EXEC = the low 32 lanes", out, cyc, sweeps);
  run<9>("9 asm, 5 slots, s_nop 0 for the writelane", out, cyc, sweeps);
  run<10>("10 asm, 6 slots, s_nop 0 x2 for s_nop 1", out, cyc, sweeps);
  run<11>("11 asm, 6 slots, two v_mov for s_nop 1", out, cyc, sweeps);
  run<13>("13 neighbour DPP fast path, pipelined broadcast", out, cyc, sweeps);
  run<14>("14 the same without the capture", out, cyc, sweeps);
  run<15>("15 capture by ds_write_b32 [row][lane]", out, cyc, sweeps);
  return 0;
}
