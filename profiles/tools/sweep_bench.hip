// Micro-benchmark of the MOTOR BLOCK of one projected-Gauss-Seidel sweep (25 rows, one per lane 1..25), two codings:
//   rows    the kernel's 25 dependent 5-slot rows (v_med3, v_writelane, v_readlane, s_nop 1, v_fmac): 125 slots, one chain
//   matrix  the block as ONE linear map of the sweep's input, valid while no row changes its clamp status:
//             raw = R y_m,  y' = y + C y_m   (R 25 x 25 unit lower triangular, C 64 x 25; column k of both in a register PAIR)
//           = 25 x (v_readlane of y_k, v_pk_fma_f32 of the pair with the scalar), no dependency between the 25, + the check
//           (one v_med3 / v_cmp over the raw values) and the commit
// cycles per SWEEP for a wave alone on its SIMD and with 2 / 4 waves per SIMD; the matrix form is checked against the rows
// on the host (same inputs, no row at a bound).
// Build: hipcc -O3 --offload-arch=gfx950 -o sweep_bench sweep_bench.hip ; run: ./sweep_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NR = 25;

__device__ __forceinline__ float rl(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// five columns of the matrix form: the scalars first (five v_readlane into even SGPRs), then five v_pk_fma_f32 - the
// two wait states between a v_readlane and the VALU instruction that reads its SGPR are covered by the other readlanes
#define MX5(K0)                                                                                                    \
  asm volatile("v_readlane_b32 s20, %[y], %[l0]\n\t"                                                               \
               "v_readlane_b32 s22, %[y], %[l1]\n\t"                                                               \
               "v_readlane_b32 s24, %[y], %[l2]\n\t"                                                               \
               "v_readlane_b32 s26, %[y], %[l3]\n\t"                                                               \
               "v_readlane_b32 s28, %[y], %[l4]\n\t"                                                               \
               "v_pk_fma_f32 %[a0], %[c0], s[20:21], %[a0] op_sel_hi:[1,0,1]\n\t"                                  \
               "v_pk_fma_f32 %[a1], %[c1], s[22:23], %[a1] op_sel_hi:[1,0,1]\n\t"                                  \
               "v_pk_fma_f32 %[a0], %[c2], s[24:25], %[a0] op_sel_hi:[1,0,1]\n\t"                                  \
               "v_pk_fma_f32 %[a1], %[c3], s[26:27], %[a1] op_sel_hi:[1,0,1]\n\t"                                  \
               "v_pk_fma_f32 %[a0], %[c4], s[28:29], %[a0] op_sel_hi:[1,0,1]\n\t"                                  \
               : [a0] "+v"(acc0), [a1] "+v"(acc1)                                                                  \
               : [y] "v"(y), [c0] "v"(CR[K0]), [c1] "v"(CR[K0 + 1]), [c2] "v"(CR[K0 + 2]), [c3] "v"(CR[K0 + 3]),    \
                 [c4] "v"(CR[K0 + 4]), [l0] "n"(K0 + 1), [l1] "n"(K0 + 2), [l2] "n"(K0 + 3), [l3] "n"(K0 + 4),      \
                 [l4] "n"(K0 + 5)                                                                                  \
               : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29")

// the same with two plain v_fmac per column (C and R in separate registers)
#define MF5(K0)                                                                                                    \
  asm volatile("v_readlane_b32 s20, %[y], %[l0]\n\t"                                                               \
               "v_readlane_b32 s22, %[y], %[l1]\n\t"                                                               \
               "v_readlane_b32 s24, %[y], %[l2]\n\t"                                                               \
               "v_readlane_b32 s26, %[y], %[l3]\n\t"                                                               \
               "v_readlane_b32 s28, %[y], %[l4]\n\t"                                                               \
               "v_fmac_f32_e32 %[ya], s20, %[c0]\n\t"                                                              \
               "v_fmac_f32_e32 %[ra], s20, %[r0]\n\t"                                                              \
               "v_fmac_f32_e32 %[ya], s22, %[c1]\n\t"                                                              \
               "v_fmac_f32_e32 %[ra], s22, %[r1]\n\t"                                                              \
               "v_fmac_f32_e32 %[ya], s24, %[c2]\n\t"                                                              \
               "v_fmac_f32_e32 %[ra], s24, %[r2]\n\t"                                                              \
               "v_fmac_f32_e32 %[ya], s26, %[c3]\n\t"                                                              \
               "v_fmac_f32_e32 %[ra], s26, %[r3]\n\t"                                                              \
               "v_fmac_f32_e32 %[ya], s28, %[c4]\n\t"                                                              \
               "v_fmac_f32_e32 %[ra], s28, %[r4]\n\t"                                                              \
               : [ya] "+v"(ya), [ra] "+v"(ra)                                                                      \
               : [y] "v"(y), [c0] "v"(CR[K0].x), [c1] "v"(CR[K0 + 1].x), [c2] "v"(CR[K0 + 2].x), [c3] "v"(CR[K0 + 3].x), \
                 [c4] "v"(CR[K0 + 4].x), [r0] "v"(CR[K0].y), [r1] "v"(CR[K0 + 1].y), [r2] "v"(CR[K0 + 2].y),          \
                 [r3] "v"(CR[K0 + 3].y), [r4] "v"(CR[K0 + 4].y), [l0] "n"(K0 + 1), [l1] "n"(K0 + 2), [l2] "n"(K0 + 3), \
                 [l3] "n"(K0 + 4), [l4] "n"(K0 + 5)                                                                 \
               : "s20", "s22", "s24", "s26", "s28")


// the matrix form with the 25 scalars taken from LDS instead of 25 v_readlane: y goes to LDS once (one ds_write_b32), every lane
// reads it back at ONE address (broadcast) as pairs, and the pair is the multiplier of v_pk_fma_f32 (op_sel picks the half)
#define LB_LO(ACC, K, P) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(ACC) : "v"(CR[K]), "v"(P))
#define LB_HI(ACC, K, P) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(ACC) : "v"(CR[K]), "v"(P))

// VARIANT 0: rows; 1: matrix, v_pk_fma_f32; 2: matrix, two v_fmac per column; 3: matrix, scalars through LDS
template <int VARIANT>
__global__ __launch_bounds__(64) void k(float *out, long long *cyc, int sweeps, float hi, const float *Bin, const float *CRin) {
  const int tid = threadIdx.x;
  float B[NR];
  f2 CR[NR];
#pragma unroll
  for (int j = 0; j < NR; j++) {
    B[j] = Bin[64 * j + tid];
    CR[j].x = CRin[128 * j + tid]; CR[j].y = CRin[128 * j + 64 + tid];
  }
  float y = 0.01f * (float)((tid * 5) % 17 - 8), lam = 0.f, lam_c = 0.f;
  const float mhi = hi;
  int dvec = 0, sprev = 0;
  unsigned bad_any = 0u;
  __shared__ float ybuf[64];
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < sweeps; it++) {
    const float blo = (-mhi - lam) + lam_c, bhi = (mhi - lam) + lam_c;
    if (VARIANT == 0) {
#pragma unroll
      for (int j = 0; j < NR; j++) {
        float d_;
        asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_writelane_b32 %1, %3, %7\n\tv_readlane_b32 %3, %2, %8\n\ts_nop 1\n\tv_fmac_f32_e32 %0, %3, %6"
                     : "+v"(y), "+v"(dvec), "=&v"(d_), "+s"(sprev) : "v"(blo), "v"(bhi), "v"(B[j]), "n"(j == 0 ? 63 : j), "n"(j + 1));
      }
      asm volatile("v_writelane_b32 %0, %1, 25" : "+v"(dvec) : "s"(sprev));
    } else {
      float yn, raw;
      if (VARIANT == 1) {
        f2 acc0 = {y, 0.f}, acc1 = {0.f, 0.f};
        MX5(0); MX5(5); MX5(10); MX5(15); MX5(20);
        yn = acc0.x + acc1.x; raw = acc0.y + acc1.y;
      } else if (VARIANT == 3) {
        ybuf[tid] = y;
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): one wave, no barrier needed
        const f2 *yb = reinterpret_cast<const f2 *>(ybuf);
        f2 p[13];
#pragma unroll
        for (int i = 0; i < 13; i++) p[i] = yb[i];
        f2 acc0 = {y, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NR; kk++) {      // column kk multiplies y of lane kk + 1
          const int l = kk + 1;
          if (kk & 1) { if (l & 1) LB_HI(acc1, kk, p[l >> 1]); else LB_LO(acc1, kk, p[l >> 1]); }
          else { if (l & 1) LB_HI(acc0, kk, p[l >> 1]); else LB_LO(acc0, kk, p[l >> 1]); }
        }
        yn = acc0.x + acc1.x; raw = acc0.y + acc1.y;
      } else {
        float ya = y, ra = 0.f;
        MF5(0); MF5(5); MF5(10); MF5(15); MF5(20);
        yn = ya; raw = ra;
      }
      // the check: every row inside its bounds (the benchmark runs with bounds that never bind); then commit
      const unsigned bad = (unsigned)__ballot(__builtin_amdgcn_fmed3f(raw, blo, bhi) != raw) & 0x3fffffeu;
      bad_any |= bad;
      y = yn;
      dvec = __float_as_int((tid >= 1 && tid <= NR) ? raw : 0.f);
    }
    {   // lam += dvec, compensated
      const float y_ = __int_as_float(dvec) - lam_c, t_ = lam + y_;
      lam_c = (t_ - lam) - y_;
      lam = t_;
    }
    // (a stand-in for the rest of the sweep that keeps the iteration from settling at zero)
    y += 0.001f * (float)((tid + it) % 7 - 3);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 128 + tid] = y;
  out[blockIdx.x * 128 + 64 + tid] = lam - lam_c + (bad_any ? 1e30f : 0.f);
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char *name, float *out, long long *cyc, int sweeps, const float *B, const float *CR, std::vector<float> *keep) {
  printf("%-40s", name);
  for (int wps : {1, 2, 4}) {
    const int nb = 1024 * wps;
    hipLaunchKernelGGL(k<V>, dim3(nb), dim3(64), 0, 0, out, cyc, sweeps, 1e30f, B, CR);
    hipLaunchKernelGGL(k<V>, dim3(nb), dim3(64), 0, 0, out, cyc, sweeps, 1e30f, B, CR);
    hipDeviceSynchronize();
    std::vector<long long> h(nb);
    hipMemcpy(h.data(), cyc, nb * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("  %d/SIMD: %7.1f cyc/sweep/wave (%6.1f per SIMD)", wps, s / nb / sweeps, s / nb / sweeps / wps);
  }
  printf("\n");
  // a short run for the comparison of the results
  hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, out, cyc, 40, 1e30f, B, CR);
  hipDeviceSynchronize();
  keep->resize(128);
  hipMemcpy(keep->data(), out, 128 * sizeof(float), hipMemcpyDeviceToHost);
}

int main() {
  // B: a contraction-like operator (diagonal -1, small couplings); C, R from it by the recurrences of the matrix form
  std::vector<double> B(64 * NR), C(64 * NR, 0.0), R(64 * NR, 0.0);
  for (int j = 0; j < NR; j++)
    for (int s = 0; s < 64; s++) B[64 * j + s] = (s == j + 1) ? -1.0 : 0.02 * (double)(((s * 7 + j * 13) % 11) - 5) / (1.0 + 0.2 * std::abs(s - j - 1));
  // T_k = response of the block to e_k; R[j][k] = T_k[lane of row j] when row j is visited
  for (int k = 0; k < NR; k++) {
    std::vector<double> T(64, 0.0);
    T[k + 1] = 1.0;
    for (int j = 0; j < NR; j++) {
      const double s = T[j + 1];
      R[64 * k + j + 1] = s;
      for (int l = 0; l < 64; l++) T[l] += B[64 * j + l] * s;
    }
    for (int l = 0; l < 64; l++) C[64 * k + l] = T[l] - (l == k + 1 ? 1.0 : 0.0);
  }
  std::vector<float> Bf(64 * NR), CRf(128 * NR);
  for (int j = 0; j < NR; j++)
    for (int s = 0; s < 64; s++) { Bf[64 * j + s] = (float)B[64 * j + s]; CRf[128 * j + s] = (float)C[64 * j + s]; CRf[128 * j + 64 + s] = (float)R[64 * j + s]; }
  float *out, *dB, *dCR; long long *cyc;
  hipMalloc(&out, 8192 * 128 * sizeof(float));
  hipMalloc(&cyc, 8192 * sizeof(long long));
  hipMalloc(&dB, Bf.size() * sizeof(float));
  hipMalloc(&dCR, CRf.size() * sizeof(float));
  hipMemcpy(dB, Bf.data(), Bf.size() * sizeof(float), hipMemcpyHostToDevice);
  hipMemcpy(dCR, CRf.data(), CRf.size() * sizeof(float), hipMemcpyHostToDevice);
  const int sweeps = 2000;
  std::vector<float> r0, r1, r2, r3;
  run<0>("0 rows (25 x 5 slots)", out, cyc, sweeps, dB, dCR, &r0);
  run<1>("1 matrix, v_pk_fma_f32 (25 x 2 slots)", out, cyc, sweeps, dB, dCR, &r1);
  run<2>("2 matrix, 2 v_fmac (25 x 3 slots)", out, cyc, sweeps, dB, dCR, &r2);
  run<3>("3 matrix, scalars through LDS (25 pk)", out, cyc, sweeps, dB, dCR, &r3);
  double e3 = 0;
  for (int i = 0; i < 64 + 1 + NR; i++) if (i != 64) e3 = std::fmax(e3, std::fabs(r3[i] - r0[i]));
  printf("matrix (LDS) - rows %.3g\n", e3);
  double e1 = 0, e2 = 0, sc = 0;
  for (int i = 0; i < 64 + 1 + NR; i++) { if (i == 64) continue;   // y of all lanes, lam of lanes 1..25
    e1 = std::fmax(e1, std::fabs(r1[i] - r0[i])); e2 = std::fmax(e2, std::fabs(r2[i] - r0[i])); sc = std::fmax(sc, std::fabs(r0[i])); }
  printf("after 40 sweeps: largest |y, lam| %.4g; matrix (pk) - rows %.3g; matrix (fmac) - rows %.3g\n", sc, e1, e2);
  return 0;
}
