// Trainer-side per-step kernels around the batched physics step (include/trex_policy.h).
//
// What a PPO2 trainer does per env step in the reference - VecNormalize.step_wait + MlpPolicy.step through
// baselines / TF (trex_train.py:41-49) - is ~30 small framework kernels per step in a stock PyTorch policy, which
// halves the rollout rate of an 11 M env-steps/s env. Here it is two launches per step:
//
//   observe_kernel   per-column batch moments of the [N, D | reward | done] row block (f64 partial sums per
//                    workgroup, merged IN FIXED ORDER by the last workgroup to end: deterministic) -> Chan's parallel
//                    update of the running mean / variance (f64 state, as numpy's in VecNormalize), f32 mean / rstd
//                    for the policy kernel; ret = ret*gamma + rew; ret[done] = 0.
//   act_kernel       one workgroup = (tile of 32 envs, net), 2 waves that own half of a hidden layer's neurons each. The net's
//                    parameters and the normalised observation tile are staged in LDS; the 75 -> 64 -> 64 -> {25, 1} tanh MLPs run on the matrix cores with
//                    v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: exact f32, the dense contraction of this path)
//                    in the TRANSPOSED form H^T = W^T X^T: the result tile has the env on the lane and the neurons in
//                    the 16 accumulator registers, which is exactly the B operand of the next layer's MFMA - a wave's own
//                    half never leaves the registers, the other wave's half arrives through LDS (one barrier per layer).
//                    Weights are stored [in][out]: an A operand is 32 consecutive floats per half wave.
//
// plus the rollout's GAE(lambda) and the optimiser step (global-norm clip + TF-form Adam) as one launch each.
// The arithmetic is the one restated by oracle/ppo_oracle.py; tests/test_gpu_policy.py compares the two.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/trex_batch.h"
#include "../../include/trex_policy.h"
#include "internal.hpp"
#include "policy_common.h"

namespace {

// ---------------------------------------------------------------- act
struct ActArgs {
  const float *theta, *rows, *norm, *noise;
  float *actions, *obs_out, *act_out, *logp_out, *value_out;
  int n, row_stride, value_only;
  float clip_obs;
  Layout lay;
};

// One workgroup = (tile of 32 envs, net): 2 waves, each owning one HALF of the 64 neurons of a hidden layer (u = wave: 32
// output rows = one MFMA tile); grid (tiles, 2) = 256 workgroups at 4096 envs - the whole chip (until round 4: 64 workgroups of
// four waves, a wave running one net of one tile alone: 22.8 us, 176 MFMAs in a row on 64 CUs). The layers are the learner's
// forward pass (ppo_learner.hip: same operands, same order of accumulation): H^T = W^T X^T, the env on the lane, the neurons
// in the 16 accumulator registers = the B operand of the next layer; the half the other wave owns arrives through LDS, where
// every activation tile is parked as [row][env]. Only this net's parameters are staged. No control flow between the first
// load and the first barrier (a branch around loads costs an s_waitcnt vmcnt(0) at the join): out-of-range lanes read a clamped
// address and select afterwards.
constexpr int ACT_UF = 29;             // float4s of theta per thread of the staging trip (128 threads: obs_dim <= 126, actions <= 32)
constexpr int ACT_UX = 32;             // observation elements per thread (32 rows x <= 128 columns)
constexpr int ACT_XS = MAXD + 1, ACT_TS = TILE + 1;
__host__ __device__ inline int act_lds_floats(const Layout &lay) {
  return ((lay.count + 3) & ~3) + TILE * ACT_XS + 2 * HID * ACT_TS + 32;      // theta image | X | H1, H2 parked | sd per action
}
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) void act_kernel(ActArgs g) {      // (one wave per SIMD: the whole register file, no spills)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, u = tid >> 6, net = blockIdx.y;      // net 0: policy, 1: value
  if (g.value_only && net == 0) return;
  const int D = g.lay.D, A = g.lay.A;
  const int Dp8 = (D + 7) & ~7;         // K of the first layer, padded to four MFMA k-steps (k = 2 each); X is zero from column D on
  float *th = lds;                                       // parameters, theta's own layout (the other net's part stays unwritten)
  float *X = lds + ((g.lay.count + 3) & ~3);             // [TILE][MAXD + 1] normalised observations (+ zero pad)
  float *H1 = X + TILE * ACT_XS, *H2 = H1 + HID * ACT_TS, *SD = H2 + HID * ACT_TS;
  const int e0 = blockIdx.x * TILE;
  const int col = lane & 31, h = lane >> 5;
  const int e = e0 + col, ec = e < g.n ? e : g.n - 1;
  // ---- loads: observation rows, this net's parameters, the noise draws (their round trip hides behind the MFMA phases)
  float raw[ACT_UX], mean[ACT_UX], rstd[ACT_UX];
#pragma unroll
  for (int q = 0; q < ACT_UX; q++) {
    const int idx = min(tid + 128 * q, TILE * Dp8 - 1), ii = idx / Dp8, kk = idx - ii * Dp8;
    const int kc = kk < D ? kk : D - 1, er = min(e0 + ii, g.n - 1);
    raw[q] = g.rows[(size_t)er * g.row_stride + kc];
    mean[q] = g.norm[kc]; rstd[q] = g.norm[D + kc];
  }
  float tpx[ACT_UF], tpy[ACT_UF], tpz[ACT_UF], tpw[ACT_UF];     // (as one float4 array the compiler kept it in scratch)
  const float4 *s4 = reinterpret_cast<const float4 *>(g.theta);
  const int lo4 = (net ? g.lay.vW1 : 0) >> 2, hi4 = (net ? g.lay.logstd : g.lay.vW1) >> 2;
  const int ls4 = g.lay.logstd >> 2, le4 = g.lay.count >> 2;
  const int n4 = (hi4 - lo4) + (net ? 0 : le4 - ls4);
  auto theta_src = [&](int q) {          // the float4 of theta that slot q of this thread stages (clamped beyond the last)
    const int i = tid + 128 * q;
    const int sq = i < hi4 - lo4 ? lo4 + i : ls4 + (i - (hi4 - lo4));
    return i < n4 ? sq : lo4;
  };
#pragma unroll
  for (int q = 0; q < ACT_UF; q++) { const float4 t = s4[theta_src(q)]; tpx[q] = t.x; tpy[q] = t.y; tpz[q] = t.z; tpw[q] = t.w; }
  const float *nzp = g.noise ? g.noise : g.theta;      // (value-only calls pass no noise: any readable float)
  f32x16 nz_in;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int a = rowmap(r, h);
    nz_in[r] = nzp[g.noise ? (size_t)ec * A + (a < A ? a : A - 1) : 0];
  }
  const float ls_raw = g.theta[g.lay.logstd + (lane < A ? lane : A - 1)];
  LDS_ISSUED();
  // ---- stage the tile: normalise, clip, keep a copy for the rollout buffer
#pragma unroll
  for (int q = 0; q < ACT_UX; q++) {
    const int idx = tid + 128 * q, ii = idx / Dp8, kk = idx - ii * Dp8;
    const bool in = idx < TILE * Dp8, ok = in && kk < D && e0 + ii < g.n;
    const float x = ok ? fminf(fmaxf((raw[q] - mean[q]) * rstd[q], -g.clip_obs), g.clip_obs) : 0.f;
    if (ok && net == 0 && g.obs_out) g.obs_out[(size_t)(e0 + ii) * D + kk] = x;
    X[in ? ii * ACT_XS + kk : TILE * ACT_XS - 1] = in ? x : 0.f;         // (the dump slot is a pad column: never read)
  }
#pragma unroll
  for (int q = 0; q < ACT_UF; q++) {
    if (tid + 128 * q < n4) reinterpret_cast<float4 *>(th)[theta_src(q)] = make_float4(tpx[q], tpy[q], tpz[q], tpw[q]);
  }
  if (u == 1 && lane < 32) SD[lane] = lane < A ? expf(ls_raw) : 1.f;       // sd of every action, once per workgroup
  asm volatile("" : "+v"(nz_in));       // (pinned: the compiler would sink the loads to their use after the last barrier)
  __syncthreads();
  const float *W1 = th + (net ? g.lay.vW1 : g.lay.pW1), *b1 = th + (net ? g.lay.vb1 : g.lay.pb1);
  const float *W2 = th + (net ? g.lay.vW2 : g.lay.pW2), *b2 = th + (net ? g.lay.vb2 : g.lay.pb2);
  // ---- layer 1 (own half): h1^T = tanh(W1^T x^T + b1); B operand = the observation of env `col` from LDS. All LDS operands
  // of a chain are read in one batch before its first MFMA (ppo_learner.hip, LDS_ISSUED)
  f32x16 h1o, h2o;
#pragma unroll
  for (int r = 0; r < 16; r++) h1o[r] = b1[32 * u + rowmap(r, h)];
  {
    const int nch = Dp8 >> 3;                            // chunks of four k-steps, <= 16
    float wa[64], xb[64];
#pragma unroll
    for (int c = 0; c < 16; c++) {
      if (c < nch) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int k = 8 * c + 2 * j + h;
          xb[4 * c + j] = X[col * ACT_XS + k];           // (zero from column D on)
          const int kc = k < D ? k : D - 1;              // (such a weight row does not exist: any finite value, times 0)
          wa[4 * c + j] = W1[kc * HID + 32 * u + col];
        }
      }
    }
    LDS_ISSUED();
#pragma unroll
    for (int c = 0; c < 16; c++) {
      if (c < nch) {
#pragma unroll
        for (int j = 0; j < 4; j++) h1o = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[4 * c + j], xb[4 * c + j], h1o, 0, 0, 0);
      }
    }
  }
  tanh16(h1o);
#pragma unroll
  for (int r = 0; r < 16; r++) H1[(32 * u + rowmap(r, h)) * ACT_TS + col] = h1o[r];
  __syncthreads();
  // ---- layer 2 (own half; k ascending: t = 0, 1)
#pragma unroll
  for (int r = 0; r < 16; r++) h2o[r] = b2[32 * u + rowmap(r, h)];
  {
    float wa[32], hb[16];
#pragma unroll
    for (int s = 0; s < 32; s++) wa[s] = W2[(32 * (s >> 4) + rowmap(s & 15, h)) * HID + 32 * u + col];
#pragma unroll
    for (int s = 0; s < 16; s++) hb[s] = H1[(32 * (1 - u) + rowmap(s, h)) * ACT_TS + col];      // the half the other wave owns
    LDS_ISSUED();
#pragma unroll
    for (int t = 0; t < 2; t++) {
      if (t == u) {
#pragma unroll
        for (int s = 0; s < 16; s++) h2o = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[16 * t + s], h1o[s], h2o, 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < 16; s++) h2o = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[16 * t + s], hb[s], h2o, 0, 0, 0);
      }
    }
  }
  tanh16(h2o);
  if (u == 1) {
#pragma unroll
    for (int r = 0; r < 16; r++) H2[(32 + rowmap(r, h)) * ACT_TS + col] = h2o[r];
  }
  __syncthreads();
  if (u == 1) return;
  // ---- wave 0: the output layer
  if (net == 1) {
    // value head: one output; every lane sums its 32 neurons (own half, then the parked half: k ascending), the two halves of an env meet by a swap
    const float *w = th + g.lay.vW3;
    float v = 0.f;
#pragma unroll
    for (int s = 0; s < 16; s++) v = __builtin_fmaf(h2o[s], w[rowmap(s, h)], v);
#pragma unroll
    for (int s = 0; s < 16; s++) v = __builtin_fmaf(H2[(32 + rowmap(s, h)) * ACT_TS + col], w[32 + rowmap(s, h)], v);
    v += __shfl_xor(v, 32, 64);
    v += th[g.lay.vb3];
    if (h == 0 && e < g.n && g.value_out) g.value_out[e] = v;
    return;
  }
  // policy head: mean^T[a][env] (a = rowmap(reg, h) < A), then the Gaussian sample and its log-probability
  const float *W3 = th + g.lay.pW3, *b3 = th + g.lay.pb3, *ls = th + g.lay.logstd;
  f32x16 mu;
#pragma unroll
  for (int r = 0; r < 16; r++) { const int a = rowmap(r, h); mu[r] = b3[a < A ? a : A - 1]; }
#pragma unroll
  for (int r = 0; r < 16; r++) mu[r] = rowmap(r, h) < A ? mu[r] : 0.f;
  {
    float wa[32], hb[16];
#pragma unroll
    for (int s = 0; s < 32; s++) wa[s] = W3[(32 * (s >> 4) + rowmap(s & 15, h)) * A + (col < A ? col : A - 1)];      // A operand: row = action `col`
#pragma unroll
    for (int s = 0; s < 16; s++) hb[s] = H2[(32 + rowmap(s, h)) * ACT_TS + col];
    LDS_ISSUED();
#pragma unroll
    for (int s = 0; s < 32; s++) wa[s] = col < A ? wa[s] : 0.f;
#pragma unroll
    for (int s = 0; s < 16; s++) mu = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], h2o[s], mu, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 16; s++) mu = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[16 + s], hb[s], mu, 0, 0, 0);
  }
  float zz = 0.f, sum_ls = 0.f;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int a = rowmap(r, h);
    if (a < A) {
      const float l = ls[a], sd = SD[a];
      sum_ls += l;
      if (e < g.n) {
        const float nz = nz_in[r];
        const float act = __builtin_fmaf(sd, nz, mu[r]);
        const float z = (act - mu[r]) / sd;            // as the learner recomputes it from the stored action
        zz = __builtin_fmaf(z, z, zz);
        g.actions[(size_t)e * A + a] = act;
        if (g.act_out) g.act_out[(size_t)e * A + a] = act;
      }
    }
  }
  zz += __shfl_xor(zz, 32, 64);
  sum_ls += __shfl_xor(sum_ls, 32, 64);
  if (h == 0 && e < g.n && g.logp_out) g.logp_out[e] = -0.5f * zz - sum_ls - 0.5f * LOG_2PI * (float)A;
}

// ---------------------------------------------------------------- observe (VecNormalize)
// stats (f64): [0, D) obs mean, [D, 2D) obs var, [2D] obs count, [2D+1] ret mean, [2D+2] ret var, [2D+3] ret count,
//              [2D+4] sum of raw rewards.  norm (f32): [0, D) mean, [D, 2D) 1/sqrt(var + eps), [2D] reward scale.
struct ObserveArgs {
  const float *rows;
  double *stats, *partial;     // partial [G][D + 2][2]: per workgroup and column: sum and sum of squares about the running mean
  float *norm, *ret, *raw_rew_out, *done_out, *rew_scale_out;
  unsigned *counter;
  int n, row_stride, D, with_reward;
  float gamma, epsilon;
};

// One workgroup = OBS_ROWS rows x 4 row groups of 128 threads (thread c of a group: column c); every thread takes 8 rows
// per trip with the 8 loads in flight together. (Version 1 walked 64 rows one dependent load after the other and merged
// 64 partials the same way: 48 us per launch.)
// TWO launches since round 4 (MERGE = false: the per-workgroup partial sums; MERGE = true, one workgroup: their merge in
// workgroup order and the update of the running statistics). As ONE launch whose last workgroup to end did the merge - two
// device-scope fences, an atomic hand-over and a workgroup that starts its second job only when the slowest has ended - this
// took 24 us for 1.26 MB.
constexpr int OBS_GROUPS = 4;
template <bool MERGE>
__global__ __launch_bounds__(128 * OBS_GROUPS) void observe_kernel(ObserveArgs g, int G) {
  __shared__ double red[OBS_GROUPS][128][3];
  const int tid = threadIdx.x, c = tid & 127, grp = tid >> 7, D = g.D;
  double s = 0.0, ss = 0.0, sr = 0.0;
  if (!MERGE) {
  const int per = OBS_ROWS / OBS_GROUPS;
  const int r0 = blockIdx.x * OBS_ROWS + grp * per, r1 = min(r0 + per, g.n);
  // thread c < D: observation column c. thread D: the discounted returns (+ the raw rewards, for logging).
  // Sums are taken about the RUNNING mean (a shift that every workgroup knows): no cancellation in the variance.
  if (c < D) {
    for (int r = r0; r < r1; r += 16) {       // 16 rows per row group: ONE trip
      float x[16];
#pragma unroll
      for (int u = 0; u < 16; u++) x[u] = r + u < r1 ? g.rows[(size_t)(r + u) * g.row_stride + c] : 0.f;
      const double shift = g.stats[c];          // (behind the row loads: one round trip for both)
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (r + u < r1) { const double d = (double)x[u] - shift; s += d; ss += d * d; }
    }
  } else if (c == D && g.with_reward) {
    const double shift = g.stats[2 * D + 1];
    for (int r = r0; r < r1; r += 8) {
      float rew[8], done[8], ret[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const bool ok = r + u < r1;
        rew[u] = ok ? g.rows[(size_t)(r + u) * g.row_stride + D] : 0.f;
        done[u] = ok ? g.rows[(size_t)(r + u) * g.row_stride + D + 1] : 0.f;
        ret[u] = ok ? g.ret[r + u] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (r + u < r1) {
          const float nr = ret[u] * g.gamma + rew[u];         // VecNormalize: ret = ret * gamma + rews (f32, as numpy's array)
          const double d = (double)nr - shift;
          s += d; ss += d * d; sr += (double)rew[u];
          g.ret[r + u] = done[u] != 0.f ? 0.f : nr;           // ... ret[news] = 0 after the statistics saw it
          if (g.raw_rew_out) g.raw_rew_out[r + u] = rew[u];
          if (g.done_out) g.done_out[r + u] = done[u];
        }
    }
  }
  red[grp][c][0] = s; red[grp][c][1] = ss; red[grp][c][2] = sr;
  __syncthreads();
  if (grp == 0 && c <= D) {     // the row groups of this workgroup, in order
    double a = 0.0, b = 0.0, d = 0.0;
#pragma unroll
    for (int q = 0; q < OBS_GROUPS; q++) { a += red[q][c][0]; b += red[q][c][1]; d += red[q][c][2]; }
    double *p = g.partial + ((size_t)blockIdx.x * (D + 2) + c) * 2;
    p[0] = a; p[1] = b;
    if (c == D) { p[2] = d; p[3] = 0.0; }
  }
  return;
  }
  // ---- MERGE: the partials in workgroup order (fixed: deterministic)
  // row group q sums the workgroups q, q + 4, ... (8 loads in flight per trip), then group 0 adds the four in order
  if (c <= D) {
    for (int w = grp; w < G; w += 8 * OBS_GROUPS) {
      double a[8], b[8], d[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int wu = w + u * OBS_GROUPS;
        const double *p = g.partial + ((size_t)(wu < G ? wu : 0) * (D + 2) + c) * 2;
        a[u] = wu < G ? p[0] : 0.0; b[u] = wu < G ? p[1] : 0.0; d[u] = (wu < G && c == D) ? p[2] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) { s += a[u]; ss += b[u]; sr += d[u]; }
    }
  }
  __syncthreads();
  red[grp][c][0] = s; red[grp][c][1] = ss; red[grp][c][2] = sr;
  __syncthreads();
  if (grp == 0 && (c < D || (c == D && g.with_reward))) {
    s = 0.0; ss = 0.0; sr = 0.0;
#pragma unroll
    for (int q = 0; q < OBS_GROUPS; q++) { s += red[q][c][0]; ss += red[q][c][1]; sr += red[q][c][2]; }
    const int im = c < D ? c : 2 * D + 1, iv = c < D ? D + c : 2 * D + 2, ic = c < D ? 2 * D : 2 * D + 3;
    const double mean = g.stats[im], var = g.stats[iv], count = g.stats[ic];
    const double bc = (double)g.n;
    const double bm_rel = s / bc;                       // batch mean relative to the running mean = "delta"
    const double bv = ss / bc - bm_rel * bm_rel;        // population variance of the batch
    const double tot = count + bc;
    const double m2 = var * count + bv * bc + bm_rel * bm_rel * count * bc / tot;
    const double nmean = mean + bm_rel * bc / tot, nvar = m2 / tot;
    g.stats[im] = nmean; g.stats[iv] = nvar;
    if (c < D) {
      g.norm[c] = (float)nmean;
      g.norm[D + c] = (float)(1.0 / sqrt(nvar + (double)g.epsilon));
    } else {
      const float sc = (float)(1.0 / sqrt(nvar + (double)g.epsilon));
      g.norm[2 * D] = sc;
      if (g.rew_scale_out) *g.rew_scale_out = sc;
      g.stats[2 * D + 4] += sr;
    }
  }
  __syncthreads();
  if (tid == 0) {   // counts last: every column read the old one above
    g.stats[2 * D] += (double)g.n;
    if (g.with_reward) g.stats[2 * D + 3] += (double)g.n;
  }
}

__global__ void refresh_norm_kernel(const double *stats, float *norm, int D, float epsilon) {
  const int t = threadIdx.x;
  if (t < D) { norm[t] = (float)stats[t]; norm[D + t] = (float)(1.0 / sqrt(stats[D + t] + (double)epsilon)); }
  if (t == D) norm[2 * D] = (float)(1.0 / sqrt(stats[2 * D + 2] + (double)epsilon));
}

// ---------------------------------------------------------------- GAE
__global__ void gae_kernel(const float *raw_rew, const float *scale, const float *done, const float *val, float *adv, float *ret,
                           int T, int n, float gamma, float lam, float clip_rew) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float last = 0.f, nextv = val[(size_t)T * n + e];
  for (int t = T - 1; t >= 0; t--) {
    const size_t i = (size_t)t * n + e;
    const float r = fminf(fmaxf(raw_rew[i] * scale[t], -clip_rew), clip_rew);
    const float nonterm = 1.f - done[i], v = val[i];
    const float delta = r + gamma * nextv * nonterm - v;
    last = delta + gamma * lam * nonterm * last;
    adv[i] = last;
    ret[i] = last + v;
    nextv = v;
  }
}

// ---------------------------------------------------------------- clip + Adam (TensorFlow's form), one workgroup
__global__ __launch_bounds__(1024) void adam_kernel(float *theta, float *grad, float *m, float *v, int P, int *step, float lr,
                                                    float b1, float b2, float eps, float max_norm, float *norm_out) {
  __shared__ double red[16];
  __shared__ float lr_sh;
  const int tid = threadIdx.x;
  constexpr int MAXV = 8;                      // float4s per thread: P <= 32768
  const int n4 = P >> 2;
  float4 gv[MAXV];
  const float4 *g4 = reinterpret_cast<const float4 *>(grad);
  double s = 0.0;
#pragma unroll
  for (int u = 0; u < MAXV; u++) {             // every load of the thread in flight together
    const int i = tid + 1024 * u;
    gv[u] = i < n4 ? g4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < MAXV; u++)
    s += (double)gv[u].x * gv[u].x + (double)gv[u].y * gv[u].y + (double)gv[u].z * gv[u].z + (double)gv[u].w * gv[u].w;
  for (int i = (n4 << 2) + tid; i < P; i += 1024) s += (double)grad[i] * grad[i];
  // wave sums by shuffles, then the 16 waves in order
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  if (tid == 0) {
    const int t = *step + 1;
    lr_sh = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
  }
  __syncthreads();
  double tot = 0.0;
#pragma unroll
  for (int w = 0; w < 16; w++) tot += red[w];
  const float norm = (float)sqrt(tot);
  const float scale = max_norm > 0.f ? max_norm / fmaxf(norm, max_norm) : 1.f;    // tf.clip_by_global_norm
  const float lr_t = lr_sh;
  float4 *t4 = reinterpret_cast<float4 *>(theta), *m4 = reinterpret_cast<float4 *>(m), *v4 = reinterpret_cast<float4 *>(v);
  float4 *gw = reinterpret_cast<float4 *>(grad);
  float4 tv[MAXV], mv[MAXV], vv[MAXV];
#pragma unroll
  for (int u = 0; u < MAXV; u++) {
    const int i = tid + 1024 * u;
    if (i < n4) { tv[u] = t4[i]; mv[u] = m4[i]; vv[u] = v4[i]; }
  }
  auto upd = [&](float g, float &mi, float &vi, float &th) {
    const float gi = g * scale;
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    th -= lr_t * mi / (sqrtf(vi) + eps);
  };
#pragma unroll
  for (int u = 0; u < MAXV; u++) {
    const int i = tid + 1024 * u;
    if (i < n4) {
      upd(gv[u].x, mv[u].x, vv[u].x, tv[u].x); upd(gv[u].y, mv[u].y, vv[u].y, tv[u].y);
      upd(gv[u].z, mv[u].z, vv[u].z, tv[u].z); upd(gv[u].w, mv[u].w, vv[u].w, tv[u].w);
      t4[i] = tv[u]; m4[i] = mv[u]; v4[i] = vv[u]; gw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  for (int i = (n4 << 2) + tid; i < P; i += 1024) {
    float mi = m[i], vi = v[i], th = theta[i];
    upd(grad[i], mi, vi, th);
    m[i] = mi; v[i] = vi; theta[i] = th; grad[i] = 0.f;
  }
  if (tid == 0) { *step = *step + 1; if (norm_out) *norm_out = norm; }
}

}  // namespace

// ---------------------------------------------------------------- C-ABI
namespace {
int init_stats(TrexPolicy *p, hipStream_t s) {
  // RunningMeanStd(epsilon = 1e-4): mean 0, var 1, count 1e-4 (both)
  std::vector<double> st(2 * p->D + 5, 0.0);
  for (int k = 0; k < p->D; k++) st[p->D + k] = 1.0;
  st[2 * p->D] = 1e-4; st[2 * p->D + 2] = 1.0; st[2 * p->D + 3] = 1e-4;
  HIP_TRY(hipMemcpyAsync(p->stats, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  hipLaunchKernelGGL(refresh_norm_kernel, dim3(1), dim3(128), 0, s, p->stats, p->norm, p->D, p->epsilon);
  HIP_TRY(hipGetLastError());
  return TREX_OK;
}
}  // namespace

extern "C" {

int trex_policy_create(int num_envs, int obs_dim, int act_dim, int hidden, int device, TrexPolicy **out) {
  if (!out) return trex_fail(TREX_E_INVALID, "trex_policy_create: null argument");
  *out = nullptr;
  if (num_envs <= 0) return trex_fail(TREX_E_INVALID, "num_envs must be positive");
  if (hidden != HID) return trex_fail(TREX_E_UNSUPPORTED, "the policy kernel is written for hidden = 64 (baselines' MlpPolicy)");
  if (act_dim < 1 || act_dim > 32) return trex_fail(TREX_E_UNSUPPORTED, "act_dim must be in [1, 32]");
  if (obs_dim < 1 || obs_dim > MAXD - 2) return trex_fail(TREX_E_UNSUPPORTED, "obs_dim must be in [1, 126]");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
    return trex_fail(TREX_E_HIP, "no HIP device available (the policy step has no CPU fallback)");
  if (device < 0 || device >= count) return trex_fail(TREX_E_INVALID, "device index out of range");
  TrexDeviceGuard guard(device);
  if (!guard.ok) return trex_fail(TREX_E_HIP, "hipSetDevice failed");
  auto p = std::make_unique<TrexPolicy>();
  p->n = num_envs; p->D = obs_dim; p->A = act_dim; p->device = device;
  p->G = (num_envs + OBS_ROWS - 1) / OBS_ROWS;
  p->lay = make_layout(obs_dim, act_dim);
  hipError_t r = hipSuccess;
  auto A = [&](size_t bytes, void **q) {
    if (r != hipSuccess) return;
    r = hipMalloc(q, bytes);
    if (r == hipSuccess) { p->allocs.push_back(*q); r = hipMemset(*q, 0, bytes); }
  };
  A((2 * obs_dim + 5) * sizeof(double), (void **)&p->stats);
  A((size_t)p->G * (obs_dim + 2) * 2 * sizeof(double), (void **)&p->partial);
  A((2 * obs_dim + 2) * sizeof(float), (void **)&p->norm);
  A((size_t)num_envs * sizeof(float), (void **)&p->ret);
  A(sizeof(unsigned), (void **)&p->counter);
  A(sizeof(int), (void **)&p->adam_step);
  if (r != hipSuccess) {
    for (void *q : p->allocs) (void)hipFree(q);
    return trex_fail(TREX_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(r));
  }
  if (int c = init_stats(p.get(), nullptr)) { for (void *q : p->allocs) (void)hipFree(q); return c; }
  HIP_TRY(hipDeviceSynchronize());
  *out = p.release();
  return TREX_OK;
}

void trex_policy_destroy(TrexPolicy *p) {
  if (!p) return;
  TrexDeviceGuard guard(p->device);
  (void)hipDeviceSynchronize();
  for (void *q : p->allocs) (void)hipFree(q);
  delete p;
}

int trex_policy_param_count(const TrexPolicy *p) { return p ? p->lay.count : trex_fail(TREX_E_INVALID, "null policy"); }

int trex_policy_param_offsets(const TrexPolicy *p, int o[13]) {
  if (!p || !o) return trex_fail(TREX_E_INVALID, "null argument");
  const Layout &l = p->lay;
  const int v[13] = {l.pW1, l.pb1, l.pW2, l.pb2, l.pW3, l.pb3, l.vW1, l.vb1, l.vW2, l.vb2, l.vW3, l.vb3, l.logstd};
  std::memcpy(o, v, sizeof v);
  return TREX_OK;
}

int trex_policy_get_stats(TrexPolicy *p, double *host, void *stream) {
  if (!p || !host) return trex_fail(TREX_E_INVALID, "null argument");
  TrexDeviceGuard guard(p->device);
  HIP_TRY(hipMemcpyAsync(host, p->stats, (2 * p->D + 5) * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return TREX_OK;
}

int trex_policy_set_stats(TrexPolicy *p, const double *host, void *stream) {
  if (!p || !host) return trex_fail(TREX_E_INVALID, "null argument");
  TrexDeviceGuard guard(p->device);
  HIP_TRY(hipMemcpyAsync(p->stats, host, (2 * p->D + 5) * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  hipLaunchKernelGGL(refresh_norm_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, p->stats, p->norm, p->D, p->epsilon);
  HIP_TRY(hipGetLastError());
  return TREX_OK;
}

int trex_policy_get_returns(TrexPolicy *p, float *ret_dev, void *stream) {
  if (!p || !ret_dev) return trex_fail(TREX_E_INVALID, "null argument");
  TrexDeviceGuard guard(p->device);
  BUF_TRY(ret_dev, (size_t)p->n * sizeof(float), "trex_policy_get_returns: ret");
  HIP_TRY(hipMemcpyAsync(ret_dev, p->ret, (size_t)p->n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return TREX_OK;
}

int trex_policy_observe(TrexPolicy *p, const float *rows_dev, int row_stride, int with_reward, float gamma, float *raw_rew_out,
                        float *done_out, float *rew_scale_out, void *stream) {
  if (!p || !rows_dev) return trex_fail(TREX_E_INVALID, "trex_policy_observe: null argument");
  if (row_stride < p->D + (with_reward ? 2 : 0)) return trex_fail(TREX_E_INVALID, "trex_policy_observe: row_stride too small");
  TrexDeviceGuard guard(p->device);
  const size_t n = (size_t)p->n;
  BUF_TRY(rows_dev, ((n - 1) * row_stride + p->D + (with_reward ? 2 : 0)) * sizeof(float), "trex_policy_observe: rows");
  BUF_TRY(raw_rew_out, n * sizeof(float), "trex_policy_observe: raw_rew_out");
  BUF_TRY(done_out, n * sizeof(float), "trex_policy_observe: done_out");
  BUF_TRY(rew_scale_out, sizeof(float), "trex_policy_observe: rew_scale_out");
  ObserveArgs a{rows_dev, p->stats, p->partial, p->norm, p->ret, raw_rew_out, done_out, rew_scale_out, p->counter,
                p->n, row_stride, p->D, with_reward ? 1 : 0, gamma, p->epsilon};
  hipLaunchKernelGGL(observe_kernel<false>, dim3(p->G), dim3(128 * OBS_GROUPS), 0, (hipStream_t)stream, a, p->G);
  hipLaunchKernelGGL(observe_kernel<true>, dim3(1), dim3(128 * OBS_GROUPS), 0, (hipStream_t)stream, a, p->G);
  HIP_TRY(hipGetLastError());
  return TREX_OK;
}

int trex_policy_act(TrexPolicy *p, const float *theta_dev, const float *rows_dev, int row_stride, float clip_obs,
                    const float *noise_dev, float *actions_dev, float *obs_out, float *act_out, float *logp_out, float *value_out,
                    int value_only, void *stream) {
  if (!p || !theta_dev || !rows_dev) return trex_fail(TREX_E_INVALID, "trex_policy_act: null argument");
  if (!value_only && (!noise_dev || !actions_dev)) return trex_fail(TREX_E_INVALID, "trex_policy_act: noise / actions are null");
  if (value_only && !value_out) return trex_fail(TREX_E_INVALID, "trex_policy_act: value_out is null");
  if (row_stride < p->D) return trex_fail(TREX_E_INVALID, "trex_policy_act: row_stride < obs_dim");
  TrexDeviceGuard guard(p->device);
  const size_t n = (size_t)p->n;
  BUF_TRY(theta_dev, (size_t)p->lay.count * sizeof(float), "trex_policy_act: theta");
  BUF_TRY(rows_dev, ((n - 1) * row_stride + p->D) * sizeof(float), "trex_policy_act: rows");
  BUF_TRY(noise_dev, n * p->A * sizeof(float), "trex_policy_act: noise");
  BUF_TRY(actions_dev, n * p->A * sizeof(float), "trex_policy_act: actions");
  BUF_TRY(obs_out, n * p->D * sizeof(float), "trex_policy_act: obs_out");
  BUF_TRY(act_out, n * p->A * sizeof(float), "trex_policy_act: act_out");
  BUF_TRY(logp_out, n * sizeof(float), "trex_policy_act: logp_out");
  BUF_TRY(value_out, n * sizeof(float), "trex_policy_act: value_out");
  ActArgs a{theta_dev, rows_dev, p->norm, noise_dev, actions_dev, obs_out, act_out, logp_out, value_out,
            p->n, row_stride, value_only ? 1 : 0, clip_obs, p->lay};
  const size_t lds_bytes = (size_t)act_lds_floats(p->lay) * sizeof(float);
  hipLaunchKernelGGL(act_kernel, dim3((p->n + TILE - 1) / TILE, 2), dim3(128), lds_bytes, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return TREX_OK;
}

int trex_policy_gae(TrexPolicy *p, const float *raw_rew_dev, const float *rew_scale_dev, const float *done_dev,
                    const float *values_dev, float *adv_dev, float *ret_dev, int T, float gamma, float lam, float clip_rew,
                    void *stream) {
  if (!p || !raw_rew_dev || !rew_scale_dev || !done_dev || !values_dev || !adv_dev || !ret_dev || T <= 0)
    return trex_fail(TREX_E_INVALID, "trex_policy_gae: bad argument");
  TrexDeviceGuard guard(p->device);
  const size_t n = (size_t)p->n, tn = (size_t)T * n * sizeof(float);
  BUF_TRY(raw_rew_dev, tn, "trex_policy_gae: raw_rew");
  BUF_TRY(rew_scale_dev, (size_t)T * sizeof(float), "trex_policy_gae: rew_scale");
  BUF_TRY(done_dev, tn, "trex_policy_gae: done");
  BUF_TRY(values_dev, tn + n * sizeof(float), "trex_policy_gae: values");
  BUF_TRY(adv_dev, tn, "trex_policy_gae: adv");
  BUF_TRY(ret_dev, tn, "trex_policy_gae: ret");
  hipLaunchKernelGGL(gae_kernel, dim3((p->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw_rew_dev, rew_scale_dev, done_dev,
                     values_dev, adv_dev, ret_dev, T, p->n, gamma, lam, clip_rew);
  HIP_TRY(hipGetLastError());
  return TREX_OK;
}

int trex_policy_adam(TrexPolicy *p, float *theta_dev, float *grad_dev, float *m_dev, float *v_dev, float lr, float beta1,
                     float beta2, float eps, float max_grad_norm, float *grad_norm_out, void *stream) {
  if (!p || !theta_dev || !grad_dev || !m_dev || !v_dev) return trex_fail(TREX_E_INVALID, "trex_policy_adam: null argument");
  TrexDeviceGuard guard(p->device);
  const size_t bytes = (size_t)p->lay.count * sizeof(float);
  BUF_TRY(theta_dev, bytes, "trex_policy_adam: theta");
  BUF_TRY(grad_dev, bytes, "trex_policy_adam: grad");
  BUF_TRY(m_dev, bytes, "trex_policy_adam: m");
  BUF_TRY(v_dev, bytes, "trex_policy_adam: v");
  BUF_TRY(grad_norm_out, sizeof(float), "trex_policy_adam: grad_norm_out");
  hipLaunchKernelGGL(adam_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, theta_dev, grad_dev, m_dev, v_dev, p->lay.count,
                     p->adam_step, lr, beta1, beta2, eps, max_grad_norm, grad_norm_out);
  HIP_TRY(hipGetLastError());
  return TREX_OK;
}

int trex_policy_adam_reset(TrexPolicy *p, void *stream) {
  if (!p) return trex_fail(TREX_E_INVALID, "null policy");
  TrexDeviceGuard guard(p->device);
  HIP_TRY(hipMemsetAsync(p->adam_step, 0, sizeof(int), (hipStream_t)stream));
  return TREX_OK;
}

}  // extern "C"
