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
//   act_kernel       one workgroup = 2 tiles of 32 envs x 2 waves (policy net, value net). The parameter vector (84 KB)
//                    and the normalised observation tiles are staged in LDS; the 75 -> 64 -> 64 -> {25, 1} tanh MLPs run on the matrix cores with
//                    v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: exact f32, the dense contraction of this path)
//                    in the TRANSPOSED form H^T = W^T X^T: the result tile has the env on the lane and the neurons in
//                    the 16 accumulator registers, which is exactly the B operand of the next layer's MFMA - the
//                    activations never leave the registers (no LDS round trip, no barrier between layers).
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

// hidden layer in transposed form: out^T[32 u + row][env] = b[..] + sum_k W[k][32 u + row] * in^T[k][env], in^T given as
// two accumulator tiles (k = 32 t + rowmap(s, h) sits in register s of tile t on the lanes of half h): the B operand
// of MFMA step (t, s) is the lane's OWN register, and the A operand (from the LDS copy of W) carries the matching k.
__device__ __forceinline__ void hidden_layer(const float *W, const float *bias, const f32x16 (&in)[2], f32x16 (&out)[2], int col, int h) {
#pragma unroll
  for (int u = 0; u < 2; u++) {
#pragma unroll
    for (int r = 0; r < 16; r++) out[u][r] = bias[32 * u + rowmap(r, h)];
  }
#pragma unroll
  for (int t = 0; t < 2; t++) {
#pragma unroll
    for (int s = 0; s < 16; s++) {
      const int k = 32 * t + rowmap(s, h);
      const float b = in[t][s];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const float a = W[k * HID + 32 * u + col];
        out[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, out[u], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 2; u++) {
#pragma unroll
    for (int r = 0; r < 16; r++) out[u][r] = tanh_fast(out[u][r]);
  }
}

// One workgroup = 4 waves = 2 tiles of 32 envs x {policy net, value net}. The whole parameter vector (21 k floats,
// 84 KB) is copied to LDS once per workgroup with 16-byte loads that are all in flight together - read from L2 inside
// the MFMA loops, every A operand was a dependent round trip of its own (32 us per launch; 172 MFMAs per wave).
constexpr int ACT_TILES = 2;
__global__ __launch_bounds__(256) void act_kernel(ActArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int net = wave & 1, tile = wave >> 1;          // net 0: policy, 1: value
  const int D = g.lay.D, A = g.lay.A;
  const int Dp8 = (D + 7) & ~7;         // K of the first layer, padded to four MFMA k-steps (k = 2 each); X is zero from column D on
  float *th = lds;                                       // [count] parameters
  float *Xall = lds + ((g.lay.count + 3) & ~3);          // [ACT_TILES][TILE][MAXD + 1] normalised observations (+ zero pad)
  {
    // (11 loads per thread in flight per trip: left as a plain loop, every 16-byte load waited for its own round trip)
    const float4 *src = reinterpret_cast<const float4 *>(g.theta);
    float4 *dst = reinterpret_cast<float4 *>(th);
    const int n4 = g.lay.count >> 2;
    for (int i0 = tid; i0 < n4; i0 += 256 * 11) {     // 21 float4 per thread: two trips of 11
      float4 t[11];
#pragma unroll
      for (int u = 0; u < 11; u++) { const int i = i0 + 256 * u; t[u] = src[i < n4 ? i : 0]; }
#pragma unroll
      for (int u = 0; u < 11; u++) { const int i = i0 + 256 * u; if (i < n4) dst[i] = t[u]; }
    }
    for (int i = (n4 << 2) + tid; i < g.lay.count; i += 256) th[i] = g.theta[i];
  }
  // ---- stage the tiles: normalise, clip, keep a copy for the rollout buffer
  const int eb = blockIdx.x * (ACT_TILES * TILE);
  constexpr int UX = 10;      // 19 elements per thread at D = 75: two trips
  for (int idx0 = tid; idx0 < ACT_TILES * TILE * Dp8; idx0 += 256 * UX) {
    float raw[UX], mean[UX], rstd[UX];
    int ii[UX], kk[UX];
    bool ok[UX];
#pragma unroll
    for (int u = 0; u < UX; u++) {
      const int idx = idx0 + 256 * u;
      ii[u] = idx / Dp8; kk[u] = idx - ii[u] * Dp8;
      ok[u] = idx < ACT_TILES * TILE * Dp8 && kk[u] < D && eb + ii[u] < g.n;
      const int kc = kk[u] < D ? kk[u] : 0;
      raw[u] = ok[u] ? g.rows[(size_t)(eb + ii[u]) * g.row_stride + kc] : 0.f;
      mean[u] = g.norm[kc]; rstd[u] = g.norm[D + kc];
    }
#pragma unroll
    for (int u = 0; u < UX; u++) {
      if (idx0 + 256 * u >= ACT_TILES * TILE * Dp8) continue;
      float x = 0.f;
      if (ok[u]) {
        x = fminf(fmaxf((raw[u] - mean[u]) * rstd[u], -g.clip_obs), g.clip_obs);
        if (g.obs_out && !g.value_only) g.obs_out[(size_t)(eb + ii[u]) * D + kk[u]] = x;
      }
      Xall[ii[u] * (MAXD + 1) + kk[u]] = x;
    }
  }
  __syncthreads();
  if (g.value_only && net == 0) return;
  const float *X = Xall + tile * TILE * (MAXD + 1);
  const int e0 = eb + tile * TILE;
  const int col = lane & 31, h = lane >> 5;
  const float *W1 = th + (net ? g.lay.vW1 : g.lay.pW1), *b1 = th + (net ? g.lay.vb1 : g.lay.pb1);
  const float *W2 = th + (net ? g.lay.vW2 : g.lay.pW2), *b2 = th + (net ? g.lay.vb2 : g.lay.pb2);
  // the sample's noise draws, fetched now: their round trip hides behind the MFMA phases
  const int e = e0 + col;
  f32x16 nz_in;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int a = rowmap(r, h);
    nz_in[r] = (net == 0 && !g.value_only && a < A && e < g.n) ? g.noise[(size_t)e * A + a] : 0.f;
  }
  // ---- layer 1: h1^T = tanh(W1^T x^T + b1); B operand = the observation of env `col` from LDS
  f32x16 h1[2], h2[2];
#pragma unroll
  for (int u = 0; u < 2; u++) {
#pragma unroll
    for (int r = 0; r < 16; r++) h1[u][r] = b1[32 * u + rowmap(r, h)];
  }
  // four k-steps per trip: their 12 LDS operand reads are issued together, then the 8 MFMAs (one k-step at a time,
  // every MFMA waited for its own LDS round trip: 3.3 us of the kernel)
  for (int k0 = 0; k0 < Dp8; k0 += 8) {
    float a0[4], a1[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = k0 + 2 * j + h;
      b[j] = X[col * (MAXD + 1) + k];                  // (zero from column D on)
      const int kc = k < D ? k : D - 1;                // (such a weight row does not exist: any finite value, times 0)
      a0[j] = W1[kc * HID + col]; a1[j] = W1[kc * HID + 32 + col];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      h1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b[j], h1[0], 0, 0, 0);
      h1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b[j], h1[1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int u = 0; u < 2; u++) {
#pragma unroll
    for (int r = 0; r < 16; r++) h1[u][r] = tanh_fast(h1[u][r]);
  }
  // ---- layer 2, activations from registers
  hidden_layer(W2, b2, h1, h2, col, h);
  if (net == 1) {
    // ---- value head: one output; every lane sums its 32 neurons, the two halves of an env meet by a swap
    const float *w = th + g.lay.vW3;
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
      for (int s = 0; s < 16; s++) v = __builtin_fmaf(h2[t][s], w[32 * t + rowmap(s, h)], v);
    }
    v += __shfl_xor(v, 32, 64);
    v += th[g.lay.vb3];
    if (h == 0 && e < g.n && g.value_out) g.value_out[e] = v;
    return;
  }
  // ---- policy head: mean^T[a][env] (a = rowmap(reg, h) < A), then the Gaussian sample and its log-probability
  const float *W3 = th + g.lay.pW3, *b3 = th + g.lay.pb3, *ls = th + g.lay.logstd;
  f32x16 mu;
#pragma unroll
  for (int r = 0; r < 16; r++) { const int a = rowmap(r, h); mu[r] = a < A ? b3[a] : 0.f; }
#pragma unroll
  for (int t = 0; t < 2; t++) {
#pragma unroll
    for (int s = 0; s < 16; s++) {
      const int k = 32 * t + rowmap(s, h);
      const float a = col < A ? W3[k * A + col] : 0.f;     // A operand: row = action `col`
      mu = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h2[t][s], mu, 0, 0, 0);
    }
  }
  float zz = 0.f, sum_ls = 0.f;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int a = rowmap(r, h);
    if (a < A) {
      const float l = ls[a], sd = expf(l);
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
  const size_t lds_bytes = (((size_t)p->lay.count + 3) & ~(size_t)3) * sizeof(float) + (size_t)ACT_TILES * TILE * (MAXD + 1) * sizeof(float);
  hipLaunchKernelGGL(act_kernel, dim3((p->n + ACT_TILES * TILE - 1) / (ACT_TILES * TILE)), dim3(256), lds_bytes, (hipStream_t)stream, a);
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
