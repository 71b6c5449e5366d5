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
//   act_kernel       one workgroup = 32 envs x 2 waves (policy net, value net). The normalised observation tile is
//                    staged in LDS; the 75 -> 64 -> 64 -> {25, 1} tanh MLPs run on the matrix cores with
//                    v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: exact f32, the dense contraction of this path)
//                    in the TRANSPOSED form H^T = W^T X^T: the result tile has the env on the lane and the neurons in
//                    the 16 accumulator registers, which is exactly the B operand of the next layer's MFMA - the
//                    activations never leave the registers (no LDS round trip, no barrier between layers).
//                    Weights are read [in][out] from the flat parameter vector: 32 consecutive floats per half wave.
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

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HID = 64;        // hidden width (two 32-row MFMA tiles)
constexpr int TILE = 32;       // envs per workgroup of act_kernel
constexpr int MAXD = 128;      // LDS row of the observation tile (obs_dim + pad <= MAXD)
constexpr int OBS_ROWS = 64;   // rows per workgroup of observe_kernel
constexpr float LOG_2PI = 1.8378770664093453f;

struct Layout {   // offsets into theta, trex_policy.h order
  int D, A;
  int pW1, pb1, pW2, pb2, pW3, pb3, vW1, vb1, vW2, vb2, vW3, vb3, logstd, count;
};

__host__ __device__ inline Layout make_layout(int D, int A) {
  Layout l{};
  l.D = D; l.A = A;
  int o = 0;
  l.pW1 = o; o += D * HID; l.pb1 = o; o += HID; l.pW2 = o; o += HID * HID; l.pb2 = o; o += HID; l.pW3 = o; o += HID * A; l.pb3 = o; o += A;
  l.vW1 = o; o += D * HID; l.vb1 = o; o += HID; l.vW2 = o; o += HID * HID; l.vb2 = o; o += HID; l.vW3 = o; o += HID; l.vb3 = o; o += 1;
  l.logstd = o; o += A;
  l.count = o;
  return l;
}

// row of accumulator register `reg` on a lane of half h = lane >> 5 (C/D layout of the 32x32 MFMA forms)
__device__ __forceinline__ constexpr int rowmap(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

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
// of MFMA step (t, s) is the lane's OWN register, and the A operand carries the matching k.
__device__ __forceinline__ void hidden_layer(const float *__restrict__ W, const float *__restrict__ bias, const f32x16 (&in)[2],
                                             f32x16 (&out)[2], int col, int h) {
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
    for (int r = 0; r < 16; r++) out[u][r] = tanhf(out[u][r]);
  }
}

__global__ __launch_bounds__(128) void act_kernel(ActArgs g) {
  __shared__ float X[TILE][MAXD + 1];   // normalised, clipped observations of the tile (+ zero pad column)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;   // wave 0: policy net, wave 1: value net
  const int e0 = blockIdx.x * TILE;
  const int D = g.lay.D, A = g.lay.A;
  const int Dp = (D + 1) & ~1;          // K of the first layer, padded to the MFMA's k = 2
  // ---- stage the tile: normalise, clip, keep a copy for the rollout buffer
  for (int idx = tid; idx < TILE * Dp; idx += 128) {
    const int i = idx / Dp, k = idx - i * Dp, e = e0 + i;
    float x = 0.f;
    if (k < D && e < g.n) {
      x = (g.rows[(size_t)e * g.row_stride + k] - g.norm[k]) * g.norm[D + k];
      x = fminf(fmaxf(x, -g.clip_obs), g.clip_obs);
      if (g.obs_out && !g.value_only) g.obs_out[(size_t)e * D + k] = x;
    }
    X[i][k] = x;
  }
  __syncthreads();
  if (g.value_only && wave == 0) return;
  const int col = lane & 31, h = lane >> 5;
  const float *th = g.theta;
  const float *W1 = th + (wave ? g.lay.vW1 : g.lay.pW1), *b1 = th + (wave ? g.lay.vb1 : g.lay.pb1);
  const float *W2 = th + (wave ? g.lay.vW2 : g.lay.pW2), *b2 = th + (wave ? g.lay.vb2 : g.lay.pb2);
  // ---- layer 1: h1^T = tanh(W1^T x^T + b1); B operand = the observation of env `col` from LDS
  f32x16 h1[2], h2[2];
#pragma unroll
  for (int u = 0; u < 2; u++) {
#pragma unroll
    for (int r = 0; r < 16; r++) h1[u][r] = b1[32 * u + rowmap(r, h)];
  }
  for (int kk = 0; kk < Dp; kk += 2) {
    const int k = kk + h;
    const float b = X[col][k];                       // (k = D on the pad column: 0)
    const int kc = k < D ? k : D - 1;                // (its weight row does not exist: any finite value, times 0)
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const float a = W1[kc * HID + 32 * u + col];
      h1[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, h1[u], 0, 0, 0);
    }
  }
#pragma unroll
  for (int u = 0; u < 2; u++) {
#pragma unroll
    for (int r = 0; r < 16; r++) h1[u][r] = tanhf(h1[u][r]);
  }
  // ---- layer 2, activations from registers
  hidden_layer(W2, b2, h1, h2, col, h);
  const int e = e0 + col;
  if (wave == 1) {
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
        const float nz = g.noise[(size_t)e * A + a];
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

__global__ __launch_bounds__(128) void observe_kernel(ObserveArgs g) {
  const int tid = threadIdx.x, D = g.D, G = gridDim.x;
  const int r0 = blockIdx.x * OBS_ROWS, r1 = min(r0 + OBS_ROWS, g.n);
  // thread c < D: observation column c. thread D: the discounted returns. thread D + 1: the raw rewards (logging).
  // Sums are taken about the RUNNING mean (a shift that every workgroup knows): no cancellation in the variance.
  if (tid < D) {
    const double shift = g.stats[tid];
    double s = 0.0, ss = 0.0;
    for (int r = r0; r < r1; r++) {
      const double x = (double)g.rows[(size_t)r * g.row_stride + tid] - shift;
      s += x; ss += x * x;
    }
    double *p = g.partial + ((size_t)blockIdx.x * (D + 2) + tid) * 2;
    p[0] = s; p[1] = ss;
  } else if (tid == D && g.with_reward) {
    const double shift = g.stats[2 * D + 1];
    double s = 0.0, ss = 0.0, sr = 0.0;
    for (int r = r0; r < r1; r++) {
      const float rew = g.rows[(size_t)r * g.row_stride + D], done = g.rows[(size_t)r * g.row_stride + D + 1];
      const float ret = g.ret[r] * g.gamma + rew;          // VecNormalize: ret = ret * gamma + rews (f32, as numpy's array)
      const double x = (double)ret - shift;
      s += x; ss += x * x; sr += (double)rew;
      g.ret[r] = done != 0.f ? 0.f : ret;                  // ... ret[news] = 0 after the statistics saw it
      if (g.raw_rew_out) g.raw_rew_out[r] = rew;
      if (g.done_out) g.done_out[r] = done;
    }
    double *p = g.partial + ((size_t)blockIdx.x * (D + 2) + D) * 2;
    p[0] = s; p[1] = ss; p[2] = sr; p[3] = 0.0;
  }
  // ---- the last workgroup to end merges the partials, in workgroup order
  __shared__ bool last;
  __threadfence();
  __syncthreads();
  if (tid == 0) last = atomicAdd(g.counter, 1u) == (unsigned)(G - 1);
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (tid < D || (tid == D && g.with_reward)) {
    double s = 0.0, ss = 0.0;
    for (int w = 0; w < G; w++) {
      const double *p = g.partial + ((size_t)w * (D + 2) + tid) * 2;
      s += p[0]; ss += p[1];
    }
    const int im = tid < D ? tid : 2 * D + 1, iv = tid < D ? D + tid : 2 * D + 2, ic = tid < D ? 2 * D : 2 * D + 3;
    const double mean = g.stats[im], var = g.stats[iv], count = g.stats[ic];
    const double bc = (double)g.n;
    const double bm_rel = s / bc;                       // batch mean relative to the running mean = "delta"
    const double bv = ss / bc - bm_rel * bm_rel;        // population variance of the batch
    const double tot = count + bc;
    const double m2 = var * count + bv * bc + bm_rel * bm_rel * count * bc / tot;
    const double nmean = mean + bm_rel * bc / tot, nvar = m2 / tot;
    g.stats[im] = nmean; g.stats[iv] = nvar;
    if (tid < D) {
      g.norm[tid] = (float)nmean;
      g.norm[D + tid] = (float)(1.0 / sqrt(nvar + (double)g.epsilon));
    } else {
      const float sc = (float)(1.0 / sqrt(nvar + (double)g.epsilon));
      g.norm[2 * D] = sc;
      if (g.rew_scale_out) *g.rew_scale_out = sc;
      double sr = 0.0;
      for (int w = 0; w < G; w++) sr += g.partial[((size_t)w * (D + 2) + D) * 2 + 2];
      g.stats[2 * D + 4] += sr;
    }
  }
  __syncthreads();
  if (tid == 0) {   // counts last: every column read the old one above
    g.stats[2 * D] += (double)g.n;
    if (g.with_reward) g.stats[2 * D + 3] += (double)g.n;
    *g.counter = 0u;
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
  __shared__ double red[1024];
  const int tid = threadIdx.x;
  double s = 0.0;
  for (int i = tid; i < P; i += 1024) { const double x = grad[i]; s += x * x; }
  red[tid] = s;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  const float norm = (float)sqrt(red[0]);
  const float scale = max_norm > 0.f ? max_norm / fmaxf(norm, max_norm) : 1.f;    // tf.clip_by_global_norm
  const int t = *step + 1;
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t));
  for (int i = tid; i < P; i += 1024) {
    const float gi = grad[i] * scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    theta[i] -= (float)lr_t * mi / (sqrtf(vi) + eps);
    grad[i] = 0.f;
  }
  __syncthreads();
  if (tid == 0) { *step = t; if (norm_out) *norm_out = norm; }
}

}  // namespace

// ---------------------------------------------------------------- C-ABI
struct TrexPolicy {
  int n = 0, D = 0, A = 0, device = 0, G = 0;
  Layout lay{};
  double *stats = nullptr, *partial = nullptr;
  float *norm = nullptr, *ret = nullptr;
  unsigned *counter = nullptr;
  int *adam_step = nullptr;
  float epsilon = 1e-8f;        // VecNormalize's
  std::vector<void *> allocs;
  std::vector<TrexSeen> seen;
};

namespace {
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) return trex_fail(TREX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
#define BUF_TRY(ptr, bytes, what)                                                                       \
  do {                                                                                                  \
    if (int _c = trex_check_device_buffer(p->device, p->seen, (ptr), (size_t)(bytes), (what))) return _c; \
  } while (0)

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
  hipLaunchKernelGGL(observe_kernel, dim3(p->G), dim3(128), 0, (hipStream_t)stream, a);
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
  hipLaunchKernelGGL(act_kernel, dim3((p->n + TILE - 1) / TILE), dim3(128), 0, (hipStream_t)stream, a);
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
