// PPO2 minibatch step as three launches (include/trex_policy.h: trex_policy_minibatch_step).
//
// What baselines' ppo2 Model.train does per minibatch through TensorFlow (trex_train.py:49-61: forward of both
// MLPs, clipped surrogate + clipped value loss, backward, global-norm clip, Adam) costs a stock autograd framework ~90
// small kernels (0.66 ms per 4096-sample minibatch even replayed from a HIP graph). Here:
//
//   learn_grad_kernel   one workgroup = one tile of 32 samples of ONE net, 4 waves: a serial-chain wave and three helpers (see
//                       the kernel). Forward exactly as
//                       the rollout's act_kernel (transposed MFMA form, activations in registers). The backward pass
//                       needs two kinds of contractions: over NEURONS (delta^T = W . delta_next^T: the B operand is
//                       again the lane's own accumulator register, the A operand the LDS copy of W read transposed -
//                       W2 is staged with a row stride of 65 words so that this read is conflict-free) and over
//                       SAMPLES (grad W[k][j] = sum_env act[env][k] delta[env][j]: the sample index has to become
//                       the MFMA's k dimension, so the activation tile and the delta tile make ONE trip through a
//                       private LDS buffer of the wave, written [row][env] and read back with the env as k). Every
//                       tile writes its gradient contribution to its own slice of a partial buffer: no atomics.
//   learn_reduce_kernel sums the per-tile partials IN TILE ORDER (deterministic), adds the entropy term, writes the gradient
//                       and each workgroup's squared norm;
//   learn_adam_kernel   takes the global norm (workgroup order), clips and applies Adam in TensorFlow's form to the flat
//                       parameter vector, 256 float4s per workgroup.
//
// The arithmetic is the one restated in f64 by oracle/ppo_oracle.py::ppo_loss_and_grads / clip_by_global_norm / Adam.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>

#include "../../include/trex_policy.h"
#include "policy_common.h"

namespace {

constexpr int LEARN_UF = 13;           // float4s of theta per thread of learn_grad_kernel's staging trip
constexpr int W2S = HID + 1;           // LDS row stride of W2 (transposed reads in the backward pass)
constexpr int XS = MAXD + 1;           // LDS row stride of the observation tile
constexpr int TS = TILE + 1;           // LDS row stride of the [row][env] transposition buffers

// The LDS copy of theta MIRRORS the global layout (policy_common.h: every block on a multiple of 4 floats) except that the
// rows of the two W2 matrices are padded to W2S words: a global float at offset i sits at i + shift(i), where the shift
// grows by HID after each W2 (its 64 pad words) and, inside a W2, by one per row. One flat loop stages everything.
struct LdsLayout {
  int X, buf, vec, total;      // observation tile; parked tiles; small vectors; floats in all
};
__host__ __device__ inline LdsLayout make_lds_layout(const Layout &lay) {
  LdsLayout l{};
  int o = lay.count + 2 * HID;                         // theta + the pad words of the two W2s
  o = (o + 3) & ~3;
  l.X = o; o += TILE * XS;
  l.buf = o; o += (4 * HID + 2 * TILE) * TS;  // parked tiles [row][env]: H1, H2 [64] | D3 [32] | D2, D1 [64] | DL [32]
  l.vec = o; o += 128;                        // per-env loss terms [32] | 1 / sd per action [32] | sum of logstd [1]
  l.total = o;
  return l;
}
// LDS offset of the global parameter at offset i (i a multiple of 4 for the float4 path: a float4 never crosses a W2 row)
__device__ __forceinline__ int lds_of(const Layout &lay, int i) {
  int o = i;
  if (i >= lay.pW2) o += (i < lay.pb2) ? (i - lay.pW2) / HID : HID;
  if (i >= lay.vW2) o += (i < lay.vb2) ? (i - lay.vW2) / HID : HID;
  return o;
}

struct LearnArgs {
  const float *theta, *obs, *act, *logp0, *val0, *adv, *ret;
  const long long *perm;
  const float *adv_stats;      // [2]: mean and 1 / (std + 1e-8) of the minibatch's advantages
  float *partial;              // [tiles][stride]
  int first, mb, stride;
  float cliprange, vf_coef;
  Layout lay;
};

// sum over the 32 lanes of a half wave (same h), result on all of them: four DPP steps inside the 16-lane rows, then
// gfx950's v_permlane16_swap (rows 0<->1, 2<->3)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float half_sum(float v) {
  v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);   // row_half_mirror
  v += dpp_f<0x140>(v);   // row_mirror
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// One workgroup = one (tile of 32 samples, net): 4 waves, grid (tiles, 2) = 256 workgroups for a 4096-sample minibatch - the
// whole chip (round 3: 128 workgroups of 2 waves, half the CUs idle, 44 us). The SERIAL chain - forward, loss, delta3 ->
// delta2 -> delta1, each needing the one before - is run by waves 0 and 1, each owning one HALF of the 64 neurons of a layer
// (u = wave: 32 output rows = one MFMA tile; the other half of the previous layer arrives through LDS, where every
// activation and delta is parked as [row][env] the moment it exists); the weight gradients, which are independent outer
// products over the samples (grad W = act^T delta: the sample index is the MFMA's k), are taken by waves 2 and 3 as soon as
// their two operands are parked, beside the chain:
//   B1: H1 parked | B2: H2 parked | (wave 0: output layer, loss) | B3: D3 / dv parked
//   delta2 (waves 0, 1)  beside  grad W3 (policy: 2 tiles, waves 2, 3)      | B4: D2 parked
//   delta1 (waves 0, 1)  beside  grad W2 (4 tiles), grad b2 (waves 2, 3)    | B5: D1 parked
//   grad W1 (6 tiles, all four waves), grad b1
// Only this net's parameters are staged (by 256 threads: one trip of loads). Every tile writes its gradient slice to
// its own row of the partial buffer: no atomics, fixed order in the reduction.
#if TREX_LEARN_STAMPS     // diagnostic build only: s_memtime at the barriers of every workgroup's wave 0 (scripts/learn_phases.py)
__device__ unsigned long long learn_stamps[512 * 12];
#define LSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) learn_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 12 + (i)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define LSTAMP(i) do {} while (0)
#endif
__global__ __launch_bounds__(256) void learn_grad_kernel(LearnArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  LSTAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, net = blockIdx.y;     // net 0: policy, 1: value
  const int D = g.lay.D, A = g.lay.A;
  const LdsLayout L = make_lds_layout(g.lay);
  // ---- staging. Three global round trips feed the first barrier - the tile's sample rows (perm), this net's parameters, the
  // observation rows of those samples (which need the perm values) - and they are issued so that they overlap: perm first, the
  // parameters behind it, one barrier for the row table while the parameters are in flight, then the observations, and only
  // then the LDS writes of the parameters (loads return in order: the wait for the row table does not wait for theta).
  // The LDS image keeps theta's layout (lds_of; the other net's part stays unwritten): policy = [pW1, vW1) + logstd, value = [vW1, logstd)
  const int s0 = blockIdx.x * TILE;                       // first sample of the tile within the minibatch
  float *X = lds + L.X;
  constexpr int XC = 96;                                  // columns of X that the weight-gradient tiles read (3 x 32)
  __shared__ long long tile_rows[TILE];
  // (No control flow between the first load and the barrier that opens layer 1: loads under a branch make the compiler wait for
  // EVERY outstanding load at the join - s_waitcnt vmcnt(0) - and the overlap is gone. Out-of-range lanes read a clamped address
  // and select afterwards, out-of-range LDS writes go to a dump slot.)
  const long long my_row_raw = g.perm[g.first + min(s0 + min(tid, TILE - 1), g.mb - 1)];
  const long long my_row = (s0 + tid < g.mb) ? my_row_raw : -1;
  constexpr int UF = LEARN_UF;                            // 2 670 float4s by 256 threads: ONE trip (the host entry checks)
  float4 tp[UF];
  int src[UF];
  const float4 *s4 = reinterpret_cast<const float4 *>(g.theta);
  const int lo4 = (net ? g.lay.vW1 : 0) >> 2, hi4 = (net ? g.lay.logstd : g.lay.vW1) >> 2;
  const int ls4 = g.lay.logstd >> 2, le4 = g.lay.count >> 2;
  const int n4 = (hi4 - lo4) + (net ? 0 : le4 - ls4);
  // (measured, not kept: layer 1 started on W1 and b1 alone, the rest of the parameters landing during it and written to LDS
  // before the next barrier - no gain: the observation rows, which wait for the row table, arrive last either way)
#pragma unroll
  for (int u = 0; u < UF; u++) {
    const int i = tid + 256 * u;
    src[u] = i < hi4 - lo4 ? lo4 + i : ls4 + (i - (hi4 - lo4));
    src[u] = i < n4 ? src[u] : lo4;
  }
#pragma unroll
  for (int u = 0; u < UF; u++) tp[u] = s4[src[u]];
  LDS_ISSUED();
  if (tid < TILE) tile_rows[tid] = my_row;
  __syncthreads();
  LSTAMP(9);
  constexpr int UX = 12;                                  // TILE * XC / 256 = 12 elements per thread: one trip
  float raw[UX];
#pragma unroll
  for (int u = 0; u < UX; u++) {
    const int idx = tid + 256 * u, ii = idx / XC, kk = idx - ii * XC;
    const long long row = tile_rows[ii];
    const float v = g.obs[(size_t)(row >= 0 ? row : 0) * D + (kk < D ? kk : D - 1)];
    raw[u] = (kk < D && row >= 0) ? v : 0.f;
  }
  const int col = lane & 31, h = lane >> 5;
  const bool valid = s0 + col < g.mb;
  // the sample's loss inputs (wave 0 runs the output layer and the loss), fetched NOW by every wave (a branch around the loads
  // would cost the overlap, above; left inside `if (wave == 0)` the compiler also SANK them to their use after the second
  // barrier, a full round trip in the middle of the chain): their round trips hide behind the staging
  f32x16 act_in;
  float adv_in, logp0_in, val0_in, ret_in, advm, advr;
  {
    const long long sidx_raw = tile_rows[col];
    const long long sidx = sidx_raw >= 0 ? sidx_raw : 0;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int a = rowmap(r, h);
      act_in[r] = g.act[(size_t)sidx * A + (a < A ? a : A - 1)];
    }
    adv_in = g.adv[sidx]; logp0_in = g.logp0[sidx]; val0_in = g.val0[sidx]; ret_in = g.ret[sidx];
    advm = g.adv_stats[0]; advr = g.adv_stats[1];
  }
  float *VEC = lds + L.vec;             // [0..31] per-env loss term | [32..63] 1 / sd per action | [64] sum of logstd | [96..99] dump slot
  const float ls_raw = g.theta[g.lay.logstd + (lane < A ? lane : A - 1)];      // (policy net, wave 3 uses it) logstd of action `lane`
  LDS_ISSUED();
#pragma unroll
  for (int u = 0; u < UF; u++) {
    float *d = lds + (tid + 256 * u < n4 ? lds_of(g.lay, 4 * src[u]) : L.vec + 96);
    d[0] = tp[u].x; d[1] = tp[u].y; d[2] = tp[u].z; d[3] = tp[u].w;
  }
#pragma unroll
  for (int u = 0; u < UX; u++) {
    const int idx = tid + 256 * u, ii = idx / XC, kk = idx - ii * XC;
    X[ii * XS + kk] = raw[u];
  }
  // (the loss inputs are pinned to registers HERE - without this the compiler sinks the loads to their use)
  asm volatile("" : "+v"(act_in), "+v"(adv_in), "+v"(logp0_in), "+v"(val0_in), "+v"(ret_in));
  __syncthreads();
  LSTAMP(1);
  const float *W1 = lds + lds_of(g.lay, net ? g.lay.vW1 : g.lay.pW1), *b1 = lds + lds_of(g.lay, net ? g.lay.vb1 : g.lay.pb1);
  const float *W2 = lds + lds_of(g.lay, net ? g.lay.vW2 : g.lay.pW2), *b2 = lds + lds_of(g.lay, net ? g.lay.vb2 : g.lay.pb2);
  const float *W3 = lds + lds_of(g.lay, net ? g.lay.vW3 : g.lay.pW3), *b3 = lds + lds_of(g.lay, net ? g.lay.vb3 : g.lay.pb3);
  // parked tiles, [row][env] with a row stride of TS words: H1, H2 (activations), D3, D2, D1 (deltas); DV = D3 (value net: dv[env])
  float *H1 = lds + L.buf, *H2 = H1 + HID * TS, *D3 = H2 + HID * TS, *D2 = D3 + TILE * TS, *D1 = D2 + HID * TS, *DL = D1 + HID * TS;
  float *out = g.partial + (size_t)blockIdx.x * g.stride;
  const float inv_mb = 1.0f / (float)g.mb;
  const int Dp8 = (D + 7) & ~7;     // (<= XC: the host entry refuses obs_dim > 96)
  const int oW2 = net ? g.lay.vW2 : g.lay.pW2, ob2 = net ? g.lay.vb2 : g.lay.pb2;
  const int oW1 = net ? g.lay.vW1 : g.lay.pW1, ob1 = net ? g.lay.vb1 : g.lay.pb1;
  // one 32 x 32 tile of a weight gradient: grad[(32 u + row)][32 v + col] = sum_env Pa[(32 u + row)][env] Pb[(32 v + col)][env]
  // (Pa, Pb parked as [row][env], stride sa / sb; for grad W1 Pa is the observation tile, [env][column])
  auto wgrad = [&](const float *Pa, int sa_row, int sa_env, const float *Pb, int ofs, int ldw, int rows, int cols) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    float pa[16], pb[16];
#pragma unroll
    for (int s = 0; s < 16; s++) {
      const int env = 2 * s + h;
      pa[s] = Pa[col * sa_row + env * sa_env]; pb[s] = Pb[col * TS + env];
    }
    LDS_ISSUED();
#pragma unroll
    for (int s = 0; s < 16; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[s], pb[s], acc, 0, 0, 0);
    if (col < cols) {
#pragma unroll
      for (int r = 0; r < 16; r++)
        if (rowmap(r, h) < rows) out[ofs + rowmap(r, h) * ldw + col] = acc[r];
    }
  };
  // sum over the 32 envs of row `lane` of a parked tile (bias gradients), fixed order
  auto row_sum = [&](const float *P, int row_mask = 63) {      // (row_mask 31: a 32-row tile, lanes 32.. repeat rows 0..)
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < TILE; e++) t += P[(lane & row_mask) * TS + e];
    return t;
  };
  const int u = wave & 1;                  // chain waves: the half of the neurons this wave owns
  const bool chain = wave < 2;
  f32x16 h1o, h2o, d2o, d3;                // own halves: rows 32 u + rowmap(r, h) of h1, h2, d2; wave 0: d3 (policy)
  float dv = 0.f;                          // value net: dL/dv of this lane's env
  // ================================================================ forward, layer 1 (own half)
  if (chain) {
#pragma unroll
    for (int r = 0; r < 16; r++) h1o[r] = b1[32 * u + rowmap(r, h)];
    // k-steps in chunks of four (8 columns; policy_step.hip, act_kernel: the same order of accumulation); all operands first
    const int nch = Dp8 >> 3;                            // <= XC / 8 = 12
    float wa[48], xb[48];
#pragma unroll
    for (int c = 0; c < 12; c++) {
      if (c < nch) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int k = 8 * c + 2 * j + h;
          xb[4 * c + j] = X[col * XS + k];               // (zero from column D on; XC = 96 columns are staged)
          const int kc = k < D ? k : D - 1;
          wa[4 * c + j] = W1[kc * HID + 32 * u + col];
        }
      }
    }
    LDS_ISSUED();
#pragma unroll
    for (int c = 0; c < 12; c++) {
      if (c < nch) {
#pragma unroll
        for (int j = 0; j < 4; j++) h1o = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[4 * c + j], xb[4 * c + j], h1o, 0, 0, 0);
      }
    }
    tanh16(h1o);
#pragma unroll
    for (int r = 0; r < 16; r++) H1[(32 * u + rowmap(r, h)) * TS + col] = h1o[r];
  }
  if (net == 0 && wave == 3) {
    // 1 / sd of every action and the sum of logstd, once per workgroup (until round 4 the loss took 16 expf per lane)
    const float ls_in = lane < A ? ls_raw : 0.f;
    if (lane < 32) VEC[32 + lane] = lane < A ? expf(-ls_in) : 0.f;
    const float sl = half_sum(lane < 32 ? ls_in : 0.f);
    if (lane == 0) VEC[64] = sl;
  }
  __syncthreads();     // ---- B1: H1 is parked
  LSTAMP(2);
  // ================================================================ forward, layer 2 (own half; k ascending: t = 0, 1)
  if (chain) {
#pragma unroll
    for (int r = 0; r < 16; r++) h2o[r] = b2[32 * u + rowmap(r, h)];
    float wa[32], hb[16];
#pragma unroll
    for (int s = 0; s < 32; s++) wa[s] = W2[(32 * (s >> 4) + rowmap(s & 15, h)) * W2S + 32 * u + col];
#pragma unroll
    for (int s = 0; s < 16; s++) hb[s] = H1[(32 * (1 - u) + rowmap(s, h)) * TS + col];      // the half the other chain wave owns
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
    tanh16(h2o);
#pragma unroll
    for (int r = 0; r < 16; r++) H2[(32 * u + rowmap(r, h)) * TS + col] = h2o[r];
  }
  __syncthreads();     // ---- B2: H2 is parked
  LSTAMP(3);
  // ================================================================ wave 0: output layer, loss, delta3 (policy) / dv (value)
  if (wave == 0) {
    if (net == 0) {
      f32x16 mu;
#pragma unroll
      for (int r = 0; r < 16; r++) { const int a = rowmap(r, h); mu[r] = a < A ? b3[a] : 0.f; }
      float wa[32], hb[16];
#pragma unroll
      for (int s = 0; s < 32; s++) wa[s] = W3[(32 * (s >> 4) + rowmap(s & 15, h)) * A + (col < A ? col : A - 1)];      // (unconditional reads: no exec-mask branches)
#pragma unroll
      for (int s = 0; s < 16; s++) hb[s] = H2[(32 + rowmap(s, h)) * TS + col];
      LDS_ISSUED();
#pragma unroll
      for (int s = 0; s < 32; s++) wa[s] = col < A ? wa[s] : 0.f;
#pragma unroll
      for (int s = 0; s < 16; s++) mu = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], h2o[s], mu, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 16; s++) mu = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[16 + s], hb[s], mu, 0, 0, 0);
      LSTAMP(8);
      f32x16 z, isd;
      float zz = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int a = rowmap(r, h);
        isd[r] = VEC[32 + a];                           // 0 beyond the last action
        const float ac = valid ? act_in[r] : mu[r];
        z[r] = (ac - mu[r]) * isd[r];
        zz = __builtin_fmaf(z[r], z[r], zz);
      }
      zz += __shfl_xor(zz, 32, 64);
      const float sum_ls = VEC[64];
      const float logp = -0.5f * zz - sum_ls - 0.5f * LOG_2PI * (float)A;
      const float an = valid ? (adv_in - advm) * advr : 0.f;
      const float ratio = valid ? expf(logp - logp0_in) : 1.f;
      const float rc = fminf(fmaxf(ratio, 1.f - g.cliprange), 1.f + g.cliprange);
      const float l1 = -an * ratio, l2 = -an * rc;
      const bool through = l1 >= l2 || rc == ratio;      // the branch of max() that carries a gradient w.r.t. ratio
      const float dlogp = valid && through ? -an * ratio * inv_mb : 0.f;     // dL/dlogp = dL/dratio * ratio
      // L_pg = mean(max(-a r, -a clip(r))), r = exp(logp - logp0): dL/dlogp = (-a) r / mb on the live branch;
      // dlogp/dmu = z / sd; dlogp/dlogstd = z^2 - 1. The sums over the tile's samples (grad b3, grad logstd, the surrogate term)
      // are taken by the helper waves from the parked tiles, beside delta 2 (until round 4: 35 DPP reductions right here)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int a = rowmap(r, h);
        d3[r] = dlogp * z[r] * isd[r];
        D3[a * TS + col] = d3[r];
        DL[a * TS + col] = a < A ? dlogp * (z[r] * z[r] - 1.f) : 0.f;
      }
      if (h == 0) VEC[col] = valid ? fmaxf(l1, l2) : 0.f;
    } else {
      // ---- value head: v = w . h2 + b (k ascending); clipped value loss; its gradients are rank-1 in the sample
      float v = 0.f;
#pragma unroll
      for (int s = 0; s < 16; s++) v = __builtin_fmaf(h2o[s], W3[rowmap(s, h)], v);
#pragma unroll
      for (int s = 0; s < 16; s++) v = __builtin_fmaf(H2[(32 + rowmap(s, h)) * TS + col], W3[32 + rowmap(s, h)], v);
      v += __shfl_xor(v, 32, 64);
      v += b3[0];
      const float v0 = valid ? val0_in : v, rt = valid ? ret_in : v;
      const float dvc = fminf(fmaxf(v - v0, -g.cliprange), g.cliprange);
      const float vclip = v0 + dvc;
      const float e1 = (v - rt) * (v - rt), e2 = (vclip - rt) * (vclip - rt);
      // 0.5 mean(max(e1, e2)): gradient (v - R) on the unclipped branch, (vclip - R) [clip inactive] on the other
      dv = e1 >= e2 ? (v - rt) : ((dvc == v - v0) ? (vclip - rt) : 0.f);
      dv = valid ? g.vf_coef * dv * inv_mb : 0.f;
      if (h == 0) { D3[col] = dv; VEC[col] = valid ? 0.5f * fmaxf(e1, e2) : 0.f; }     // DV[env], the env's loss term: for waves 1 - 3
    }
  }
  __syncthreads();     // ---- B3: D3 (policy) / dv (value) is parked
  LSTAMP(4);
  // ================================================================ delta of layer 2 (own half)  |  grad W3
  if (chain) {
    if (net == 0) {
      // d2^T = (W3 d3^T) * (1 - h2^2); k = action = rowmap(s, h): B is d3 - wave 0's own registers, wave 1 reads the parked tile
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] = 0.f;
      float wa[16], db[16];
#pragma unroll
      for (int s = 0; s < 16; s++) {
        const int a = rowmap(s, h);
        wa[s] = W3[(32 * u + col) * A + (a < A ? a : A - 1)];
        db[s] = D3[a * TS + col];                           // (wave 0 holds the same values in d3)
      }
      LDS_ISSUED();
#pragma unroll
      for (int s = 0; s < 16; s++) wa[s] = rowmap(s, h) < A ? wa[s] : 0.f;
      if (wave == 0) {
#pragma unroll
        for (int s = 0; s < 16; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], d3[s], acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < 16; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], db[s], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; r++) d2o[r] = acc[r] * (1.f - h2o[r] * h2o[r]);
    } else {
      if (wave == 1) dv = D3[col];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int k = 32 * u + rowmap(r, h);
        d2o[r] = W3[k] * dv * (1.f - h2o[r] * h2o[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; r++) D2[(32 * u + rowmap(r, h)) * TS + col] = d2o[r];
  } else if (net == 0) {
    // ---- grad W3[n2][a] = sum_env h2[env][n2] d3[env][a]: row tile (wave - 2) of H2 against D3; the sums over the samples
    wgrad(H2 + 32 * (wave - 2) * TS, TS, 1, D3, g.lay.pW3 + 32 * (wave - 2) * A, A, 32, A);
    if (wave == 2) {
      const float gb = row_sum(D3, 31);
      if (lane < A) out[g.lay.pb3 + lane] = gb;
    } else {
      const float gl = row_sum(DL, 31);
      if (lane < A) out[g.lay.logstd + lane] = gl;
      const float pg = half_sum(lane < 32 ? VEC[lane] : 0.f);
      if (lane == 0) out[g.stride - 4] = pg;              // sum of the tile's surrogate terms (the mean is taken later)
    }
  } else if (wave == 2) {
    // ---- value net: grad W3[k] = sum_env h2[env][k] dv[env] (lane = k)
    float gw = 0.f;
#pragma unroll
    for (int e = 0; e < TILE; e++) gw = __builtin_fmaf(H2[lane * TS + e], D3[e], gw);
    out[g.lay.vW3 + lane] = gw;
  } else {
    const float dv_l = lane < 32 ? D3[lane] : 0.f, vl_l = lane < 32 ? VEC[lane] : 0.f;
    const float gb = half_sum(dv_l), vl = half_sum(vl_l);
    if (lane == 0) { out[g.stride - 3] = vl; out[g.lay.vb3] = gb; }
  }
  __syncthreads();     // ---- B4: D2 is parked
  LSTAMP(5);
  // ================================================================ delta of layer 1 (own half)  |  grad W2, grad b2
  if (chain) {
    // d1^T = (W2 d2^T) * (1 - h1^2); W2 read transposed (row stride 65: conflict-free); k ascending: t = 0, 1
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    float wa[32], db[16];
#pragma unroll
    for (int s = 0; s < 32; s++) wa[s] = W2[(32 * u + col) * W2S + 32 * (s >> 4) + rowmap(s & 15, h)];
#pragma unroll
    for (int s = 0; s < 16; s++) db[s] = D2[(32 * (1 - u) + rowmap(s, h)) * TS + col];
    LDS_ISSUED();
#pragma unroll
    for (int t = 0; t < 2; t++) {
      if (t == u) {
#pragma unroll
        for (int s = 0; s < 16; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[16 * t + s], d2o[s], acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < 16; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[16 * t + s], db[s], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; r++) D1[(32 * u + rowmap(r, h)) * TS + col] = acc[r] * (1.f - h1o[r] * h1o[r]);
  } else {
    // ---- grad W2[n1][n2] = sum_env h1[env][n1] d2[env][n2]: four tiles (uu, v), two per helper wave; grad b2 = row sums of D2
    for (int t = wave - 2; t < 4; t += 2) {
      const int uu = t >> 1, v = t & 1;
      wgrad(H1 + 32 * uu * TS, TS, 1, D2 + 32 * v * TS, oW2 + 32 * uu * HID + 32 * v, HID, 32, 32);
    }
    if (wave == 3) out[ob2 + lane] = row_sum(D2);
  }
  __syncthreads();     // ---- B5: D1 is parked
  LSTAMP(6);
  // ---- grad W1[k][n1] = sum_env x[env][k] d1[env][n1]: six tiles (q, v) over the four waves; grad b1 = row sums of D1
  for (int t = wave; t < 6; t += 4) {
    const int q = t >> 1, v = t & 1;
    wgrad(X + 32 * q, 1, XS, D1 + 32 * v * TS, oW1 + 32 * q * HID + 32 * v, HID, D - 32 * q, 32);
  }
  if (wave == 2) out[ob1 + lane] = row_sum(D1);
  LSTAMP(7);
}

// ---------------------------------------------------------------- advantage statistics of every minibatch of an epoch
__global__ __launch_bounds__(1024) void adv_stats_kernel(const float *adv, const long long *perm, int mb, float *out) {
  __shared__ double red[16];
  __shared__ double mean_sh;
  const int tid = threadIdx.x;
  const long long *p = perm + (size_t)blockIdx.x * mb;
  double s = 0.0;
  for (int i = tid; i < mb; i += 1024) s += (double)adv[p[i]];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) { double t = 0.0; for (int w = 0; w < 16; w++) t += red[w]; mean_sh = t / mb; }
  __syncthreads();
  const double mean = mean_sh;
  s = 0.0;
  for (int i = tid; i < mb; i += 1024) { const double d = (double)adv[p[i]] - mean; s += d * d; }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; w++) t += red[w];
    out[2 * blockIdx.x] = (float)mean;
    out[2 * blockIdx.x + 1] = (float)(1.0 / (sqrt(t / mb) + 1e-8));       // numpy's std: population
  }
}

// ---------------------------------------------------------------- partial sums -> gradient -> clip -> Adam
struct ApplyArgs {
  const float *partial;
  float *theta, *grad, *m, *v, *losses;     // losses [2]: running sums of the surrogate / value loss MEANS per minibatch
  double *red;                              // [workgroups] partial squared norms
  unsigned *counter;
  int *step;
  int tiles, stride, count, logstd_off, A;
  float ent_coef, lr, b1, b2, eps, max_norm, inv_mb;
  float grad_scale;      // the gradient (entropy term included) and the loss sums are multiplied by this: 1 / ranks for a data-parallel step
  int advance;           // != 0: workgroup 0 advances the Adam step count and forms its step size (the Adam launch follows)
};

// Two launches (measured in round 4 and not kept: ONE launch with a grid barrier between the column sums and Adam - arrival
// count + generation word, all 83 workgroups resident - took 21 us against 4.8 + 5.0; and no last-workgroup serial tail: a single workgroup updating 21 k parameters in six dependent trips
// behind two device-scope fences was most of the old single launch's 31 us):
//   learn_reduce_kernel  one float4 column of the partial rows per thread, 8 tile groups per workgroup (two trips of 8
//                        loads), groups added in order -> grad, + the workgroup's squared norm; workgroup 0 advances
//                        the Adam step count
//   learn_adam_kernel    every workgroup sums the squared norms (workgroup order), clips and updates its 256 float4s
constexpr int AP_COLS = 64, AP_GROUPS = 8;
constexpr int LR_SLOT = 4095;          // learn_red[LR_SLOT]: the Adam step size of the current minibatch step (the buffer holds 4096 doubles)
__global__ __launch_bounds__(AP_COLS * AP_GROUPS) void learn_reduce_kernel(ApplyArgs g) {
  __shared__ float4 part[AP_GROUPS][AP_COLS];
  __shared__ double wred[AP_COLS * AP_GROUPS / 64];
  const int tid = threadIdx.x, c = tid & (AP_COLS - 1), grp = tid / AP_COLS;
  const int n4 = g.stride >> 2;                         // float4 columns of a partial row (the last one: the loss sums)
  const int c4 = blockIdx.x * AP_COLS + c;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < n4) {
    const float4 *P = reinterpret_cast<const float4 *>(g.partial) + c4;
    for (int t0 = grp; t0 < g.tiles; t0 += 8 * AP_GROUPS) {
      float4 x[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int t = t0 + u * AP_GROUPS; x[u] = t < g.tiles ? P[(size_t)t * n4] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
      for (int u = 0; u < 8; u++) { s.x += x[u].x; s.y += x[u].y; s.z += x[u].z; s.w += x[u].w; }
    }
  }
  part[grp][c] = s;
  __syncthreads();
  double sq = 0.0;
  if (grp == 0 && c4 < n4) {
    float4 t = part[0][c];
#pragma unroll
    for (int q = 1; q < AP_GROUPS; q++) { t.x += part[q][c].x; t.y += part[q][c].y; t.z += part[q][c].z; t.w += part[q][c].w; }
    if (c4 == n4 - 1) {                                 // the loss sums of the minibatch (x: surrogate, y: value)
      if (g.losses) { g.losses[0] += t.x * g.inv_mb * g.grad_scale; g.losses[1] += t.y * g.inv_mb * g.grad_scale; }
    } else {
      float e[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int i = 4 * c4 + k;
        if (i >= g.count) e[k] = 0.f;
        else if (i >= g.logstd_off && i < g.logstd_off + g.A) e[k] -= g.ent_coef;      // d(-ent_coef * entropy)/dlogstd
        e[k] *= g.grad_scale;                                                         // (1: a multiplication by one is exact)
        sq += (double)e[k] * e[k];
      }
      if (4 * c4 + 3 < g.count) reinterpret_cast<float4 *>(g.grad)[c4] = make_float4(e[0], e[1], e[2], e[3]);
      else
        for (int k = 0; k < 4; k++)
          if (4 * c4 + k < g.count) g.grad[4 * c4 + k] = e[k];
    }
  }
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  if ((tid & 63) == 0) wred[tid >> 6] = sq;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < AP_COLS * AP_GROUPS / 64; w++) t += wred[w];
    g.red[blockIdx.x] = t;
    if (blockIdx.x == 0 && g.advance) {
      // the Adam step count and TensorFlow's step size for it, formed HERE, off the Adam launch's critical path (two double
      // pow() by one thread cost that launch about a microsecond): lr_t = lr sqrt(1 - b2^t) / (1 - b1^t)
      const int st = *g.step + 1;
      *g.step = st;
      g.red[LR_SLOT] = (double)g.lr * sqrt(1.0 - pow((double)g.b2, (double)st)) / (1.0 - pow((double)g.b1, (double)st));
    }
  }
}

__global__ __launch_bounds__(256) void learn_adam_kernel(ApplyArgs g, int reduce_groups) {
  __shared__ double rsh[4];
  const int tid = threadIdx.x;
  // this thread's parameters, fetched before the norm is known
  const int i = blockIdx.x * 256 + tid, P4 = g.count >> 2;
  float4 *t4 = reinterpret_cast<float4 *>(g.theta), *m4 = reinterpret_cast<float4 *>(g.m), *v4 = reinterpret_cast<float4 *>(g.v);
  float4 gv = make_float4(0.f, 0.f, 0.f, 0.f), tv = gv, mv = gv, vv = gv;
  if (i < P4) { gv = reinterpret_cast<const float4 *>(g.grad)[i]; tv = t4[i]; mv = m4[i]; vv = v4[i]; }
  // the global norm: the workgroups' squared norms added in a FIXED tree (the same in every workgroup: deterministic) -
  // each thread one or more partials, butterfly inside the wave, the four waves in order
  double t = 0.0;
  for (int w = tid; w < reduce_groups; w += 256) t += g.red[w];
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
  if ((tid & 63) == 0) rsh[tid >> 6] = t;
  __syncthreads();
  const float norm = (float)sqrt(((rsh[0] + rsh[1]) + rsh[2]) + rsh[3]);
  const float scale = g.max_norm > 0.f ? g.max_norm / fmaxf(norm, g.max_norm) : 1.f;        // tf.clip_by_global_norm
  const float lr_t = (float)g.red[LR_SLOT];       // lr sqrt(1 - b2^t) / (1 - b1^t), formed by the reduce launch
  auto upd = [&](float gr, float &mi, float &vi, float &th) {
    const float gi = gr * scale;
    mi = g.b1 * mi + (1.f - g.b1) * gi;
    vi = g.b2 * vi + (1.f - g.b2) * gi * gi;
    th -= lr_t * mi / (sqrtf(vi) + g.eps);
  };
  if (i < P4) {
    upd(gv.x, mv.x, vv.x, tv.x); upd(gv.y, mv.y, vv.y, tv.y); upd(gv.z, mv.z, vv.z, tv.z); upd(gv.w, mv.w, vv.w, tv.w);
    t4[i] = tv; m4[i] = mv; v4[i] = vv;
  }
  if (blockIdx.x == 0)
    for (int k = (P4 << 2) + tid; k < g.count; k += 256) {
      float mi = g.m[k], vi = g.v[k], th = g.theta[k];
      upd(g.grad[k], mi, vi, th);
      g.m[k] = mi; g.v[k] = vi; g.theta[k] = th;
    }
}

}  // namespace

extern "C" {

#if TREX_LEARN_STAMPS
__attribute__((visibility("default"))) int trex_policy_debug_learn_stamps(unsigned long long *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(learn_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif

int trex_policy_minibatch_stats(TrexPolicy *p, const float *adv_dev, int64_t num_samples, const int64_t *perm_dev, int num_minibatches,
                                int mb, float *stats_out_dev, void *stream) {
  if (!p || !adv_dev || !perm_dev || !stats_out_dev || num_minibatches <= 0 || mb <= 0 || num_samples <= 0)
    return trex_fail(TREX_E_INVALID, "trex_policy_minibatch_stats: bad argument");
  TrexDeviceGuard guard(p->device);
  BUF_TRY(perm_dev, (size_t)num_minibatches * mb * sizeof(int64_t), "trex_policy_minibatch_stats: perm");
  BUF_TRY(stats_out_dev, (size_t)num_minibatches * 2 * sizeof(float), "trex_policy_minibatch_stats: stats_out");
  BUF_TRY(adv_dev, (size_t)num_samples * sizeof(float), "trex_policy_minibatch_stats: adv");
  hipLaunchKernelGGL(adv_stats_kernel, dim3(num_minibatches), dim3(1024), 0, (hipStream_t)stream, adv_dev,
                     reinterpret_cast<const long long *>(perm_dev), mb, stats_out_dev);
  HIP_TRY(hipGetLastError());
  return TREX_OK;
}

// gradient of one minibatch: learn_grad_kernel + learn_reduce_kernel -> grad_dev (UNclipped, times grad_scale)
static int launch_minibatch_grad(TrexPolicy *p, const char *who, float *theta_dev, float *grad_dev, float *m_dev, float *v_dev,
                                 const float *obs_dev, const float *act_dev, const float *logp_dev, const float *val_dev,
                                 const float *adv_dev, const float *ret_dev, int64_t num_samples, const int64_t *perm_dev, int first,
                                 int mb, const float *adv_stats_dev, float cliprange, float ent_coef, float vf_coef, float lr,
                                 float beta1, float beta2, float eps, float max_grad_norm, float grad_scale, int advance,
                                 float *loss_sums_dev, void *stream, ApplyArgs *out_args, int *out_groups) {
  if (!p || !theta_dev || !grad_dev || !obs_dev || !act_dev || !logp_dev || !val_dev || !adv_dev || !ret_dev || !perm_dev || !adv_stats_dev)
    return trex_fail(TREX_E_INVALID, std::string(who) + ": null argument");
  if (mb <= 0 || first < 0 || num_samples <= 0) return trex_fail(TREX_E_INVALID, std::string(who) + ": bad sizes");
  if (p->D > 96) return trex_fail(TREX_E_INVALID, std::string(who) + ": the learner stages 96 observation columns (obs_dim <= 96)");
  if ((p->lay.vW1 + (p->lay.count - p->lay.logstd)) / 4 > 256 * LEARN_UF || (p->lay.logstd - p->lay.vW1) / 4 > 256 * LEARN_UF)
    return trex_fail(TREX_E_INVALID, std::string(who) + ": the learner stages one net's parameters in one trip of 256 x 13 float4s");
  const size_t P = (size_t)p->lay.count, N = (size_t)num_samples;
  BUF_TRY(theta_dev, P * sizeof(float), "minibatch step: theta");
  BUF_TRY(grad_dev, P * sizeof(float), "minibatch step: grad");
  BUF_TRY(m_dev, P * sizeof(float), "minibatch step: m");
  BUF_TRY(v_dev, P * sizeof(float), "minibatch step: v");
  BUF_TRY(obs_dev, N * p->D * sizeof(float), "minibatch step: obs");
  BUF_TRY(act_dev, N * p->A * sizeof(float), "minibatch step: act");
  BUF_TRY(logp_dev, N * sizeof(float), "minibatch step: logp");
  BUF_TRY(val_dev, N * sizeof(float), "minibatch step: val");
  BUF_TRY(adv_dev, N * sizeof(float), "minibatch step: adv");
  BUF_TRY(ret_dev, N * sizeof(float), "minibatch step: ret");
  BUF_TRY(perm_dev, ((size_t)first + mb) * sizeof(int64_t), "minibatch step: perm");
  BUF_TRY(adv_stats_dev, 2 * sizeof(float), "minibatch step: adv_stats");
  BUF_TRY(loss_sums_dev, 2 * sizeof(float), "minibatch step: loss_sums");
  const int tiles = (mb + TILE - 1) / TILE;
  const int stride = (int)(((P + 3) & ~(size_t)3) + 4);       // parameters, padded to float4, + one float4 of loss sums
  const int apply_groups = ((stride >> 2) + AP_COLS - 1) / AP_COLS;
  if (tiles > p->grad_tiles) {      // workspace, grown on demand (never inside a graph capture: the first call is eager)
    float *buf = nullptr;
    HIP_TRY(hipMalloc((void **)&buf, (size_t)tiles * stride * sizeof(float)));
    p->allocs.push_back(buf);
    HIP_TRY(hipMemset(buf, 0, (size_t)tiles * stride * sizeof(float)));     // (pad elements are never written: they stay 0)
    p->grad_partial = buf; p->grad_tiles = tiles;
    if (!p->learn_counter) {
      HIP_TRY(hipMalloc((void **)&p->learn_counter, sizeof(unsigned)));
      p->allocs.push_back(p->learn_counter);
      HIP_TRY(hipMemset(p->learn_counter, 0, sizeof(unsigned)));
      HIP_TRY(hipMalloc((void **)&p->learn_red, 4096 * sizeof(double)));
      p->allocs.push_back(p->learn_red);
    }
  }
  const LdsLayout L = make_lds_layout(p->lay);
  LearnArgs a{theta_dev, obs_dev, act_dev, logp_dev, val_dev, adv_dev, ret_dev, reinterpret_cast<const long long *>(perm_dev),
              adv_stats_dev, p->grad_partial, first, mb, stride, cliprange, vf_coef, p->lay};
  hipLaunchKernelGGL(learn_grad_kernel, dim3(tiles, 2), dim3(256), (size_t)L.total * sizeof(float), (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  ApplyArgs b{p->grad_partial, theta_dev, grad_dev, m_dev, v_dev, loss_sums_dev, p->learn_red, p->learn_counter, p->adam_step,
              tiles, stride, p->lay.count, p->lay.logstd, p->A, ent_coef, lr, beta1, beta2, eps, max_grad_norm, 1.0f / (float)mb,
              grad_scale, advance};
  hipLaunchKernelGGL(learn_reduce_kernel, dim3(apply_groups), dim3(AP_COLS * AP_GROUPS), 0, (hipStream_t)stream, b);
  HIP_TRY(hipGetLastError());
  if (out_args) *out_args = b;
  if (out_groups) *out_groups = apply_groups;
  return TREX_OK;
}

int trex_policy_minibatch_step(TrexPolicy *p, float *theta_dev, float *grad_dev, float *m_dev, float *v_dev, const float *obs_dev,
                               const float *act_dev, const float *logp_dev, const float *val_dev, const float *adv_dev,
                               const float *ret_dev, int64_t num_samples, const int64_t *perm_dev, int first, int mb,
                               const float *adv_stats_dev, float cliprange, float ent_coef, float vf_coef, float lr, float beta1,
                               float beta2, float eps, float max_grad_norm, float *loss_sums_dev, void *stream) {
  if (!p || !m_dev || !v_dev) return trex_fail(TREX_E_INVALID, "trex_policy_minibatch_step: null argument");
  TrexDeviceGuard guard(p->device);
  ApplyArgs b{};
  int groups = 0;
  if (int rc = launch_minibatch_grad(p, "trex_policy_minibatch_step", theta_dev, grad_dev, m_dev, v_dev, obs_dev, act_dev, logp_dev,
                                     val_dev, adv_dev, ret_dev, num_samples, perm_dev, first, mb, adv_stats_dev, cliprange, ent_coef,
                                     vf_coef, lr, beta1, beta2, eps, max_grad_norm, 1.0f, 1, loss_sums_dev, stream, &b, &groups))
    return rc;
  hipLaunchKernelGGL(learn_adam_kernel, dim3(((p->lay.count >> 2) + 255) / 256), dim3(256), 0, (hipStream_t)stream, b, groups);
  HIP_TRY(hipGetLastError());
  return TREX_OK;
}

int trex_policy_minibatch_grad(TrexPolicy *p, const float *theta_dev, float *grad_dev, const float *obs_dev, const float *act_dev,
                               const float *logp_dev, const float *val_dev, const float *adv_dev, const float *ret_dev,
                               int64_t num_samples, const int64_t *perm_dev, int first, int mb, const float *adv_stats_dev,
                               float cliprange, float ent_coef, float vf_coef, float grad_scale, float *loss_sums_dev, void *stream) {
  if (!p) return trex_fail(TREX_E_INVALID, "trex_policy_minibatch_grad: null argument");
  TrexDeviceGuard guard(p->device);
  return launch_minibatch_grad(p, "trex_policy_minibatch_grad", const_cast<float *>(theta_dev), grad_dev, nullptr, nullptr, obs_dev,
                               act_dev, logp_dev, val_dev, adv_dev, ret_dev, num_samples, perm_dev, first, mb, adv_stats_dev,
                               cliprange, ent_coef, vf_coef, 0.f, 0.9f, 0.999f, 1e-5f, 0.f, grad_scale, 0, loss_sums_dev, stream,
                               nullptr, nullptr);
}

}  // extern "C"
