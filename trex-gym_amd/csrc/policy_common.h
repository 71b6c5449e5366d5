// Shared by policy_step.hip and ppo_learner.hip (the trainer-side kernels of include/trex_policy.h).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/trex_batch.h"
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
  // every region starts on a multiple of 4 floats (16-byte loads when the kernels stage the parameters in LDS); the
  // pad elements are ordinary, unused parameters: zero gradient, never read
  Layout l{};
  l.D = D; l.A = A;
  int o = 0;
  auto take = [&](int n) { const int at = o; o = (o + n + 3) & ~3; return at; };
  l.pW1 = take(D * HID); l.pb1 = take(HID); l.pW2 = take(HID * HID); l.pb2 = take(HID); l.pW3 = take(HID * A); l.pb3 = take(A);
  l.vW1 = take(D * HID); l.vb1 = take(HID); l.vW2 = take(HID * HID); l.vb2 = take(HID); l.vW3 = take(HID); l.vb3 = take(1);
  l.logstd = take(A);
  l.count = o;
  return l;
}

// tanh in ~10 instructions, |error| <= 2e-7 absolute (libm's tanhf expands to ~50 with branches; the policy step
// evaluates 128 per lane): 1 - 2 / (exp(2|x|) + 1) away from 0, the odd series below |x| = 0.1 where that form cancels
__device__ __forceinline__ float tanh_fast(float x) {
  const float ax = fabsf(x), x2 = x * x;
  const float e = __builtin_amdgcn_exp2f(ax * 2.885390081777927f);           // exp(2|x|); inf for large |x| -> t = 1
  const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  const float p = x * (1.0f + x2 * (-0.3333333333f + x2 * (0.1333333333f + x2 * (-0.05396825397f))));
  return ax < 0.1f ? p : copysignf(t, x);
}

// tanh_fast of 16 accumulator registers, stage by stage: the two quarter-rate transcendentals of one value
// overlap with those of the others (one value after the other the chain of each costs ~165 cycles: profiles/tools/mfma_chain_bench.hip)
__device__ __forceinline__ void tanh16(f32x16 &v) {
  float e[16];
#pragma unroll
  for (int r = 0; r < 16; r++) e[r] = __builtin_amdgcn_exp2f(fabsf(v[r]) * 2.885390081777927f);
#pragma unroll
  for (int r = 0; r < 16; r++) e[r] = __builtin_amdgcn_rcpf(e[r] + 1.0f);
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const float x = v[r], x2 = x * x;
    const float t = 1.0f - 2.0f * e[r];
    const float p = x * (1.0f + x2 * (-0.3333333333f + x2 * (0.1333333333f + x2 * (-0.05396825397f))));
    v[r] = fabsf(x) < 0.1f ? p : copysignf(t, x);
  }
}

// Every MFMA loop of the policy / learner kernels takes its LDS operands in ONE batch before the first MFMA (LDS_ISSUED is a compiler fence: the reads may
// not sink of the policy / learner kernels it). Left to itself the compiler issued each read right before the MFMA that uses it - read, s_waitcnt
// lgkmcnt(0), MFMA, 170 - 190 cycles per step instead of the MFMA's 64 (scripts/learn_phases.py: layer 1 7.8 k cycles for 40 steps).
#define LDS_ISSUED() asm volatile("" ::: "memory")

// row of accumulator register `reg` on a lane of half h = lane >> 5 (C/D layout of the 32x32 MFMA forms)
__device__ __forceinline__ constexpr int rowmap(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

}  // namespace

struct TrexPolicy {
  int n = 0, D = 0, A = 0, device = 0, G = 0;
  Layout lay{};
  double *stats = nullptr, *partial = nullptr;
  float *norm = nullptr, *ret = nullptr;
  unsigned *counter = nullptr;
  int *adam_step = nullptr;
  float epsilon = 1e-8f;        // VecNormalize's
  // learner workspace (ppo_learner.hip), allocated on first use: per-tile gradient partials [tiles][count4]
  float *grad_partial = nullptr;
  int grad_tiles = 0;
  unsigned *learn_counter = nullptr;
  double *learn_red = nullptr;
  std::vector<void *> allocs;
  std::vector<TrexSeen> seen;
};

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) return trex_fail(TREX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
#define BUF_TRY(ptr, bytes, what)                                                                       \
  do {                                                                                                  \
    if (int _c = trex_check_device_buffer(p->device, p->seen, (ptr), (size_t)(bytes), (what))) return _c; \
  } while (0)
