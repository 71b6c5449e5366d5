// K1/K2: fused batched physics step for the T-rex env on gfx950 (MI355X, CDNA4).
//
// Replaces, per env and per launch, what TrexBulletEnv.step() (trex_env.py:128-154) asks of pybullet
// and the robot adapter: clip action -> substeps x [position-motor rows (trex_robot.py:397-422) +
// stepSimulation (trex_env.py:150)] -> observations (trex_robot.py:359-365) + reward
// (trex_env.py:186-196); with RESET: TrexBulletEnv.reset() (trex_env.py:98-122).
//
// Mapping to the hardware
//   * ONE ENV PER 64-LANE WAVEFRONT - one wavefront per workgroup, or two envs = two wavefronts per workgroup with split roles in
//     the middle of a substep (PAIR, below: the step launch of even batches up to 4096 envs) -, at most 128 registers per lane and
//     under 10 KB of LDS per env: 4 waves per SIMD (16 per CU), so that at the headline 4096 envs every
//     env is resident at once (4096 waves = 1024 SIMDs x 4) and the serial Gauss-Seidel chain of one env
//     hides behind the three other waves of its SIMD. No inter-wave synchronisation exists.
//   * lanes 0..25 = the 26 bodies (lane b = body b = joint b) for the tree sweeps; for the constraint
//     solve ONE CONSTRAINT ROW PER LANE: motor row j (with joint j's limit row riding on it) on lane j
//     (1..25), the 3 x 13 contact rows on lanes 26..63 and 0.
//   * every spatial quantity is expressed in WORLD-ALIGNED axes about the body's OWN frame origin
//     (the joint axis passes through it). Parent<->child sweeps therefore need no rotations - only
//     the translation by the joint offset d - and no quantity is a difference of m*r^2-sized terms
//     (f32-safe: D_i = a.(I a) directly). Base-to-tip passes move 6..12 registers per level with
//     wavefront shuffles, the tip-to-base articulated-inertia pass stages 27 floats per body through LDS.
//   * the env's base state (pose, twist) is wave-uniform and lives in SGPRs; what later phases need of a
//     body (axis, origin, U/D, 1/D, updated joint rate, parent) is PARKED in LDS as one 80-byte record
//     per body instead of being carried in registers across phases.
//   * M^-1 is never formed by repeated sweeps: the ABA factorisation M^-1 = A^T B A is kept
//     DISTRIBUTED. Every constraint row walks ITS OWN chain once, on its own lane (motor rows and
//     contact rows in the same pass, reading the body records), and keeps a descriptor (chain nodes,
//     entries u, u/D, base force r0, I0^-1 r0); any entry of the Delassus matrix J M^-1 J^T is then 12
//     multiply-adds of two descriptors, the column's one read from LDS at one address by all lanes.
//   * projected Gauss-Seidel runs in Delassus (residual) form: a row's impulse change reaches all other
//     rows as one v_readlane (SGPR broadcast) + one FMA per lane - no reduction, no LDS in a row. The rows are
//     hand-placed: 5 issue slots per motor row, 27 per live contact point (bounds shifted by the impulse).
//   * which wave runs which env, and at which issue priority, is decided inside the kernel (contact counts of the
//     previous launch; the SIMD's arbiter breaks ties by wave age, which has to be countered): set_sweep_priority.
//   * HBM traffic per env-step is the state row in/out + action in + obs/reward out (912 B); the
//     kernel is bound by VALU issue, not by bandwidth.
//
// The arithmetic is the one restated by oracle/trex_oracle.c; tests/ compare the two.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "device_model.h"

#define TL TREX_TL
#define MAXD TREX_MAXD
#define MAXCH TREX_MAXCH
#define MAXC TREX_MAXC

#ifndef TREX_STAMPS
#define TREX_STAMPS 0
#endif
#ifndef TREX_PAIR_MAX
#define TREX_PAIR_MAX 4096                   // largest (even) batch the pair form steps
#endif
#ifndef TREX_PAIR_LAUNCH
#define TREX_PAIR_LAUNCH (!TREX_STAMPS)      // 0: every batch through the single-env launch (A/B builds; the stamped diagnostic build)
#endif
// diagnostic variants for scripts/parity_ablation.sh (which round-2 arithmetic shortcut costs what in one-step error)
#ifndef TREX_ABLATE_EXACT_MATH
#define TREX_ABLATE_EXACT_MATH 0
#endif
#ifndef TREX_ABLATE_EXACT_QUAT
#define TREX_ABLATE_EXACT_QUAT 0
#endif
#ifndef TREX_ABLATE_PLAIN_COMMIT
#define TREX_ABLATE_PLAIN_COMMIT 0
#endif
#ifndef TREX_PRIO_MODE
#define TREX_PRIO_MODE 1   // 0: no priorities (ablation), 1: the policy described at set_sweep_priority
#endif
// Diagnostic build only (make stamps): s_memtime at phase boundaries of workgroup 0, accumulated into the
// debug buffer at [3000 + phase] as cycles. Never compiled into the product library.
#if TREX_STAMPS
// per-wave phase cycles: debug[4096 + phase * n_envs + wave] accumulates over the substeps of the launch
#define STAMP(i)                                                                          \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    const unsigned long long _t = __builtin_amdgcn_s_memtime();                           \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                   \
    if (args.debug && (threadIdx.x & 63) == 0) args.debug[4096 + (i) * args.n_envs + wg] += (float)(_t - stamp_last); \
    stamp_last = _t;                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
// sub-phase stamp inside a phase: adds the cycles since the last stamp to slot i, then continues the phase clock
#define SUBSTAMP(i) STAMP(i)
#else
#define STAMP(i) asm volatile("; ---- phase mark " #i)
#define SUBSTAMP(i) do {} while (0)
#endif

namespace {

// ---------------------------------------------------------------- wave (64-lane) primitives
__device__ __forceinline__ float wshfl(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int wshfl(int v, int src) { return __shfl(v, src, 64); }
// value held by lane `src` (src WAVE-uniform) through an SGPR: one v_readlane
__device__ __forceinline__ float rl(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ int rl(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ unsigned rl(unsigned v, int src) { return (unsigned)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// the same through an asm the optimiser cannot see through: a readfirstlane of a value it KNOWS to be uniform is folded away, and the
// value then stays in the vector register that produced it
// (with the wait states the hazard rules ask for and the compiler does not add around an asm: one between the VALU write of the
// source and v_readfirstlane, two before a VALU may read the scalar result)
__device__ __forceinline__ float uni_sgpr(float v) {
  float r;
  asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 1" : "=s"(r) : "v"(v));
  return r;
}
// all-reduce over the 64 lanes on the VALU (no LDS round trips): four DPP steps inside the 16-lane rows,
// then gfx950's v_permlane16_swap (rows 0<->1, 2<->3) and v_permlane32_swap (halves).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ float wsum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  u = __float_as_uint(v);
  r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wminf(float v) {
  v = fminf(v, dpp_mov<0xB1>(v));
  v = fminf(v, dpp_mov<0x4E>(v));
  v = fminf(v, dpp_mov<0x141>(v));
  v = fminf(v, dpp_mov<0x140>(v));
  unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  u = __float_as_uint(v);
  r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ int wmini(int v) {   // non-negative values
  v = min(v, dpp_mov_i<0xB1>(v));
  v = min(v, dpp_mov_i<0x4E>(v));
  v = min(v, dpp_mov_i<0x141>(v));
  v = min(v, dpp_mov_i<0x140>(v));
  auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  v = min((int)r[0], (int)r[1]);
  r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  return min((int)r[0], (int)r[1]);
}
// all-reduce inside aligned lane GROUPS of G = 8, 16, 32 or 64 lanes (G wave-uniform): the first DPP steps of the
// wave reductions above
__device__ __forceinline__ float gmaxf(float v, int G) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  if (G >= 16) v = fmaxf(v, dpp_mov<0x140>(v));
  if (G >= 32) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  if (G >= 64) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  return v;
}
__device__ __forceinline__ int gmini(int v, int G) {
  v = min(v, dpp_mov_i<0xB1>(v));
  v = min(v, dpp_mov_i<0x4E>(v));
  v = min(v, dpp_mov_i<0x141>(v));
  if (G >= 16) v = min(v, dpp_mov_i<0x140>(v));
  if (G >= 32) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = min((int)r[0], (int)r[1]);
  }
  if (G >= 64) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    v = min((int)r[0], (int)r[1]);
  }
  return v;
}
// wave arg-max with ties to the lowest index: returns the winning (score, index) on every lane
__device__ __forceinline__ void wargmax(float &score, int &index) {
  const float best = -wminf(-score);
  index = uni(wmini(score == best ? index : 0x7fffffff));
  score = uni(best);
}

__device__ __forceinline__ void cross3(const float *a, const float *b, float *o) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
__device__ __forceinline__ float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ float dot6(const float *a, const float *b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
__device__ __forceinline__ void matvec3(const float *m, const float *v, float *o) {
  float x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  float y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  float z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
__device__ __forceinline__ void matmul3(const float *a, const float *b, float *o) {
  float t[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
#pragma unroll
  for (int i = 0; i < 9; i++) o[i] = t[i];
}
__device__ __forceinline__ void quat_to_mat(const float *q, float *m) {
  float x = q[0], y = q[1], z = q[2], w = q[3];
  m[0] = 1 - 2 * (y * y + z * z); m[1] = 2 * (x * y - z * w); m[2] = 2 * (x * z + y * w);
  m[3] = 2 * (x * y + z * w); m[4] = 1 - 2 * (x * x + z * z); m[5] = 2 * (y * z - x * w);
  m[6] = 2 * (x * z - y * w); m[7] = 2 * (y * z + x * w); m[8] = 1 - 2 * (x * x + y * y);
}

// Symmetric 6x6 stored as A(6: xx xy xz yy yz zz) | B(9, row-major upper-right block) | C(6):
//   M = [[A, B], [B^T, C]]
struct Sym6 { float A[6], B[9], C[6]; };

__device__ __forceinline__ void sym3_mul(const float *s, const float *v, float *o) {
  o[0] = s[0] * v[0] + s[1] * v[1] + s[2] * v[2];
  o[1] = s[1] * v[0] + s[3] * v[1] + s[4] * v[2];
  o[2] = s[2] * v[0] + s[4] * v[1] + s[5] * v[2];
}
__device__ __forceinline__ void sym6_mul(const Sym6 &m, const float *v, float *o) {
  float a[3], b[3], c[3], d[3];
  sym3_mul(m.A, v, a);
  matvec3(m.B, v + 3, b);
  // B^T w
  c[0] = m.B[0] * v[0] + m.B[3] * v[1] + m.B[6] * v[2];
  c[1] = m.B[1] * v[0] + m.B[4] * v[1] + m.B[7] * v[2];
  c[2] = m.B[2] * v[0] + m.B[5] * v[1] + m.B[8] * v[2];
  sym3_mul(m.C, v + 3, d);
#pragma unroll
  for (int i = 0; i < 3; i++) { o[i] = a[i] + b[i]; o[3 + i] = c[i] + d[i]; }
}
// full 6x6 from Sym6 (row-major)
__device__ __forceinline__ void sym6_full(const Sym6 &m, float *f) {
  const int sidx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      f[6 * r + c] = m.A[sidx[r][c]];
      f[6 * r + 3 + c] = m.B[3 * r + c];
      f[6 * (3 + r) + c] = m.B[3 * c + r];
      f[6 * (3 + r) + 3 + c] = m.C[sidx[r][c]];
    }
}
// M -= U Us^T with Us = U * s already formed: one fma per entry
__device__ __forceinline__ void sym6_rank1_sub(Sym6 &m, const float *U, const float *Us) {
  const int ia[6] = {0, 0, 0, 1, 1, 2}, ib[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
  for (int k = 0; k < 6; k++) {
    m.A[k] = __builtin_fmaf(-U[ia[k]], Us[ib[k]], m.A[k]);
    m.C[k] = __builtin_fmaf(-U[3 + ia[k]], Us[3 + ib[k]], m.C[k]);
  }
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) m.B[3 * r + c] = __builtin_fmaf(-U[r], Us[3 + c], m.B[3 * r + c]);
}
// o += a x b, two fma per entry
__device__ __forceinline__ void cross3_acc(const float *a, const float *b, float &o0, float &o1, float &o2) {
  o0 = __builtin_fmaf(a[1], b[2], __builtin_fmaf(-a[2], b[1], o0));
  o1 = __builtin_fmaf(a[2], b[0], __builtin_fmaf(-a[0], b[2], o1));
  o2 = __builtin_fmaf(a[0], b[1], __builtin_fmaf(-a[1], b[0], o2));
}

// SPD 6x6 systems (the base's articulated inertia): Cholesky factor L kept as its 15 strictly-lower entries
// (row-major, l[i (i - 1) / 2 + j], j < i) and the 6 RECIPROCAL diagonal entries; a solve is a forward and a back
// substitution, 30 fma + 12 mul. No explicit inverse, no division, no IEEE sqrt expansion: the reciprocal roots
// come from v_rsq_f32 with one Newton step.
struct Chol6 { float l[15], il[6]; };
__device__ __forceinline__ constexpr int lidx(int i, int j) { return i * (i - 1) / 2 + j; }
__device__ __forceinline__ float rsqrt_nr(float s) {
#if TREX_ABLATE_EXACT_MATH   // diagnostic variant (scripts/parity_ablation.sh): IEEE sqrt and division
  return 1.0f / sqrtf(s);
#endif
  const float r0 = __builtin_amdgcn_rsqf(s);
  const float e = __builtin_fmaf(-s * r0, r0, 1.0f);   // 1 - s r0^2
  return __builtin_fmaf(0.5f * r0, e, r0);
}
__device__ __forceinline__ void chol6_factor(const float *a, Chol6 &c) {   // a: full 6x6, row-major (lower part read)
#pragma unroll
  for (int j = 0; j < 6; j++) {
    float s = a[6 * j + j];
#pragma unroll
    for (int k = 0; k < j; k++) s -= c.l[lidx(j, k)] * c.l[lidx(j, k)];
    c.il[j] = rsqrt_nr(s);
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      float t = a[6 * i + j];
#pragma unroll
      for (int k = 0; k < j; k++) t -= c.l[lidx(i, k)] * c.l[lidx(j, k)];
      c.l[lidx(i, j)] = t * c.il[j];
    }
  }
}
__device__ __forceinline__ void chol6_solve(const Chol6 &c, const float *r, float *z) {
  float w[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float t = r[i];
#pragma unroll
    for (int k = 0; k < i; k++) t -= c.l[lidx(i, k)] * w[k];
    w[i] = t * c.il[i];
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    float t = w[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) t -= c.l[lidx(k, i)] * z[k];
    z[i] = t * c.il[i];
  }
}

// ---------------------------------------------------------------- LDS layout (per wave = one env)
constexpr int NJMAX = 25;                // hinge joints at most (26 bodies)
constexpr int NROW = NJMAX + 3 * MAXC;   // constraint rows of one env: 25 motor rows + 13 x (normal, 2 friction) = 64
static_assert(NROW == 64, "one constraint row per lane");
#define KROW_LANE(k) ((k) < 3 * MAXC - 1 ? NJMAX + 1 + (k) : 0)
constexpr int CLANE0 = NJMAX + 1;        // contact row k lives on lane CLANE0 + k (k < 38) and on lane 0 (k = 38)
// Body record, parked after the tree phases and read by the row walks (float4 reads, 80-byte stride:
// 16 lanes reading 16 different records hit 16 different bank quartets):
//   q0 = joint axis (world) xyz | 1/D      q1 = origin r (rel. base origin) xyz | updated joint rate
//   q2 = (U/D)[0..3]                       q3 = (U/D)[4..5] | parent + 256 * depth (int) | -
//   q4 = offset from the parent's origin xyz | children, 8 bits each (int)
constexpr int BREC = 5;                  // float4s per record
struct WaveLds {
  float4 body[32 * BREC];     // 2560 B; after the row walks: z0 stash [6][64] for the base twist change
  union {
    struct {
      float aba[32][28];      // tip-to-base pass: articulated inertia (21) + bias force (6) per body      3584 B
      float rtab[32][9];      // PAIR launches: body rotations, handed to the wave that runs the tree phases 1152 B
    } t;
    float4 desc[64][4];       // B build: column descriptor of the row on lane L: chain | zc[6] | z0[6]     4096 B
    float4 cg[160];           // contact generation (single-env launches): CgLds (below)                     2560 B
  } u;
  float cpt[MAXC][8];         // contact points: body, x, y, z (rel. base origin), distance     416 B
  float st[6][TL];            // per body lane, parked across the phases: q, qd, motor torque, target, updated rate, 1/M^-1_jj  768 B
  float xch[44];              // PAIR launches, between the two waves of a workgroup: [0..5] base twist w, v | [6] env | [7] substeps
                              // of this step | [8..28] Cholesky factor of the base's articulated inertia | [29..34] base acceleration
};
enum { ST_Q, ST_QD, ST_TAU, ST_TARGET, ST_NQD, ST_MDG };
static_assert(sizeof(WaveLds) <= 10240, "16 envs per CU need <= 10 KB of LDS each");
// Contact generation works in the union area (the inertia slots are written after it):
constexpr int CG_WORDS = TREX_CM_WORDS;  // in-margin mask words per ENV, packed per body (device_model.h: cm_pack)
struct CgLds {
  unsigned long long best[TL];           // per body: min of (ordered distance << 32 | vertex)                256 B
  float4 ent[TL][2];                     // near-hull table: end position | vertex - position | body | body v0 ; Rz, zb  1024 B
  unsigned cm[CG_WORDS];                 // per body b, from word cm_pack[b] >> 8: word w, bit j <-> vertex hull_start[b] + P j + w inside the margin, P = 8 or 32   1280 B
};
static_assert(sizeof(CgLds) <= sizeof(WaveLds::u), "contact-generation scratch fits the union area");

struct KernelArgs {
  const TrexDeviceModel *model;
  TrexBatchArrays arr;
  int n_envs;
  const float *actions;   // [N, J]
  float *obs;             // [N, 3J] nullable; row e starts at obs + e * obs_stride
  float *reward;          // [N] nullable; element e at reward[e * scal_stride]
  uint8_t *done;          // [N] nullable
  float *done_f;          // done as 0.0 / 1.0 at done_f[e * scal_stride] (row-block output), nullable
  int obs_stride, scal_stride;
  float *penalties;       // [N, 3] nullable
  const uint8_t *reset_mask;  // RESET only, nullable = all
  int32_t *bal;               // rank lists of the wave balance (step launches), nullable = workgroup k runs env k
  float w_distance, w_energy, w_drift;
  float *debug;           // diagnostics of env 0's last substep (tests), nullable
  // MULTI launches (trex_batch_step_many): n_steps env-steps per launch; step s reads actions + s * N * J and writes the
  // row block at + s * step_rows floats (obs, reward, done_f alike), done bytes at + s * N, penalties at + s * 3 N
  int n_steps;
  long long step_rows;
  int pen_in_rows;        // row-block launches of a batch with trex_batch_set_penalties_in_rows: the three penalties follow done in the row
};

}  // namespace

// DEBUG instantiations carry the diagnostics dump (scripts/gpu_debug.py, phase stamps); the product launches
// use DEBUG = false so that none of the dump's address arithmetic exists in the shipped kernels.
// MULTI: the launch advances every env by args.n_steps env-steps (open-loop action sequences): the state stays in
// SGPRs / LDS between the steps and - what it is for - no wave ever waits for the slowest wave of a step: with one step
// per launch the SIMDs idle a fifth of the launch behind its heaviest envs.
// (Measured in round 4 and NOT kept - DESIGN.md 6: a PERSISTENT launch for batches beyond the 4096 wave slots, 4096 workgroups
// that draw env after env off the rank lists through an atomic cursor, heaviest first. Bitwise the same rows; 10.7 M env-steps/s
// at 8192 envs and 13.0 M at 32768 against 11.4 M / 13.3 M for one workgroup per env: the dispatcher refills the slots at
// least as well, and the env loop around this body made the compiler hoist constants out of it - 7 spilled registers.)
// PAIR (trex_step_pair_kernel, the step launch of an even batch of at most 4096 envs): a workgroup of TWO waves = two envs, a
// HEAVY one (wave 0: rank p from the heavy end of the rank lists) and a LIGHT one (wave 1: rank n - 1 - p). Kinematics, row walks,
// B build, sweeps and integration stay per wave, each for its own env. In between the two waves split ROLES, each working
// for BOTH envs at the same time:
//   wave 1  generates the contacts of env 0, then of env 1 (64 lanes each; the body rotations, origins and the base height come
//           from LDS, its scratch is the workgroup's);
//   wave 0  runs the four phases that work with one lane per BODY - velocities / inertias / bias forces, ABA pass 2, the base's
//           Cholesky factor, ABA pass 3: a fifth of a wave's cycles with 26 of 64 lanes busy - ONCE for both envs: lanes 0..31
//           the bodies of its own env, lanes 32..63 those of its partner's, every LDS address and shuffle source offset by the half.
// A substep loses the SHORTER of the two phases from its critical path (the single-env launch runs them one after the other)
// and the instruction stream of the tree phases is issued once for two envs; it pays two workgroup barriers, the hand-over of
// rotations, twists and the base factor through LDS, and the wait of the env that is done first - which is why the pairs are
// heavy + light: the wave that generates contacts does it for both envs one after the other, and two contact-heavy envs in one
// workgroup leave nothing of the overlap (round 4, first form: the partner merely WAITED during the tree phases - 11.40 M against
// 11.53 M; roles with adjacent ranks paired +1.7 % at 2048 envs, heavy + light +7.7 %; at 4096 envs, where four waves share a
// SIMD, 11.70 M against 11.50 M). The same arithmetic per lane: BITWISE the rows of the single-env launch (scripts/state_digest.py,
// 300 steps of 4096 envs; the test-suite compares even batches - this form - with step_many, resets and odd batches - that form).
#define WSYNC() do { if (PAIR) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); else __syncthreads(); } while (0)
template <bool RESET, bool DEBUG, bool MULTI, bool PAIR = false>
__device__ __forceinline__ void trex_step_body(const KernelArgs &args, const int wg_in) {   // wg_in: blockIdx.x
  static_assert(!PAIR || (!RESET && !DEBUG && !MULTI), "the pair form exists for the product step launch only");
  __shared__ WaveLds Wpair[PAIR ? 2 : 1];
  __shared__ __attribute__((aligned(16))) unsigned char Gpair[PAIR ? sizeof(CgLds) : 16];   // PAIR: contact-generation scratch of the workgroup (wave 1)
  const int wave = PAIR ? uni((int)threadIdx.x >> 6) : 0;
  // (PAIR: the LDS of this wave's env is addressed through ONE base register the compiler takes for lane-dependent - a
  // wave-uniform base made it precompute an address per access as scalars: 226 spilled scalar registers, 80 reloads in the
  // B build alone)
  int wave_v = PAIR ? (int)threadIdx.x >> 6 : 0;
  if (PAIR) asm volatile("" : "+v"(wave_v));
  WaveLds &W = Wpair[wave_v];
  const int wg = PAIR ? 2 * wg_in + wave : wg_in;       // the index the env-to-wave deal and the priorities go by
  const int tid = (int)threadIdx.x & 63;
  const TrexDeviceModel *__restrict__ M = args.model;
  // Which env this wave runs. All waves of the headline launch are resident at once and a SIMD is done when its
  // slowest wave is, so the envs are dealt by the contact count of their PREVIOUS step launch: every wave filed its
  // env under its count at the end of that launch (below), and wave k now takes rank r(k) of those lists, heaviest
  // count first. Ranks 0..1023 go to workgroups 0..1023 in order, every later block of 1024 in REVERSE: SIMD j
  // (workgroups j, 1024 + j, ...) gets the j-th heaviest env together with the j-th lightest of each later block -
  // the sums of work per SIMD are level. Device-side state only (phase, counts, lists): nothing to launch before
  // the step, and a captured graph replays correctly.
  int env = wg;
  int bal_phase = 0;
  if (args.bal) {
    const int32_t *B = args.bal;
    bal_phase = uni(B[TREX_BAL_PHASE]);
    const int32_t *cnt = B + TREX_BAL_COUNTS + TREX_BAL_BINS * bal_phase;
    const int k = wg, q = k >> 10, m = min(1024, args.n_envs - (q << 10));
    int r = q == 0 ? k : (q << 10) + (m - 1 - (k & 1023));
    if (PAIR) {
      // a workgroup pairs a HEAVY env with a LIGHT one: wave 0 takes rank p from the heavy end, wave 1 rank n - 1 - p from the
      // light end. The wave that generates the contacts does it for both envs one after the other, beside the tree dynamics of
      // both: two contact-heavy envs in one workgroup would leave nothing of the overlap. Workgroups b, b + 512, ... share a
      // SIMD pair: pair indices go to them like ranks go to single-env workgroups - first block in order, later blocks reversed.
      const int bq = wg_in >> 9, half = args.n_envs >> 1, bm = min(512, half - (bq << 9));
      const int pidx = bq == 0 ? wg_in : (bq << 9) + (bm - 1 - (wg_in & 511));
      // (measured, 4096 envs: this deal 11.69 M; heaviest with the median env 11.55 M; workgroups in plain rank order 11.67 M; the
      // light env's wave running the tree dynamics and the heavy one the contacts 11.58 M)
      r = wave == 0 ? pidx : args.n_envs - 1 - pidx;
    }
    // (not better, measured: SIMD j taking rank j and the 3 LIGHTEST envs still to be dealt - 11.07 M against 11.13 M
    // at 4096 envs, 13.02 M against 13.18 M at 32768: which light mates a heavy wave has does not matter)
    const int lane_ = tid;
    const int mine = lane_ < TREX_BAL_BINS ? cnt[lane_] : 0;   // the 16 counts in one load, lane c holds count c
    int b = TREX_BAL_BINS - 1, total = 0;
#pragma unroll
    for (int c = 0; c < TREX_BAL_BINS; c++) total += rl(mine, c);
    for (; b > 0; b--) {
      const int c = rl(mine, b);
      if (r < c) break;
      r -= c;
    }
    // The lists are sound exactly when they hold every env once: the counts sum to n_envs (every wave sees the same
    // counts and takes the same decision). If they do not - a launch that did not complete left them half filed -
    // this launch keeps env k in workgroup k instead of stepping one env twice and another not at all; its waves
    // still file their envs below, so the next launch finds sound lists again.
    if (total == args.n_envs) {
      env = B[TREX_BAL_LISTS + (size_t)(bal_phase * TREX_BAL_BINS + b) * args.n_envs + r];
      if (env < 0 || env >= args.n_envs) env = wg;   // (unreachable with sound lists; never an out-of-range row)
    }
  }
  env = uni(env);

  const int nb = M->nb, maxdepth = M->maxdepth;
  // (PAIR: scalars - as values the compiler loads with vector loads, dt and the products it hoists out of the substep loop,
  // 0.5 dt and 0.25 dt^2, sat in vector registers for the whole kernel and were spilled to SCRATCH: 12 MB of traffic per launch)
  const float dt = PAIR ? uni_sgpr(M->prm[TP_DT]) : M->prm[TP_DT];
  const float inv_dt = PAIR ? uni_sgpr(M->inv_dt) : M->inv_dt;
  const float dt_half = PAIR ? uni_sgpr(0.5f * dt) : 0.5f * dt, dt2_quarter = PAIR ? uni_sgpr(0.25f * dt * dt) : 0.25f * dt * dt;
  const int nj = nb - 1;
  // Everything about the model is (re)read from the L2-resident struct in the phase that uses it, through an
  // opaque pointer, and every lane-derived mask / index is re-derived from an opaque copy of the lane id, so
  // that nothing loop-invariant is hoisted out of the substep loop and then spilled across the solver.
  auto Mo = [&]() { const TrexDeviceModel *Mi = M; asm volatile("" : "+s"(Mi)); return Mi; };
  auto lane_id = [&]() {   // (volatile: recomputed at every use site, never kept live or spilled)
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };

  // ---- per-env state: the base pose and twist are wave-uniform (SGPRs); q / qd / motor torque / target of
  // body lane b are parked in LDS (W.st) and read where a phase needs them
  float pos[3], quat[4], bv[3], bw[3];
  const float mu = args.arr.domain ? uni(args.arr.friction[env]) : (PAIR ? uni_sgpr(M->prm[TP_FRICTION]) : M->prm[TP_FRICTION]);
  bool motors_on;
  int flags_in = 0, steps_in = 0;      // base row words 13 and 15 (device_model.h)
  bool do_reset = false;
  bool bad;
  if (RESET) do_reset = args.reset_mask ? (args.reset_mask[env] != 0) : true;
  if (RESET && !do_reset) {
    // an env that is not reset only hands out its observation again: no physics, no kinematics, no LDS. (The
    // episode-limit reset of a VecEnv runs with a mask every step: about N/1000 envs reset, the rest end here.)
    // Its motor-torque columns are LEFT as the caller's row holds them - the last step wrote them there; the batch keeps no
    // copy of its own (until round 3 a [N][32] row, written by every step and read only here: 0.5 MB of the 3.7 MB a
    // 4096-env step launch wrote).
    if (args.obs && tid >= 1 && tid < nb) {
      float *o = args.obs + (size_t)env * args.obs_stride;
      const int slot = M->obs_slot[tid];
      o[slot] = args.arr.q[(size_t)env * TL + tid];
      o[nj + slot] = args.arr.qd[(size_t)env * TL + tid];
    }
    return;
  }
  {
    const bool is_body = tid < nb, is_joint = tid >= 1 && tid < nb;
    float q = 0.f, qd = 0.f, mtau = 0.f, target = 0.f;
    if (RESET && do_reset) {
#pragma unroll
      for (int c = 0; c < 3; c++) { pos[c] = uni(M->base_pos0[c]); bv[c] = 0.f; bw[c] = 0.f; }
#pragma unroll
      for (int c = 0; c < 4; c++) quat[c] = uni(M->base_quat0[c]);
      q = is_body ? M->q_start[tid & (TL - 1)] : 0.f;
      motors_on = false;  // remove_joint_control, trex_robot.py:309
    } else {
      const float *b = args.arr.base + (size_t)env * 16;
#pragma unroll
      for (int c = 0; c < 3; c++) { pos[c] = uni(b[c]); bv[c] = uni(b[7 + c]); bw[c] = uni(b[10 + c]); }
#pragma unroll
      for (int c = 0; c < 4; c++) quat[c] = uni(b[3 + c]);
      flags_in = uni(__float_as_int(b[TREX_BASE_FLAGS]));
      if (tid < TL) {
        q = args.arr.q[(size_t)env * TL + tid];
        qd = args.arr.qd[(size_t)env * TL + tid];
        // (the stored motor torque is only handed out again by a reset launch that leaves the env alone; a step
        // overwrites it in its first substep: no load)
      }
      motors_on = RESET ? ((flags_in & TREX_MOTORS_BIT) != 0) : true;
    }
    steps_in = uni(__float_as_int(args.arr.base[(size_t)env * 16 + TREX_BASE_STEPS]));
    // non-finite input state (checked here as well as after the step: fminf/fmaxf clamps launder NaNs)
    bool badl = !(fabsf(q) < 3.0e38f) || !(fabsf(qd) < 3.0e38f);
#pragma unroll
    for (int k = 0; k < 3; k++) badl |= !(fabsf(pos[k]) < 3.0e38f) || !(fabsf(bv[k]) < 3.0e38f) || !(fabsf(bw[k]) < 3.0e38f);
#pragma unroll
    for (int k = 0; k < 4; k++) badl |= !(fabsf(quat[k]) < 3.0e38f);
    bad = __ballot(badl) != 0ull;
    (void)is_joint;   // (the action -> joint target of an env-step is read at the top of the step loop below)
    if (tid < TL) {
      W.st[ST_Q][tid] = q; W.st[ST_QD][tid] = qd; W.st[ST_TAU][tid] = mtau; W.st[ST_TARGET][tid] = target;
      W.st[ST_NQD][tid] = 0.f;
    }
  }
  WSYNC();
  const int n_sub = RESET ? (do_reset ? 1 : 0) : M->n_substeps;
  // Wave priority: the launch lasts as long as its slowest wave, and with one env per wave that is an env with
  // many contact rows. During its sweeps such a wave wins the issue arbitration against the lighter waves of
  // its SIMD, which fill the slots its dependency chain leaves empty. (Mode 2, priority for the whole substep,
  // starved the light waves instead: 3.84 M against 4.12 M env-steps/s.)
  // ---- issue priority. The SIMD's arbiter serves the highest s_setprio level first and, within a level, the OLDEST
  // wave. Measured (bench.py, 4096 envs): no priorities 8.5 M env-steps/s; waves with contact rows
  // first during their sweeps (levels by contact count) 9.5 M - their row chains are the longest; and on top of that
  // the age rule has to be countered: workgroup k sits on SIMD k mod 1024, so the waves of workgroup blocks 2 and 3
  // are the two YOUNGEST of their SIMD and lose every tie - they ended 0.2 M cycles after their mates and the SIMD
  // ran one wave for a fifth of the launch. In the sweeps they get one level more (10.1 M); outside the sweeps the
  // two older and the two younger waves take turns at level 1, substep by substep (10.5 M). (Not better: the bump in
  // the sweeps half of the time, for three waves instead of two, +2, the pairs taking turns there too, a rotating top
  // wave, distinct static levels per wave, levels outside the sweeps by contact count or for the young pair only, the
  // older pair first, a change of places in the middle of the tree phases too, other contact-count thresholds.)
  // Only where the launch is resident at once, 4096 envs or fewer: beyond that a workgroup's index says nothing
  // about its age among the waves of its SIMD.
#ifndef TREX_PRIO_T1
#define TREX_PRIO_T1 1
#define TREX_PRIO_T2 3
#define TREX_PRIO_T3 6
#endif
  const bool aged_launch = args.n_envs <= 4096;
  const int wave_pair = (wg >> 11) & 1;      // 0: the two older waves of the SIMD, 1: the two younger
  auto set_sweep_priority = [&](int contacts) {
    int v = contacts >= TREX_PRIO_T3 ? 3 : (contacts >= TREX_PRIO_T2 ? 2 : (contacts >= TREX_PRIO_T1 ? 1 : 0));
    v += (aged_launch && wave_pair == 1) ? 1 : 0;
    if (v <= 0) __builtin_amdgcn_s_setprio(0);
    else if (v == 1) __builtin_amdgcn_s_setprio(1);
    else if (v == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
  };
#ifndef TREX_TREE_HEAVY
#define TREX_TREE_HEAVY 12
#endif
  int prio_nc = 0;      // contact points of the env's last substep (before the first one: of its last step)
  // (PAIR: the heaviest envs' waves no longer generate their own contacts, and the top level for them outside the sweeps stopped
  // paying - 11.71 M without it against 11.68 M; priorities BY ROLE were measured too: the tree-dynamics wave one level up 11.44 M,
  // the contact wave one level up 11.65 M, the tree-dynamics wave at level 2 over the alternation 11.43 M; no priorities 10.07 M)
  auto set_tree_priority = [&](int substep) {   // outside the sweeps: the pairs take turns
    if (!PAIR && TREX_TREE_HEAVY > 0 && prio_nc >= TREX_TREE_HEAVY) { __builtin_amdgcn_s_setprio(3); return; }
    if (aged_launch && ((wave_pair + substep) & 1)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
  };
#if TREX_PRIO_MODE == 1
  if (!RESET) { prio_nc = flags_in & 255; set_tree_priority(0); }
#endif

  const float floor_z = PAIR ? uni_sgpr(M->prm[TP_FLOOR_Z]) : M->prm[TP_FLOOR_Z], margin = PAIR ? uni_sgpr(M->prm[TP_CONTACT_MARGIN]) : M->prm[TP_CONTACT_MARGIN];
  const int iters = M->n_iterations;
  int maxc = M->max_contacts;
  if (maxc > MAXC) maxc = MAXC;

  int stat_nc = 0;
  float stat_imp = 0.f;

  // FK: world rotation R and origin r (relative to the base origin) of every body, the offset dpar from the
  // parent's origin and the joint axis Sa in world axes (motion subspace about the body's own origin
  // S = [Sa; 0]).  Base-to-tip, parent data via shuffles.
  auto forward_kinematics = [&](int lt, int psrc, int depth, float *R, float *r, float *dpar, float *Sa) {
    const TrexDeviceModel *Mi = Mo();
    const int bl = lt & (TL - 1);
    float axis[3], jpos[3], jrot[9];
#pragma unroll
    for (int c = 0; c < 3; c++) { axis[c] = Mi->axis[c][bl]; jpos[c] = Mi->jpos[c][bl]; }
#pragma unroll
    for (int c = 0; c < 9; c++) jrot[c] = Mi->jrot[c][bl];
    const float q = W.st[ST_Q][bl];
    float Rl[9];
    {
      // jrot * Rot(axis, q)
      float c = cosf(q), s = sinf(q), t = 1.f - c, rq[9];
      rq[0] = t * axis[0] * axis[0] + c;           rq[1] = t * axis[0] * axis[1] - s * axis[2]; rq[2] = t * axis[0] * axis[2] + s * axis[1];
      rq[3] = t * axis[0] * axis[1] + s * axis[2]; rq[4] = t * axis[1] * axis[1] + c;           rq[5] = t * axis[1] * axis[2] - s * axis[0];
      rq[6] = t * axis[0] * axis[2] - s * axis[1]; rq[7] = t * axis[1] * axis[2] + s * axis[0]; rq[8] = t * axis[2] * axis[2] + c;
      matmul3(jrot, rq, Rl);
    }
    quat_to_mat(quat, R);
    r[0] = r[1] = r[2] = 0.f;
    dpar[0] = dpar[1] = dpar[2] = 0.f;
    for (int d = 1; d <= maxdepth; d++) {
      float pR[9], pr[3];
#pragma unroll
      for (int c = 0; c < 9; c++) pR[c] = wshfl(R[c], psrc);
#pragma unroll
      for (int c = 0; c < 3; c++) pr[c] = wshfl(r[c], psrc);
      if (depth == d) {
        float o[3];
        matmul3(pR, Rl, R);
        matvec3(pR, jpos, o);
#pragma unroll
        for (int c = 0; c < 3; c++) { dpar[c] = o[c]; r[c] = pr[c] + o[c]; }
      }
    }
    matvec3(R, axis, Sa);
    if (!(lt >= 1 && lt < nb)) { Sa[0] = Sa[1] = Sa[2] = 0.f; }
  };

#if TREX_STAMPS
  unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
  int stamp_alive = 0, stamp_lamnz = 0;
  if (args.debug && (threadIdx.x & 63) == 0) args.debug[4096 + 11 * args.n_envs + wg] = (float)env;   // the env of this wave
#endif
  // Episode limit of the harness (the reference never terminates, trex_env.py:183-184; a VecEnv auto-resets): the env
  // whose count reaches the limit with this step finishes the step - reward, done = 1 - and then, IN THE SAME LAUNCH,
  // goes to the start pose and takes the reset's un-actuated settle substep, so that its observation is the first
  // one of the new episode (baselines' VecEnv semantics) and no separate reset launch sits between two steps.
  int age = 0;
  bool time_up = false;
  bool env_bad = false;
  float lift = 0.f, drift = 0.f, energy = 0.f;
  // the end of an env-step: head position (needs FK at the new pose: getLinkState(computeForwardKinematics=1)), reward
  // terms, failure containment
  auto finish_step = [&]() {
    const int lt = lane_id();
    const int bl = lt & (TL - 1);
    const bool is_body = lt < nb, is_joint = lt >= 1 && lt < nb;
    float head[3];
    {
      const int parent = is_body ? M->parent[bl] : 0;
      const int psrc = parent < 0 ? 0 : parent;
      const int depth = is_body ? M->depth[bl] : -1;
      float R[9], r[3], dpar[3], Sa[3];
      forward_kinematics(lt, psrc, depth, R, r, dpar, Sa);
      const float hp[3] = {M->head_point[0], M->head_point[1], M->head_point[2]};
      float o[3];
      matvec3(R, hp, o);
      const int hb = M->head_body;
#pragma unroll
      for (int k = 0; k < 3; k++) head[k] = rl(pos[k] + r[k] + o[k], hb);
    }
    float q = W.st[ST_Q][bl], qd = W.st[ST_QD][bl], mtau = W.st[ST_TAU][bl];
    if (lt >= TL) { q = 0.f; qd = 0.f; mtau = 0.f; }
    const float power = wsum(is_joint ? fabsf(qd * mtau) : 0.f);
    // (wave-uniform: kept on the scalar side across the rest of the substep loop)
    lift = uni(args.w_distance * (2.5f - head[2]) * (2.5f - head[2]));
    drift = uni(args.w_drift * (head[0] * head[0] + head[1] * head[1]));
    energy = uni(args.w_energy * power);
    // failure containment (no reference counterpart): an env whose state stopped being finite is put
    // back on the start pose with zero velocities and reports done = 1 once, with a finite reward of 0 (one env
    // per wave: no other env can be affected).
    bool badl = !(fabsf(q) < 3.0e38f) || !(fabsf(qd) < 3.0e38f);
#pragma unroll
    for (int k = 0; k < 3; k++) badl |= !(fabsf(pos[k]) < 3.0e38f) || !(fabsf(bv[k]) < 3.0e38f) || !(fabsf(bw[k]) < 3.0e38f);
#pragma unroll
    for (int k = 0; k < 4; k++) badl |= !(fabsf(quat[k]) < 3.0e38f);
    env_bad = bad || (__ballot(badl) != 0ull);
  };
  auto to_start_pose = [&]() {
#pragma unroll
    for (int k = 0; k < 3; k++) { pos[k] = uni(M->base_pos0[k]); bv[k] = 0.f; bw[k] = 0.f; }
#pragma unroll
    for (int k = 0; k < 4; k++) quat[k] = uni(M->base_quat0[k]);
    const int l = lane_id();   // (not tid: an address formed from it would be kept - and spilled - from the prologue on)
    if (l < TL) {
      W.st[ST_Q][l] = l < nb ? Mo()->q_start[l] : 0.f;
      W.st[ST_QD][l] = 0.f; W.st[ST_TAU][l] = 0.f;
    }
    WSYNC();
  };
  const int n_launch_steps = MULTI ? args.n_steps : 1;
#pragma unroll 1
  for (int ls = 0; ls < n_launch_steps; ls++) {
  // ================================================================ one env-step
  if (!RESET) {
    // the joint targets of this step: clip(action) (np.clip, trex_env.py:147), parked per body lane
    const int l = lane_id();
    if (l < TL) {
      float target = 0.f;
      if (l >= 1 && l < nb) {
        const TrexDeviceModel *Mi = Mo();
        const float a = args.actions[((size_t)(MULTI ? ls : 0) * args.n_envs + env) * nj + Mi->obs_slot[l]];
        target = fminf(fmaxf(a, Mi->lower[l]), Mi->upper[l]);
      }
      W.st[ST_TARGET][l] = target;
    }
    if (MULTI && ls > 0) { motors_on = true; bad = false; env_bad = false; }
  }
  age = 0; time_up = false;
  if (!RESET && args.arr.max_episode_steps > 0) {
    age = steps_in + 1;
    time_up = age >= args.arr.max_episode_steps;
  }
  const int n_total = n_sub + ((!RESET && time_up) ? 1 : 0);
  int n_loop = n_total;
  int nt_pair[2] = {n_total, n_total};
  if (PAIR) {   // the two envs of the workgroup run the same number of loop trips (an env whose episode ends takes one more substep)
    if (lane_id() == 0) W.xch[7] = __int_as_float(n_total);
    __syncthreads();
    nt_pair[0] = uni(__float_as_int(Wpair[0].xch[7])); nt_pair[1] = uni(__float_as_int(Wpair[PAIR ? 1 : 0].xch[7]));
    n_loop = max(nt_pair[0], nt_pair[1]);
  }
#pragma unroll 1
  for (int sub = 0; sub < n_loop; sub++) {
    // lane id and what derives from it are RE-derived at the start of every phase (RELANE): a value that
    // lived from the top of the substep would be spilled across the phases in between
    int lt, bl;                        // bl: index into the [32]-wide model / state rows (lanes >= 32 alias, never used)
    bool is_body, is_joint;
#define RELANE() do { lt = lane_id(); bl = lt & (TL - 1); is_body = lt < nb; is_joint = lt >= 1 && lt < nb; } while (0)
    int psrc, depth;
    float R[9], r[3];
    float Sa[3], dpar[3];   // joint axis (world) and offset from the parent's origin: re-read from the record per phase
    int nc = 0;
    const bool act = !PAIR || sub < n_total;   // (PAIR: a wave whose env is done with this step only keeps the barriers)
    if (act) {
    if (!RESET && sub == n_sub) {   // time is up: the step is complete, the new episode starts (settle substep follows)
      finish_step();
      to_start_pose();
      motors_on = false;            // remove_joint_control, trex_robot.py:309
    }
    RELANE();
    {
      const TrexDeviceModel *Mi = Mo();
      const int parent = is_body ? Mi->parent[bl] : 0;
      psrc = parent < 0 ? 0 : parent;
      depth = is_body ? Mi->depth[bl] : -1;
    }
    {
      float dpar0[3], Sa0[3];
      forward_kinematics(lt, psrc, depth, R, r, dpar0, Sa0);
      if (lt < TL) {   // what later phases need of this body's pose and place in the tree goes to its record now
        float4 *rec = &W.body[BREC * lt];
        rec[0] = make_float4(Sa0[0], Sa0[1], Sa0[2], 0.f);
        rec[1] = make_float4(r[0], r[1], r[2], 0.f);
        rec[3] = make_float4(0.f, 0.f, __int_as_float(psrc + 256 * (depth < 0 ? 255 : depth)), 0.f);
        // children, 8 bits each (255 = none): the tip-to-base pass reads them here, not from the model (4 dependent
        // L2 round trips per level)
        const TrexDeviceModel *Mi = Mo();
        unsigned ch4 = 0u;
#pragma unroll
        for (int k = 0; k < MAXCH; k++) {
          const int c = is_body ? Mi->child[k][bl] : -1;
          ch4 |= (unsigned)(c < 0 ? 255 : c) << (8 * k);
        }
        rec[4] = make_float4(dpar0[0], dpar0[1], dpar0[2], __uint_as_float(ch4));
      }
    }
    // (axis, parent offset, parent and depth are re-read from the record by the phases that sweep the tree)
#define REAXIS() do { const float4 q0_ = W.body[BREC * bl], q4_ = W.body[BREC * bl + 4];                          \
                      Sa[0] = is_joint ? q0_.x : 0.f; Sa[1] = is_joint ? q0_.y : 0.f; Sa[2] = is_joint ? q0_.z : 0.f; \
                      dpar[0] = q4_.x; dpar[1] = q4_.y; dpar[2] = q4_.z; } while (0)
#define RETREE() do { const int lk_ = __float_as_int(reinterpret_cast<const float *>(&W.body[BREC * bl + 3])[2]); \
                      psrc = lk_ & 255; depth = is_body ? (lk_ >> 8) : -1; } while (0)
    if (PAIR) {   // hand this env's rotations, base twist and base height to the workgroup's LDS
      const int l_ = lane_id();
      if (l_ < TL) {
        float *rt = W.u.t.rtab[l_];
#pragma unroll
        for (int c = 0; c < 9; c++) rt[c] = R[c];
      }
      if (l_ == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { W.xch[c] = bw[c]; W.xch[3 + c] = bv[c]; }
        W.xch[6] = __int_as_float(env);
        W.xch[36] = pos[2];
      }
    }
    }   // act (kinematics)
    STAMP(0);
    if (PAIR) { __syncthreads(); SUBSTAMP(16); }      // both envs' body records, state rows, rotations, twists and heights are in LDS
    // (measured, not kept: FEEDBACK by arrival order at this barrier - the later wave of the pair one level up in its next sweeps,
    // the earlier one down: 11.73 against 11.69 M at 4096 envs, 6.72 against 6.75 M at 2048; the stamped pair build shows the heavy
    // env's wave waiting here 11 % of its time for the light env's wave, whose sweeps the priorities starve - but a waiting wave's
    // issue slots go to its SIMD mates, so levelling the pair moves nothing)
    // (measured, not kept: both waves at the priority of the pair's heavier env between the two barriers - 11.69 against 11.67 M
    // at 4096 envs, 6.73 against 6.94 M at 2048)
    // ROLES (PAIR): wave 1 generates the contacts of BOTH envs, one after the other, WHILE wave 0 runs the lane-per-body
    // dynamics of both (below): neither waits for the other's phase, a substep loses the shorter of the two
    if (!PAIR || wave == 1) {
    for (int e = 0; e < (PAIR ? 2 : 1); e++) {
    if (PAIR && !(sub < nt_pair[e])) continue;
    WaveLds &E = Wpair[PAIR ? e : 0];
    float posz = pos[2];
    if (PAIR) {
      RELANE();
      posz = E.xch[36];
      if (lt < TL) {
#pragma unroll
        for (int c = 0; c < 9; c++) R[c] = E.u.t.rtab[lt][c];
        const float4 q1_ = E.body[BREC * lt + 1];
        r[0] = q1_.x; r[1] = q1_.y; r[2] = q1_.z;
      }
    }

    // ================================================================ contact generation
    // hull vertices against z <= floor_z. Pass A finds, per body, WHICH vertices are inside the margin (bit masks
    // in LDS) and its DEEPEST such vertex (= the first point the selection rule keeps):
    //   * broad phase per HULL (scan unit), one lane each: the lowest point of the hull's oriented bounding box
    //     (only the z row of the body's rotation is needed);
    //   * the vertices of all near hulls form ONE list that the 64 lanes stride over together (lane l takes list
    //     positions l, l + 64, ...; a cursor walks the near-hull table), 4 loads in flight per lane: two feet on
    //     the ground are 700 vertices = 11 per lane, where a lane per body scanned up to 96 one after the other;
    //   * results by LDS atomics, which commute: or into the body's mask, min of (distance, vertex) - ties go to
    //     the lowest vertex index, as in the oracle.
    // Pass B revisits a body's in-margin vertices only when more than one point per body is kept (K >= 2).
    // The points go to LDS (W.cpt) in contact order; only their number nc stays in a register.
    nc = 0;
    {
      const TrexDeviceModel *Mi = Mo();
      const int hull_v0 = Mi->hull_start[is_body ? lt : nb], hull_v1 = Mi->hull_start[is_body ? lt + 1 : nb];
      CgLds &G = PAIR ? *reinterpret_cast<CgLds *>(Gpair) : *reinterpret_cast<CgLds *>(&W.u);
      // ---- broad phase, one hull per lane
      const int nchunk = Mi->nchunk;
      const bool is_chunk = lt < nchunk;
      const int cbody = Mi->chunk_body[bl], cv0 = Mi->chunk_v0[bl], cv1 = is_chunk ? Mi->chunk_v1[bl] : 0;
      float Rz[3], zbc;
      {
        const float zb = posz + r[2] - floor_z;    // body origin above the floor (body lanes)
#pragma unroll
        for (int c = 0; c < 3; c++) Rz[c] = wshfl(R[6 + c], cbody);
        zbc = wshfl(zb, cbody);
      }
      bool near = false;
      if (is_chunk) {
        const float cz = Rz[0] * Mi->chunk_c[0][bl] + Rz[1] * Mi->chunk_c[1][bl] + Rz[2] * Mi->chunk_c[2][bl];
        const float reach = fabsf(Rz[0]) * Mi->chunk_h[0][bl] + fabsf(Rz[1]) * Mi->chunk_h[1][bl] + fabsf(Rz[2]) * Mi->chunk_h[2][bl];
        near = cv1 > cv0 && (zbc + cz - reach < margin);
      }
      const unsigned near_mask = (unsigned)__ballot(near);
      if (near_mask != 0u) {   // (an env with no hull near the floor - every second one under random actions - is done here)
      // clear the masks and the minima
      {
        unsigned *z = &G.cm[0];
#pragma unroll
        for (int i = 0; i < (CG_WORDS + 63) / 64; i++)
          if (lt + 64 * i < CG_WORDS) z[lt + 64 * i] = 0u;
        if (lt < TL) {   // (the all-ones key made here, not hoisted out of the substep loop as a register pair)
          int ones = -1;
          asm volatile("" : "+v"(ones));
          *reinterpret_cast<int2 *>(&G.best[lt]) = make_int2(ones, ones);
        }
      }
      // ---- table of the near hulls: position of their first vertex in the list, ... (prefix sum over the set bits)
      int total = 0;
      {
        int my_off = 0;
        for (unsigned m = near_mask; m != 0u; m &= m - 1u) {
          const int k = __ffs(m) - 1;
          if (lt == k) my_off = total;
          total += rl(cv1 - cv0, k);
        }
        if (near) {
          const int e = __popc(near_mask & ((1u << bl) - 1u));
          G.ent[e][0] = make_float4(__int_as_float(my_off + (cv1 - cv0)), __int_as_float(cv0 - my_off),
                                    __int_as_float(cbody | (Mi->cm_pack[cbody] << 8)), __int_as_float(Mi->hull_start[cbody]));
          // (.z: body in bits 0..7, its mask's log2 period in 8..15, its first mask word from bit 16)
          G.ent[e][1] = make_float4(Rz[0], Rz[1], Rz[2], zbc);
        }
      }
      WSYNC();
      SUBSTAMP(9);    // broad phase + table
      // ---- the scan
      if (total > 0) {
        int cur = 0;
        float4 e0 = G.ent[0][0], e1 = G.ent[0][1];
        constexpr int UN = 4;
        for (int f0 = 0; f0 < total; f0 += 64 * UN) {
          float4 h[UN], q[UN];
          int vtx[UN], rel[UN], bod[UN], mlg[UN], mof[UN];
#pragma unroll
          for (int u = 0; u < UN; u++) {
            const int f = f0 + 64 * u + lt;
            // advance the cursor to the hull that holds list position f (hulls hold >= 1 vertex: a few steps at most)
            while (f < total && f >= __float_as_int(e0.x)) { cur++; e0 = G.ent[cur][0]; e1 = G.ent[cur][1]; }
            vtx[u] = f < total ? f + __float_as_int(e0.y) : -1;
            bod[u] = __float_as_int(e0.z) & 255;
            mlg[u] = (__float_as_int(e0.z) >> 8) & 255;
            mof[u] = __float_as_int(e0.z) >> 16;
            rel[u] = vtx[u] - __float_as_int(e0.w);
            q[u] = e1;
            h[u] = args.arr.hull[vtx[u] < 0 ? 0 : vtx[u]];
          }
#pragma unroll
          for (int u = 0; u < UN; u++) {
            // only the height decides; h.w = support radius (0 for a hull vertex): the sphere's lowest point
            const float dd = q[u].w + (q[u].x * h[u].x + q[u].y * h[u].y + q[u].z * h[u].z) - h[u].w;
            if (vtx[u] >= 0 && dd < margin) {
              if (mlg[u] != 0 && rel[u] < (32 << mlg[u])) atomicOr(&G.cm[mof[u] + (rel[u] & ((1 << mlg[u]) - 1))], 1u << (rel[u] >> mlg[u]));
              unsigned ub = __float_as_uint(dd);
              ub ^= (ub >> 31) ? 0xffffffffu : 0x80000000u;    // order-preserving map of the float to unsigned
              atomicMin(&G.best[bod[u]], ((unsigned long long)ub << 32) | (unsigned)vtx[u]);
            }
          }
        }
      }
      WSYNC();
      SUBSTAMP(10);   // scan
      // ---- per body: its deepest vertex
      unsigned active_mask = 0u;
      float a_x[3] = {0.f, 0.f, 0.f}, a_d = 0.f;   // lane b: deepest candidate of body b
      int a_v = -1;
      {
        const unsigned long long key = lt < TL ? G.best[bl] : ~0ull;
        if (is_body && key != ~0ull) {
          a_v = (int)(unsigned)(key & 0xffffffffull);
          const float4 hw = args.arr.hull[a_v];
          const float hv[3] = {hw.x, hw.y, hw.z};
          float w[3];
          matvec3(R, hv, w);
          a_x[0] = r[0] + w[0]; a_x[1] = r[1] + w[1]; a_x[2] = r[2] + w[2] - hw.w;
          a_d = posz + a_x[2] - floor_z;
        }
        active_mask = (unsigned)__ballot(a_v >= 0);
      }
      int n_active = __popc(active_mask);
      int K = n_active > 0 ? maxc / n_active : 0;
      K = K > 4 ? 4 : (K < 1 ? 1 : K);
      if (n_active > maxc) {
        // more touching bodies than contact rows: keep the maxc bodies whose deepest vertex is deepest
        // (ties -> lower body index), one point each; they stay in body order.
        int rank = 0;
        for (unsigned m = active_mask; m != 0u; m &= m - 1u) {
          const int b2 = __ffs(m) - 1;
          const float d2 = rl(a_d, b2);
          rank += (d2 < a_d || (d2 == a_d && b2 < lt)) ? 1 : 0;
        }
        active_mask = (unsigned)__ballot(lt < 32 && ((active_mask >> bl) & 1u) && rank < maxc);
        n_active = maxc;
      }
      if (K < 2) {
        // one point per touching body (the standing case): contact c is the c-th touching body's deepest vertex
        const bool mine = lt < 32 && ((active_mask >> bl) & 1u);
        const int slot = __popc(active_mask & ((1u << bl) - 1u));
        if (mine) {
          float *o = E.cpt[slot];
          o[0] = __int_as_float(lt); o[1] = a_x[0]; o[2] = a_x[1]; o[3] = a_x[2]; o[4] = a_d;
        }
        nc = n_active;
      } else {
        // ---- K >= 2 points per body, at most 6 touching bodies: the bodies are processed SIDE BY SIDE, one aligned
        // lane group each (64, 32, 16 or 8 lanes); lane g of a group owns the body's vertices g, g + GS, ... and finds
        // them in the mask words g, g + GS, ... (< 32). Every pass ends in group-wide DPP reductions: max of the
        // score, ties to the lowest vertex index (as the oracle's scan order gives), then the winner's position.
        const int GS = n_active <= 1 ? 64 : (n_active <= 2 ? 32 : (n_active <= 4 ? 16 : 8));
        const int g = lt & (GS - 1), gi = lt / GS;
        int b = 0;
        bool act = false;
        {
          int i = 0;
          for (unsigned am = active_mask; am != 0u; am &= am - 1u, i++)
            if (gi == i) { b = __ffs(am) - 1; act = true; }
        }
        float Rb[9], rb[3], px[4][3];
        int sel[3] = {-1, -1, -1};
#pragma unroll
        for (int c = 0; c < 9; c++) Rb[c] = wshfl(R[c], b);
#pragma unroll
        for (int c = 0; c < 3; c++) { rb[c] = wshfl(r[c], b); px[0][c] = wshfl(a_x[c], b); }
        sel[0] = act ? wshfl(a_v, b) : -1;
        const int v0 = wshfl(hull_v0, b), v1 = wshfl(hull_v1, b);
        const int cmp = Mi->cm_pack[b & (TL - 1)], mlog = cmp & 255;
        const bool masked = mlog != 0;
        // this lane's candidate words (masked bodies); a body without a mask is swept. Lane g of a group owns a FIXED subset
        // of the body's vertices - which one does not matter (the passes pick by score, ties by vertex index):
        //   period 32: the vertices congruent to g modulo the group size (words g, g + GS, ...; GS = 64: every other bit);
        //   period 8:  word g & 7, and of its bits those congruent to g >> 3 modulo GS / 8 - ONE word per lane
        unsigned m0 = 0u, m1 = 0u, m2 = 0u, m3 = 0u;
        if (act && masked) {
          const unsigned *cw = G.cm + (cmp >> 8);
          if (mlog == 3) {
            const unsigned pick = GS >= 64 ? 0x01010101u : (GS >= 32 ? 0x11111111u : (GS >= 16 ? 0x55555555u : 0xFFFFFFFFu));
            m0 = cw[g & 7] & (pick << (g >> 3));
          } else if (GS >= 64) m0 = cw[g & 31] & ((g >> 5) ? 0xAAAAAAAAu : 0x55555555u);
          else if (GS >= 32) m0 = cw[g];
          else if (GS >= 16) { m0 = cw[g]; m1 = cw[g + 16]; }
          else { m0 = cw[g]; m1 = cw[g + 8]; m2 = cw[g + 16]; m3 = cw[g + 24]; }
        }
        const int wstep = (mlog == 3 || GS >= 32) ? 0 : GS;
        const int vmul = mlog == 3 ? 8 : 32, wb0 = mlog == 3 ? (g & 7) : (g & 31);   // bit j of the word at base wb: vertex v0 + vmul j + wb
        int nsel = act ? 1 : 0;
        bool stop = !act;
        // A lane with at most CC in-margin vertices (every lane of a body of toe size) loads and places them ONCE and
        // keeps the positions for all passes: one trip to L2 per substep instead of one per pass, no vertex placed
        // twice. If any lane holds more (or a body is beyond the mask capacity) the passes re-read, as before.
        constexpr int CC = 4;
        SUBSTAMP(12);   // deepest vertices, ranking, group set-up
        const bool cached = __ballot(act && (!masked || __popc(m0) + __popc(m1) + __popc(m2) + __popc(m3) > CC)) == 0ull;
        float cx[CC][3];
        int cvx[CC];
#pragma unroll
        for (int u = 0; u < CC; u++) { cvx[u] = -1; cx[u][0] = cx[u][1] = cx[u][2] = 0.f; }
        if (cached) {
          unsigned c0 = m0, c1 = m1, c2 = m2, c3 = m3;
          int wb = wb0;
          float4 hc[CC];
#pragma unroll
          for (int u = 0; u < CC; u++) {
            if (c0 == 0u) { c0 = c1; c1 = c2; c2 = c3; c3 = 0u; wb += wstep; }   // next word of this lane
            const int j = c0 != 0u ? (__ffs(c0) - 1) : -1;
            c0 &= c0 - 1u;            // (0 stays 0)
            cvx[u] = j >= 0 ? v0 + vmul * j + wb : -1;
            hc[u] = args.arr.hull[cvx[u] >= 0 ? cvx[u] : v0];
          }
#pragma unroll
          for (int u = 0; u < CC; u++) {
            const float hv[3] = {hc[u].x, hc[u].y, hc[u].z};
            float w[3];
            matvec3(Rb, hv, w);
            cx[u][0] = rb[0] + w[0]; cx[u][1] = rb[1] + w[1]; cx[u][2] = rb[2] + w[2] - hc[u].w;
            const float dd = posz + cx[u][2] - floor_z;
            if (!(dd < margin)) cvx[u] = -1;
          }
        }
#pragma unroll
        for (int pass = 1; pass < 4; pass++) {
          if (pass >= K || __ballot(!stop) == 0ull) break;
          float bs = -3.0e38f;
          int bi = 0x7fffffff;
          float ex = 0.f, ey = 0.f, flip = 1.f;
          if (pass >= 2) { ex = px[1][0] - px[0][0]; ey = px[1][1] - px[0][1]; }
          if (pass == 3) {
            const float c3 = ex * (px[2][1] - px[0][1]) - ey * (px[2][0] - px[0][0]);
            flip = c3 > 0.f ? -1.f : 1.f;
          }
          float bx[3] = {0.f, 0.f, 0.f};
          auto visit = [&](int v, const float4 h) {
            const float hv[3] = {h.x, h.y, h.z};
            float w[3];
            matvec3(Rb, hv, w);
            const float x0 = rb[0] + w[0], x1 = rb[1] + w[1], x2 = rb[2] + w[2] - h.w;
            const float dd = posz + x2 - floor_z;
            if (!(dd < margin)) return;
            if (v == sel[0] || v == sel[1] || v == sel[2]) return;
            const float dx = x0 - px[0][0], dy = x1 - px[0][1];
            float score;
            if (pass == 1) score = dx * dx + dy * dy;
            else {
              const float cr = ex * dy - ey * dx;
              score = (pass == 2) ? fabsf(cr) : flip * cr;
            }
            if (score > bs || (score == bs && v < bi)) { bs = score; bi = v; bx[0] = x0; bx[1] = x1; bx[2] = x2; }
          };
          if (cached) {
#pragma unroll
            for (int u = 0; u < CC; u++) {
              const int v = cvx[u];
              if (v < 0 || stop || v == sel[0] || v == sel[1] || v == sel[2]) continue;
              const float dx = cx[u][0] - px[0][0], dy = cx[u][1] - px[0][1];
              float score;
              if (pass == 1) score = dx * dx + dy * dy;
              else {
                const float cr = ex * dy - ey * dx;
                score = (pass == 2) ? fabsf(cr) : flip * cr;
              }
              if (score > bs || (score == bs && v < bi)) { bs = score; bi = v; bx[0] = cx[u][0]; bx[1] = cx[u][1]; bx[2] = cx[u][2]; }
            }
          } else if (__ballot(act && !masked) == 0ull) {
            unsigned c0 = stop ? 0u : m0, c1 = stop ? 0u : m1, c2 = stop ? 0u : m2, c3 = stop ? 0u : m3;
            int wb = wb0;
            constexpr int UC = 2;   // candidates per trip: their loads are issued together
            while (__ballot((c0 | c1 | c2 | c3) != 0u) != 0ull) {
              int vi[UC];
              float4 hc[UC];
#pragma unroll
              for (int u = 0; u < UC; u++) {
                if (c0 == 0u) { c0 = c1; c1 = c2; c2 = c3; c3 = 0u; wb += wstep; }   // next word of this lane
                const int j = c0 != 0u ? (__ffs(c0) - 1) : -1;
                c0 &= c0 - 1u;            // (0 stays 0)
                vi[u] = j >= 0 ? v0 + vmul * j + wb : -1;
                hc[u] = args.arr.hull[vi[u] >= 0 ? vi[u] : v0];
              }
#pragma unroll
              for (int u = 0; u < UC; u++)
                if (vi[u] >= 0) visit(vi[u], hc[u]);
            }
          } else if (!stop) {
            for (int v = v0 + g; v < v1; v += GS) visit(v, args.arr.hull[v]);
          }
          const int mine = bi;
          const float best = gmaxf(bs, GS);
          bi = gmini(bs == best ? bi : 0x7fffffff, GS);
          if (bi == 0x7fffffff || !(best > 0.f)) stop = true;
          const bool own = !stop && mine == bi;   // exactly one lane of the group: its position goes to the group
          {
            const unsigned long long ob = __ballot(own) >> (lt & ~(GS - 1));   // bit i: lane i of MY group owns
            const int src = (lt & ~(GS - 1)) + (ob != 0ull ? __ffsll((unsigned long long)ob) - 1 : 0);
#pragma unroll
            for (int c = 0; c < 3; c++) {
              const float wx = wshfl(bx[c], src);
              if (!stop) px[pass][c] = wx;
            }
          }
          if (!stop) {
            if (pass < 3) sel[pass] = bi;
            nsel = pass + 1;
          }
        }
        SUBSTAMP(13);   // candidate fill and the selection passes
        // the points go out in body order (= group order), the deepest vertex of a body first
        int off = 0;
        for (int i = 0; i < n_active; i++) {
          const int ni = rl(nsel, i * GS);
          if (gi > i) off += ni;
          nc += ni;
        }
        if (act && g == 0) {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            if (k < nsel && off + k < maxc) {
              float *o = E.cpt[off + k];
              o[0] = __int_as_float(b); o[1] = px[k][0]; o[2] = px[k][1]; o[3] = px[k][2]; o[4] = posz + px[k][2] - floor_z;
            }
          }
        }
        nc = nc < maxc ? nc : maxc;
      }
      }   // near_mask != 0
    }
    nc = uni(nc);
    if (PAIR) { if (lane_id() == 0) E.xch[35] = __int_as_float(nc); }
    }   // e
    }   // contact generation (PAIR: wave 1)
    STAMP(1);
    if (!PAIR) {
    RELANE();
    RETREE();
    REAXIS();
    }

    Chol6 I0c;
    float a0[6];
    float nw[3], nv[3];
#ifndef TREX_GENERIC_TREE
#define TREX_GENERIC_TREE 0      // diagnostic: the single-env launch through the two-env form of the tree phases (with one half)
#endif
    if (!PAIR && !TREX_GENERIC_TREE) {
    // ================================================================ tree dynamics
    // ---- rigid-body spatial inertia about the body origin, bias force (both straight to the body's LDS slot:
    // the tip-to-base pass works on LDS-resident inertias), velocity-product acceleration cv (registers)
    float cv[6];
    {
      Sym6 IA;
      float pA[6];
      const float qd = W.st[ST_QD][bl];
      // spatial velocity of every body ABOUT ITS OWN ORIGIN for the base twist and the joint rates
      float vel[6];
#pragma unroll
      for (int c = 0; c < 3; c++) { vel[c] = bw[c]; vel[3 + c] = bv[c]; }
      for (int d = 1; d <= maxdepth; d++) {
        float pv[6];
#pragma unroll
        for (int c = 0; c < 6; c++) pv[c] = wshfl(vel[c], psrc);
        if (depth == d) {
          float wxd[3];
          cross3(pv, dpar, wxd);   // velocity of the parent-body point at this body's origin
#pragma unroll
          for (int c = 0; c < 3; c++) { vel[c] = pv[c] + Sa[c] * qd; vel[3 + c] = pv[3 + c] + wxd[c]; }
        }
      }
      const TrexDeviceModel *Mi = Mo();
      float comb[3], inb[6];
      const float mscale = args.arr.domain ? args.arr.mass_scale[(size_t)env * TL + bl] : 1.0f;
      const float mass = Mi->mass[bl] * mscale;
#pragma unroll
      for (int c = 0; c < 3; c++) comb[c] = Mi->com[c][bl];
#pragma unroll
      for (int c = 0; c < 6; c++) inb[c] = Mi->inertia[c][bl];
      const float grav = Mi->prm[TP_GRAVITY], kdamp = Mi->prm[TP_LINK_DAMPING];
      float comw[3], Icw[6];   // comw = COM offset from the body origin, world axes
      {
        matvec3(R, comb, comw);
        // Ic_world = R Ib R^T (symmetric)
        float t[9];
        const float Ib[9] = {inb[0], inb[1], inb[2], inb[1], inb[3], inb[4], inb[2], inb[4], inb[5]};
        matmul3(R, Ib, t);
        const int ia[6] = {0, 0, 0, 1, 1, 2}, ib[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
        for (int k = 0; k < 6; k++)
          Icw[k] = mscale * (t[3 * ia[k]] * R[3 * ib[k]] + t[3 * ia[k] + 1] * R[3 * ib[k] + 1] + t[3 * ia[k] + 2] * R[3 * ib[k] + 2]);
      }
      {
        const float cc = dot3(comw, comw);
        IA.A[0] = Icw[0] + mass * (cc - comw[0] * comw[0]);
        IA.A[1] = Icw[1] - mass * comw[0] * comw[1];
        IA.A[2] = Icw[2] - mass * comw[0] * comw[2];
        IA.A[3] = Icw[3] + mass * (cc - comw[1] * comw[1]);
        IA.A[4] = Icw[4] - mass * comw[1] * comw[2];
        IA.A[5] = Icw[5] + mass * (cc - comw[2] * comw[2]);
        // B = m * [c]x
        IA.B[0] = 0.f;              IA.B[1] = -mass * comw[2];  IA.B[2] = mass * comw[1];
        IA.B[3] = mass * comw[2];   IA.B[4] = 0.f;              IA.B[5] = -mass * comw[0];
        IA.B[6] = -mass * comw[1];  IA.B[7] = mass * comw[0];   IA.B[8] = 0.f;
        IA.C[0] = mass; IA.C[1] = 0.f; IA.C[2] = 0.f; IA.C[3] = mass; IA.C[4] = 0.f; IA.C[5] = mass;
      }
      if (!is_body) {
#pragma unroll
        for (int k = 0; k < 6; k++) { IA.A[k] = (k == 0 || k == 3 || k == 5) ? 1.f : 0.f; IA.C[k] = IA.A[k]; }
#pragma unroll
        for (int k = 0; k < 9; k++) IA.B[k] = 0.f;
      }
      {
        float h[6];
        sym6_mul(IA, vel, h);
        // v x* h
        float a[3], b[3], c[3];
        cross3(vel, h, a); cross3(vel + 3, h + 3, b); cross3(vel, h + 3, c);
#pragma unroll
        for (int k = 0; k < 3; k++) { pA[k] = a[k] + b[k]; pA[3 + k] = c[k]; }
        float f[3] = {0.f, 0.f, -mass * grav}, n[3] = {0.f, 0.f, 0.f};
        if (kdamp > 0.f) {
          float vc[3], wxc[3], Iw[3];
          cross3(vel, comw, wxc);
#pragma unroll
          for (int k = 0; k < 3; k++) vc[k] = vel[3 + k] + wxc[k];
          const float sv = sqrtf(dot3(vc, vc)), sw = sqrtf(dot3(vel, vel));
          sym3_mul(Icw, vel, Iw);
#pragma unroll
          for (int k = 0; k < 3; k++) {
            f[k] -= mass * vc[k] * (kdamp + kdamp * sv);
            n[k] -= Iw[k] * (kdamp + kdamp * sw);
          }
        }
        float cxf[3];
        cross3(comw, f, cxf);
#pragma unroll
        for (int k = 0; k < 3; k++) { pA[k] -= n[k] + cxf[k]; pA[3 + k] -= f[k]; }
        // c = vel x (S qd), S = [Sa; 0]
        float sq[3];
#pragma unroll
        for (int k = 0; k < 3; k++) sq[k] = Sa[k] * qd;
        float x0[3], x2[3];
        cross3(vel, sq, x0); cross3(vel + 3, sq, x2);
#pragma unroll
        for (int k = 0; k < 3; k++) { cv[k] = x0[k]; cv[3 + k] = x2[k]; }
        if (!is_body) {
#pragma unroll
          for (int k = 0; k < 6; k++) { pA[k] = 0.f; cv[k] = 0.f; }
        }
      }
      if (is_body) {
        float *o = W.u.t.aba[bl];
#pragma unroll
        for (int k = 0; k < 6; k++) { o[k] = IA.A[k]; o[15 + k] = IA.C[k]; o[21 + k] = pA[k]; }
#pragma unroll
        for (int k = 0; k < 9; k++) o[6 + k] = IA.B[k];
      }
    }
    WSYNC();
    STAMP(2);
    RELANE();
    RETREE();
    REAXIS();

    // ---- ABA pass 2 (tip to base) on LDS-resident inertias: slot b of W.u.t.aba holds body b's rigid-body inertia
    // (21) and bias force (6) about its own origin. Level by level, the lanes AT depth d take their slot, add
    // what their children left in theirs (already shifted to this body's origin; fixed order), form U, 1/D, u
    // (which go to the body record: pass 3 and the row walks read them there), remove the joint's freedom,
    // shift to the parent's origin and put the result back for the parent. One LDS round trip and one barrier
    // per level; nothing of this is carried in registers between levels. Level 0 is the base: it only sums.
    {
      const TrexDeviceModel *Mi = Mo();
      const float tau_j = -Mi->damp[bl] * W.st[ST_QD][bl];  // explicit joint damping torque
      for (int d = maxdepth; d >= 0; d--) {
        if (depth == d) {
          float *o = W.u.t.aba[bl];
          const unsigned ch4 = __float_as_uint(reinterpret_cast<const float *>(&W.body[BREC * bl + 4])[3]);
          float acc[27];
          {
            // own slot and first child's in flight together (most bodies have exactly one child)
            const int c0 = (int)(ch4 & 255u);
            const float *c = W.u.t.aba[c0 == 255 ? bl : c0];
            const float w0 = c0 == 255 ? 0.f : 1.f;
#pragma unroll
            for (int k = 0; k < 27; k++) acc[k] = __builtin_fmaf(w0, c[k], o[k]);
          }
#pragma unroll 1
          for (int kc = 1; kc < MAXCH; kc++) {   // further children, fixed order (packed without gaps)
            const int ch = (int)((ch4 >> (8 * kc)) & 255u);
            if (ch == 255) break;
            const float *c = W.u.t.aba[ch];
#pragma unroll
            for (int k = 0; k < 27; k++) acc[k] += c[k];
          }
          if (d == 0) {
#pragma unroll
            for (int k = 0; k < 27; k++) o[k] = acc[k];
          } else {
            Sym6 IA;
            float pA[6];
#pragma unroll
            for (int k = 0; k < 6; k++) { IA.A[k] = acc[k]; IA.C[k] = acc[15 + k]; pA[k] = acc[21 + k]; }
#pragma unroll
            for (int k = 0; k < 9; k++) IA.B[k] = acc[6 + k];
            float U[6];   // U = IA S, S = [Sa; 0]
            sym3_mul(IA.A, Sa, U);
            U[3] = IA.B[0] * Sa[0] + IA.B[3] * Sa[1] + IA.B[6] * Sa[2];
            U[4] = IA.B[1] * Sa[0] + IA.B[4] * Sa[1] + IA.B[7] * Sa[2];
            U[5] = IA.B[2] * Sa[0] + IA.B[5] * Sa[1] + IA.B[8] * Sa[2];
            const float D = dot3(Sa, U);
            const float rD = __builtin_amdgcn_rcpf(D);
#if TREX_ABLATE_EXACT_MATH
            const float invD = 1.0f / D;
#else
            const float invD = rD * __builtin_fmaf(-D, rD, 2.0f);   // v_rcp_f32 + one Newton step (no IEEE division expansion)
#endif
            const float u = tau_j - dot3(Sa, pA);
            float Ud[6];
#pragma unroll
            for (int k = 0; k < 6; k++) Ud[k] = U[k] * invD;
            {
              float4 *rec = &W.body[BREC * bl];
              rec[0] = make_float4(Sa[0], Sa[1], Sa[2], invD);
              rec[2] = make_float4(Ud[0], Ud[1], Ud[2], Ud[3]);
              rec[3] = make_float4(Ud[4], Ud[5], __int_as_float(psrc + 256 * depth), u * invD);
            }
            {   // pa = pA + Ia c + U u / D with Ia c = IA c - U (U.c) / D
              float Ic[6];
              sym6_mul(IA, cv, Ic);
              const float coef = (u - dot6(U, cv)) * invD;
#pragma unroll
              for (int k = 0; k < 6; k++) pA[k] += Ic[k] + U[k] * coef;
            }
            sym6_rank1_sub(IA, U, Ud);
            // shift both to the parent's origin (this origin = parent origin + d, d = dpar):
            //   n' = n + d x f,  B' = B + [d]x C,  A' = A + X^T + X', X = [d]x B^T, X' = [d]x B'^T
            {
              cross3_acc(dpar, pA + 3, pA[0], pA[1], pA[2]);
              const int sidx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
#pragma unroll
              for (int i = 0; i < 3; i++) {   // A_ij += X_ji = (d x row i of B)_j, j >= i
                float t[3] = {0.f, 0.f, 0.f};
                if (i == 0) cross3_acc(dpar, IA.B, IA.A[0], IA.A[1], IA.A[2]);
                else if (i == 1) cross3_acc(dpar, IA.B + 3, t[0], IA.A[3], IA.A[4]);
                else cross3_acc(dpar, IA.B + 6, t[0], t[1], IA.A[5]);
              }
#pragma unroll
              for (int j = 0; j < 3; j++) {   // column j of [d]x C = d x (column j of C)
                const float cj[3] = {IA.C[sidx[0][j]], IA.C[sidx[1][j]], IA.C[sidx[2][j]]};
                cross3_acc(dpar, cj, IA.B[j], IA.B[3 + j], IA.B[6 + j]);
              }
#pragma unroll
              for (int j = 0; j < 3; j++) {   // A_ij += X'_ij = (d x row j of B')_i, i <= j
                float t[3] = {0.f, 0.f, 0.f};
                if (j == 0) cross3_acc(dpar, IA.B, IA.A[0], t[1], t[2]);
                else if (j == 1) cross3_acc(dpar, IA.B + 3, IA.A[1], IA.A[3], t[2]);
                else cross3_acc(dpar, IA.B + 6, IA.A[2], IA.A[4], IA.A[5]);
              }
            }
#pragma unroll
            for (int k = 0; k < 6; k++) { o[k] = IA.A[k]; o[15 + k] = IA.C[k]; o[21 + k] = pA[k]; }
#pragma unroll
            for (int k = 0; k < 9; k++) o[6 + k] = IA.B[k];
          }
        }
        WSYNC();
      }
      if (lt < TL && !is_joint) {   // base and unused lanes: inert records
        float4 *rec = &W.body[BREC * lt];
        rec[0] = make_float4(0.f, 0.f, 0.f, 0.f);
        rec[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        rec[3] = make_float4(0.f, 0.f, __int_as_float(psrc + 256 * (depth < 0 ? 255 : depth)), 0.f);
      }
    }
    STAMP(3);
    RELANE();
    RETREE();
    REAXIS();

    // ---- floating base: a0 = -(IA_0)^-1 pA_0; the Cholesky factor of IA_0 is wave-uniform (SGPRs)
    {
      const float *o = W.u.t.aba[0];   // every lane reads the same words: LDS broadcast
      Sym6 I0;
#pragma unroll
      for (int k = 0; k < 6; k++) { I0.A[k] = o[k]; I0.C[k] = o[15 + k]; }
#pragma unroll
      for (int k = 0; k < 9; k++) I0.B[k] = o[6 + k];
      float p0[6];
#pragma unroll
      for (int k = 0; k < 6; k++) p0[k] = -o[21 + k];
      float full[36];
      sym6_full(I0, full);
      Chol6 c;
      chol6_factor(full, c);
#pragma unroll
      for (int k = 0; k < 15; k++) I0c.l[k] = uni(c.l[k]);
#pragma unroll
      for (int k = 0; k < 6; k++) I0c.il[k] = uni(c.il[k]);
      chol6_solve(I0c, p0, a0);
#pragma unroll
      for (int k = 0; k < 6; k++) a0[k] = uni(a0[k]);
    }
    // ---- ABA pass 3 (base to tip): accelerations; qdd = (u - U.a) / D = u/D - (U/D).a from the body record
    float qdd = 0.f;
    {
      const float4 q2 = W.body[BREC * bl + 2], q3 = W.body[BREC * bl + 3];
      const float Ud[6] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y};
      float acc[6];
#pragma unroll
      for (int k = 0; k < 6; k++) acc[k] = a0[k];
      for (int d = 1; d <= maxdepth; d++) {
        float pa[6];
#pragma unroll
        for (int k = 0; k < 6; k++) pa[k] = wshfl(acc[k], psrc);
        if (depth == d) {
          float axd[3];
          cross3(pa, dpar, axd);   // parent acceleration seen at this body's origin
#pragma unroll
          for (int k = 0; k < 3; k++) pa[3 + k] += axd[k];
#pragma unroll
          for (int k = 0; k < 6; k++) pa[k] += cv[k];
          qdd = q3.w - dot6(Ud, pa);
#pragma unroll
          for (int k = 0; k < 3; k++) acc[k] = pa[k] + Sa[k] * qdd;
#pragma unroll
          for (int k = 3; k < 6; k++) acc[k] = pa[k];
        }
      }
    }
    // ---- unconstrained velocity update
    {
      const float vmax = M->prm[TP_MAX_COORD_VEL];
      float wxv[3];
      cross3(bw, bv, wxv);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        nw[k] = uni(fminf(fmaxf(bw[k] + a0[k] * dt, -vmax), vmax));
        nv[k] = uni(fminf(fmaxf(bv[k] + (a0[3 + k] + wxv[k]) * dt, -vmax), vmax));
      }
      const float nqd = is_joint ? fminf(fmaxf(W.st[ST_QD][bl] + qdd * dt, -vmax), vmax) : 0.f;
      // ---- the last thing the row walks need of a body: its updated joint rate
      if (lt < TL) {
        reinterpret_cast<float *>(&W.body[BREC * lt + 1])[3] = nqd;
        W.st[ST_NQD][lt] = nqd;
      }
      if (DEBUG && args.debug && wg == 0) {
        float *D = args.debug;
        if (lt < TL) { D[lt] = qdd; D[64 + lt] = nqd; }
#pragma unroll
        for (int k = 0; k < 6; k++)
          if (lt == k) D[32 + k] = a0[k];
        if (lt == 0) {
          for (int k = 0; k < 3; k++) { D[64 + nb + k] = nw[k]; D[64 + nb + 3 + k] = nv[k]; }
        }
      }
    }
    } else {
      if (wave == 0) {
        // ... which runs the lane-per-body phases for BOTH: H = the LDS of the lane's half, hoff = its first lane
        WaveLds *H;
        int hoff;
#define RELANE2() do { lt = lane_id(); bl = lt & (TL - 1); hoff = PAIR ? (lt & TL) : 0; H = &Wpair[PAIR ? (lt >> 5) : 0];            \
                       is_body = (PAIR ? bl : lt) < nb; is_joint = (PAIR ? bl : lt) >= 1 && (PAIR ? bl : lt) < nb; } while (0)
#define REAXIS2() do { const float4 q0_ = H->body[BREC * bl], q4_ = H->body[BREC * bl + 4];                          \
                       Sa[0] = is_joint ? q0_.x : 0.f; Sa[1] = is_joint ? q0_.y : 0.f; Sa[2] = is_joint ? q0_.z : 0.f; \
                       dpar[0] = q4_.x; dpar[1] = q4_.y; dpar[2] = q4_.z; } while (0)
#define RETREE2() do { const int lk_ = __float_as_int(reinterpret_cast<const float *>(&H->body[BREC * bl + 3])[2]); \
                       psrc = lk_ & 255; depth = is_body ? (lk_ >> 8) : -1; } while (0)
        RELANE2();
        RETREE2();
        REAXIS2();
    // ================================================================ tree dynamics
    // ---- rigid-body spatial inertia about the body origin, bias force (both straight to the body's LDS slot:
    // the tip-to-base pass works on LDS-resident inertias), velocity-product acceleration cv (registers)
    float cv[6];
    {
      Sym6 IA;
      float pA[6];
      const float qd = H->st[ST_QD][bl];
      // spatial velocity of every body ABOUT ITS OWN ORIGIN for the base twist and the joint rates
      float vel[6];
#pragma unroll
      for (int c = 0; c < 3; c++) { vel[c] = H->xch[c]; vel[3 + c] = H->xch[3 + c]; }
      for (int d = 1; d <= maxdepth; d++) {
        float pv[6];
#pragma unroll
        for (int c = 0; c < 6; c++) pv[c] = wshfl(vel[c], psrc + hoff);
        if (depth == d) {
          float wxd[3];
          cross3(pv, dpar, wxd);   // velocity of the parent-body point at this body's origin
#pragma unroll
          for (int c = 0; c < 3; c++) { vel[c] = pv[c] + Sa[c] * qd; vel[3 + c] = pv[3 + c] + wxd[c]; }
        }
      }
      float Rh[9];
#pragma unroll
      for (int c = 0; c < 9; c++) Rh[c] = H->u.t.rtab[bl][c];
      const TrexDeviceModel *Mi = Mo();
      float comb[3], inb[6];
      const float mscale = args.arr.domain ? args.arr.mass_scale[(size_t)__float_as_int(H->xch[6]) * TL + bl] : 1.0f;
      const float mass = Mi->mass[bl] * mscale;
#pragma unroll
      for (int c = 0; c < 3; c++) comb[c] = Mi->com[c][bl];
#pragma unroll
      for (int c = 0; c < 6; c++) inb[c] = Mi->inertia[c][bl];
      const float grav = Mi->prm[TP_GRAVITY], kdamp = Mi->prm[TP_LINK_DAMPING];
      float comw[3], Icw[6];   // comw = COM offset from the body origin, world axes
      {
        matvec3(Rh, comb, comw);
        // Ic_world = R Ib R^T (symmetric)
        float t[9];
        const float Ib[9] = {inb[0], inb[1], inb[2], inb[1], inb[3], inb[4], inb[2], inb[4], inb[5]};
        matmul3(Rh, Ib, t);
        const int ia[6] = {0, 0, 0, 1, 1, 2}, ib[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
        for (int k = 0; k < 6; k++)
          Icw[k] = mscale * (t[3 * ia[k]] * Rh[3 * ib[k]] + t[3 * ia[k] + 1] * Rh[3 * ib[k] + 1] + t[3 * ia[k] + 2] * Rh[3 * ib[k] + 2]);
      }
      {
        const float cc = dot3(comw, comw);
        IA.A[0] = Icw[0] + mass * (cc - comw[0] * comw[0]);
        IA.A[1] = Icw[1] - mass * comw[0] * comw[1];
        IA.A[2] = Icw[2] - mass * comw[0] * comw[2];
        IA.A[3] = Icw[3] + mass * (cc - comw[1] * comw[1]);
        IA.A[4] = Icw[4] - mass * comw[1] * comw[2];
        IA.A[5] = Icw[5] + mass * (cc - comw[2] * comw[2]);
        // B = m * [c]x
        IA.B[0] = 0.f;              IA.B[1] = -mass * comw[2];  IA.B[2] = mass * comw[1];
        IA.B[3] = mass * comw[2];   IA.B[4] = 0.f;              IA.B[5] = -mass * comw[0];
        IA.B[6] = -mass * comw[1];  IA.B[7] = mass * comw[0];   IA.B[8] = 0.f;
        IA.C[0] = mass; IA.C[1] = 0.f; IA.C[2] = 0.f; IA.C[3] = mass; IA.C[4] = 0.f; IA.C[5] = mass;
      }
      if (!is_body) {
#pragma unroll
        for (int k = 0; k < 6; k++) { IA.A[k] = (k == 0 || k == 3 || k == 5) ? 1.f : 0.f; IA.C[k] = IA.A[k]; }
#pragma unroll
        for (int k = 0; k < 9; k++) IA.B[k] = 0.f;
      }
      {
        float h[6];
        sym6_mul(IA, vel, h);
        // v x* h
        float a[3], b[3], c[3];
        cross3(vel, h, a); cross3(vel + 3, h + 3, b); cross3(vel, h + 3, c);
#pragma unroll
        for (int k = 0; k < 3; k++) { pA[k] = a[k] + b[k]; pA[3 + k] = c[k]; }
        float f[3] = {0.f, 0.f, -mass * grav}, n[3] = {0.f, 0.f, 0.f};
        if (kdamp > 0.f) {
          float vc[3], wxc[3], Iw[3];
          cross3(vel, comw, wxc);
#pragma unroll
          for (int k = 0; k < 3; k++) vc[k] = vel[3 + k] + wxc[k];
          const float sv = sqrtf(dot3(vc, vc)), sw = sqrtf(dot3(vel, vel));
          sym3_mul(Icw, vel, Iw);
#pragma unroll
          for (int k = 0; k < 3; k++) {
            f[k] -= mass * vc[k] * (kdamp + kdamp * sv);
            n[k] -= Iw[k] * (kdamp + kdamp * sw);
          }
        }
        float cxf[3];
        cross3(comw, f, cxf);
#pragma unroll
        for (int k = 0; k < 3; k++) { pA[k] -= n[k] + cxf[k]; pA[3 + k] -= f[k]; }
        // c = vel x (S qd), S = [Sa; 0]
        float sq[3];
#pragma unroll
        for (int k = 0; k < 3; k++) sq[k] = Sa[k] * qd;
        float x0[3], x2[3];
        cross3(vel, sq, x0); cross3(vel + 3, sq, x2);
#pragma unroll
        for (int k = 0; k < 3; k++) { cv[k] = x0[k]; cv[3 + k] = x2[k]; }
        if (!is_body) {
#pragma unroll
          for (int k = 0; k < 6; k++) { pA[k] = 0.f; cv[k] = 0.f; }
        }
      }
      if (is_body) {
        float *o = H->u.t.aba[bl];
#pragma unroll
        for (int k = 0; k < 6; k++) { o[k] = IA.A[k]; o[15 + k] = IA.C[k]; o[21 + k] = pA[k]; }
#pragma unroll
        for (int k = 0; k < 9; k++) o[6 + k] = IA.B[k];
      }
    }
    WSYNC();
    RELANE2();
    RETREE2();
    REAXIS2();

    // ---- ABA pass 2 (tip to base) on LDS-resident inertias: slot b of H->u.t.aba holds body b's rigid-body inertia
    // (21) and bias force (6) about its own origin. Level by level, the lanes AT depth d take their slot, add
    // what their children left in theirs (already shifted to this body's origin; fixed order), form U, 1/D, u
    // (which go to the body record: pass 3 and the row walks read them there), remove the joint's freedom,
    // shift to the parent's origin and put the result back for the parent. One LDS round trip and one barrier
    // per level; nothing of this is carried in registers between levels. Level 0 is the base: it only sums.
    {
      const TrexDeviceModel *Mi = Mo();
      const float tau_j = -Mi->damp[bl] * H->st[ST_QD][bl];  // explicit joint damping torque
      for (int d = maxdepth; d >= 0; d--) {
        if (depth == d) {
          float *o = H->u.t.aba[bl];
          const unsigned ch4 = __float_as_uint(reinterpret_cast<const float *>(&H->body[BREC * bl + 4])[3]);
          float acc[27];
          {
            // own slot and first child's in flight together (most bodies have exactly one child)
            const int c0 = (int)(ch4 & 255u);
            const float *c = H->u.t.aba[c0 == 255 ? bl : c0];
            const float w0 = c0 == 255 ? 0.f : 1.f;
#pragma unroll
            for (int k = 0; k < 27; k++) acc[k] = __builtin_fmaf(w0, c[k], o[k]);
          }
#pragma unroll 1
          for (int kc = 1; kc < MAXCH; kc++) {   // further children, fixed order (packed without gaps)
            const int ch = (int)((ch4 >> (8 * kc)) & 255u);
            if (ch == 255) break;
            const float *c = H->u.t.aba[ch];
#pragma unroll
            for (int k = 0; k < 27; k++) acc[k] += c[k];
          }
          if (d == 0) {
#pragma unroll
            for (int k = 0; k < 27; k++) o[k] = acc[k];
          } else {
            Sym6 IA;
            float pA[6];
#pragma unroll
            for (int k = 0; k < 6; k++) { IA.A[k] = acc[k]; IA.C[k] = acc[15 + k]; pA[k] = acc[21 + k]; }
#pragma unroll
            for (int k = 0; k < 9; k++) IA.B[k] = acc[6 + k];
            float U[6];   // U = IA S, S = [Sa; 0]
            sym3_mul(IA.A, Sa, U);
            U[3] = IA.B[0] * Sa[0] + IA.B[3] * Sa[1] + IA.B[6] * Sa[2];
            U[4] = IA.B[1] * Sa[0] + IA.B[4] * Sa[1] + IA.B[7] * Sa[2];
            U[5] = IA.B[2] * Sa[0] + IA.B[5] * Sa[1] + IA.B[8] * Sa[2];
            const float D = dot3(Sa, U);
            const float rD = __builtin_amdgcn_rcpf(D);
#if TREX_ABLATE_EXACT_MATH
            const float invD = 1.0f / D;
#else
            const float invD = rD * __builtin_fmaf(-D, rD, 2.0f);   // v_rcp_f32 + one Newton step (no IEEE division expansion)
#endif
            const float u = tau_j - dot3(Sa, pA);
            float Ud[6];
#pragma unroll
            for (int k = 0; k < 6; k++) Ud[k] = U[k] * invD;
            {
              float4 *rec = &H->body[BREC * bl];
              rec[0] = make_float4(Sa[0], Sa[1], Sa[2], invD);
              rec[2] = make_float4(Ud[0], Ud[1], Ud[2], Ud[3]);
              rec[3] = make_float4(Ud[4], Ud[5], __int_as_float(psrc + 256 * depth), u * invD);
            }
            {   // pa = pA + Ia c + U u / D with Ia c = IA c - U (U.c) / D
              float Ic[6];
              sym6_mul(IA, cv, Ic);
              const float coef = (u - dot6(U, cv)) * invD;
#pragma unroll
              for (int k = 0; k < 6; k++) pA[k] += Ic[k] + U[k] * coef;
            }
            sym6_rank1_sub(IA, U, Ud);
            // shift both to the parent's origin (this origin = parent origin + d, d = dpar):
            //   n' = n + d x f,  B' = B + [d]x C,  A' = A + X^T + X', X = [d]x B^T, X' = [d]x B'^T
            {
              cross3_acc(dpar, pA + 3, pA[0], pA[1], pA[2]);
              const int sidx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
#pragma unroll
              for (int i = 0; i < 3; i++) {   // A_ij += X_ji = (d x row i of B)_j, j >= i
                float t[3] = {0.f, 0.f, 0.f};
                if (i == 0) cross3_acc(dpar, IA.B, IA.A[0], IA.A[1], IA.A[2]);
                else if (i == 1) cross3_acc(dpar, IA.B + 3, t[0], IA.A[3], IA.A[4]);
                else cross3_acc(dpar, IA.B + 6, t[0], t[1], IA.A[5]);
              }
#pragma unroll
              for (int j = 0; j < 3; j++) {   // column j of [d]x C = d x (column j of C)
                const float cj[3] = {IA.C[sidx[0][j]], IA.C[sidx[1][j]], IA.C[sidx[2][j]]};
                cross3_acc(dpar, cj, IA.B[j], IA.B[3 + j], IA.B[6 + j]);
              }
#pragma unroll
              for (int j = 0; j < 3; j++) {   // A_ij += X'_ij = (d x row j of B')_i, i <= j
                float t[3] = {0.f, 0.f, 0.f};
                if (j == 0) cross3_acc(dpar, IA.B, IA.A[0], t[1], t[2]);
                else if (j == 1) cross3_acc(dpar, IA.B + 3, IA.A[1], IA.A[3], t[2]);
                else cross3_acc(dpar, IA.B + 6, IA.A[2], IA.A[4], IA.A[5]);
              }
            }
#pragma unroll
            for (int k = 0; k < 6; k++) { o[k] = IA.A[k]; o[15 + k] = IA.C[k]; o[21 + k] = pA[k]; }
#pragma unroll
            for (int k = 0; k < 9; k++) o[6 + k] = IA.B[k];
          }
        }
        WSYNC();
      }
      if ((PAIR || lt < TL) && !is_joint) {   // base and unused lanes: inert records
        float4 *rec = &H->body[BREC * bl];
        rec[0] = make_float4(0.f, 0.f, 0.f, 0.f);
        rec[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        rec[3] = make_float4(0.f, 0.f, __int_as_float(psrc + 256 * (depth < 0 ? 255 : depth)), 0.f);
      }
    }
    RELANE2();
    RETREE2();
    REAXIS2();

    // ---- floating base: a0 = -(IA_0)^-1 pA_0; every lane of a half factors its env's matrix (the same arithmetic on the
    // same words); lane 0 of the half hands the factor and a0 to the env's own wave
    float a0h[6];
    {
      const float *o = H->u.t.aba[0];
      Sym6 I0;
#pragma unroll
      for (int k = 0; k < 6; k++) { I0.A[k] = o[k]; I0.C[k] = o[15 + k]; }
#pragma unroll
      for (int k = 0; k < 9; k++) I0.B[k] = o[6 + k];
      float p0[6];
#pragma unroll
      for (int k = 0; k < 6; k++) p0[k] = -o[21 + k];
      float full[36];
      sym6_full(I0, full);
      Chol6 c;
      chol6_factor(full, c);
      chol6_solve(c, p0, a0h);
      if (bl == 0) {
        float *x = H->xch + 8;
#pragma unroll
        for (int k = 0; k < 15; k++) x[k] = c.l[k];
#pragma unroll
        for (int k = 0; k < 6; k++) { x[15 + k] = c.il[k]; x[21 + k] = a0h[k]; }
      }
    }
    // ---- ABA pass 3 (base to tip): accelerations; qdd = (u - U.a) / D = u/D - (U/D).a from the body record
    float qdd = 0.f;
    {
      const float4 q2 = H->body[BREC * bl + 2], q3 = H->body[BREC * bl + 3];
      const float Ud[6] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y};
      float acc[6];
#pragma unroll
      for (int k = 0; k < 6; k++) acc[k] = a0h[k];
      for (int d = 1; d <= maxdepth; d++) {
        float pa[6];
#pragma unroll
        for (int k = 0; k < 6; k++) pa[k] = wshfl(acc[k], psrc + hoff);
        if (depth == d) {
          float axd[3];
          cross3(pa, dpar, axd);   // parent acceleration seen at this body's origin
#pragma unroll
          for (int k = 0; k < 3; k++) pa[3 + k] += axd[k];
#pragma unroll
          for (int k = 0; k < 6; k++) pa[k] += cv[k];
          qdd = q3.w - dot6(Ud, pa);
#pragma unroll
          for (int k = 0; k < 3; k++) acc[k] = pa[k] + Sa[k] * qdd;
#pragma unroll
          for (int k = 3; k < 6; k++) acc[k] = pa[k];
        }
      }
    }
    // ---- unconstrained joint rates of both envs (each wave forms its own base twist below)
    {
      const float vmax = M->prm[TP_MAX_COORD_VEL];
      const float nqd = is_joint ? fminf(fmaxf(H->st[ST_QD][bl] + qdd * dt, -vmax), vmax) : 0.f;
      if (PAIR || lt < TL) {
        reinterpret_cast<float *>(&H->body[BREC * bl + 1])[3] = nqd;
        H->st[ST_NQD][bl] = nqd;
      }
    }
#undef RELANE2
#undef REAXIS2
#undef RETREE2
      }
      if (PAIR) { SUBSTAMP(17); __syncthreads(); SUBSTAMP(18); } else WSYNC();      // records (U/D, 1/D, u/D, updated rates), base factor and base acceleration are in LDS
      if (act) {
        nc = uni(__float_as_int(W.xch[35]));
        const float *x = W.xch + 8;      // (every lane reads the same words: LDS broadcast)
#pragma unroll
        for (int k = 0; k < 15; k++) I0c.l[k] = uni(x[k]);
#pragma unroll
        for (int k = 0; k < 6; k++) { I0c.il[k] = uni(x[15 + k]); a0[k] = uni(x[21 + k]); }
        const float vmax = M->prm[TP_MAX_COORD_VEL];
        float wxv[3];
        cross3(bw, bv, wxv);
#pragma unroll
        for (int k = 0; k < 3; k++) {
          nw[k] = uni(fminf(fmaxf(bw[k] + a0[k] * dt, -vmax), vmax));
          nv[k] = uni(fminf(fmaxf(bv[k] + (a0[3 + k] + wxv[k]) * dt, -vmax), vmax));
        }
      }
    }
    if (act) {
    WSYNC();
    STAMP(4);
    RELANE();

    // ================================================================ constraint rows
    // Rows -> lanes: motor row j on lane j (1..25, joint j's limit row riding on it); contact row k = 3 s + a
    // of point SLOT s (a: normal z, friction x, friction y) on lane 26 + k for k < 38 and on lane 0 for k = 38.
    // The nc points of this substep take the LAST nc slots (slot = MAXC - nc + contact number), so that a
    // sweep is one jump into the unrolled chain of point blocks and no per-point branch.
    const int s0 = uni(MAXC - nc);                               // first slot in use
    bool mlane, mrow;                                            // motor-row lane; live motor row
    int cdir;
#define REROW() do { mlane = lt >= 1 && lt <= NJMAX; mrow = mlane && lt < nb; cdir = mlane ? 0 : (lt == 0 ? 2 : (lt - CLANE0) % 3); } while (0)
    REROW();
    // Every row walks its chain to the base ONCE: the generalised force J^T is pushed through the ABA
    // factorisation (u_a = -a_a . n, p += (U/D)_a u_a), which yields the row's column of A (u), the same
    // divided by D (zc), the base wrench r0 and z0 = I0^-1 r0 (two triangular solves) - and, for a contact row, the plain Jacobian
    // entries for J.v on the way. A motor row is the unit force on its own joint: u = 1 at its own level.
    // ca0: the chain, 5 bits per level (depth-1 ancestor in the low bits, 0 = none): two rows share the joints
    // of their common prefix, so "same joint at level d" is "the lowest differing bit lies above field d"
    unsigned ca0 = 0u;
    static_assert(5 * MAXD <= 32 && NJMAX < 32, "chain packs into one register");
    float u0[MAXD], zc0[MAXD], r00[6], z00[6], inv0 = 0.f, y = 0.f;
    float mhi = 0.f, ldir = 0.f, lr = 0.f;
    {
      const int cslot = mlane ? 0 : (lt == 0 ? MAXC - 1 : (lt - CLANE0) / 3);
      const bool crow = !mlane && cslot >= s0;                   // live contact row
      float cx[3] = {0.f, 0.f, 0.f}, cdist = 0.f;
      int cb = mrow ? lt : 0;
      if (crow) {
        const float *cp = W.cpt[cslot - s0];
        cb = __float_as_int(cp[0]); cx[0] = cp[1]; cx[1] = cp[2]; cx[2] = cp[3]; cdist = cp[4];
      }
      const float dir[3] = {cdir == 1 ? 1.f : 0.f, cdir == 2 ? 1.f : 0.f, cdir == 0 ? 1.f : 0.f};
      float p[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, po[3];
      {
        const float4 q1 = W.body[BREC * cb + 1];
        po[0] = q1.x; po[1] = q1.y; po[2] = q1.z;
      }
      if (crow) {
        const float xrel[3] = {cx[0] - po[0], cx[1] - po[1], cx[2] - po[2]};
        float xd[3];
        cross3(xrel, dir, xd);
#pragma unroll
        for (int k = 0; k < 3; k++) { p[k] = -xd[k]; p[3 + k] = -dir[k]; }
      }
      const bool live = mrow || crow;
      float diag = 0.f, jv = 0.f;
      int cur = cb;
#pragma unroll
      for (int d = MAXD; d >= 1; d--) {
        u0[d - 1] = 0.f; zc0[d - 1] = 0.f;
        if (d <= maxdepth) {
          const float4 q0 = W.body[BREC * cur], q1 = W.body[BREC * cur + 1], q2 = W.body[BREC * cur + 2], q3 = W.body[BREC * cur + 3];
          const int lk = __float_as_int(q3.z);
          if (live && (lk >> 8) == d) {
            const float aa[3] = {q0.x, q0.y, q0.z}, ra[3] = {q1.x, q1.y, q1.z};
            const float Uda[6] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y};
            float dd[3], dxf[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { dd[k] = po[k] - ra[k]; po[k] = ra[k]; }
            cross3(dd, p + 3, dxf);
#pragma unroll
            for (int k = 0; k < 3; k++) p[k] += dxf[k];
            const float ua = (mlane && cur == lt) ? 1.f : -dot3(aa, p);
            const float zc = ua * q0.w;
            diag += ua * zc;
#pragma unroll
            for (int k = 0; k < 6; k++) p[k] += Uda[k] * ua;
            ca0 |= (unsigned)cur << (5 * (d - 1)); u0[d - 1] = ua; zc0[d - 1] = zc;
            if (crow) {   // plain Jacobian entry of joint `cur`: a . ((x - r_a) x dir)
              const float xr[3] = {cx[0] - ra[0], cx[1] - ra[1], cx[2] - ra[2]};
              float xd[3];
              cross3(xr, dir, xd);
              jv += dot3(aa, xd) * q1.w;
            }
            cur = lk & 255;
          }
        }
      }
      {
        float dxf[3];
        cross3(po, p + 3, dxf);   // on to the base origin O
#pragma unroll
        for (int k = 0; k < 3; k++) p[k] += dxf[k];
      }
#pragma unroll
      for (int k = 0; k < 6; k++) r00[k] = live ? -p[k] : 0.f;
      chol6_solve(I0c, r00, z00);
      diag += dot6(r00, z00);
      inv0 = live ? 1.0f / diag : 0.f;
      const TrexDeviceModel *Mi = Mo();
      if (crow) {
        const float cerp = Mi->prm[TP_CONTACT_ERP];
        float tv = 0.f;
        if (cdir == 0) tv = (cdist > 0.f) ? -cdist * inv_dt : -cdist * cerp * inv_dt;
        float wxx[3];
        cross3(nw, cx, wxx);
        const float pv[3] = {nv[0] + wxx[0], nv[1] + wxx[1], nv[2] + wxx[2]};
        y = (tv - (jv + dot3(dir, pv))) * inv0;
      }
      if (mrow) {
        const float erp = Mi->prm[TP_ERP], kp = Mi->prm[TP_MOTOR_KP], kd = Mi->prm[TP_MOTOR_KD];
        const float max_imp = Mi->motor_max_impulse;
        const float q = W.st[ST_Q][bl], nqd = W.st[ST_NQD][bl], target = W.st[ST_TARGET][bl];
        W.st[ST_MDG][bl] = diag;
        // btMultiBodyJointMotor velocity target: kp*(target-q)/dt + qd + kd*(0-qd), minus current qd
        y = (kp * (target - q) * inv_dt + kd * (0.f - nqd)) * inv0;
        mhi = motors_on ? max_imp : 0.f;
        const float q_lo = Mi->lower[bl], q_hi = Mi->upper[bl];
        float pen = 0.f;
        if (q - q_lo <= 0.f) { pen = q - q_lo; ldir = 1.f; }
        else if (q_hi - q <= 0.f) { pen = q_hi - q; ldir = -1.f; }
        const float lim_rhs = (-pen * erp * inv_dt - ldir * nqd) * inv0;
        lr = lim_rhs - ldir * y;
      }
    }
    const unsigned lim_mask = (unsigned)__ballot(ldir != 0.f);
    WSYNC();   // the body records are dead: the z0 stash may overwrite them; so are the inertia slots
    {
      float *zs = reinterpret_cast<float *>(W.body);
#pragma unroll
      for (int k = 0; k < 6; k++) zs[64 * k + lt] = z00[k];
      // column side of this lane's row, for every other lane to read (one address per column: LDS broadcast)
      float4 *dc = W.u.desc[lt];
      dc[0] = make_float4(__uint_as_float(ca0), zc0[0], zc0[1], zc0[2]);
      dc[1] = make_float4(zc0[3], zc0[4], zc0[5], z00[0]);
      dc[2] = make_float4(z00[1], z00[2], z00[3], z00[4]);
      dc[3] = make_float4(z00[5], 0.f, 0.f, 0.f);
      if (PAIR) {
        // the sweeps' inputs of this lane's row wait in LDS while the B entries are built (the 1024 bytes of the body-record area
        // that the z0 stash leaves): the pair form carries the LDS base of its env in a register, the B build - 64 entries, a
        // column descriptor in flight, the row's own descriptor - is the phase with the fewest to spare, and what did not fit
        // went to SCRATCH (12 MB of HBM traffic per launch of 4096 envs)
        zs[64 * 6 + lt] = y; zs[64 * 7 + lt] = mhi; zs[64 * 8 + lt] = lr; zs[64 * 9 + lt] = ldir;
        asm volatile("" : "=v"(y), "=v"(mhi), "=v"(lr), "=v"(ldir));     // (dead until they are read back)
      }
    }
    WSYNC();
    STAMP(5);
    RELANE();

    // ---- B entries of this lane's row against every column: B_sr = -(J_s M^-1 J_r^T) / diag_s with
    //     J_s M^-1 J_r^T = r0_s . z0_r + sum_d [ca_s[d] == ca_r[d]] u_s[d] zc_r[d].
    // The column's descriptor (chain, zc, z0: 13 words) is read from LDS at ONE address by all lanes (broadcast,
    // no VALU) - the v_readlane form of it cost 13 VALU per column. Unused chain levels hold u = zc = 0, so a
    // "match" of two empty levels adds nothing.
    auto krow_lane = [](int k) { return k < 3 * MAXC - 1 ? CLANE0 + k : 0; };   // lane of contact row k
    auto column = [&](int L, const float *m, float4 d0, float4 d1, float4 d2, float4 d3) {
      float a0_ = r00[0] * d1.w;
      a0_ = __builtin_fmaf(r00[1], d2.x, a0_); a0_ = __builtin_fmaf(r00[2], d2.y, a0_);
      a0_ = __builtin_fmaf(r00[3], d2.z, a0_); a0_ = __builtin_fmaf(r00[4], d2.w, a0_);
      a0_ = __builtin_fmaf(r00[5], d3.x, a0_);
      a0_ = __builtin_fmaf(m[0], d0.y, a0_); a0_ = __builtin_fmaf(m[1], d0.z, a0_); a0_ = __builtin_fmaf(m[2], d0.w, a0_);
      a0_ = __builtin_fmaf(m[3], d1.x, a0_); a0_ = __builtin_fmaf(m[4], d1.y, a0_); a0_ = __builtin_fmaf(m[5], d1.z, a0_);
      return -inv0 * a0_;
    };
    auto chain_mask = [&](float4 d0, float *m) {   // u of this row on the levels it shares with the column's chain
      // lowest differing bit of the two packed chains (30 bits); bit 30 is set so that identical chains give 30
      // without a special case: every level then counts as shared
      const unsigned lowdiff = (unsigned)__builtin_ctz((ca0 ^ __float_as_uint(d0.x)) | 0x40000000u);
#pragma unroll
      for (int d = 0; d < MAXD; d++) m[d] = (lowdiff >= 5u * (d + 1)) ? u0[d] : 0.f;
    };
    // (one column's 4 reads in flight while the previous column is evaluated - pinned by sched_barrier: left to
    // itself the scheduler hoists the reads of ALL columns, 400 registers)
    float Bm[NJMAX];
    {
      float4 n0 = W.u.desc[1][0], n1 = W.u.desc[1][1], n2 = W.u.desc[1][2], n3 = W.u.desc[1][3];
#pragma unroll
      for (int j = 1; j <= NJMAX; j++) {
        const float4 d0 = n0, d1 = n1, d2 = n2, d3 = n3;
        if (j < NJMAX) { n0 = W.u.desc[j + 1][0]; n1 = W.u.desc[j + 1][1]; n2 = W.u.desc[j + 1][2]; n3 = W.u.desc[j + 1][3]; }
        float m[MAXD];
        chain_mask(d0, m);
        Bm[j - 1] = column(j, m, d0, d1, d2, d3);
        asm volatile("" : "+v"(Bm[j - 1]));   // evaluated HERE (not sunk to its first use in the sweeps)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float Bc[3 * MAXC];
    // Slots below s0 are never read. All 39 entries are first "defined" by an empty asm (no instruction): the
    // live slots are then overwritten, the dead ones cost neither zeros to materialise nor a value carried
    // around the substep loop.
#pragma unroll
    for (int k = 0; k < 3 * MAXC; k++) asm volatile("" : "=v"(Bc[k]));
#pragma unroll
    for (int s = 0; s < MAXC; s++) {
      if (s >= s0) {
        float m[MAXD];
#pragma unroll
        for (int a = 0; a < 3; a++) {
          const float4 *dc = W.u.desc[krow_lane(3 * s + a)];
          const float4 d0 = dc[0], d1 = dc[1], d2 = dc[2], d3 = dc[3];
          if (a == 0) chain_mask(d0, m);   // the three rows of a point share its chain
          Bc[3 * s + a] = column(0, m, d0, d1, d2, d3);
          asm volatile("" : "+v"(Bc[3 * s + a]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // (the descriptors stay where they are. Until round 3 the limit rows read their B column from LDS - 25 columns staged
    // over the descriptors, 6400 B, whenever a joint sat on a stop; they now take it from the lane's own Bm register)
    STAMP(6);
    RELANE();

    // ---- projected Gauss-Seidel in DELASSUS (residual) form. Each row s lives on ONE lane and keeps
    //     y_s = rhs_s - (J_s dv) / diag_s            (its unclamped Gauss-Seidel increment)
    // so a row visit is  nl = clamp(lam_r + y_r); d = nl - lam_r; lam_r = nl;  y_s += B_sr d  for all s
    // (B_rr = -1) - the same iteration as Bullet's dv form (and the oracle's), but the row's impulse change
    // reaches the other rows as ONE v_readlane (SGPR broadcast) + ONE fma per lane. y (not z = lam + y) is
    // what is accumulated: it is small where lam is large, and the rounding of a row's own update stays in y.
    // Row order (the oracle's): limit rows (ascending joint), motor rows, then per point normal, friction x, y.
    if (PAIR) {
      const float *zs = reinterpret_cast<const float *>(W.body);
      y = zs[64 * 6 + lt]; mhi = zs[64 * 7 + lt]; lr = zs[64 * 8 + lt]; ldir = zs[64 * 9 + lt];
    }
    float lam = 0.f, lam_c = 0.f, lim_lam = 0.f;
    int dvec = 0;   // lane j: the impulse change of motor row j in the current sweep (the other lanes stay 0)
    {
#define TREX_ROW(LANE, BCOL, LO, HI)                                                                   \
  {                                                                                                    \
    const float nl_ = __builtin_amdgcn_fmed3f(lam + y, (LO), (HI));                                    \
    const float d_ = nl_ - lam;                                                                        \
    const float sd_ = rl(d_, (LANE));                                                                  \
    if (vs == (LANE)) lam = nl_;                                                                       \
    y = __builtin_fmaf((BCOL), sd_, y);                                                                \
  }
// A point with no normal impulse before its visit (lam_n = 0) and none after it (lam_n + y_n <= 0) changes
// nothing: its normal row gives d = 0, its friction rows are clamped to 0 and hold 0 already (they were visited
// after the normal row lost its impulse). Most candidate points are like that - inside the 2 cm margin, not
// pressing. ONE vector test over all normal-row lanes therefore precedes the point blocks: a dead
// point costs a scalar bit test, and the test is repeated after every point that was processed (it changed y).
// Bitwise the same result as visiting every row: the skipped updates would add B * 0.
// A live point is one hand-placed block of 24 slots. Like the motor rows, its rows work with bounds SHIFTED by the
// impulse: normal d = max(y, -lam) (one instruction; lam + d = 0 exactly when the contact lets go), friction
// d = med3(y, -hi - lam, hi - lam) with hi = mu * the new normal impulse, the two shifted bounds formed once for
// both friction lanes; the three changes are captured by v_writelane - which doubles as the wait state between a
// v_med3 and the v_readlane of its result - into `dvc`, which is committed ONCE per sweep after the last point (a
// row is visited once per sweep, and a point block reads lam on its own three lanes only: 27 slots with a
// v_mov 0 / v_add per point). For the same reason the liveness of the points still to come needs nothing but
// the new y: lam != 0 is settled per sweep (`lamnz`), and y > -lam is ONE compare against `thr` = -lam on the
// normal lanes of the live slots, +inf elsewhere (formed per sweep), so vcc needs no masking.
#define TREX_POINT_TEXT(P)                                                                             \
               "s_bitcmp1_b64 %[al], %[ln" #P "]\n\t"                                                  \
               "s_cbranch_scc0 " #P "f\n\t"                                                            \
               /* normal row, bounds shifted by the impulse: d = max(y, -lam); the new impulse lam + d */ \
               "v_max_f32_e64 %[d], %[y], -%[lam]\n\t"                                                 \
               "v_add_f32_e32 %[t], %[lam], %[d]\n\t"                                                  \
               "v_readlane_b32 %[sd], %[d], %[ln" #P "]\n\t"                                           \
               "v_readlane_b32 %[snl], %[t], %[ln" #P "]\n\t"                                          \
               "s_nop 0\n\t"                                                                           \
               "v_fmac_f32_e32 %[y], %[sd], %[b0" #P "]\n\t"                                           \
               "v_mul_f32_e32 %[hi], %[snl], %[mu]\n\t"                                                \
               /* friction bounds -hi - lam, hi - lam for both friction lanes at once */               \
               "v_sub_f32_e64 %[t], -%[hi], %[lam]\n\t"                                                \
               "v_sub_f32_e32 %[hi], %[hi], %[lam]\n\t"                                                \
               /* friction x */                                                                        \
               "v_med3_f32 %[d], %[y], %[t], %[hi]\n\t"                                                \
               "v_writelane_b32 %[dv], %[sd], %[ln" #P "]\n\t"                                         \
               "v_readlane_b32 %[snl], %[d], %[lx" #P "]\n\t"                                          \
               "s_nop 1\n\t"                                                                           \
               "v_fmac_f32_e32 %[y], %[snl], %[b1" #P "]\n\t"                                          \
               /* friction y */                                                                        \
               "v_med3_f32 %[d], %[y], %[t], %[hi]\n\t"                                                \
               "v_writelane_b32 %[dv], %[snl], %[lx" #P "]\n\t"                                        \
               "v_readlane_b32 %[sd], %[d], %[ly" #P "]\n\t"                                           \
               "s_nop 1\n\t"                                                                           \
               "v_fmac_f32_e32 %[y], %[sd], %[b2" #P "]\n\t"                                           \
               "v_writelane_b32 %[dv], %[sd], %[ly" #P "]\n\t"                                         \
               /* which of the points still to come can change anything now */                         \
               "v_cmp_gt_f32_e32 vcc, %[y], %[thr]\n\t"                   /* lam + y > 0 */           \
               "s_or_b64 %[al], vcc, %[lnz]\n\t"                                                       \
               #P ":\n\t"
#define TREX_POINT_OUTS [y] "+v"(y), [dv] "+v"(dvc), [al] "+s"(alive), [t] "=&v"(pt_), [d] "=&v"(pd_), [hi] "=&v"(ph_), \
                        [snl] "=&s"(psn_), [sd] "=&s"(psd_)
#define TREX_POINT_INS [lam] "v"(lam), [mu] "v"(mu_v), [thr] "v"(thr), [lnz] "s"(lamnz)
#define TREX_POINT_OPS(P, S) [b0##P] "v"(Bc[3 * (S)]), [b1##P] "v"(Bc[3 * (S) + 1]), [b2##P] "v"(Bc[3 * (S) + 2]),     \
                             [ln##P] "n"(KROW_LANE(3 * (S))), [lx##P] "n"(KROW_LANE(3 * (S) + 1)), [ly##P] "n"(KROW_LANE(3 * (S) + 2))
// (several point slots per asm statement: the compiler closes every statement with an s_nop of its own)
#define TREX_POINTS3(S)                                                                                \
  asm volatile(TREX_POINT_TEXT(0) TREX_POINT_TEXT(1) TREX_POINT_TEXT(2)                                \
               : TREX_POINT_OUTS                                                                       \
               : TREX_POINT_INS, TREX_POINT_OPS(0, S), TREX_POINT_OPS(1, (S) + 1), TREX_POINT_OPS(2, (S) + 2) \
               : "vcc", "scc");
#define TREX_POINTS1(S)                                                                                \
  asm volatile(TREX_POINT_TEXT(0) : TREX_POINT_OUTS : TREX_POINT_INS, TREX_POINT_OPS(0, S) : "vcc", "scc");
      // lanes that hold the normal row of a live point slot
      float mu_v = mu;        // the friction coefficient as a vector operand of the point blocks
      if (PAIR) asm volatile("" : "+v"(mu_v));     // (made HERE: hoisted out of the substep loop it was a register carried - and spilled - through the whole kernel)
      const bool is_nrm = lt >= CLANE0 && (lt - CLANE0) % 3 == 0 && (lt - CLANE0) / 3 >= s0;
      const unsigned long long nrm_mask = __ballot(is_nrm);
#if TREX_PRIO_MODE == 1
      set_sweep_priority(nc);
#endif
#pragma unroll 1
      for (int it = 0; it < iters; it++) {
        // limit rows: the row of joint j rides on motor lane j, whose y gives dv_j / diag = rhs - y
        for (unsigned m = lim_mask; m != 0u; m &= m - 1u) {
          // the lane id, opaque: `vs == j` is then one v_cmp here, not a mask hoisted out of the loops and spilled.
          // (A lane mask built on the scalar unit - s_lshl_b64 + v_cndmask - measured SLOWER than v_cmp +
          // v_cndmask: 17.6 against 15.2 cycles per row and SIMD at 4 waves per SIMD, profiles/tools/row_bench.hip.)
          int vs = lt;
          asm volatile("" : "+v"(vs));
          const int j = __ffs(m) - 1;
          const float nl = fmaxf(lim_lam + (lr + ldir * y), 0.f);
          const float dl = (nl - lim_lam) * ldir;
          if (vs == j) lim_lam = nl;
          const float sd = rl(dl, j);
          // column j of B sits in register Bm[j - 1] of every lane and j is wave-uniform: a computed jump into a table of
          // (v_fmac, s_branch) pairs, 8 bytes each - five scalar instructions and ONE v_fmac per limit-row visit. (As a C
          // switch the compiler lowered this to ~30 flag tests per visit; until round 3 the column came from LDS, where
          // all 25 were staged over the row descriptors - 6400 B - whenever a joint sat on a stop.) s_getpc yields the
          // address of the instruction that follows it; the four instructions up to and including s_setpc are 16
          // bytes, so entry j (1-based) sits at pc + 16 + 8 (j - 1) = pc + 8 j + 8.
          {
            int jt_;
            static_assert(NJMAX == 25, "the jump table below has 25 entries");
            asm volatile("s_getpc_b64 vcc\n\t"
                         "s_lshl3_add_u32 %[t], %[j], 8\n\t"
                         "s_add_u32 vcc_lo, vcc_lo, %[t]\n\t"
                         "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
                         "s_setpc_b64 vcc\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b1]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b2]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b3]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b4]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b5]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b6]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b7]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b8]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b9]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b10]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b11]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b12]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b13]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b14]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b15]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b16]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b17]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b18]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b19]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b20]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b21]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b22]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b23]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b24]\n\t"
                         "s_branch 9f\n\t"
                         "v_fmac_f32_e32 %[y], %[sd], %[b25]\n\t"
                         "9:\n\t"
                         : [y] "+v"(y), [t] "=&s"(jt_)
                         : [sd] "s"(sd), [j] "s"(j), [b1] "v"(Bm[0]), [b2] "v"(Bm[1]), [b3] "v"(Bm[2]), [b4] "v"(Bm[3]), [b5] "v"(Bm[4]), [b6] "v"(Bm[5]), [b7] "v"(Bm[6]), [b8] "v"(Bm[7]), [b9] "v"(Bm[8]), [b10] "v"(Bm[9]), [b11] "v"(Bm[10]), [b12] "v"(Bm[11]), [b13] "v"(Bm[12]), [b14] "v"(Bm[13]), [b15] "v"(Bm[14]), [b16] "v"(Bm[15]), [b17] "v"(Bm[16]), [b18] "v"(Bm[17]), [b19] "v"(Bm[18]), [b20] "v"(Bm[19]), [b21] "v"(Bm[20]), [b22] "v"(Bm[21]), [b23] "v"(Bm[22]), [b24] "v"(Bm[23]), [b25] "v"(Bm[24])
                         : "vcc", "scc");
          }
        }
        // motor rows (joints beyond nb are null rows: y = 0, bounds 0), hand-placed: 5 issue slots per row (the compiler's
        // form of TREX_ROW takes 8). With the bounds SHIFTED by the impulse, d_j = clamp(lam_j + y_j) - lam_j =
        // med3(y_j, lo - lam_j, hi - lam_j) is one instruction; a motor row is visited once per sweep, so the shifted
        // bounds are formed for all lanes at once before the block. d_j, which sits in an SGPR for the broadcast anyway,
        // is captured into lane j of `dvec` with v_writelane and the 25 impulses are committed after the block,
        // lam += dvec - compensated, the rounding error kept in lam_c: the sum of the d's that the other rows have
        // seen and the stored impulse must not drift apart over 60 sweeps (an unsaturated row adds y itself, not
        // fl(lam + y) - lam). The v_writelane of row j-1 is the wait state between v_med3 and the v_readlane of its
        // result; s_nop 1 covers the two wait states between v_readlane and the v_fmac that reads the SGPR.
        // (Not taken: accumulating z = lam + y instead of y saves the add too but costs 30x the one-step error of an
        // airborne env - z is as large as a saturated impulse, y is small; a speculative UNCLAMPED block - d_j = y_j,
        // committed only if no bound was crossed - runs twice too often: under random actions 7 percent of the motor
        // rows sit at 3e5 N m.)
        {
          static_assert(NJMAX == 25, "the blocks below are written out for 25 motor rows");
          int sa_, sb_;
          float d_;
          const float blo = (-mhi - lam) + lam_c, bhi = (mhi - lam) + lam_c;
          asm volatile("v_med3_f32 %2, %0, %5, %6\n\t"
                       "s_nop 0\n\t"
                       "v_readlane_b32 %3, %2, 1\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %3, %7\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %3, 1\n\t"
                       "v_readlane_b32 %4, %2, 2\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %4, %8\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %4, 2\n\t"
                       "v_readlane_b32 %3, %2, 3\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %3, %9\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %3, 3\n\t"
                       "v_readlane_b32 %4, %2, 4\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %4, %10\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %4, 4\n\t"
                       "v_readlane_b32 %3, %2, 5\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %3, %11\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %3, 5\n\t"
                       "v_readlane_b32 %4, %2, 6\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %4, %12\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %4, 6\n\t"
                       "v_readlane_b32 %3, %2, 7\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %3, %13\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %3, 7\n\t"
                       "v_readlane_b32 %4, %2, 8\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %4, %14\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %4, 8\n\t"
                       "v_readlane_b32 %3, %2, 9\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %3, %15\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %3, 9\n\t"
                       "v_readlane_b32 %4, %2, 10\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %4, %16\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %4, 10\n\t"
                       "v_readlane_b32 %3, %2, 11\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %3, %17\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %3, 11\n\t"
                       "v_readlane_b32 %4, %2, 12\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %4, %18\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %4, 12\n\t"
                       "v_readlane_b32 %3, %2, 13\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %3, %19\n\t"
                       : "+v"(y), "+v"(dvec), "=&v"(d_), "=&s"(sa_), "=&s"(sb_)
                       : "v"(blo), "v"(bhi), "v"(Bm[0]), "v"(Bm[1]), "v"(Bm[2]), "v"(Bm[3]), "v"(Bm[4]), "v"(Bm[5]), "v"(Bm[6]), "v"(Bm[7]), "v"(Bm[8]), "v"(Bm[9]), "v"(Bm[10]), "v"(Bm[11]), "v"(Bm[12]));
          asm volatile("v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %3, 13\n\t"
                       "v_readlane_b32 %4, %2, 14\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %4, %7\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %4, 14\n\t"
                       "v_readlane_b32 %3, %2, 15\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %3, %8\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %3, 15\n\t"
                       "v_readlane_b32 %4, %2, 16\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %4, %9\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %4, 16\n\t"
                       "v_readlane_b32 %3, %2, 17\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %3, %10\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %3, 17\n\t"
                       "v_readlane_b32 %4, %2, 18\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %4, %11\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %4, 18\n\t"
                       "v_readlane_b32 %3, %2, 19\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %3, %12\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %3, 19\n\t"
                       "v_readlane_b32 %4, %2, 20\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %4, %13\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %4, 20\n\t"
                       "v_readlane_b32 %3, %2, 21\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %3, %14\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %3, 21\n\t"
                       "v_readlane_b32 %4, %2, 22\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %4, %15\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %4, 22\n\t"
                       "v_readlane_b32 %3, %2, 23\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %3, %16\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %3, 23\n\t"
                       "v_readlane_b32 %4, %2, 24\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %4, %17\n\t"
                       "v_med3_f32 %2, %0, %5, %6\n\t"
                       "v_writelane_b32 %1, %4, 24\n\t"
                       "v_readlane_b32 %3, %2, 25\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_e32 %0, %3, %18\n\t"
                       "v_writelane_b32 %1, %3, 25\n\t"
                       : "+v"(y), "+v"(dvec), "=&v"(d_), "+s"(sa_), "=&s"(sb_)
                       : "v"(blo), "v"(bhi), "v"(Bm[13]), "v"(Bm[14]), "v"(Bm[15]), "v"(Bm[16]), "v"(Bm[17]), "v"(Bm[18]), "v"(Bm[19]), "v"(Bm[20]), "v"(Bm[21]), "v"(Bm[22]), "v"(Bm[23]), "v"(Bm[24]));
          {   // lam += dvec, compensated (Kahan): lam - lam_c is the sum of the d's the other rows have seen
            const float y_ = __int_as_float(dvec) - lam_c, t_ = lam + y_;
            lam_c = (t_ - lam) - y_;
            lam = t_;
          }
        }
        // the live point slots, in order (dead slots have no bit in `alive`)
        unsigned long long alive = 0ull, lamnz = 0ull;
        if (nrm_mask != 0ull) {   // (an airborne env has no point rows at all)
          lamnz = nrm_mask & __ballot(lam != 0.f);
          alive = lamnz | (nrm_mask & __ballot(y > -lam));
        }
#if TREX_STAMPS   // per wave: point slots alive / holding an impulse at the start of a sweep, summed over the launch
        stamp_alive += __popcll(alive); stamp_lamnz += __popcll(lamnz);
#endif
        if (alive != 0ull) {
          float pt_, pd_, ph_;
          int psn_, psd_;
          const float thr = is_nrm ? -lam : __builtin_inff();
          float dvc = 0.f;      // the impulse changes of this sweep's point rows, by lane
          // (a dead point costs its bit test and a TAKEN branch, ~16 cycles, and of the slots of an env on 12 points one
          // or two are alive in a sweep: a group of slots without a live one is passed in one test)
          constexpr unsigned long long NB = 1ull << CLANE0;      // normal row of slot S: lane CLANE0 + 3 S
          constexpr unsigned long long G0 = NB * 0111ull, G3 = G0 << 9, G6 = NB << 18, G7 = G0 << 21, G10 = G0 << 30;
          if ((alive & (G0 | G3 | G6)) != 0ull) {
            if ((alive & G0) != 0ull) { TREX_POINTS3(0) }
            if ((alive & G3) != 0ull) { TREX_POINTS3(3) }
            if ((alive & G6) != 0ull) { TREX_POINTS1(6) }
          }
          if ((alive & G7) != 0ull) { TREX_POINTS3(7) }
          if ((alive & G10) != 0ull) { TREX_POINTS3(10) }
          lam += dvc;
        }
      }
#undef TREX_ROW
#undef TREX_POINT_TEXT
#undef TREX_POINTS3
#undef TREX_POINTS1
      lam -= lam_c;
    }
#if TREX_PRIO_MODE == 1
    prio_nc = nc;
    set_tree_priority(sub);
#endif
    STAMP(7);
    RELANE();
    REROW();

    // ---- results. Joint lanes: dv_j / diag_j = -sum_r B_jr lam_r, summed afresh from the final impulses
    // (rhs_j - y_j holds the same number, but as a difference of large terms when the motor is saturated).
    // Base twist change = sum_r lam_r z0_r.
    const float lt0 = lam + (mlane ? ldir * lim_lam : 0.f);   // motor + limit impulse of the joint
    float dvj = 0.f;
#pragma unroll
    for (int j = 1; j <= NJMAX; j++) dvj -= Bm[j - 1] * rl(lt0, j);
#pragma unroll
    for (int s = 0; s < MAXC; s++) {
      if (s >= s0) {
#pragma unroll
        for (int a = 0; a < 3; a++) dvj -= Bc[3 * s + a] * rl(lam, krow_lane(3 * s + a));
      }
    }
    float dvb[6];
    {
      const float *zs = reinterpret_cast<const float *>(W.body);
#pragma unroll
      for (int k = 0; k < 6; k++) dvb[k] = uni(wsum(lt0 * zs[64 * k + lt]));
    }
    const float nimp = uni(wsum((!mlane && cdir == 0) ? lam : 0.f));
    const float mdg = W.st[ST_MDG][bl];
    const float dv = mrow ? dvj * mdg : 0.f;

    if (DEBUG && args.debug && wg == 0) {
      float *D = args.debug;
      if (lt < TL) D[96 + lt] = dv;
      if (lt == 0) {
        for (int k = 0; k < 6; k++) D[96 + nb + k] = dvb[k];
        D[128] = (float)nc; D[129] = (float)lim_mask;
      }
      // joint block of M^-1 recovered from the staged columns, contact points and their impulses
      if (lt < TL) {
#pragma unroll
        for (int j = 1; j <= NJMAX; j++) D[160 + 32 * (j - 1) + lt] = mrow ? -Bm[j - 1] * mdg : 0.f;
      }
      for (int c = 0; c < nc; c++) {
        const float l0 = rl(lam, krow_lane(3 * (s0 + c))), l1 = rl(lam, krow_lane(3 * (s0 + c) + 1)), l2 = rl(lam, krow_lane(3 * (s0 + c) + 2));
        if (lt == 0) {
          float *C = D + 960 + c * 16;
          C[0] = (float)__float_as_int(W.cpt[c][0]); C[1] = W.cpt[c][1]; C[2] = W.cpt[c][2]; C[3] = W.cpt[c][3]; C[4] = W.cpt[c][4];
          C[11] = l0; C[12] = l1; C[13] = l2;
        }
      }
    }
    WSYNC();

    // ---- commit velocities, integrate positions
    if (lt < TL) {
      const float qd = is_joint ? W.st[ST_NQD][lt] + dv : 0.f;
      W.st[ST_QD][lt] = qd;
      W.st[ST_TAU][lt] = (mrow && motors_on) ? lam * inv_dt : 0.f;
      W.st[ST_Q][lt] += qd * dt;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { bw[k] = uni(nw[k] + dvb[k]); bv[k] = uni(nv[k] + dvb[3 + k]); }
#pragma unroll
    for (int k = 0; k < 3; k++) pos[k] = uni(pos[k] + bv[k] * dt);
    {
      // exponential map of w dt: dq = (w sin(h)/|w|, cos(h)), h = |w| dt / 2. Below h = 1/4 (|w| < 250 rad/s at
      // dt = 2 ms: every physical state) the series in h^2 are exact to f32 rounding and need neither |w| nor a
      // division: sin(h)/|w| = dt/2 (1 - h^2/6 + h^4/120 - h^6/5040 + h^8/362880)
      const float w2 = dot3(bw, bw), h2 = dt2_quarter * w2;
      float dq[4];
      if (h2 < 0.0625f && !TREX_ABLATE_EXACT_QUAT) {
        const float sc = 1.f + h2 * (-1.f / 6.f + h2 * (1.f / 120.f + h2 * (-1.f / 5040.f + h2 * (1.f / 362880.f))));
        const float sh = dt_half * sc;
        dq[0] = bw[0] * sh; dq[1] = bw[1] * sh; dq[2] = bw[2] * sh;
        dq[3] = 1.f + h2 * (-0.5f + h2 * (1.f / 24.f + h2 * (-1.f / 720.f + h2 * (1.f / 40320.f))));
      } else {
        const float wn = sqrtf(w2), sh = wn > 1e-12f ? sinf(0.5f * wn * dt) / wn : dt_half;
        dq[0] = bw[0] * sh; dq[1] = bw[1] * sh; dq[2] = bw[2] * sh; dq[3] = cosf(0.5f * wn * dt);
      }
      float o[4];
      o[3] = dq[3] * quat[3] - dq[0] * quat[0] - dq[1] * quat[1] - dq[2] * quat[2];
      o[0] = dq[3] * quat[0] + dq[0] * quat[3] + dq[1] * quat[2] - dq[2] * quat[1];
      o[1] = dq[3] * quat[1] - dq[0] * quat[2] + dq[1] * quat[3] + dq[2] * quat[0];
      o[2] = dq[3] * quat[2] + dq[0] * quat[1] - dq[1] * quat[0] + dq[2] * quat[3];
      const float qn = 1.0f / sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
#pragma unroll
      for (int k = 0; k < 4; k++) quat[k] = uni(o[k] * qn);
    }
    stat_nc = nc;
    stat_imp = nimp;
    WSYNC();
    STAMP(8);
    }   // act (rows, sweeps, integration)
#undef RELANE
#undef RETREE
#undef REAXIS
#undef REROW
  }

#if TREX_STAMPS
  if (args.debug && (threadIdx.x & 63) == 0) {
    args.debug[4096 + 14 * args.n_envs + wg] = (float)stamp_alive;
    args.debug[4096 + 15 * args.n_envs + wg] = (float)stamp_lamnz;
  }
#endif
  // ---- end of the env-step: outputs
  if (RESET || !time_up) finish_step();
  if (env_bad && !time_up) to_start_pose();     // (a time-limit reset already left a sound state)
  const int steps_out = (RESET || args.arr.max_episode_steps > 0) ? ((RESET || time_up || env_bad) ? 0 : age) : steps_in;
  {
    const int lt = lane_id();
    const int bl = lt & (TL - 1);
    const bool is_joint = lt >= 1 && lt < nb;
    const size_t so = MULTI ? (size_t)ls * (size_t)args.step_rows : 0;      // this step's row block
    if (args.obs && is_joint) {
      const float q = W.st[ST_Q][bl], qd = W.st[ST_QD][bl], mtau = W.st[ST_TAU][bl];
      float *o = args.obs + so + (size_t)env * args.obs_stride;
      const int obs_slot = M->obs_slot[bl];
      o[obs_slot] = q; o[nj + obs_slot] = qd; o[2 * nj + obs_slot] = mtau;
    }
    if (lt == 0) {
      // (RESET launches carry the reward / done pointers only for trex_batch_reset_rows: the env that was reset
      // starts its episode with reward 0, done 0 in the caller's row block)
      if (args.reward) args.reward[so + (size_t)env * args.scal_stride] = (RESET || env_bad) ? 0.f : -lift - drift - energy;
      // should_terminate() is constant False (trex_env.py:183-184): done only flags the harness's episode limit and
      // a contained non-finite env
      if (args.done) args.done[(size_t)(MULTI ? ls : 0) * args.n_envs + env] = (!RESET && (env_bad || time_up)) ? 1 : 0;
      if (args.done_f) args.done_f[so + (size_t)env * args.scal_stride] = (!RESET && (env_bad || time_up)) ? 1.f : 0.f;
      if (args.penalties) {
        float *pn = args.penalties + ((size_t)(MULTI ? ls : 0) * args.n_envs + env) * 3;
        pn[0] = env_bad ? 0.f : lift; pn[1] = env_bad ? 0.f : drift; pn[2] = env_bad ? 0.f : energy;
      }
      if (args.pen_in_rows) {     // ... obs | reward | done | lifting, station keeping, energy: one aligned 320-byte row at J = 25
        float *pn = args.done_f + so + (size_t)env * args.scal_stride + 1;
        const bool zero = RESET || env_bad;
        pn[0] = zero ? 0.f : lift; pn[1] = zero ? 0.f : drift; pn[2] = zero ? 0.f : energy;
      }
    }
  }
  steps_in = steps_out;
  }   // the env-steps of this launch

  // ---- epilogue: the state goes back to HBM
  const int lt = lane_id();
  const int bl = lt & (TL - 1);
  float q = W.st[ST_Q][bl], qd = W.st[ST_QD][bl];
  if (lt >= TL) { q = 0.f; qd = 0.f; }
  const bool store_state = RESET ? do_reset : true;
  if (store_state) {
    // base row: pos(3) quat(4) v(3) w(3); lane k < 13 stores word k (static selects: a dynamically indexed
    // register array would be demoted to scratch memory)
    float *b = args.arr.base + (size_t)env * 16;
    // ... + contact count | motors flag, summed normal impulse, episode steps: the whole 64-byte line in one store
    const float row[16] = {pos[0], pos[1], pos[2], quat[0], quat[1], quat[2], quat[3], bv[0], bv[1], bv[2], bw[0], bw[1], bw[2],
                           __int_as_float((stat_nc & 255) | (motors_on ? TREX_MOTORS_BIT : 0)), stat_imp, __int_as_float(steps_in)};
    float word = row[0];
#pragma unroll
    for (int k = 1; k < 16; k++) word = (lt == k) ? row[k] : word;
    if (lt < 16) b[lt] = word;
    if (lt < TL) {
      args.arr.q[(size_t)env * TL + lt] = q;
      args.arr.qd[(size_t)env * TL + lt] = qd;
    }
  }
  if (lt == 0) {
    if (!RESET && args.bal) {
      // file this env under its contact count for the next launch (the other phase's lists); the LAST wave of the
      // launch - every wave has read the phase and filed its env by then - clears the counts this launch read and
      // flips the phase
      int32_t *B = args.bal;
      const int w = bal_phase ^ 1;
      // (filed under the contact count. Not better, measured: under a work class from a least-squares fit of the
      // lone wave's cycles - 470 k + 11.8 k per contact point + 96 k per point ALIVE in a sweep, scripts/wave_phases.py
      // 1024, residual 19 k against 37 k for the count alone - in 32 classes of 16 k cycles: 11.55 M against 11.59 M)
      const int bin = stat_nc < 0 ? 0 : (stat_nc >= TREX_BAL_BINS ? TREX_BAL_BINS - 1 : stat_nc);
      const int at = atomicAdd(&B[TREX_BAL_COUNTS + TREX_BAL_BINS * w + bin], 1);
      if (at < args.n_envs) B[TREX_BAL_LISTS + (size_t)(w * TREX_BAL_BINS + bin) * args.n_envs + at] = env;
      // (no fence: the lists are read by the NEXT launch only; within this launch the last wave needs nothing but
      // the count of ended waves, an atomic)
      if (atomicAdd(&B[TREX_BAL_FINISHED], 1) == args.n_envs - 1) {
        B[TREX_BAL_FINISHED] = 0;
        for (int i = 0; i < TREX_BAL_BINS; i++) B[TREX_BAL_COUNTS + TREX_BAL_BINS * bal_phase + i] = 0;
        B[TREX_BAL_PHASE] = w;
      }
    }
  }
}

template <bool RESET, bool DEBUG>
__global__ __launch_bounds__(64, 4) void trex_step_kernel(KernelArgs args) { trex_step_body<RESET, DEBUG, false>(args, (int)blockIdx.x); }
// two envs per workgroup with split roles between the barriers of a substep (PAIR above): even batches of at most 4096 envs
__global__ __launch_bounds__(128, 4) void trex_step_pair_kernel(KernelArgs args) { trex_step_body<false, false, false, true>(args, (int)blockIdx.x); }
// S env-steps per launch (trex_batch_step_many)
__global__ __launch_bounds__(64, 4) void trex_step_many_kernel(KernelArgs args) { trex_step_body<false, false, true>(args, (int)blockIdx.x); }

// ---------------------------------------------------------------- small utility kernels
__global__ void trex_pack_state_kernel(const TrexDeviceModel *M, TrexBatchArrays arr, int n, float *out, int pack) {
  // pack=1: internal -> [N, 13+2J]; pack=0: [N, 13+2J] -> internal
  const int env = blockIdx.x * (blockDim.x / TL) + threadIdx.x / TL;
  const int lane = threadIdx.x & (TL - 1);
  if (env >= n) return;
  const int nj = M->nb - 1, width = 13 + 2 * nj;
  float *row = out + (size_t)env * width;
  float *b = arr.base + env * 16;
  if (pack) {
    if (lane < 13) row[lane] = b[lane];
    if (lane >= 1 && lane < M->nb) {
      const int s = M->obs_slot[lane];
      row[13 + s] = arr.q[env * TL + lane];
      row[13 + nj + s] = arr.qd[env * TL + lane];
    }
  } else {
    if (lane < 13) b[lane] = row[lane];
    float qv = 0.f, qdv = 0.f;
    if (lane >= 1 && lane < M->nb) {
      const int s = M->obs_slot[lane];
      qv = row[13 + s]; qdv = row[13 + nj + s];
    }
    arr.q[env * TL + lane] = qv;
    arr.qd[env * TL + lane] = qdv;
  }
}

__global__ void trex_head_kernel(KernelArgs args, float *out) {
  // FK only; one team per env. Reuses nothing from the step kernel to stay simple: serial per lane 0.
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= args.n_envs) return;
  const TrexDeviceModel *M = args.model;
  const float *b = args.arr.base + env * 16;
  float quat[4] = {b[3], b[4], b[5], b[6]};
  // walk from the head body up to the base collecting the chain, then compose base-to-tip
  int chain[MAXD + 1], n = 0;
  for (int i = M->head_body; i > 0; i = M->parent[i]) chain[n++] = i;
  float R[9], r[3] = {0.f, 0.f, 0.f};
  quat_to_mat(quat, R);
  for (int k = n - 1; k >= 0; k--) {
    const int i = chain[k];
    float ax[3] = {M->axis[0][i], M->axis[1][i], M->axis[2][i]}, jp[3] = {M->jpos[0][i], M->jpos[1][i], M->jpos[2][i]};
    float jr[9], rq[9], t[9], o[3];
    for (int c = 0; c < 9; c++) jr[c] = M->jrot[c][i];
    const float q = args.arr.q[env * TL + i];
    const float c = cosf(q), s = sinf(q), tt = 1.f - c;
    rq[0] = tt * ax[0] * ax[0] + c;         rq[1] = tt * ax[0] * ax[1] - s * ax[2]; rq[2] = tt * ax[0] * ax[2] + s * ax[1];
    rq[3] = tt * ax[0] * ax[1] + s * ax[2]; rq[4] = tt * ax[1] * ax[1] + c;         rq[5] = tt * ax[1] * ax[2] - s * ax[0];
    rq[6] = tt * ax[0] * ax[2] - s * ax[1]; rq[7] = tt * ax[1] * ax[2] + s * ax[0]; rq[8] = tt * ax[2] * ax[2] + c;
    matvec3(R, jp, o);
    for (int k2 = 0; k2 < 3; k2++) r[k2] += o[k2];
    matmul3(R, jr, t);
    matmul3(t, rq, R);
  }
  const float hp[3] = {M->head_point[0], M->head_point[1], M->head_point[2]};
  float o[3];
  matvec3(R, hp, o);
  for (int k = 0; k < 3; k++) out[env * 3 + k] = b[k] + r[k] + o[k];
}

// Rollout export: world pose of every URDF link. One 64-thread block per env: lanes < nb walk their
// body's chain from the base (<= 6 hinges) and park R, p in LDS; then the block strides over the links.
__global__ __launch_bounds__(64) void trex_link_transforms_kernel(KernelArgs args, float *out, int L, const int *frame_body,
                                                                  const float *frame_tf) {
  __shared__ float bodyR[TL][9], bodyP[TL][3];
  const int env = blockIdx.x;
  const TrexDeviceModel *M = args.model;
  const int t = threadIdx.x;
  const float *b = args.arr.base + env * 16;
  if (t < M->nb) {
    const float quat[4] = {b[3], b[4], b[5], b[6]};
    float R[9], p[3] = {b[0], b[1], b[2]};
    quat_to_mat(quat, R);
    int chain[MAXD + 1], n = 0;
    for (int i = t; i > 0; i = M->parent[i]) chain[n++] = i;
    for (int k = n - 1; k >= 0; k--) {
      const int i = chain[k];
      const float ax[3] = {M->axis[0][i], M->axis[1][i], M->axis[2][i]}, jp[3] = {M->jpos[0][i], M->jpos[1][i], M->jpos[2][i]};
      float jr[9], rq[9], tmp[9], o[3];
      for (int c = 0; c < 9; c++) jr[c] = M->jrot[c][i];
      const float q = args.arr.q[env * TL + i];
      const float c = cosf(q), s = sinf(q), tt = 1.f - c;
      rq[0] = tt * ax[0] * ax[0] + c;         rq[1] = tt * ax[0] * ax[1] - s * ax[2]; rq[2] = tt * ax[0] * ax[2] + s * ax[1];
      rq[3] = tt * ax[0] * ax[1] + s * ax[2]; rq[4] = tt * ax[1] * ax[1] + c;         rq[5] = tt * ax[1] * ax[2] - s * ax[0];
      rq[6] = tt * ax[0] * ax[2] - s * ax[1]; rq[7] = tt * ax[1] * ax[2] + s * ax[0]; rq[8] = tt * ax[2] * ax[2] + c;
      matvec3(R, jp, o);
      for (int k2 = 0; k2 < 3; k2++) p[k2] += o[k2];
      matmul3(R, jr, tmp);
      matmul3(tmp, rq, R);
    }
    for (int c = 0; c < 9; c++) bodyR[t][c] = R[c];
    for (int c = 0; c < 3; c++) bodyP[t][c] = p[c];
  }
  __syncthreads();
  // the frames to export: the URDF link frames (trex_batch_link_transforms) or the <visual> meshes
  // (trex_batch_visual_transforms), each given by its body and its transform in that body's frame
  for (int l = t; l < L; l += blockDim.x) {
    const int body = frame_body[l];
    const float *tf = frame_tf + 12 * l;
    float R[9], o[3];
    matmul3(bodyR[body], tf, R);
    matvec3(bodyR[body], tf + 9, o);
    float *w = out + ((size_t)env * L + l) * 7;
    for (int c = 0; c < 3; c++) w[c] = bodyP[body][c] + o[c];
    // rotation matrix -> quaternion xyzw (w >= 0)
    float qx, qy, qz, qw;
    const float tr = R[0] + R[4] + R[8];
    if (tr > 0.f) {
      const float s = sqrtf(tr + 1.f) * 2.f;
      qw = 0.25f * s; qx = (R[7] - R[5]) / s; qy = (R[2] - R[6]) / s; qz = (R[3] - R[1]) / s;
    } else if (R[0] >= R[4] && R[0] >= R[8]) {
      const float s = sqrtf(1.f + R[0] - R[4] - R[8]) * 2.f;
      qw = (R[7] - R[5]) / s; qx = 0.25f * s; qy = (R[1] + R[3]) / s; qz = (R[2] + R[6]) / s;
    } else if (R[4] >= R[8]) {
      const float s = sqrtf(1.f + R[4] - R[0] - R[8]) * 2.f;
      qw = (R[2] - R[6]) / s; qx = (R[1] + R[3]) / s; qy = 0.25f * s; qz = (R[5] + R[7]) / s;
    } else {
      const float s = sqrtf(1.f + R[8] - R[0] - R[4]) * 2.f;
      qw = (R[3] - R[1]) / s; qx = (R[2] + R[6]) / s; qy = (R[5] + R[7]) / s; qz = 0.25f * s;
    }
    const float sg = qw < 0.f ? -1.f : 1.f;
    w[3] = sg * qx; w[4] = sg * qy; w[5] = sg * qz; w[6] = sg * qw;
  }
}

// the per-env scalars of the base row (device_model.h): read out / set by the C-ABI's accessors
__global__ void trex_scalars_get_kernel(const float *base, int n, int32_t *count, float *impulse, int32_t *steps) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const float *b = base + (size_t)e * 16;
  if (count) count[e] = __float_as_int(b[TREX_BASE_FLAGS]) & 255;
  if (impulse) impulse[e] = b[TREX_BASE_IMPULSE];
  if (steps) steps[e] = __float_as_int(b[TREX_BASE_STEPS]);
}
__global__ void trex_scalars_set_kernel(float *base, int n, const int32_t *steps, int set_steps, int motors) {
  // set_steps: word 15 <- steps[e] (or 0 if steps == null); motors >= 0: the motors flag <- motors
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float *b = base + (size_t)e * 16;
  if (set_steps) b[TREX_BASE_STEPS] = __int_as_float(steps ? steps[e] : 0);
  if (motors >= 0) {
    const int f = __float_as_int(b[TREX_BASE_FLAGS]);
    b[TREX_BASE_FLAGS] = __int_as_float(motors ? (f | TREX_MOTORS_BIT) : (f & ~TREX_MOTORS_BIT));
  }
}

__global__ void trex_fill_kernel(float *p, float v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void trex_fill_u8_kernel(uint8_t *p, uint8_t v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void trex_copy_mass_scale_kernel(const float *src, float *dst, int n, int nb) {
  // [N, nb] -> [N, 32]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * TL) return;
  const int e = i / TL, l = i % TL;
  dst[i] = l < nb ? src[e * nb + l] : 1.0f;
}

// ---------------------------------------------------------------- host launchers (called by capi.cpp)
extern "C" {

hipError_t trex_launch_step(const TrexDeviceModel *model, TrexBatchArrays arr, int n, const float *actions,
                            float *obs, float *reward, uint8_t *done, float *penalties, float wd, float we,
                            float wk, float *debug, hipStream_t stream, float *done_f, int obs_stride, int scal_stride,
                            int balance, int pen_in_rows) {
  // balance: the env-to-wave assignment by contact rank (trex_batch_set_wave_balance decides; capi.cpp). Diagnostics
  // launches keep env k in workgroup k (the stamped build is balanced like the product: it reports the env of every wave)
  int32_t *perm = ((debug && !TREX_STAMPS) || !balance) ? nullptr : arr.balance;
  KernelArgs a{model, arr, n, actions, obs, reward, done, done_f, obs_stride, scal_stride, penalties, nullptr, perm, wd, we, wk, debug,
               1, 0, pen_in_rows};

#if TREX_STAMPS   // diagnostic build: the PRODUCT instantiation, stamped (the dump of <false, true> would change its code)
  if (TREX_PAIR_LAUNCH && (n & 1) == 0 && n <= TREX_PAIR_MAX) hipLaunchKernelGGL(trex_step_pair_kernel, dim3(n / 2), dim3(128), 0, stream, a);
  else hipLaunchKernelGGL((trex_step_kernel<false, false>), dim3(n), dim3(64), 0, stream, a);
#else
  if (debug) hipLaunchKernelGGL((trex_step_kernel<false, true>), dim3(n), dim3(64), 0, stream, a);
  else if (TREX_PAIR_LAUNCH && (n & 1) == 0 && n <= TREX_PAIR_MAX) hipLaunchKernelGGL(trex_step_pair_kernel, dim3(n / 2), dim3(128), 0, stream, a);
  else hipLaunchKernelGGL((trex_step_kernel<false, false>), dim3(n), dim3(64), 0, stream, a);
#endif
  return hipGetLastError();
}

// S env-steps per launch (open-loop action sequences): actions [S, N, J], rows [S, N, row_stride] = obs | reward | done,
// penalties [S, N, 3] and done bytes [S, N] nullable
hipError_t trex_launch_step_many(const TrexDeviceModel *model, TrexBatchArrays arr, int n, const float *actions, float *rows,
                                 int row_stride, int n_steps, float *penalties, uint8_t *done, float wd, float we, float wk,
                                 hipStream_t stream, int balance, int nj, int pen_in_rows) {
  float *rew = rows + 3 * nj;
  KernelArgs a{model, arr, n, actions, rows, rew, done, rew + 1, row_stride, row_stride, penalties, nullptr,
               balance ? arr.balance : nullptr, wd, we, wk, nullptr, n_steps, (long long)n * row_stride, pen_in_rows};
  hipLaunchKernelGGL(trex_step_many_kernel, dim3(n), dim3(64), 0, stream, a);
  return hipGetLastError();
}

hipError_t trex_launch_reset(const TrexDeviceModel *model, TrexBatchArrays arr, int n, const uint8_t *mask,
                             float *obs, float wd, float we, float wk, float *debug, hipStream_t stream, int obs_stride,
                             float *reward, float *done_f, int scal_stride, int nj, int pen_in_rows) {
  KernelArgs a{model, arr, n, nullptr, obs, reward, nullptr, done_f, obs_stride, scal_stride, nullptr, mask, nullptr, wd, we, wk, debug,
               1, 0, (reward && done_f && pen_in_rows) ? 1 : 0};
  hipLaunchKernelGGL((trex_step_kernel<true, false>), dim3(n), dim3(64), 0, stream, a);
  return hipGetLastError();
}

hipError_t trex_launch_pack_state(const TrexDeviceModel *model, TrexBatchArrays arr, int n, float *state, int pack,
                                  hipStream_t stream) {
  hipLaunchKernelGGL(trex_pack_state_kernel, dim3((n + 7) / 8), dim3(256), 0, stream, model, arr, n, state, pack);
  return hipGetLastError();
}

hipError_t trex_launch_head(const TrexDeviceModel *model, TrexBatchArrays arr, int n, float *out, hipStream_t stream) {
  KernelArgs a{model, arr, n, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, nullptr};
  hipLaunchKernelGGL(trex_head_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a, out);
  return hipGetLastError();
}

hipError_t trex_launch_link_transforms(const TrexDeviceModel *model, TrexBatchArrays arr, int n, float *out, hipStream_t stream,
                                       int visuals) {
  KernelArgs a{model, arr, n, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, nullptr};
  if (visuals) hipLaunchKernelGGL(trex_link_transforms_kernel, dim3(n), dim3(64), 0, stream, a, out, arr.num_visuals, arr.visual_body, arr.visual_tf);
  else hipLaunchKernelGGL(trex_link_transforms_kernel, dim3(n), dim3(64), 0, stream, a, out, arr.num_links, arr.link_body, arr.link_tf);
  return hipGetLastError();
}

hipError_t trex_launch_scalars_get(TrexBatchArrays arr, int n, int32_t *count, float *impulse, int32_t *steps, hipStream_t stream) {
  hipLaunchKernelGGL(trex_scalars_get_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, arr.base, n, count, impulse, steps);
  return hipGetLastError();
}
hipError_t trex_launch_scalars_set(TrexBatchArrays arr, int n, const int32_t *steps, int set_steps, int motors, hipStream_t stream) {
  hipLaunchKernelGGL(trex_scalars_set_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, arr.base, n, steps, set_steps, motors);
  return hipGetLastError();
}

hipError_t trex_launch_fill(float *p, float v, int n, hipStream_t stream) {
  hipLaunchKernelGGL(trex_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, v, n);
  return hipGetLastError();
}
hipError_t trex_launch_fill_u8(uint8_t *p, uint8_t v, int n, hipStream_t stream) {
  hipLaunchKernelGGL(trex_fill_u8_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, v, n);
  return hipGetLastError();
}
hipError_t trex_launch_copy_mass_scale(const float *src, float *dst, int n, int nb, hipStream_t stream) {
  hipLaunchKernelGGL(trex_copy_mass_scale_kernel, dim3((n * TL + 255) / 256), dim3(256), 0, stream, src, dst, n, nb);
  return hipGetLastError();
}

// launch shape of the step launch for a batch of n envs (trex_batch_launch_info): two envs per workgroup - the pair form - for an
// even batch that is resident at once, one otherwise
int trex_step_envs_per_workgroup(int n) { return (TREX_PAIR_LAUNCH && (n & 1) == 0 && n <= TREX_PAIR_MAX) ? 2 : 1; }
int trex_step_lds_bytes(int n) { return trex_step_envs_per_workgroup(n) == 2 ? (int)(2 * sizeof(WaveLds) + sizeof(CgLds)) : (int)sizeof(WaveLds); }

}  // extern "C"
