// K1/K2: fused batched physics step for the T-rex env on gfx950 (MI355X, CDNA4).
//
// Replaces, per env and per launch, what TrexBulletEnv.step() (trex_env.py:128-154) asks of pybullet
// and the robot adapter: clip action -> substeps x [position-motor rows (trex_robot.py:397-422) +
// stepSimulation (trex_env.py:150)] -> observations (trex_robot.py:359-365) + reward
// (trex_env.py:186-196); with RESET: TrexBulletEnv.reset() (trex_env.py:98-122).
//
// Mapping to the hardware
//   * one 64-lane wavefront = one workgroup = TWO envs. Tree phases: one env per 32-lane half ("team"):
//     the tree has 26 bodies and 25 + 6 = 31 degrees of freedom, so a team's lanes are the bodies 0..25
//     (lanes 1..25 = joint of that body, lanes 26..31 = base angular xyz / linear xyz for the
//     factorisation). Constraint solve: one env at a time on all 64 lanes, ONE CONSTRAINT ROW PER LANE
//     (25 motor rows + 3 x 13 contact rows = 64).  No inter-wave synchronisation exists.
//   * every spatial quantity is expressed in WORLD-ALIGNED axes about the body's OWN frame origin
//     (the joint axis passes through it). Parent<->child sweeps therefore need no rotations - only
//     the translation by the joint offset d - and, unlike a single common origin, no quantity is a
//     difference of m*r^2-sized terms (f32-safe: D_i = a.(I a) directly). Base-to-tip passes move
//     6..12 registers per level with wavefront shuffles (ds_bpermute within the half), the
//     tip-to-base articulated-inertia pass stages 27 floats per body through LDS.
//   * M^-1 is never formed by repeated sweeps: the ABA factorisation M^-1 = A^T B A is kept
//     DISTRIBUTED (lane j holds column j of A: <= 6 ancestor entries + 6 base entries); any entry of
//     the Delassus matrix J M^-1 J^T is then 12 multiply-adds of two row descriptors.
//   * projected Gauss-Seidel runs in Delassus (residual) form: a row's impulse change reaches all other
//     rows as one v_readlane (SGPR broadcast) + one FMA per lane - no reduction, no LDS in a row.
//   * HBM traffic per env-step is the state row in/out + action in + obs/reward out (912 B); the
//     kernel is bound by the instruction issue of its heaviest wave, not by bandwidth.
//
// The arithmetic is the one restated by oracle/trex_oracle.c; tests/ compare the two.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

#include <type_traits>

#include "device_model.h"

#define TL TREX_TL
#define MAXD TREX_MAXD
#define MAXCH TREX_MAXCH
#define MAXC TREX_MAXC

#ifndef TREX_STAMPS
#define TREX_STAMPS 0
#endif
#ifndef TREX_PRIO_MODE
#define TREX_PRIO_MODE 2   // 0 none, 1 per-env priority during its sweeps, 2 per-wave priority from contact generation on
#endif
// Diagnostic build only (make stamps): s_memtime at phase boundaries of workgroup 0, written to the
// debug buffer at [3000 + 16*substep + phase] as cycle deltas. Never compiled into the product library.
#if TREX_STAMPS
#define STAMP(i)                                                                          \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    const unsigned long long _t = __builtin_amdgcn_s_memtime();                           \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                   \
    if (DEBUG && args.debug && blockIdx.x == 0 && threadIdx.x == 0) args.debug[3000 + 16 * sub + (i)] += (float)(_t - stamp_last); \
    stamp_last = _t;                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
#define STAMP2(i)                                                                         \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    const unsigned long long _t = __builtin_amdgcn_s_memtime();                           \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                   \
    if (DEBUG && args.debug && blockIdx.x == 0 && threadIdx.x == 0) args.debug[3100 + 8 * sub + (i)] = (float)(_t - stamp_last); \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
#else
#define STAMP(i) do {} while (0)
#define STAMP2(i) do {} while (0)
#endif

namespace {

// ---------------------------------------------------------------- team (32-lane) primitives
__device__ __forceinline__ float tshfl(float v, int src) { return __shfl(v, src, TL); }
__device__ __forceinline__ int tshfl(int v, int src) { return __shfl(v, src, TL); }
// all-reduce sum over the 32 lanes of a team on the VALU (no LDS round trips): four DPP adds inside
// the 16-lane rows, then gfx950's v_permlane16_swap exchanges row 0<->1 and 2<->3.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float tsum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wsum(float v) {   // all-reduce over the 64 lanes of the wave
  v = tsum(v);
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float tminf(float v) {
  v = fminf(v, dpp_mov<0xB1>(v));
  v = fminf(v, dpp_mov<0x4E>(v));
  v = fminf(v, dpp_mov<0x141>(v));
  v = fminf(v, dpp_mov<0x140>(v));
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ int tmini(int v) {   // non-negative values
  v = min(v, dpp_mov_i<0xB1>(v));
  v = min(v, dpp_mov_i<0x4E>(v));
  v = min(v, dpp_mov_i<0x141>(v));
  v = min(v, dpp_mov_i<0x140>(v));
  const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  return min((int)r[0], (int)r[1]);
}
// team arg-max with ties to the lowest index: returns the winning (score, index) on every lane
__device__ __forceinline__ void targmax(float &score, int &index) {
  const float best = -tminf(-score);
  index = tmini(score == best ? index : 0x7fffffff);
  score = best;
}
// value held by team lane `src` (src WAVE-uniform), through SGPRs: two v_readlane + one select
__device__ __forceinline__ float tbcast(float v, int src) {
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
  const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src + 32));
  return (threadIdx.x & 32) ? b : a;
}
__device__ __forceinline__ unsigned tballot(bool p) {
  unsigned long long b = __ballot(p);
  return (unsigned)(b >> (threadIdx.x & 32));
}
__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

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
// M -= U U^T * s
__device__ __forceinline__ void sym6_rank1_sub(Sym6 &m, const float *U, float s) {
  const int ia[6] = {0, 0, 0, 1, 1, 2}, ib[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
  for (int k = 0; k < 6; k++) {
    m.A[k] -= U[ia[k]] * U[ib[k]] * s;
    m.C[k] -= U[3 + ia[k]] * U[3 + ib[k]] * s;
  }
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) m.B[3 * r + c] -= U[r] * U[3 + c] * s;
}

// inverse of an SPD 6x6 (Cholesky), result as 21 unique entries of the symmetric inverse, row-major
// upper triangle: inv[tri(r,c)], r<=c
__device__ __forceinline__ int tri(int r, int c) { return r * 6 - r * (r - 1) / 2 + (c - r); }
__device__ __forceinline__ void spd6_inverse(const Sym6 &m, float *inv21) {
  float a[36], l[36];
  sym6_full(m, a);
#pragma unroll
  for (int i = 0; i < 36; i++) l[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) {
      float s = a[6 * i + j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= l[6 * i + k] * l[6 * j + k];
      if (i == j) l[6 * i + j] = sqrtf(s);
      else l[6 * i + j] = s / l[6 * j + j];
    }
  // Linv (lower triangular), then inv = Linv^T Linv
  float li[36];
#pragma unroll
  for (int i = 0; i < 36; i++) li[i] = 0.f;
#pragma unroll
  for (int c = 0; c < 6; c++) {
#pragma unroll
    for (int i = c; i < 6; i++) {
      float s = (i == c) ? 1.f : 0.f;
#pragma unroll
      for (int k = c; k < i; k++) s -= l[6 * i + k] * li[6 * k + c];
      li[6 * i + c] = s / l[6 * i + i];
    }
  }
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int c = r; c < 6; c++) {
      float s = 0.f;
#pragma unroll
      for (int k = c; k < 6; k++) s += li[6 * k + r] * li[6 * k + c];
      inv21[tri(r, c)] = s;
    }
}
__device__ __forceinline__ void inv21_mul(const float *inv21, const float *v, float *o) {
#pragma unroll
  for (int r = 0; r < 6; r++) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 6; c++) s += inv21[r <= c ? tri(r, c) : tri(c, r)] * v[c];
    o[r] = s;
  }
}

// ---------------------------------------------------------------- LDS layout (per wave = two envs)
constexpr int NJMAX = TL - 7;            // 25 hinge joints at most (26 bodies + 6 base dofs = 32 lanes)
constexpr int NROW = NJMAX + 3 * MAXC;   // constraint rows of one env: 25 motor rows + 13 x (normal, 2 friction) = 64
static_assert(NROW <= 64, "one constraint row per lane");
// Row descriptors. Column side, read as a wave-wide broadcast: chain nodes ca[6] | zc[6] = u/D at those
// nodes | z0[6] = I0^-1 r0. Own side, read once by the lane that owns the row: u[6] | r0[6] | 1/diag | scaled
// right-hand side. Records: motor rows of env 0, motor rows of env 1 (staged once per substep, straight
// after the factorisation), contact rows of the env being solved, one null record.
constexpr int CROW0 = 2 * NJMAX;            // first contact record
constexpr int NREC = CROW0 + 3 * MAXC + 1;  // 90 records, the last one null
struct RowStage {
  float4 col[NREC][5];        // 20 words per row (2 pad), 16-byte aligned broadcast reads      7200 B
  float own[NREC][15];        // odd stride: conflict-free per-lane reads                       5400 B
};
struct WaveLds {
  union {
    float aba[2][TL][28];     // per team: tip-to-base staging, Ia (21) + pa (6) per body       7168 B
    RowStage rows;            // afterwards: the row descriptors                               12600 B
  } u;
  float jcol[NJMAX][64];      // [j-1][row lane]: B entries of motor column j                   6400 B
  float i0inv[2][24];         // per team: inverse of the base's articulated inertia (21)        192 B
};
static_assert(sizeof(WaveLds) <= 20480, "8 workgroups per CU need <= 20 KB of LDS each");

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
  const int32_t *perm;        // wave slot -> env id (step launches), nullable = identity
  float w_distance, w_energy, w_drift;
  float *debug;           // diagnostics of env 0's last substep (tests), nullable
};

}  // namespace

// DEBUG instantiations carry the diagnostics dump (tests, phase stamps); the product launches use
// DEBUG = false so that none of the dump's address arithmetic exists in the shipped kernels.
template <bool RESET, bool DEBUG>
__global__ __launch_bounds__(64, 2) void trex_step_kernel(KernelArgs args) {
  __shared__ WaveLds W;
#if !TREX_STAMPS
#define WPH(i) do {} while (0)
#endif
#if TREX_STAMPS
  const unsigned long long wave_t0 = __builtin_amdgcn_s_memtime();
  int dbg_bodies = 0, dbg_passes = 0, dbg_trips = 0;   // pass B: body iterations, passes, candidate trips
  unsigned long long wph[5] = {0, 0, 0, 0, 0}, wph_t = wave_t0;   // per-wave phase cycles: tree, contacts, walk+build, sweeps, rest
#define WPH(i) do { const unsigned long long _w = __builtin_amdgcn_s_memtime(); wph[i] += _w - wph_t; wph_t = _w; } while (0)
  unsigned long long wave_cg = 0, wave_cg1 = 0, wave_cg2 = 0;   // cycles in contact generation: all, small-hull scan, large-hull scan
#endif
  const int lane = threadIdx.x & (TL - 1);
  const int team = threadIdx.x >> 5;
  const TrexDeviceModel *__restrict__ M = args.model;
  int env = blockIdx.x * 2 + team;   // wave slot
  const bool env_ok = env < args.n_envs;
  if (!env_ok) env = args.n_envs - 1;  // duplicate the last env's work, never store it
  if (args.perm) env = args.perm[env];

  const int nb = M->nb, maxdepth = M->maxdepth;
  const float dt = M->prm[TP_DT];
  const float inv_dt = 1.0f / dt;
  const bool is_body = lane < nb;
  const bool is_joint = lane >= 1 && lane < nb;
  const int bdof = lane - nb;  // 0..5 on base dof lanes
  const bool is_base_dof = bdof >= 0 && bdof < 6;

  // ---- model constants of this lane's body. Topology (integers) stays in registers for the whole
  // launch; the geometric constants (24 floats) are re-read from the L2-resident model at the top of
  // every substep instead of being carried - and spilled - across the solver loop.
  const int parent = is_body ? M->parent[lane] : 0;
  const int psrc = parent < 0 ? 0 : parent;
  const int depth = is_body ? M->depth[lane] : -1;
  // Everything else about the model is (re)read from the L2-resident struct in the phase that uses
  // it, through an opaque pointer, so that no constant is live - and spilled - across the solver loop.
  auto Mo = [&]() { const TrexDeviceModel *Mi = M; asm volatile("" : "+s"(Mi)); return Mi; };
  int anc[MAXD];
  float axis[3], jpos[3], jrot[9], comb[3], inb[6];
  float mass = 0.f, mscale = 1.f, jdamp = 0.f;
  auto load_body_constants = [&]() {
    const TrexDeviceModel *Mi = Mo();
    mscale = args.arr.mass_scale[env * TL + lane];
    mass = Mi->mass[lane] * mscale;
    jdamp = Mi->damp[lane];
#pragma unroll
    for (int c = 0; c < 3; c++) { axis[c] = Mi->axis[c][lane]; jpos[c] = Mi->jpos[c][lane]; comb[c] = Mi->com[c][lane]; }
#pragma unroll
    for (int c = 0; c < 9; c++) jrot[c] = Mi->jrot[c][lane];
#pragma unroll
    for (int c = 0; c < 6; c++) inb[c] = Mi->inertia[c][lane];
  };
  const int nj = nb - 1;

  // ---- per-env state
  float pos[3], quat[4], bv[3], bw[3], q, qd, mtau = 0.f;
  const float mu = args.arr.friction[env];
  bool motors_on;
  bool do_reset = false;
  if (RESET) do_reset = args.reset_mask ? (args.reset_mask[env] != 0) : true;
  if (RESET && do_reset) {
#pragma unroll
    for (int c = 0; c < 3; c++) { pos[c] = M->base_pos0[c]; bv[c] = 0.f; bw[c] = 0.f; }
#pragma unroll
    for (int c = 0; c < 4; c++) quat[c] = M->base_quat0[c];
    q = M->q_start[lane]; qd = 0.f;
    motors_on = false;  // remove_joint_control, trex_robot.py:309
  } else {
    const float *b = args.arr.base + env * 16;
#pragma unroll
    for (int c = 0; c < 3; c++) { pos[c] = b[c]; bv[c] = b[7 + c]; bw[c] = b[10 + c]; }
#pragma unroll
    for (int c = 0; c < 4; c++) quat[c] = b[3 + c];
    q = args.arr.q[env * TL + lane];
    qd = args.arr.qd[env * TL + lane];
    mtau = args.arr.tau[env * TL + lane];
    motors_on = RESET ? (args.arr.motors_on[env] != 0) : true;
  }
  // non-finite input state (checked here as well as after the step: fminf/fmaxf clamps launder NaNs)
  bool bad = !(fabsf(q) < 3.0e38f) || !(fabsf(qd) < 3.0e38f);
#pragma unroll
  for (int k = 0; k < 3; k++) bad |= !(fabsf(pos[k]) < 3.0e38f) || !(fabsf(bv[k]) < 3.0e38f) || !(fabsf(bw[k]) < 3.0e38f);
#pragma unroll
  for (int k = 0; k < 4; k++) bad |= !(fabsf(quat[k]) < 3.0e38f);
  float target = 0.f;
  if (!RESET && is_joint) {
    const float a = args.actions[env * nj + M->obs_slot[lane]];
    target = fminf(fmaxf(a, M->lower[lane]), M->upper[lane]);  // np.clip, trex_env.py:147
  }
  const int n_sub = RESET ? (do_reset ? 1 : 0) : (int)M->prm[TP_SUBSTEPS];
  // a team that does not reset still walks through the loop when its wave partner resets
  const int n_sub_wave = RESET ? (wave_any(do_reset) ? 1 : 0) : n_sub;

  const float grav = M->prm[TP_GRAVITY], kdamp = M->prm[TP_LINK_DAMPING], vmax = M->prm[TP_MAX_COORD_VEL];
  const float floor_z = M->prm[TP_FLOOR_Z], margin = M->prm[TP_CONTACT_MARGIN];
  const float erp = M->prm[TP_ERP], cerp = M->prm[TP_CONTACT_ERP];
  const float kp = M->prm[TP_MOTOR_KP], kd = M->prm[TP_MOTOR_KD], max_imp = M->prm[TP_MOTOR_MAX_FORCE] * dt;
  const int iters = (int)M->prm[TP_ITERATIONS];
  int maxc = (int)M->prm[TP_MAX_CONTACTS];
  if (maxc > MAXC) maxc = MAXC;

  // kinematic quantities of this lane's body
  float R[9], r[3], dpar[3] = {0.f, 0.f, 0.f}, S[6];   // dpar = r - r(parent), world axes
  int stat_nc = 0;
  float stat_imp = 0.f;

  // FK: world rotation R and origin r (relative to the base origin) of every body; joint motion
  // subspace about the body's own origin S = [a; 0].  (base-to-tip, parent data via shuffles)
  auto forward_kinematics = [&]() {
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
    for (int d = 1; d <= maxdepth; d++) {
      float pR[9], pr[3];
#pragma unroll
      for (int c = 0; c < 9; c++) pR[c] = tshfl(R[c], psrc);
#pragma unroll
      for (int c = 0; c < 3; c++) pr[c] = tshfl(r[c], psrc);
      if (depth == d) {
        float o[3];
        matmul3(pR, Rl, R);
        matvec3(pR, jpos, o);
#pragma unroll
        for (int c = 0; c < 3; c++) { dpar[c] = o[c]; r[c] = pr[c] + o[c]; }
      }
    }
    float a[3];
    matvec3(R, axis, a);
    S[0] = a[0]; S[1] = a[1]; S[2] = a[2];
    S[3] = S[4] = S[5] = 0.f;
    if (!is_joint) {
#pragma unroll
      for (int c = 0; c < 6; c++) S[c] = 0.f;
    }
  };

  // spatial velocity of every body ABOUT ITS OWN ORIGIN for base twist (w, v) and joint rates
  auto body_velocities = [&](const float *w, const float *v, float rate, float *vel) {
#pragma unroll
    for (int c = 0; c < 3; c++) { vel[c] = w[c]; vel[3 + c] = v[c]; }
    for (int d = 1; d <= maxdepth; d++) {
      float pv[6];
#pragma unroll
      for (int c = 0; c < 6; c++) pv[c] = tshfl(vel[c], psrc);
      if (depth == d) {
        float wxd[3];
        cross3(pv, dpar, wxd);   // velocity of the parent-body point at this body's origin
#pragma unroll
        for (int c = 0; c < 3; c++) { vel[c] = pv[c] + S[c] * rate; vel[3 + c] = pv[3 + c] + wxd[c]; }
      }
    }
  };

  for (int sub = 0; sub < n_sub_wave; sub++) {
    const bool live = sub < n_sub;  // this team really advances
#if TREX_STAMPS
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#endif
    // `ls` = lane, opaque to the optimiser once per substep: lane-derived masks and unit vectors are
    // then recomputed where used (1 VALU) instead of being hoisted out of the loop and spilled.
    int ls = lane;
    asm volatile("" : "+v"(ls));
    const int bdof_s = ls - nb;
    const bool is_base_dof_s = bdof_s >= 0 && bdof_s < 6;
    load_body_constants();
    forward_kinematics();
    float vel[6];
    body_velocities(bw, bv, qd, vel);

    STAMP(0);
    // ---- rigid-body spatial inertia about the body origin, bias force, velocity-product acceleration
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
    Sym6 IA;
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
    float pA[6], cv[6];
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
      // c = vel x (S qd)
      float sq[6];
#pragma unroll
      for (int k = 0; k < 6; k++) sq[k] = S[k] * qd;
      float x0[3], x1[3], x2[3];
      cross3(vel, sq, x0); cross3(vel, sq + 3, x1); cross3(vel + 3, sq, x2);
#pragma unroll
      for (int k = 0; k < 3; k++) { cv[k] = x0[k]; cv[3 + k] = x1[k] + x2[k]; }
      if (!is_body) {
#pragma unroll
        for (int k = 0; k < 6; k++) { pA[k] = 0.f; cv[k] = 0.f; }
      }
    }

    STAMP(1);
    // ---- ABA pass 2 (tip to base): articulated inertias and bias forces through LDS
    float U[6], Ud[6], invD = 0.f, u = 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++) { U[k] = 0.f; Ud[k] = 0.f; }
    const float tau_j = -jdamp * qd;  // explicit joint damping torque
    int child[MAXCH];
    {
      const TrexDeviceModel *Mi = Mo();
#pragma unroll
      for (int k = 0; k < MAXCH; k++) child[k] = is_body ? Mi->child[k][lane] : -1;
    }
    for (int d = maxdepth; d >= 1; d--) {
      if (depth == d) {
        sym6_mul(IA, S, U);
        invD = 1.0f / dot6(S, U);
#pragma unroll
        for (int k = 0; k < 6; k++) Ud[k] = U[k] * invD;
        u = tau_j - dot6(S, pA);
        float Ic[6];
        sym6_mul(IA, cv, Ic);
        const float uc = dot6(U, cv);
        Sym6 Ia = IA;
        sym6_rank1_sub(Ia, U, invD);
        float pa[6];
#pragma unroll
        for (int k = 0; k < 6; k++) pa[k] = pA[k] + Ic[k] + U[k] * (u - uc) * invD;
        // shift both to the parent's origin (this origin = parent origin + dpar):
        //   B' = B + [d]x C,  A' = A - B [d]x + [d]x B'^T,  C' = C,  n' = n + d x f
        {
          const int sidx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
          float Bn[9], BD[9], DBt[9];
#pragma unroll
          for (int j = 0; j < 3; j++) {   // column j of [d]x C = d x (column j of C)
            const float cj[3] = {Ia.C[sidx[0][j]], Ia.C[sidx[1][j]], Ia.C[sidx[2][j]]};
            float t[3];
            cross3(dpar, cj, t);
#pragma unroll
            for (int i = 0; i < 3; i++) Bn[3 * i + j] = Ia.B[3 * i + j] + t[i];
          }
#pragma unroll
          for (int i = 0; i < 3; i++) {   // row i of B [d]x = -(d x row i of B)
            float t[3];
            cross3(dpar, Ia.B + 3 * i, t);
#pragma unroll
            for (int j = 0; j < 3; j++) BD[3 * i + j] = -t[j];
          }
#pragma unroll
          for (int j = 0; j < 3; j++) {   // column j of [d]x B'^T = d x (row j of B')
            float t[3];
            cross3(dpar, Bn + 3 * j, t);
#pragma unroll
            for (int i = 0; i < 3; i++) DBt[3 * i + j] = t[i];
          }
          const int ia6[6] = {0, 0, 0, 1, 1, 2}, ib6[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
          for (int k = 0; k < 6; k++) {   // symmetric part (exactly symmetric in exact arithmetic)
            const int ij = 3 * ia6[k] + ib6[k], ji = 3 * ib6[k] + ia6[k];
            Ia.A[k] += 0.5f * ((DBt[ij] - BD[ij]) + (DBt[ji] - BD[ji]));
          }
#pragma unroll
          for (int k = 0; k < 9; k++) Ia.B[k] = Bn[k];
          float dxf[3];
          cross3(dpar, pa + 3, dxf);
#pragma unroll
          for (int k = 0; k < 3; k++) pa[k] += dxf[k];
        }
        float *o = W.u.aba[team][lane];
#pragma unroll
        for (int k = 0; k < 6; k++) { o[k] = Ia.A[k]; o[15 + k] = Ia.C[k]; o[21 + k] = pa[k]; }
#pragma unroll
        for (int k = 0; k < 9; k++) o[6 + k] = Ia.B[k];
      }
      __syncthreads();
      if (depth == d - 1) {
#pragma unroll
        for (int kc = 0; kc < MAXCH; kc++) {
          const int ch = child[kc];
          if (ch >= 0) {
            const float *o = W.u.aba[team][ch];
#pragma unroll
            for (int k = 0; k < 6; k++) { IA.A[k] += o[k]; IA.C[k] += o[15 + k]; pA[k] += o[21 + k]; }
#pragma unroll
            for (int k = 0; k < 9; k++) IA.B[k] += o[6 + k];
          }
        }
      }
      __syncthreads();
    }

    STAMP(2);
    // ---- floating base: a0 = -(IA_0)^-1 pA_0 ; broadcast the inverse to the whole team
    float I0inv[21], a0[6];
    {
      float inv_l[21];
      spd6_inverse(IA, inv_l);
#pragma unroll
      for (int k = 0; k < 21; k++) I0inv[k] = tshfl(inv_l[k], 0);
      float p0[6];
#pragma unroll
      for (int k = 0; k < 6; k++) p0[k] = -tshfl(pA[k], 0);
      inv21_mul(I0inv, p0, a0);
      if (lane < 21) {   // parked for the contact rows: not carried in registers across contact generation
        float v = I0inv[0];
#pragma unroll
        for (int k = 1; k < 21; k++) v = (lane == k) ? I0inv[k] : v;
        W.i0inv[team][lane] = v;
      }
    }
    STAMP(3);
    // ---- ABA pass 3 (base to tip): accelerations
    float qdd = 0.f;
    {
      float acc[6];
#pragma unroll
      for (int k = 0; k < 6; k++) acc[k] = a0[k];
      for (int d = 1; d <= maxdepth; d++) {
        float pa[6];
#pragma unroll
        for (int k = 0; k < 6; k++) pa[k] = tshfl(acc[k], psrc);
        if (depth == d) {
          float axd[3];
          cross3(pa, dpar, axd);   // parent acceleration seen at this body's origin
#pragma unroll
          for (int k = 0; k < 3; k++) pa[3 + k] += axd[k];
#pragma unroll
          for (int k = 0; k < 6; k++) pa[k] += cv[k];
          qdd = (u - dot6(U, pa)) * invD;
#pragma unroll
          for (int k = 0; k < 6; k++) acc[k] = pa[k] + S[k] * qdd;
        }
      }
    }
    // ---- unconstrained velocity update; vg = this dof lane's generalised velocity
    float nw[3], nv[3];
    {
      float wxv[3];
      cross3(bw, bv, wxv);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        nw[k] = fminf(fmaxf(bw[k] + a0[k] * dt, -vmax), vmax);
        nv[k] = fminf(fmaxf(bv[k] + (a0[3 + k] + wxv[k]) * dt, -vmax), vmax);
      }
    }
    float nqd = fminf(fmaxf(qd + qdd * dt, -vmax), vmax);
    float vg = is_joint ? nqd : 0.f;
    // dof-lane motion subspace about the dof's own origin: joint lanes S about r, base dof lanes unit
    // vectors about O
    float Sd[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      Sd[k] = is_joint ? S[k] : ((is_base_dof_s && bdof_s == k) ? 1.f : 0.f);
      if (is_base_dof_s && bdof_s == k) vg = (k < 3) ? nw[k] : nv[k - 3];
    }

    STAMP(4);
    // ---- distributed factorisation M^-1 = A^T B A: this lane's column of A
    //   Aanc[d-1] = entry at its ancestor of depth d (1 at its own depth), A0 = base block entry,
    //   Z[d-1]    = Aanc[d-1] / D(ancestor), g = I0inv * A0
    float Aanc[MAXD], Z[MAXD], A0[6], g[6];
    {
      const TrexDeviceModel *Mi = Mo();
#pragma unroll
      for (int d = 0; d < MAXD; d++) anc[d] = is_body ? Mi->anc[d][lane] : -1;
    }
    {
      float p[6], po[3];   // force p about the point po (starts at this joint's origin, walks up)
#pragma unroll
      for (int k = 0; k < 6; k++) p[k] = is_joint ? Ud[k] : 0.f;
#pragma unroll
      for (int k = 0; k < 3; k++) po[k] = r[k];
#pragma unroll
      for (int d = MAXD; d >= 1; d--) {
        Aanc[d - 1] = 0.f; Z[d - 1] = 0.f;
        if (d <= maxdepth) {
          const int a = anc[d - 1] < 0 ? 0 : anc[d - 1];
          float aa[3], Uda[6], ra[3];
#pragma unroll
          for (int k = 0; k < 3; k++) { aa[k] = tshfl(S[k], a); ra[k] = tshfl(r[k], a); }
#pragma unroll
          for (int k = 0; k < 6; k++) Uda[k] = tshfl(Ud[k], a);
          const float invDa = tshfl(invD, a);
          if (is_joint && depth == d) { Aanc[d - 1] = 1.f; Z[d - 1] = invD; }
          else if (is_joint && depth > d) {
            float dd[3], dxf[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { dd[k] = po[k] - ra[k]; po[k] = ra[k]; }
            cross3(dd, p + 3, dxf);
#pragma unroll
            for (int k = 0; k < 3; k++) p[k] += dxf[k];
            const float ua = -dot3(aa, p);
            Aanc[d - 1] = ua; Z[d - 1] = ua * invDa;
#pragma unroll
            for (int k = 0; k < 6; k++) p[k] += Uda[k] * ua;
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
      for (int k = 0; k < 6; k++) A0[k] = is_joint ? -p[k] : ((is_base_dof_s && bdof_s == k) ? 1.f : 0.f);
      inv21_mul(I0inv, A0, g);
    }
    // Diagonal of M^-1 on the joint lanes (motor / limit rows are unit rows): a^T B a of this lane's own
    // column of A.
    STAMP(5);
    float mdiag = 1.f;
    if (is_joint) {
      mdiag = dot6(A0, g);
#pragma unroll
      for (int d = 0; d < MAXD; d++) mdiag += Aanc[d] * Z[d];
    }

    STAMP(6);
    // ---- joint rows: limits (unilateral, ERP) and position motors
    const float inv_mdiag = 1.0f / mdiag;
    float lim_dir = 0.f, lim_rhs = 0.f, lim_lam = 0.f;
    if (is_joint) {
      const TrexDeviceModel *Mi = Mo();
      const float q_lo = Mi->lower[lane], q_hi = Mi->upper[lane];
      float pen = 0.f;
      if (q - q_lo <= 0.f) { pen = q - q_lo; lim_dir = 1.f; }
      else if (q_hi - q <= 0.f) { pen = q_hi - q; lim_dir = -1.f; }
      lim_rhs = (-pen * erp * inv_dt - lim_dir * vg) * inv_mdiag;
    }
    const unsigned lim_mask = tballot(lim_dir != 0.f);
    float mot_rhs = 0.f, mot_lam = 0.f;
    const float mot_hi = (is_joint && motors_on) ? max_imp : 0.f;
    if (is_joint) {
      // btMultiBodyJointMotor velocity target: kp*(target-q)/dt + qd + kd*(0-qd), minus current qd
      const float tv = kp * (target - q) * inv_dt + vg + kd * (0.f - vg);
      mot_rhs = (tv - vg) * inv_mdiag;
    }

    STAMP(7);
    WPH(0);
#if TREX_STAMPS
    const unsigned long long cg_t0 = __builtin_amdgcn_s_memtime();
#endif
    // ---- contact generation: hull vertices against z <= floor_z
    // Pass A walks the near bodies once: it finds whether a body has any vertex inside the margin and,
    // in the same sweep, its DEEPEST such vertex (= the first point the selection rule keeps), parked in
    // lane b's registers. Pass B revisits a body's vertices only when more than one point per body is
    // kept (K >= 2, i.e. fewer than 8 bodies touch).
    int nc = 0;
    int cbody = 0;
    float cx[3] = {0.f, 0.f, 0.f}, cdist = 0.f;
    {
      const TrexDeviceModel *Mi = Mo();
      const int hull_v0 = Mi->hull_start[lane < nb ? lane : nb], hull_v1 = Mi->hull_start[lane < nb ? lane + 1 : nb];
      float sc[3];
      float sph[3], boxh[3];   // read here, not at the top of the substep: not carried across the ABA passes
#pragma unroll
      for (int c = 0; c < 3; c++) { sph[c] = Mi->sphere[c][lane]; boxh[c] = Mi->box_half[c][lane]; }
      matvec3(R, sph, sc);
      // broad phase: lowest point of the hull's oriented bounding box (conservative, much tighter than
      // a sphere for the long bones): z_centre - sum_k |R_zk| half_k
      const float reach = fabsf(R[6]) * boxh[0] + fabsf(R[7]) * boxh[1] + fabsf(R[8]) * boxh[2];
      const bool near = is_body && hull_v1 > hull_v0 && (pos[2] + r[2] + sc[2] - reach - floor_z < margin);
      unsigned active_mask = 0u;
      float a_x[3] = {0.f, 0.f, 0.f}, a_d = 0.f;   // lane b: deepest candidate of body b
      int a_v = -1;
      // (i) small hulls (<= 96 vertices: every foot, shank, neck and tail segment): the BODY LANE scans its
      // own vertices serially with its own R, r - all near bodies in parallel, no shuffles, no reductions.
      constexpr int SMALL_HULL = 96;
      const int nverts = hull_v1 - hull_v0;
      const bool small = near && nverts <= SMALL_HULL;
      // In-margin vertex sets, kept for the point selection (pass B never sweeps a hull again): a small body's
      // lane keeps bit i of cm[i / 32] for its vertex i; for the (at most two) large bodies every team lane
      // keeps bit i for its strided vertex v0 + lane + 32 i.
      unsigned cm0 = 0u, cm1 = 0u, cm2 = 0u;
      unsigned imL0 = 0u, imL1 = 0u;
      int bL0 = -1, bL1 = -1;
      if (wave_any(small)) {
        float bs = -3.0e38f;
        constexpr int UN = 16;   // 16 independent 16-B loads in flight per lane (256 contiguous bytes)
        for (int i0 = 0; wave_any(small && i0 < nverts); i0 += UN) {
          unsigned bm = 0u;
          float4 h[UN];
#pragma unroll
          for (int u = 0; u < UN; u++) {
            const int i = min(i0 + u, nverts - 1);
            h[u] = args.arr.hull[hull_v0 + (small ? max(i, 0) : 0)];
          }
#pragma unroll
          for (int u = 0; u < UN; u++) {
            const int i = i0 + u;
            // only the height decides; the winner's position is formed once, after the scan.
            // h.w = support radius (0 for a hull vertex): the contact point is the sphere's lowest point
            const float wz = R[6] * h[u].x + R[7] * h[u].y + R[8] * h[u].z;
            const float dd = pos[2] + r[2] + wz - h[u].w - floor_z;
            const bool in = small && i < nverts && dd < margin;
            bm |= in ? (1u << u) : 0u;
            if (in && -dd > bs) { bs = -dd; a_v = hull_v0 + i; a_d = dd; }
          }
          const unsigned add = bm << (i0 & 16);
          if ((i0 >> 5) == 0) cm0 |= add;
          else if ((i0 >> 5) == 1) cm1 |= add;
          else cm2 |= add;
        }
        if (small && a_v >= 0) {
          const float4 hw = args.arr.hull[a_v];
          const float hv[3] = {hw.x, hw.y, hw.z};
          float w[3];
          matvec3(R, hv, w);
          a_x[0] = r[0] + w[0]; a_x[1] = r[1] + w[1]; a_x[2] = r[2] + w[2] - hw.w;
        }
        active_mask |= tballot(small && a_v >= 0);
      }
#if TREX_STAMPS
      const unsigned long long cg_t1 = __builtin_amdgcn_s_memtime();
      wave_cg1 += cg_t1 - cg_t0;
#endif
      STAMP2(0);
      // (ii) large hulls (cranium, pelvis+ribcage): the team strides over the vertices together
      unsigned near_mask = tballot(near && !small);
      while (wave_any(near_mask != 0u)) {
        const bool valid = near_mask != 0u;
        const int b = valid ? (__ffs(near_mask) - 1) : 0;
        near_mask &= near_mask - 1u;
        float Rb[9], rb[3];
#pragma unroll
        for (int c = 0; c < 9; c++) Rb[c] = tshfl(R[c], b);
#pragma unroll
        for (int c = 0; c < 3; c++) rb[c] = tshfl(r[c], b);
        const int v0 = tshfl(hull_v0, b), v1 = tshfl(hull_v1, b);
        float bs = -3.0e38f, bx[3] = {0.f, 0.f, 0.f};
        int bi = 0x7fffffff;
        unsigned im = 0u;
        constexpr int UL = 4;   // 4 coalesced 512-B loads in flight per team
        for (int it0 = 0; wave_any(valid && v0 + TL * it0 < v1); it0 += UL) {
          float4 h[UL];
#pragma unroll
          for (int u = 0; u < UL; u++) {
            const int v = v0 + lane + TL * (it0 + u);
            h[u] = args.arr.hull[(valid && v < v1) ? v : v0];
          }
#pragma unroll
          for (int u = 0; u < UL; u++) {
            const int v = v0 + lane + TL * (it0 + u);
            const float wz = Rb[6] * h[u].x + Rb[7] * h[u].y + Rb[8] * h[u].z;   // the height decides
            const float dd = pos[2] + rb[2] + wz - h[u].w - floor_z;
            if (valid && v < v1 && dd < margin) {
              im |= (it0 + u < 32) ? (1u << (it0 + u)) : 0u;
              if (-dd > bs) { bs = -dd; bi = v; }
            }
          }
        }
        if (bi != 0x7fffffff) {   // position of this lane's deepest vertex, once
          const float4 hw = args.arr.hull[bi];
          const float hv[3] = {hw.x, hw.y, hw.z};
          float w[3];
          matvec3(Rb, hv, w);
          bx[0] = rb[0] + w[0]; bx[1] = rb[1] + w[1]; bx[2] = rb[2] + w[2] - hw.w;
        }
        if (valid) {
          if (bL0 < 0) { bL0 = b; imL0 = im; }
          else if (bL1 < 0) { bL1 = b; imL1 = im; }
        }
        const int mine = bi;
        targmax(bs, bi);
        if (valid && bi != 0x7fffffff) {
          active_mask |= 1u << b;
          const int win = __ffs(tballot(mine == bi)) - 1;   // the lane that holds the winner's position
          const float wx = tshfl(bx[0], win), wy = tshfl(bx[1], win), wz = tshfl(bx[2], win);
          if (lane == b) { a_x[0] = wx; a_x[1] = wy; a_x[2] = wz; a_d = -bs; a_v = bi; }
        }
      }
#if TREX_STAMPS
      wave_cg2 += __builtin_amdgcn_s_memtime() - cg_t1;
#endif
      STAMP2(1);
      const int n_active = __popc(active_mask);
      int K = n_active > 0 ? maxc / n_active : 0;
      K = K > 4 ? 4 : (K < 1 ? 1 : K);
      STAMP2(2);
      if (!wave_any(n_active > 0 && K >= 2)) {
        // fast path (one point per touching body in both envs of the wave - the standing case): contact
        // slot c takes the c-th touching body; every lane finds its body and gathers its point at once
        unsigned m = active_mask;
        for (int i = 0; i < lane && m != 0u; i++) m &= m - 1u;
        nc = min(n_active, maxc);
        const int bsrc = m != 0u ? (__ffs(m) - 1) : 0;
        const float gx = tshfl(a_x[0], bsrc), gy = tshfl(a_x[1], bsrc), gz = tshfl(a_x[2], bsrc), gd = tshfl(a_d, bsrc);
        if (lane < nc) { cbody = bsrc; cx[0] = gx; cx[1] = gy; cx[2] = gz; cdist = gd; }
        active_mask = 0u;
      }
      STAMP2(3);
      while (wave_any(active_mask != 0u)) {
#if TREX_STAMPS
        dbg_bodies++;
#endif
        const bool valid = active_mask != 0u;
        const int b = valid ? (__ffs(active_mask) - 1) : 0;
        active_mask &= active_mask - 1u;
        int sel[4] = {-1, -1, -1, -1};
        float px[4][3], pd[4];
        sel[0] = tshfl(a_v, b);
#pragma unroll
        for (int c = 0; c < 3; c++) px[0][c] = tshfl(a_x[c], b);
        pd[0] = tshfl(a_d, b);
        int nsel = valid ? 1 : 0;
        bool stop = !valid || K < 2;
        if (wave_any(!stop)) {
          float Rb[9], rb[3];
#pragma unroll
          for (int c = 0; c < 9; c++) Rb[c] = tshfl(R[c], b);
#pragma unroll
          for (int c = 0; c < 3; c++) rb[c] = tshfl(r[c], b);
          const int v0 = tshfl(hull_v0, b), v1 = tshfl(hull_v1, b);
          // this lane's candidates of body b: bit i <-> vertex v0 + lane + 32 i (in the margin during pass A)
          unsigned im;
          // (a team without an active body in this trip has stop = true: it must not sweep body 0's hull)
          const bool masked = stop || ((v1 - v0) <= 1024 && ((v1 - v0) <= SMALL_HULL || b == bL0 || b == bL1));
          if ((v1 - v0) <= SMALL_HULL) {
            const unsigned c0 = (unsigned)tshfl((int)cm0, b), c1 = (unsigned)tshfl((int)cm1, b), c2 = (unsigned)tshfl((int)cm2, b);
            im = ((c0 >> lane) & 1u) | (((c1 >> lane) & 1u) << 1) | (((c2 >> lane) & 1u) << 2);
          } else {
            im = (b == bL0) ? imL0 : ((b == bL1) ? imL1 : 0xffffffffu);
          }
          if (!masked) im = 0xffffffffu;   // (a third large hull, or one beyond 1024 vertices: sweep it)
          {
            const int o = sel[0] - v0;     // the deepest vertex is taken
            if (lane == (o & 31) && (o >> 5) < 32) im &= ~(1u << (o >> 5));
          }
#pragma unroll
          for (int pass = 1; pass < 4; pass++) {
            if (!wave_any(!stop)) break;
#if TREX_STAMPS
            dbg_passes++;
#endif
            float bs = -3.0e38f;
            int bi = 0x7fffffff;
            float ex = 0.f, ey = 0.f, flip = 1.f;
            if (pass >= 2) { ex = px[1][0] - px[0][0]; ey = px[1][1] - px[0][1]; }
            if (pass == 3) {
              const float c3 = ex * (px[2][1] - px[0][1]) - ey * (px[2][0] - px[0][0]);
              flip = c3 > 0.f ? -1.f : 1.f;
            }
            float bx[3] = {0.f, 0.f, 0.f};
            const int nit = (v1 - v0 - lane + TL - 1) / TL;   // strided vertices of this lane
            auto visit = [&](int v, const float4 h) {
              const float hv[3] = {h.x, h.y, h.z};
              float w[3];
              matvec3(Rb, hv, w);
              const float x0 = rb[0] + w[0], x1 = rb[1] + w[1], x2 = rb[2] + w[2] - h.w;
              const float dd = pos[2] + x2 - floor_z;
              if (!(dd < margin)) return;
              if (v == sel[0] || v == sel[1] || v == sel[2]) return;
              const float dx = x0 - px[0][0], dy = x1 - px[0][1];
              float score;
              if (pass == 1) score = dx * dx + dy * dy;
              else {
                const float cr = ex * dy - ey * dx;
                score = (pass == 2) ? fabsf(cr) : flip * cr;
              }
              if (score > bs) { bs = score; bi = v; bx[0] = x0; bx[1] = x1; bx[2] = x2; }
            };
            if (masked) {
              constexpr int UC = 4;   // candidates per trip: their loads are issued together
              for (unsigned m = stop ? 0u : im; wave_any(m != 0u);) {
#if TREX_STAMPS
                dbg_trips++;
#endif
                int vi[UC];
                float4 hc[UC];
#pragma unroll
                for (int u = 0; u < UC; u++) {
                  const int i = m != 0u ? (__ffs(m) - 1) : 32;
                  m &= m - 1u;            // (0 stays 0)
                  vi[u] = i < nit ? v0 + lane + TL * i : -1;
                  hc[u] = args.arr.hull[vi[u] >= 0 ? vi[u] : v0];
                }
#pragma unroll
                for (int u = 0; u < UC; u++)
                  if (vi[u] >= 0) visit(vi[u], hc[u]);
              }
            } else {
              for (int v = v0 + lane; !stop && v < v1; v += TL) visit(v, args.arr.hull[v]);
            }
            const int mine = bi;
            targmax(bs, bi);
            if (pass >= K || bi == 0x7fffffff || !(bs > 0.f)) stop = true;
            const int win = __ffs(tballot(mine == bi && bi != 0x7fffffff)) - 1;
            const float wx = tshfl(bx[0], win < 0 ? 0 : win), wy = tshfl(bx[1], win < 0 ? 0 : win), wz = tshfl(bx[2], win < 0 ? 0 : win);
            if (!stop) {
              sel[pass] = bi;
              px[pass][0] = wx; px[pass][1] = wy; px[pass][2] = wz;
              pd[pass] = pos[2] + wz - floor_z;
              nsel = pass + 1;
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (k < nsel && nc < maxc) {
            if (lane == nc) { cbody = b; cx[0] = px[k][0]; cx[1] = px[k][1]; cx[2] = px[k][2]; cdist = pd[k]; }
            nc++;
          }
        }
      }
    }

    STAMP2(4);
#if TREX_STAMPS
    wave_cg += __builtin_amdgcn_s_memtime() - cg_t0;
#endif
    STAMP(8);
    WPH(1);
#if TREX_PRIO_MODE == 2
    {   // wave-level priority for the rest of the substep: total contact rows of the two envs
      const int tot = __builtin_amdgcn_readlane(nc, 0) + __builtin_amdgcn_readlane(nc, 32);
      if (tot >= 12) __builtin_amdgcn_s_setprio(3);
      else if (tot >= 8) __builtin_amdgcn_s_setprio(2);
      else if (tot >= 4) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
#endif
    // ---- constraint solve: projected Gauss-Seidel in DELASSUS (residual) form, one env at a time on all
    // 64 lanes. Rows of an env: 25 motor rows (joint j, with the joint's limit row riding on the same lane)
    // and 3 rows per contact point (normal z, friction x, friction y). Each row s lives on ONE lane and keeps
    //     y_s = rhs_s - (J_s dv) / diag_s            (its unclamped Gauss-Seidel increment)
    // so a row visit is  nl = clamp(lam_r + y_r); d = nl - lam_r; lam_r = nl;  y_s += B_sr d  for all s, with
    // B_sr = -(J_s M^-1 J_r^T)/diag_s (B_rr = -1) - the same iteration as Bullet's dv form (and the oracle's),
    // but the row's impulse change reaches the other rows as ONE v_readlane (SGPR broadcast) + ONE packed fma
    // per lane instead of a 32-lane reduction per row: 6-7 instructions per env-row instead of 12 for a
    // contact row, and a light env no longer walks its wave partner's rows. y (not z = lam + y) is what is
    // accumulated: it is small where lam is large, and the rounding of a row's own update stays in y.
    // Rows -> lanes (the same for both envs of the wave; the per-lane inputs of env 1 are brought down
    // with v_permlane32_swap): motor row j on lane j (1..25); contact row k = 3c+a on lane 32+k for k < 32
    // and on the seven lanes left in the lower half (0, 26..31) for k = 32..38: 25 + 3 x 13 = 64 rows, one
    // per lane - which is where the budget of 13 contact points per env comes from.
    // B comes from the factorisation M^-1 = A^T B A: every row carries a descriptor (chain nodes ca[d],
    // entries u[d], base force r0; zc = u/D, z0 = I0^-1 r0) and
    //     J_s M^-1 J_r^T = r0_s . z0_r + sum_d [ca_s[d] == ca_r[d]] u_s[d] zc_r[d].
    // Motor columns go to LDS (read back one per motor row, and by the dynamic limit rows), contact columns
    // into registers (static index).
    float dv = 0.f, nimp = 0.f;
    const int tid = threadIdx.x;
    // motor rows of both envs: staged once, so that the factorisation's per-lane column data (Aanc, Z, A0,
    // g) is dead before the solves
    __syncthreads();
    {
      RowStage &S_ = W.u.rows;
      if (lane >= 1 && lane <= NJMAX) {
        const int row = team * NJMAX + lane - 1;
        const bool jn = is_joint;
        int ca[MAXD];
#pragma unroll
        for (int d = 0; d < MAXD; d++) ca[d] = jn ? anc[d] : -1;
        S_.col[row][0] = make_float4(__int_as_float(ca[0]), __int_as_float(ca[1]), __int_as_float(ca[2]), __int_as_float(ca[3]));
        S_.col[row][1] = make_float4(__int_as_float(ca[4]), __int_as_float(ca[5]), jn ? Z[0] : 0.f, jn ? Z[1] : 0.f);
        S_.col[row][2] = make_float4(jn ? Z[2] : 0.f, jn ? Z[3] : 0.f, jn ? Z[4] : 0.f, jn ? Z[5] : 0.f);
        S_.col[row][3] = make_float4(jn ? g[0] : 0.f, jn ? g[1] : 0.f, jn ? g[2] : 0.f, jn ? g[3] : 0.f);
        S_.col[row][4] = make_float4(jn ? g[4] : 0.f, jn ? g[5] : 0.f, 0.f, 0.f);
        float *o = S_.own[row];
#pragma unroll
        for (int d = 0; d < MAXD; d++) { o[d] = jn ? Aanc[d] : 0.f; o[6 + d] = jn ? A0[d] : 0.f; }
        o[12] = jn ? inv_mdiag : 0.f; o[13] = jn ? mot_rhs : 0.f;
      }
    }
    auto krow_lane = [](int k) { return k < 32 ? 32 + k : (k == 32 ? 0 : k - 7); };   // lane of contact row k
#pragma unroll 1
    for (int e_ = 0; e_ < 2; e_++) {
      int e = e_;
      asm volatile("" : "+s"(e));   // one copy of the solve in the instruction cache, not two
      // -- contact rows of each team's points: lane c owns point c and walks its body's chain for the
      //    three directions; per lane, no reductions. (Both teams walk; team e's result is staged.)
      const int ncE = __builtin_amdgcn_readlane(nc, 32 * e);
      const unsigned lmE = (unsigned)__builtin_amdgcn_readlane((int)lim_mask, 32 * e);
      const float muE = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mu), 32 * e));
      // per-lane inputs of the motor rows, on lanes 0..31 for either env
      auto pick = [&](float x) {
        const unsigned u = __float_as_uint(x);
        const auto r2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return __uint_as_float(e ? r2[1] : r2[0]);
      };
      const float mhi = pick(mot_hi), ldir = pick(lim_dir), lr = pick(lim_rhs - lim_dir * mot_rhs), mdg = pick(mdiag);
      // -- contact rows of env e: walk (both teams walk, team e stages) - skipped for an airborne env
      __syncthreads();
      if (ncE > 0) {
        // Every number of a row descriptor goes to the LDS stage the moment it is known (team e's point
        // lanes write, the others only walk): nothing but the three running forces p[a] is carried.
        const bool has = lane < nc;
        const bool st = (team == e) && lane < MAXC;
        int anc_w[MAXD];   // re-read (L2-resident model): not carried across contact generation and the other env's solve
        {
          const TrexDeviceModel *Mi = Mo();
#pragma unroll
          for (int d = 0; d < MAXD; d++) anc_w[d] = is_body ? Mi->anc[d][lane] : -1;
        }
        float *colf = reinterpret_cast<float *>(W.u.rows.col[CROW0 + 3 * (st ? lane : 0)]);   // 20 words per row
        float *ownf = W.u.rows.own[CROW0 + 3 * (st ? lane : 0)];                               // 15 words per row
        // updated body velocities (after the unconstrained step) for the row right-hand sides
        float nvel[6];
        body_velocities(nw, nv, nqd, nvel);
        float vb[6];
#pragma unroll
        for (int k = 0; k < 6; k++) vb[k] = tshfl(nvel[k], cbody);
        float p[3][6], diag[3] = {0.f, 0.f, 0.f};
        const float dirs[3][3] = {{0.f, 0.f, 1.f}, {1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}};
        float po[3], xrel[3];   // po: the point the forces p[a] refer to (body origin first, then up the chain)
#pragma unroll
        for (int k = 0; k < 3; k++) { po[k] = tshfl(r[k], cbody); xrel[k] = cx[k] - po[k]; }
#pragma unroll
        for (int a = 0; a < 3; a++) {
          float xd[3];
          cross3(xrel, dirs[a], xd);
#pragma unroll
          for (int k = 0; k < 3; k++) { p[a][k] = -xd[k]; p[a][3 + k] = -dirs[a][k]; }
        }
#pragma unroll
        for (int d = MAXD; d >= 1; d--) {
          int ca = -1;
          float ua[3] = {0.f, 0.f, 0.f}, zc[3] = {0.f, 0.f, 0.f};
          if (d <= maxdepth) {
            const int ab = tshfl(anc_w[d - 1], cbody);
            ca = has ? ab : -1;
            const int src = ab < 0 ? 0 : ab;
            float aa[3], Uda[6], ra[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { aa[k] = tshfl(S[k], src); ra[k] = tshfl(r[k], src); }
#pragma unroll
            for (int k = 0; k < 6; k++) Uda[k] = tshfl(Ud[k], src);
            const float invDa = tshfl(invD, src);
            if (has && ab >= 0) {
              float dd[3];
#pragma unroll
              for (int k = 0; k < 3; k++) { dd[k] = po[k] - ra[k]; po[k] = ra[k]; }
#pragma unroll
              for (int a = 0; a < 3; a++) {
                float dxf[3];
                cross3(dd, p[a] + 3, dxf);
#pragma unroll
                for (int k = 0; k < 3; k++) p[a][k] += dxf[k];
                ua[a] = -dot3(aa, p[a]);
                zc[a] = ua[a] * invDa;
                diag[a] += ua[a] * ua[a] * invDa;
#pragma unroll
                for (int k = 0; k < 6; k++) p[a][k] += Uda[k] * ua[a];
              }
            }
          }
          if (st) {
#pragma unroll
            for (int a = 0; a < 3; a++) {
              colf[20 * a + d - 1] = __int_as_float(ca);
              colf[20 * a + 6 + d - 1] = zc[a];
              ownf[15 * a + d - 1] = ua[a];
            }
          }
        }
        float pvel[3], wxx[3];
        cross3(vb, xrel, wxx);   // body velocity is about the body origin
        float I0l[21];
#pragma unroll
        for (int k = 0; k < 21; k++) I0l[k] = W.i0inv[team][k];
#pragma unroll
        for (int k = 0; k < 3; k++) pvel[k] = vb[3 + k] + wxx[k];
#pragma unroll
        for (int a = 0; a < 3; a++) {
          float rhs0[6], z0[6], dxf[3];
          cross3(po, p[a] + 3, dxf);   // on to the base origin O
#pragma unroll
          for (int k = 0; k < 3; k++) p[a][k] += dxf[k];
#pragma unroll
          for (int k = 0; k < 6; k++) rhs0[k] = -p[a][k];
          inv21_mul(I0l, rhs0, z0);
          diag[a] += dot6(rhs0, z0);
          const float inv = has ? 1.0f / diag[a] : 0.f;
          float tv = 0.f;
          if (a == 0) tv = (cdist > 0.f) ? -cdist * inv_dt : -cdist * cerp * inv_dt;
          const float rhs = (tv - dot3(dirs[a], pvel)) * inv;
          if (st) {
#pragma unroll
            for (int k = 0; k < 6; k++) { colf[20 * a + 12 + k] = has ? z0[k] : 0.f; ownf[15 * a + 6 + k] = has ? rhs0[k] : 0.f; }
            colf[20 * a + 18] = 0.f; colf[20 * a + 19] = 0.f;
            ownf[15 * a + 12] = inv; ownf[15 * a + 13] = has ? rhs : 0.f; ownf[15 * a + 14] = 0.f;
          }
        }
      } else if (team == e && lane < MAXC) {
        RowStage &S_ = W.u.rows;
        const float m1 = __int_as_float(-1);
#pragma unroll
        for (int a = 0; a < 3; a++) {
          const int row = CROW0 + 3 * lane + a;
          S_.col[row][0] = make_float4(m1, m1, m1, m1);
          S_.col[row][1] = make_float4(m1, m1, 0.f, 0.f);
          S_.col[row][2] = make_float4(0.f, 0.f, 0.f, 0.f);
          S_.col[row][3] = make_float4(0.f, 0.f, 0.f, 0.f);
          S_.col[row][4] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int k = 0; k < 15; k++) S_.own[row][k] = 0.f;
        }
      }
      STAMP(9);
      if (team == e) {
        RowStage &S_ = W.u.rows;
        if (lane == 31) {   // the null record
          const float m1 = __int_as_float(-1);
          S_.col[NREC - 1][0] = make_float4(m1, m1, m1, m1);
          S_.col[NREC - 1][1] = make_float4(m1, m1, 0.f, 0.f);
          S_.col[NREC - 1][2] = make_float4(0.f, 0.f, 0.f, 0.f);
          S_.col[NREC - 1][3] = make_float4(0.f, 0.f, 0.f, 0.f);
          S_.col[NREC - 1][4] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int k = 0; k < 15; k++) S_.own[NREC - 1][k] = 0.f;
        }
      }
      __syncthreads();
      // -- this lane's row
      int row0;
      if (tid >= 1 && tid <= NJMAX) row0 = e * NJMAX + tid - 1;
      else if (tid >= 32) row0 = CROW0 + (tid - 32);
      else row0 = CROW0 + 32 + (tid == 0 ? 0 : tid - 25);   // lanes 0, 26..31 -> contact rows 32..38
      int ca0[MAXD];
      float u0[MAXD], r00[6], zc0[MAXD], z00[6], inv0, y, lam = 0.f;
      {
        const float *c = reinterpret_cast<const float *>(W.u.rows.col[row0]);
        const float *o = W.u.rows.own[row0];
#pragma unroll
        for (int d = 0; d < MAXD; d++) { ca0[d] = __float_as_int(c[d]); zc0[d] = c[6 + d]; z00[d] = c[12 + d]; u0[d] = o[d]; r00[d] = o[6 + d]; }
        inv0 = o[12]; y = o[13];
      }
      // B entries of this lane's row against column r: the column's descriptor (ca, zc, z0: 18 words) sits in
      // the registers of the lane that owns row r and is broadcast with v_readlane into SGPRs - no LDS round
      // trip per column. Unused chain slots hold u = zc = 0, so a -1 == -1 match adds nothing.
      auto bcast_i = [&](int v, int L) { return __builtin_amdgcn_readlane(v, L); };
      auto bcast_f = [&](float v, int L) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), L)); };
      // -- motor columns -> registers (and LDS for the dynamic limit rows); column j is row lane j
      float Bm[NJMAX];
#pragma unroll
      for (int j = 1; j <= NJMAX; j++) {
        float a0_ = 0.f;
#pragma unroll
        for (int d = 0; d < 6; d++) a0_ = __builtin_fmaf(r00[d], bcast_f(z00[d], j), a0_);
#pragma unroll
        for (int d = 0; d < MAXD; d++) {
          const float m = (ca0[d] == bcast_i(ca0[d], j)) ? u0[d] : 0.f;
          a0_ = __builtin_fmaf(m, bcast_f(zc0[d], j), a0_);
        }
        Bm[j - 1] = -inv0 * a0_;
        W.jcol[j - 1][tid] = Bm[j - 1];
      }
      // -- contact columns -> registers; column k is row lane krow_lane(k), the chain is shared by a point's rows
      float Bc[3 * MAXC];
#pragma unroll
      for (int c = 0; c < MAXC; c++) {
#pragma unroll
        for (int a = 0; a < 3; a++) Bc[3 * c + a] = 0.f;
        if (c < ncE) {
          float m0[MAXD];
#pragma unroll
          for (int d = 0; d < MAXD; d++) m0[d] = (ca0[d] == bcast_i(ca0[d], krow_lane(3 * c))) ? u0[d] : 0.f;
#pragma unroll
          for (int a = 0; a < 3; a++) {
            const int k = 3 * c + a;
            const int L = krow_lane(k);
            float a0_ = 0.f;
#pragma unroll
            for (int d = 0; d < 6; d++) a0_ = __builtin_fmaf(r00[d], bcast_f(z00[d], L), a0_);
#pragma unroll
            for (int d = 0; d < MAXD; d++) a0_ = __builtin_fmaf(m0[d], bcast_f(zc0[d], L), a0_);
            Bc[k] = -inv0 * a0_;
          }
        }
      }
      __syncthreads();
      STAMP(10);
      WPH(2);
      // The launch lasts as long as its heaviest wave: let a wave with many rows win the issue arbitration
      // against its lighter SIMD partner.
#if TREX_PRIO_MODE == 1
      {
        const int groups = (ncE + 3) >> 2;
        if (groups >= 4) __builtin_amdgcn_s_setprio(3);
        else if (groups == 3) __builtin_amdgcn_s_setprio(2);
        else if (groups == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
#endif
      float lim_lam = 0.f;
#if TREX_STAMPS
      unsigned long long acc_joint = 0;
#endif
      constexpr int GP = 4;
      for (int it = 0; it < iters; it++) {
#if TREX_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
#endif
        int vs = tid;   // opaque once per sweep: `vs == j` is one v_cmp where used, not a spilled mask
        asm volatile("" : "+v"(vs));
        // limit rows (ascending joint order, as the oracle): the row of joint j rides on motor lane j,
        // whose y gives dv_j / diag = rhs - y
        for (unsigned m = lmE; m != 0u; m &= m - 1u) {
          const int j = __ffs(m) - 1;
          const float nl = fmaxf(lim_lam + (lr + ldir * y), 0.f);
          const float dl = (nl - lim_lam) * ldir;
          if (vs == j) lim_lam = nl;
          const float sd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dl), j));
          y += W.jcol[j - 1][tid] * sd;
        }
#pragma unroll
        for (int j = 1; j <= NJMAX; j++) {   // joints beyond nb are null rows (y = 0, bounds 0)
          const float nl = __builtin_amdgcn_fmed3f(lam + y, -mhi, mhi);
          const float d = nl - lam;
          const float sd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), j));
          if (vs == j) lam = nl;
          y += Bm[j - 1] * sd;
        }
#if TREX_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        acc_joint += ts1 - ts0;
#endif
#pragma unroll
        for (int g = 0; g < (MAXC + GP - 1) / GP; g++) {
          if (GP * g < ncE) {
#pragma unroll
            for (int cc = 0; cc < GP; cc++) {
              const int c = GP * g + cc;
              if (c >= MAXC) continue;
              float hi = 0.f;
#pragma unroll
              for (int a = 0; a < 3; a++) {
                const int k = 3 * c + a;
                const int L = krow_lane(k);
                float nl;
                if (a == 0) nl = fmaxf(lam + y, 0.f);
                else nl = __builtin_amdgcn_fmed3f(lam + y, -hi, hi);
                const float d = nl - lam;
                const float sd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), L));
                if (a == 0) hi = muE * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nl), L));
                if (vs == L) lam = nl;
                y += Bc[k] * sd;
              }
              __builtin_amdgcn_sched_barrier(0);   // bound live ranges: one point per window
            }
          }
        }
      }
#if TREX_PRIO_MODE == 1
      __builtin_amdgcn_s_setprio(0);
#endif
      STAMP(12);
      WPH(3);
#if TREX_STAMPS
      if (DEBUG && args.debug && blockIdx.x == 0 && threadIdx.x == 0) {
        args.debug[3000 + 16 * sub + 14] += (float)acc_joint;
        args.debug[3000 + 16 * sub + 15] = (float)lmE;
      }
#endif
      // -- results of env e. Joint lanes: dv_j / diag_j = -sum_r B_jr lam_r, summed afresh from the final
      // impulses (rhs_j - y_j holds the same number, but as a difference of large terms when the motor is
      // saturated). Base twist change = sum_r lam_r z0_r.
      const bool mrow = tid >= 1 && tid <= NJMAX;
      const float lt0 = lam + (mrow ? ldir * lim_lam : 0.f);   // motor + limit impulse of the joint
      float dvj = 0.f;
#pragma unroll
      for (int j = 1; j <= NJMAX; j++) {
        const float sl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lt0), j));
        dvj -= Bm[j - 1] * sl;
      }
#pragma unroll
      for (int g = 0; g < (MAXC + 3) / 4; g++) {
        if (4 * g < ncE) {
#pragma unroll
          for (int k = 12 * g; k < 12 * g + 12 && k < 3 * MAXC; k++) {
            const float sl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lam), krow_lane(k)));
            dvj -= Bc[k] * sl;
          }
        }
      }
      float dvb[6];
      {
        const float *c0 = reinterpret_cast<const float *>(W.u.rows.col[row0]);
#pragma unroll
        for (int k = 0; k < 6; k++) dvb[k] = wsum(lt0 * c0[12 + k]);
      }
      const bool nrm0 = !mrow && (row0 - CROW0) % 3 == 0;
      const float ni = wsum(nrm0 ? lam : 0.f);
      {
        // lanes 0..31 hold env e's joint results; every lane takes those of its own index within the team
        float dlo = dvj * mdg;
        {
          const unsigned u = __float_as_uint(dlo);
          const auto r2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
          dlo = __uint_as_float(r2[0]);
        }
        float mlo = lam;
        {
          const unsigned u = __float_as_uint(mlo);
          const auto r2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
          mlo = __uint_as_float(r2[0]);
        }
        if (team == e) {
          dv = is_joint ? dlo : 0.f;
#pragma unroll
          for (int k = 0; k < 6; k++)
            if (is_base_dof_s && bdof_s == k) dv = dvb[k];
          mot_lam = mlo;
          nimp = ni;
        }
      }
      if (DEBUG && args.debug && blockIdx.x == 0 && e == 0) {
        float *D = args.debug;
        // joint block of M^-1 recovered from the staged columns, contact impulses by row
        if (team == 0) {
#pragma unroll
          for (int j = 1; j <= NJMAX; j++) D[160 + 32 * (j - 1) + lane] = is_joint ? -W.jcol[j - 1][lane] * mdiag : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 3 * MAXC; k++) {
          const float l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lam), krow_lane(k)));
          if (threadIdx.x == 0 && k / 3 < ncE) D[960 + (k / 3) * 16 + 11 + k % 3] = l;
        }
      }
      __syncthreads();
    }

    if (DEBUG && args.debug && blockIdx.x == 0 && team == 0) {
      float *D = args.debug;
      D[lane] = qdd; D[64 + lane] = vg; D[96 + lane] = dv;
      if (lane < 6) D[32 + lane] = a0[lane];
      if (lane == 0) { D[128] = (float)nc; D[129] = (float)lim_mask; }
      if (lane < nc) {
        float *C = D + 960 + lane * 16;
        C[0] = (float)cbody; C[1] = cx[0]; C[2] = cx[1]; C[3] = cx[2]; C[4] = cdist;
      }
    }
    __syncthreads();

    // ---- commit velocities, integrate positions (only for a team that is really stepping)
    if (live) {
      vg += dv;
      qd = is_joint ? vg : 0.f;
#pragma unroll
      for (int k = 0; k < 3; k++) { bw[k] = tshfl(vg, nb + k); bv[k] = tshfl(vg, nb + 3 + k); }
      mtau = (is_joint && motors_on) ? mot_lam * inv_dt : 0.f;
      q += qd * dt;
#pragma unroll
      for (int k = 0; k < 3; k++) pos[k] += bv[k] * dt;
      const float wn = sqrtf(dot3(bw, bw)), th = wn * dt;
      float dq[4] = {0.f, 0.f, 0.f, 1.f};
      if (th > 1e-12f) {
        const float sh = sinf(0.5f * th) / wn;
        dq[0] = bw[0] * sh; dq[1] = bw[1] * sh; dq[2] = bw[2] * sh; dq[3] = cosf(0.5f * th);
      }
      float o[4];
      o[3] = dq[3] * quat[3] - dq[0] * quat[0] - dq[1] * quat[1] - dq[2] * quat[2];
      o[0] = dq[3] * quat[0] + dq[0] * quat[3] + dq[1] * quat[2] - dq[2] * quat[1];
      o[1] = dq[3] * quat[1] - dq[0] * quat[2] + dq[1] * quat[3] + dq[2] * quat[0];
      o[2] = dq[3] * quat[2] + dq[0] * quat[1] - dq[1] * quat[0] + dq[2] * quat[3];
      const float qn = 1.0f / sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
#pragma unroll
      for (int k = 0; k < 4; k++) quat[k] = o[k] * qn;
      stat_nc = nc;
      stat_imp = nimp;
    }
    STAMP(13);
    WPH(4);
  }

  // ---- head position (needs FK at the new pose: getLinkState(computeForwardKinematics=1))
  load_body_constants();
  forward_kinematics();
  float head[3];
  {
    const float hp[3] = {M->head_point[0], M->head_point[1], M->head_point[2]};
    float o[3];
    matvec3(R, hp, o);
#pragma unroll
    for (int k = 0; k < 3; k++) head[k] = tshfl(pos[k] + r[k] + o[k], M->head_body);
  }
  const float power = tsum(is_joint ? fabsf(qd * mtau) : 0.f);
  const float lift = args.w_distance * (2.5f - head[2]) * (2.5f - head[2]);
  const float drift = args.w_drift * (head[0] * head[0] + head[1] * head[1]);
  const float energy = args.w_energy * power;

#if TREX_STAMPS
  // per-wave duration and contact counts of its two envs (scripts/wave_balance.py; buffer >= 4 * 4096 floats)
  if (DEBUG && args.debug && blockIdx.x < 4096) {
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const int n1 = __shfl(stat_nc, 32);
    if (threadIdx.x == 0) {
      args.debug[4096 + blockIdx.x] = (float)(t_end - wave_t0);
      args.debug[8192 + blockIdx.x] = (float)(stat_nc + 100 * n1);
      args.debug[12288 + blockIdx.x] = (float)wave_cg;
      args.debug[16384 + blockIdx.x] = (float)wave_cg1;
      args.debug[20480 + blockIdx.x] = (float)wave_cg2;
      args.debug[24576 + blockIdx.x] = (float)(dbg_bodies + 1000 * dbg_passes + 1000000 * dbg_trips);
      for (int i = 0; i < 5; i++) args.debug[28672 + 4096 * i + blockIdx.x] = (float)wph[i];
    }
  }
#endif
  if (!env_ok) return;
  // ---- failure containment (no reference counterpart): an env whose state stopped being
  // finite is put back on the start pose with zero velocities and reports done = 1 once, with a finite
  // reward of 0; it never poisons its wave partner (every cross-lane exchange stays inside the team).
  bad |= !(fabsf(q) < 3.0e38f) || !(fabsf(qd) < 3.0e38f);
#pragma unroll
  for (int k = 0; k < 3; k++) bad |= !(fabsf(pos[k]) < 3.0e38f) || !(fabsf(bv[k]) < 3.0e38f) || !(fabsf(bw[k]) < 3.0e38f);
#pragma unroll
  for (int k = 0; k < 4; k++) bad |= !(fabsf(quat[k]) < 3.0e38f);
  const bool env_bad = tballot(bad) != 0u;
  if (env_bad) {
#pragma unroll
    for (int k = 0; k < 3; k++) { pos[k] = M->base_pos0[k]; bv[k] = 0.f; bw[k] = 0.f; }
#pragma unroll
    for (int k = 0; k < 4; k++) quat[k] = M->base_quat0[k];
    q = M->q_start[lane]; qd = 0.f; mtau = 0.f;
  }
  // ---- write back
  const bool store_state = RESET ? do_reset : true;
  if (store_state) {
    float *b = args.arr.base + env * 16;
    if (lane < 3) { b[lane] = pos[lane]; b[7 + lane] = bv[lane]; b[10 + lane] = bw[lane]; }
    if (lane < 4) b[3 + lane] = quat[lane];
    args.arr.q[env * TL + lane] = q;
    args.arr.qd[env * TL + lane] = qd;
    args.arr.tau[env * TL + lane] = mtau;
    if (lane == 0) {
      args.arr.motors_on[env] = motors_on ? 1 : 0;
      args.arr.contact_count[env] = stat_nc;
      args.arr.normal_impulse[env] = stat_imp;
    }
  }
  if (args.obs && is_joint) {
    float *o = args.obs + (size_t)env * args.obs_stride;
    const int obs_slot = M->obs_slot[lane];
    o[obs_slot] = q; o[nj + obs_slot] = qd; o[2 * nj + obs_slot] = mtau;
  }
  if (lane == 0) {
    if (args.reward) args.reward[(size_t)env * args.scal_stride] = env_bad ? 0.f : -lift - drift - energy;
    if (args.done) args.done[env] = env_bad ? 1 : 0;  // should_terminate() is constant False, trex_env.py:183-184
    if (args.done_f) args.done_f[(size_t)env * args.scal_stride] = env_bad ? 1.f : 0.f;
    if (args.penalties) {
      args.penalties[env * 3 + 0] = env_bad ? 0.f : lift; args.penalties[env * 3 + 1] = env_bad ? 0.f : drift;
      args.penalties[env * 3 + 2] = env_bad ? 0.f : energy;
    }
  }
}

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
__global__ __launch_bounds__(64) void trex_link_transforms_kernel(KernelArgs args, float *out) {
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
  const int L = args.arr.num_links;
  for (int l = t; l < L; l += blockDim.x) {
    const int body = args.arr.link_body[l];
    const float *tf = args.arr.link_tf + 12 * l;
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

// Wave pairing. A wave solves its two envs one after the other, so its time is the SUM of their row counts
// and the launch lasts as long as its heaviest wave: sort the envs by the contact count of their previous
// step (counting sort, 14 bins) and give the k-th lightest env the k-th heaviest as wave partner. The
// physics of an env does not depend on its partner (tests: permutation equivariance), so any order inside
// a bin is fine. One workgroup; N / 1024 trips per thread.
__global__ __launch_bounds__(1024) void trex_pair_kernel(const int32_t *contact_count, int32_t *perm, int n) {
  __shared__ int hist[16], start[16], fill[16];
  const int t = threadIdx.x;
  if (t < 16) { hist[t] = 0; fill[t] = 0; }
  __syncthreads();
  for (int i = t; i < n; i += 1024) {
    const int c = contact_count[i];
    atomicAdd(&hist[c < 0 ? 0 : (c > 15 ? 15 : c)], 1);
  }
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int b = 0; b < 16; b++) { start[b] = acc; acc += hist[b]; }
  }
  __syncthreads();
  const int half = (n + 1) / 2;
  for (int i = t; i < n; i += 1024) {
    const int c = contact_count[i];
    const int b = c < 0 ? 0 : (c > 15 ? 15 : c);
    const int pos = start[b] + atomicAdd(&fill[b], 1);   // rank by contact count
    perm[pos < half ? 2 * pos : 2 * (n - 1 - pos) + 1] = i;
  }
}

// ---------------------------------------------------------------- host launchers (called by capi.cpp)
extern "C" {

hipError_t trex_launch_step(const TrexDeviceModel *model, TrexBatchArrays arr, int n, const float *actions,
                            float *obs, float *reward, uint8_t *done, float *penalties, float wd, float we,
                            float wk, float *debug, hipStream_t stream, float *done_f, int obs_stride, int scal_stride) {
  // diagnostics launches keep env 0 and 1 in workgroup 0
#if TREX_STAMPS
  // diagnostic build: TREX_DEBUG_PAIR=1 keeps the pairing in debug launches (scripts/wave_balance.py)
  const bool dbg_pair = getenv("TREX_DEBUG_PAIR") != nullptr;
  const int32_t *perm = ((debug && !dbg_pair) || n < 4) ? nullptr : arr.pair_perm;
#else
  const int32_t *perm = (debug || n < 4) ? nullptr : arr.pair_perm;
#endif
  if (perm) hipLaunchKernelGGL(trex_pair_kernel, dim3(1), dim3(1024), 0, stream, arr.contact_count, arr.pair_perm, n);
  KernelArgs a{model, arr, n, actions, obs, reward, done, done_f, obs_stride, scal_stride, penalties, nullptr, perm, wd, we, wk, debug};
  if (debug) hipLaunchKernelGGL((trex_step_kernel<false, true>), dim3((n + 1) / 2), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL((trex_step_kernel<false, false>), dim3((n + 1) / 2), dim3(64), 0, stream, a);
  return hipGetLastError();
}

hipError_t trex_launch_reset(const TrexDeviceModel *model, TrexBatchArrays arr, int n, const uint8_t *mask,
                             float *obs, float wd, float we, float wk, float *debug, hipStream_t stream, int obs_stride) {
  KernelArgs a{model, arr, n, nullptr, obs, nullptr, nullptr, nullptr, obs_stride, 1, nullptr, mask, nullptr, wd, we, wk, debug};
  hipLaunchKernelGGL((trex_step_kernel<true, false>), dim3((n + 1) / 2), dim3(64), 0, stream, a);
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

hipError_t trex_launch_link_transforms(const TrexDeviceModel *model, TrexBatchArrays arr, int n, float *out, hipStream_t stream) {
  KernelArgs a{model, arr, n, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, nullptr};
  hipLaunchKernelGGL(trex_link_transforms_kernel, dim3(n), dim3(64), 0, stream, a, out);
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

int trex_step_lds_bytes(void) { return (int)sizeof(WaveLds); }

}  // extern "C"
