// C-ABI of include/trex_batch.h: model handle, batch handle, stream-ordered launches.
#include <hip/hip_runtime.h>

#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/trex_batch.h"
#include "device_model.h"
#include "internal.hpp"
#include "model.hpp"

extern "C" {
hipError_t trex_launch_step(const TrexDeviceModel *, TrexBatchArrays, int, const float *, float *, float *, uint8_t *,
                            float *, float, float, float, float *, hipStream_t, float *, int, int, int, int);
hipError_t trex_launch_reset(const TrexDeviceModel *, TrexBatchArrays, int, const uint8_t *, float *, float, float,
                             float, float *, hipStream_t, int, float *, float *, int, int, int);
hipError_t trex_launch_step_many(const TrexDeviceModel *, TrexBatchArrays, int, const float *, float *, int, int, float *, uint8_t *,
                                 float, float, float, hipStream_t, int, int, int);
hipError_t trex_launch_pack_state(const TrexDeviceModel *, TrexBatchArrays, int, float *, int, hipStream_t);
hipError_t trex_launch_head(const TrexDeviceModel *, TrexBatchArrays, int, float *, hipStream_t);
hipError_t trex_launch_link_transforms(const TrexDeviceModel *, TrexBatchArrays, int, float *, hipStream_t, int);
hipError_t trex_launch_fill(float *, float, int, hipStream_t);
hipError_t trex_launch_scalars_get(TrexBatchArrays, int, int32_t *, float *, int32_t *, hipStream_t);
hipError_t trex_launch_scalars_set(TrexBatchArrays, int, const int32_t *, int, int, hipStream_t);
hipError_t trex_launch_fill_u8(uint8_t *, uint8_t, int, hipStream_t);
hipError_t trex_launch_copy_mass_scale(const float *, float *, int, int, hipStream_t);
int trex_step_lds_bytes(int);
int trex_step_envs_per_workgroup(int);
}

struct TrexModel {
  trex::HostModel host;
};

struct TrexBatch {
  int n = 0, device = 0, nb = 0, nj = 0;
  TrexDeviceModel *dmodel = nullptr;
  TrexBatchArrays arr{};
  float wd = 1.0f, we = 0.005f, wk = 0.002f;  // trex_env.py:42-44
  bool pen_in_rows = false;                    // trex_batch_set_penalties_in_rows
  int balance_mode = -1;                       // trex_batch_set_wave_balance: -1 auto, 0 off, 1 on
  bool balance() const { return balance_mode < 0 ? n >= 2048 : balance_mode != 0; }
  std::vector<void *> allocs;
  // caller allocations already validated as memory of this device (base address, bytes known to be good):
  // the hot path pays one hash-free scan of a handful of entries, hipPointerGetAttributes only on a new one
  std::vector<TrexSeen> seen;
};

namespace {

thread_local std::string g_error;

int fail(int code, const std::string &msg) {
  g_error = msg;
  return code;
}
int hip_fail(hipError_t e, const char *what) {
  return fail(TREX_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) return hip_fail(_e, #expr);  \
  } while (0)

using DeviceGuard = TrexDeviceGuard;

void fill_device_model(const trex::HostModel &h, TrexDeviceModel &d) {
  std::memset(&d, 0, sizeof d);
  d.nb = h.nb;
  d.head_body = h.head_body;
  d.nv = (int)h.hull_xyz.size();
  const trex::Params &p = h.prm;
  const double prm[TP_COUNT] = {p.dt, p.substeps, p.iterations, p.gravity, p.motor_kp, p.motor_kd, p.motor_max_force,
                                p.floor_z, p.friction, p.erp, p.contact_erp, p.contact_margin, p.link_damping,
                                p.max_coordinate_velocity, p.max_contacts};
  for (int i = 0; i < TP_COUNT; i++) d.prm[i] = (float)prm[i];
  d.n_substeps = (int)p.substeps; d.n_iterations = (int)p.iterations; d.max_contacts = (int)p.max_contacts;
  d.inv_dt = 1.0f / (float)p.dt;
  d.motor_max_impulse = (float)p.motor_max_force * (float)p.dt;
  d.head_point[0] = (float)h.head_point.x; d.head_point[1] = (float)h.head_point.y; d.head_point[2] = (float)h.head_point.z;
  d.base_pos0[0] = (float)h.base_start_pos.x; d.base_pos0[1] = (float)h.base_start_pos.y; d.base_pos0[2] = (float)h.base_start_pos.z;
  for (int c = 0; c < 4; c++) d.base_quat0[c] = (float)h.base_start_quat[c];
  int maxdepth = 0;
  for (int l = 0; l < TREX_TL; l++) {
    d.parent[l] = -1; d.depth[l] = -1; d.obs_slot[l] = -1;
    for (int k = 0; k < TREX_MAXD; k++) d.anc[k][l] = -1;
    for (int k = 0; k < TREX_MAXCH; k++) d.child[k][l] = -1;
    d.mass[l] = 1.0f;
    d.jrot[0][l] = d.jrot[4][l] = d.jrot[8][l] = 1.0f;
    d.axis[2][l] = 1.0f;
    d.inertia[0][l] = d.inertia[3][l] = d.inertia[5][l] = 1.0f;
  }
  for (int b = 0; b < h.nb; b++) {
    d.parent[b] = h.parent[b];
    d.depth[b] = h.depth[b];
    maxdepth = std::max(maxdepth, h.depth[b]);
    for (int i = b; i > 0; i = h.parent[i]) d.anc[h.depth[i] - 1][b] = i;
    if (b > 0) {
      int p = h.parent[b];
      for (int k = 0; k < TREX_MAXCH; k++)
        if (d.child[k][p] < 0) { d.child[k][p] = b; break; }
    }
    const trex::Vec3 &a = h.joint_axis[b], &jp = h.joint_pos[b], &c = h.com[b], &sc = h.sphere_center[b];
    d.axis[0][b] = (float)a.x; d.axis[1][b] = (float)a.y; d.axis[2][b] = (float)a.z;
    d.jpos[0][b] = (float)jp.x; d.jpos[1][b] = (float)jp.y; d.jpos[2][b] = (float)jp.z;
    d.com[0][b] = (float)c.x; d.com[1][b] = (float)c.y; d.com[2][b] = (float)c.z;
    for (int k = 0; k < 9; k++) d.jrot[k][b] = (float)h.joint_rot[b].m[k];
    for (int k = 0; k < 6; k++) d.inertia[k][b] = (float)h.inertia[b][k];
    d.mass[b] = (float)h.mass[b];
    d.lower[b] = (float)h.q_lower[b]; d.upper[b] = (float)h.q_upper[b]; d.damp[b] = (float)h.joint_damping[b];
    d.q_start[b] = (float)h.q_start[b];
    d.sphere[0][b] = (float)sc.x; d.sphere[1][b] = (float)sc.y; d.sphere[2][b] = (float)sc.z;
    d.sphere[3][b] = (float)h.sphere_radius[b];
    // round the half extents up one ulp-ish: the bound must stay conservative in f32
    d.box_half[0][b] = (float)(h.box_half[b].x * (1 + 1e-6) + 1e-7); d.box_half[1][b] = (float)(h.box_half[b].y * (1 + 1e-6) + 1e-7);
    d.box_half[2][b] = (float)(h.box_half[b].z * (1 + 1e-6) + 1e-7);
    d.hull_start[b] = h.hull_start[b];
  }
  for (int b = h.nb; b <= TREX_TL; b++) d.hull_start[b] = h.hull_start[h.nb];
  {
    int off = 0;
    for (int b = 0; b < TREX_TL; b++) {
      const int nv = b < h.nb ? h.hull_start[b + 1] - h.hull_start[b] : 0;
      int lg = nv == 0 ? 0 : (nv <= 256 ? 3 : (nv <= 1024 ? 5 : 0));
      if (lg && off + (1 << lg) > TREX_CM_WORDS) lg = 0;          // no room left: this body is swept, not masked
      d.cm_pack[b] = (off << 8) | lg;
      if (lg) off += 1 << lg;
    }
  }
  {
    // scan units: the hull groups if they nest in the bodies' vertex ranges and are at most 32, else one per body
    std::vector<std::array<int, 3>> units;   // body, v0, v1
    bool ok = h.hull_group_start.size() >= 2;
    if (ok)
      for (size_t g = 0; g + 1 < h.hull_group_start.size() && ok; g++) {
        const int g0 = h.hull_group_start[g], g1 = h.hull_group_start[g + 1];
        if (g1 <= g0) continue;
        int body = -1;
        for (int b = 0; b < h.nb; b++)
          if (h.hull_start[b] <= g0 && g1 <= h.hull_start[b + 1]) body = b;
        if (body < 0) ok = false;
        else units.push_back({body, g0, g1});
      }
    if (!ok || units.size() > TREX_TL) {
      units.clear();
      for (int b = 0; b < h.nb; b++)
        if (h.hull_start[b + 1] > h.hull_start[b]) units.push_back({b, h.hull_start[b], h.hull_start[b + 1]});
    }
    d.nchunk = (int)units.size();
    for (int k = 0; k < TREX_TL; k++) { d.chunk_body[k] = 0; d.chunk_v0[k] = 0; d.chunk_v1[k] = 0; }
    for (size_t k = 0; k < units.size(); k++) {
      const int v0 = units[k][1], v1 = units[k][2];
      trex::Vec3 lo{1e300, 1e300, 1e300}, hi{-1e300, -1e300, -1e300};
      for (int v = v0; v < v1; v++) {
        const trex::Vec3 &p = h.hull_xyz[v];
        const double r = h.hull_radius[v];
        lo = {std::min(lo.x, p.x - r), std::min(lo.y, p.y - r), std::min(lo.z, p.z - r)};
        hi = {std::max(hi.x, p.x + r), std::max(hi.y, p.y + r), std::max(hi.z, p.z + r)};
      }
      d.chunk_body[k] = units[k][0]; d.chunk_v0[k] = v0; d.chunk_v1[k] = v1;
      d.chunk_c[0][k] = (float)(0.5 * (lo.x + hi.x)); d.chunk_c[1][k] = (float)(0.5 * (lo.y + hi.y)); d.chunk_c[2][k] = (float)(0.5 * (lo.z + hi.z));
      // half extents rounded up: the bound must stay conservative in f32 (centre rounding included)
      d.chunk_h[0][k] = (float)(0.5 * (hi.x - lo.x) * (1 + 1e-6) + 1e-6); d.chunk_h[1][k] = (float)(0.5 * (hi.y - lo.y) * (1 + 1e-6) + 1e-6);
      d.chunk_h[2][k] = (float)(0.5 * (hi.z - lo.z) * (1 + 1e-6) + 1e-6);
    }
  }
  d.maxdepth = maxdepth;
  for (size_t k = 0; k < h.obs_order.size(); k++) d.obs_slot[h.obs_order[k]] = (int)k;
  // dof lane l in chain of body b?  joint lanes: b is l or a descendant of l; base dof lanes: every body
  for (int l = 0; l < TREX_TL; l++) {
    unsigned m = 0;
    if (l >= 1 && l < h.nb) {
      for (int b = 0; b < h.nb; b++)
        for (int i = b; i > 0; i = h.parent[i])
          if (i == l) { m |= 1u << b; break; }
    } else if (l >= h.nb && l < h.nb + 6) {
      m = (h.nb >= 32) ? 0xffffffffu : ((1u << h.nb) - 1u);
    }
    d.desc_mask[l] = m;
  }
}

int check_batch(const TrexBatch *b) { return b ? TREX_OK : fail(TREX_E_INVALID, "null batch"); }

int check_device_buffer(TrexBatch *b, const void *p, size_t bytes, const char *what) {
  return trex_check_device_buffer(b->device, b->seen, p, bytes, what);
}
#define BUF_TRY(p, bytes, what)                                                  \
  do {                                                                           \
    if (int _c = check_device_buffer(b, (p), (size_t)(bytes), (what))) return _c; \
  } while (0)

}  // namespace

int trex_fail(int code, const std::string &msg) { return fail(code, msg); }

// A caller-owned buffer must be HIP device (or managed) memory of the given device and at least `bytes`
// long: a host pointer or a short buffer would make the kernel fault the GPU. NULL is accepted where the
// header says nullable (the caller checks non-nullable arguments first).
int trex_check_device_buffer(int device, std::vector<TrexSeen> &seen, const void *p, size_t bytes, const char *what) {
  if (!p) return TREX_OK;
  // validated ALLOCATIONS of this device: any pointer into one of them with enough room behind it passes without a
  // runtime query (a caller that walks through one large tensor - a [T, N, J] action pool - presents a new pointer
  // every step; hipPointerGetAttributes + hipMemGetAddressRange cost about 20 us)
  for (const auto &s : seen) {
    const char *lo = (const char *)s.p, *q = (const char *)p;
    if (q >= lo && q + bytes <= lo + s.bytes) return TREX_OK;
  }
  hipPointerAttribute_t at;
  std::memset(&at, 0, sizeof at);
  hipError_t e = hipPointerGetAttributes(&at, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();   // unregistered host memory reports an error: clear it
    return fail(TREX_E_INVALID, std::string(what) + ": not a HIP device pointer (host memory?)");
  }
  if (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged)
    return fail(TREX_E_INVALID, std::string(what) + ": pointer is not device memory");
  if (at.type == hipMemoryTypeDevice && at.device != device)
    return fail(TREX_E_INVALID, std::string(what) + ": buffer lives on device " + std::to_string(at.device) +
                                    ", the batch on device " + std::to_string(device));
  void *base = nullptr;
  size_t size = 0;
  const void *lo = p;
  size_t known = bytes;
  if (hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)p) == hipSuccess && base) {
    const size_t left = size - (size_t)((const char *)p - (const char *)base);
    if (left < bytes)
      return fail(TREX_E_INVALID, std::string(what) + ": buffer too small (" + std::to_string(left) + " bytes, need " +
                                      std::to_string(bytes) + ")");
    lo = base; known = size;   // the whole allocation is good
  } else {
    (void)hipGetLastError();
  }
  if (seen.size() >= 64) seen.erase(seen.begin());
  seen.push_back({lo, known});
  return TREX_OK;
}

extern "C" {

const char *trex_last_error(void) { return g_error.c_str(); }

#ifndef TREX_BUILD_ID
#define TREX_BUILD_ID "unknown"
#endif
const char *trex_build_id(void) { return TREX_BUILD_ID; }

int trex_model_load(const char *urdf_path, const char *collisions_dir, TrexModel **out) {
  if (!urdf_path || !out) return fail(TREX_E_INVALID, "trex_model_load: null argument");
  *out = nullptr;
  int code = TREX_OK;
  try {
    auto m = std::make_unique<TrexModel>();
    m->host = trex::load_model(urdf_path, collisions_dir, &code);
    *out = m.release();
    return TREX_OK;
  } catch (const std::exception &e) {
    return fail(code ? code : TREX_E_PARSE, e.what());
  }
}

void trex_model_destroy(TrexModel *m) { delete m; }

int trex_model_num_bodies(const TrexModel *m) { return m ? m->host.nb : fail(TREX_E_INVALID, "null model"); }
int trex_model_num_joints(const TrexModel *m) { return m ? m->host.nb - 1 : fail(TREX_E_INVALID, "null model"); }
int trex_model_num_urdf_joints(const TrexModel *m) { return m ? m->host.num_urdf_joints : fail(TREX_E_INVALID, "null model"); }
int trex_model_num_hull_vertices(const TrexModel *m) { return m ? (int)m->host.hull_xyz.size() : fail(TREX_E_INVALID, "null model"); }
double trex_model_total_mass(const TrexModel *m, int include_base_link) {
  if (!m) return -1.0;
  return include_base_link ? m->host.total_mass : m->host.total_mass_excluding_base;
}

int trex_model_joint_info(const TrexModel *m, int k, const char **name, int *urdf_joint_index, double *lower, double *upper) {
  if (!m) return fail(TREX_E_INVALID, "null model");
  if (k < 0 || k >= m->host.nb - 1) return fail(TREX_E_INVALID, "joint index out of range");
  int b = m->host.obs_order[k];
  if (name) *name = m->host.obs_joint_names[k].c_str();
  if (urdf_joint_index) *urdf_joint_index = m->host.revolute_joint_indices[k];
  if (lower) *lower = m->host.q_lower[b];
  if (upper) *upper = m->host.q_upper[b];
  return TREX_OK;
}

int trex_model_set_start_angle(TrexModel *m, const char *joint_name, double angle) {
  if (!m || !joint_name) return fail(TREX_E_INVALID, "null argument");
  std::string n = trex::rename_v0_name(joint_name);
  for (int i = 1; i < m->host.nb; i++)
    if (m->host.joint_names[i] == n) { m->host.q_start[i] = angle; return TREX_OK; }
  return fail(TREX_E_INVALID, std::string("unknown joint '") + joint_name + "'");
}

int trex_model_set_start_pose(TrexModel *m, const double xyz[3], const double rpy[3]) {
  if (!m || !xyz || !rpy) return fail(TREX_E_INVALID, "null argument");
  m->host.base_start_pos = {xyz[0], xyz[1], xyz[2]};
  trex::matrix_to_quat(trex::rpy_to_matrix(rpy[0], rpy[1], rpy[2]), m->host.base_start_quat);
  return TREX_OK;
}

int trex_model_use_primitive_collision(TrexModel *m, double max_radius, int max_divisions, int min_points) {
  if (!m) return fail(TREX_E_INVALID, "null model");
  if (!(max_radius > 0) || max_divisions < 0 || min_points < 1) return fail(TREX_E_INVALID, "bad primitive-fitting arguments");
  trex::use_primitive_collision(m->host, max_radius, max_divisions, min_points);
  return TREX_OK;
}

int trex_model_fit_hull_primitives(const TrexModel *m, int group, double max_radius, int max_divisions, int min_points,
                                   double *out, int capacity) {
  if (!m) return fail(TREX_E_INVALID, "null model");
  const auto &gs = m->host.hull_group_start;
  if (group < 0 || group + 1 >= (int)gs.size()) return fail(TREX_E_INVALID, "hull group out of range");
  if (!(max_radius > 0) || max_divisions < 0 || min_points < 1) return fail(TREX_E_INVALID, "bad primitive-fitting arguments");
  std::vector<trex::Vec3> pts(m->host.hull_xyz.begin() + gs[group], m->host.hull_xyz.begin() + gs[group + 1]);
  auto prims = trex::fit_primitives(pts, max_radius, max_divisions, min_points);
  if (out) {
    if (capacity < (int)prims.size()) return fail(TREX_E_INVALID, "capacity too small");
    for (size_t i = 0; i < prims.size(); i++) {
      double *o = out + 7 * i;
      o[0] = prims[i].p0.x; o[1] = prims[i].p0.y; o[2] = prims[i].p0.z;
      o[3] = prims[i].p1.x; o[4] = prims[i].p1.y; o[5] = prims[i].p1.z; o[6] = prims[i].radius;
    }
  }
  return (int)prims.size();
}

int trex_model_set_param(TrexModel *m, const char *name, double value) {
  if (!m || !name) return fail(TREX_E_INVALID, "null argument");
  double *p = m->host.prm.find(name);
  if (!p) return fail(TREX_E_INVALID, std::string("unknown parameter '") + name + "'");
  if (!std::isfinite(value)) return fail(TREX_E_INVALID, "parameter value is not finite");
  *p = value;
  return TREX_OK;
}
int trex_model_get_param(const TrexModel *m, const char *name, double *value) {
  if (!m || !name || !value) return fail(TREX_E_INVALID, "null argument");
  double *p = const_cast<TrexModel *>(m)->host.prm.find(name);
  if (!p) return fail(TREX_E_INVALID, std::string("unknown parameter '") + name + "'");
  *value = *p;
  return TREX_OK;
}

int trex_model_get_array(const TrexModel *m, const char *name, double *out, int capacity) {
  if (!m || !name) return fail(TREX_E_INVALID, "null argument");
  const trex::HostModel &h = m->host;
  std::vector<double> v;
  std::string n = name;
  auto push3 = [&](const std::vector<trex::Vec3> &a) { for (auto &p : a) { v.push_back(p.x); v.push_back(p.y); v.push_back(p.z); } };
  if (n == "parent") v.assign(h.parent.begin(), h.parent.end());
  else if (n == "depth") v.assign(h.depth.begin(), h.depth.end());
  else if (n == "joint_axis") push3(h.joint_axis);
  else if (n == "joint_pos") push3(h.joint_pos);
  else if (n == "joint_rot") { for (auto &r : h.joint_rot) v.insert(v.end(), r.m, r.m + 9); }
  else if (n == "q_lower") v = h.q_lower;
  else if (n == "q_upper") v = h.q_upper;
  else if (n == "joint_damping") v = h.joint_damping;
  else if (n == "mass") v = h.mass;
  else if (n == "com") push3(h.com);
  else if (n == "inertia") { for (auto &a : h.inertia) v.insert(v.end(), a.begin(), a.end()); }
  else if (n == "obs_order") v.assign(h.obs_order.begin(), h.obs_order.end());
  else if (n == "revolute_joint_indices") v.assign(h.revolute_joint_indices.begin(), h.revolute_joint_indices.end());
  else if (n == "head_body") v = {(double)h.head_body};
  else if (n == "link_body") v.assign(h.link_body.begin(), h.link_body.end());
  else if (n == "link_tf") { for (auto &t : h.link_tf) { v.insert(v.end(), t.R.m, t.R.m + 9); v.push_back(t.t.x); v.push_back(t.t.y); v.push_back(t.t.z); } }
  else if (n == "head_point") v = {h.head_point.x, h.head_point.y, h.head_point.z};
  else if (n == "hull_xyz") push3(h.hull_xyz);
  else if (n == "hull_start") v.assign(h.hull_start.begin(), h.hull_start.end());
  else if (n == "hull_radius") v = h.hull_radius;
  else if (n == "hull_group_start") v.assign(h.hull_group_start.begin(), h.hull_group_start.end());
  else if (n == "sphere_center") push3(h.sphere_center);
  else if (n == "sphere_radius") v = h.sphere_radius;
  else if (n == "q_start") v = h.q_start;
  else if (n == "base_start_pos") v = {h.base_start_pos.x, h.base_start_pos.y, h.base_start_pos.z};
  else if (n == "base_start_quat") v.assign(h.base_start_quat, h.base_start_quat + 4);
  else return fail(TREX_E_INVALID, "unknown array '" + n + "'");
  if (out) {
    if (capacity < (int)v.size()) return fail(TREX_E_INVALID, "capacity too small for '" + n + "'");
    std::memcpy(out, v.data(), v.size() * sizeof(double));
  }
  return (int)v.size();
}

// ------------------------------------------------------------------ batch
int trex_batch_create(const TrexModel *model, int num_envs, int device, TrexBatch **out) {
  if (!model || !out) return fail(TREX_E_INVALID, "trex_batch_create: null argument");
  *out = nullptr;
  if (num_envs <= 0) return fail(TREX_E_INVALID, "num_envs must be positive");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) return fail(TREX_E_HIP, "no HIP device available (the physics step has no CPU fallback)");
  if (device < 0 || device >= count) return fail(TREX_E_INVALID, "device index out of range");
  DeviceGuard guard(device);
  if (!guard.ok) return fail(TREX_E_HIP, "hipSetDevice failed");
  auto b = std::make_unique<TrexBatch>();
  b->n = num_envs; b->device = device; b->nb = model->host.nb; b->nj = b->nb - 1;
  auto alloc = [&](size_t bytes, void **p) -> hipError_t {
    hipError_t r = hipMalloc(p, bytes);
    if (r == hipSuccess) { b->allocs.push_back(*p); r = hipMemset(*p, 0, bytes); }
    return r;
  };
  auto cleanup = [&]() { for (void *p : b->allocs) (void)hipFree(p); };
  TrexDeviceModel dm;
  fill_device_model(model->host, dm);
  size_t n = (size_t)num_envs;
  hipError_t r = hipSuccess;
  auto A = [&](size_t bytes, void **p) { if (r == hipSuccess) r = alloc(bytes, p); };
  A(sizeof(TrexDeviceModel), (void **)&b->dmodel);
  A(n * 16 * sizeof(float), (void **)&b->arr.base);
  A(n * TREX_TL * sizeof(float), (void **)&b->arr.q);
  A(n * TREX_TL * sizeof(float), (void **)&b->arr.qd);
  A(n * TREX_TL * sizeof(float), (void **)&b->arr.mass_scale);
  A(n * sizeof(float), (void **)&b->arr.friction);
  A(TREX_BAL_WORDS(n) * sizeof(int32_t), (void **)&b->arr.balance);
  b->arr.max_episode_steps = 0;
  b->arr.domain = 0;
  size_t nv = model->host.hull_xyz.size();
  A((nv ? nv : 1) * sizeof(float4), (void **)&b->arr.hull);
  const size_t nl = model->host.link_names.size();
  int *link_body_dev = nullptr;
  float *link_tf_dev = nullptr;
  A(nl * sizeof(int), (void **)&link_body_dev);
  A(nl * 12 * sizeof(float), (void **)&link_tf_dev);
  const size_t nvis = model->host.visual_file.size();
  int *vis_body_dev = nullptr;
  float *vis_tf_dev = nullptr;
  A((nvis ? nvis : 1) * sizeof(int), (void **)&vis_body_dev);
  A((nvis ? nvis : 1) * 12 * sizeof(float), (void **)&vis_tf_dev);
  if (r != hipSuccess) { cleanup(); return hip_fail(r, "hipMalloc"); }
  {
    std::vector<int> vb(nvis);
    std::vector<float> vtf(nvis * 12);
    for (size_t v = 0; v < nvis; v++) {
      vb[v] = model->host.link_body[model->host.visual_link[v]];
      const trex::Tf &t = model->host.visual_body_tf[v];
      for (int k = 0; k < 9; k++) vtf[12 * v + k] = (float)t.R.m[k];
      vtf[12 * v + 9] = (float)t.t.x; vtf[12 * v + 10] = (float)t.t.y; vtf[12 * v + 11] = (float)t.t.z;
    }
    b->arr.num_visuals = (int)nvis; b->arr.visual_body = vis_body_dev; b->arr.visual_tf = vis_tf_dev;
    if (nvis) {
      r = hipMemcpy(vis_body_dev, vb.data(), nvis * sizeof(int), hipMemcpyHostToDevice);
      if (r == hipSuccess) r = hipMemcpy(vis_tf_dev, vtf.data(), vtf.size() * sizeof(float), hipMemcpyHostToDevice);
      if (r != hipSuccess) { cleanup(); return hip_fail(r, "hipMemcpy (visual table)"); }
    }
  }
  std::vector<float4> hull(nv ? nv : 1);
  for (size_t i = 0; i < nv; i++) hull[i] = make_float4((float)model->host.hull_xyz[i].x, (float)model->host.hull_xyz[i].y, (float)model->host.hull_xyz[i].z, (float)model->host.hull_radius[i]);
  std::vector<float> ltf(nl * 12);
  for (size_t l = 0; l < nl; l++) {
    for (int k = 0; k < 9; k++) ltf[12 * l + k] = (float)model->host.link_tf[l].R.m[k];
    ltf[12 * l + 9] = (float)model->host.link_tf[l].t.x; ltf[12 * l + 10] = (float)model->host.link_tf[l].t.y; ltf[12 * l + 11] = (float)model->host.link_tf[l].t.z;
  }
  b->arr.num_links = (int)nl; b->arr.link_body = link_body_dev; b->arr.link_tf = link_tf_dev;
  r = hipMemcpy(link_body_dev, model->host.link_body.data(), nl * sizeof(int), hipMemcpyHostToDevice);
  if (r == hipSuccess) r = hipMemcpy(link_tf_dev, ltf.data(), ltf.size() * sizeof(float), hipMemcpyHostToDevice);
  if (r == hipSuccess) r = hipMemcpy(b->dmodel, &dm, sizeof dm, hipMemcpyHostToDevice);
  if (r == hipSuccess) r = hipMemcpy(b->arr.hull, hull.data(), hull.size() * sizeof(float4), hipMemcpyHostToDevice);
  {   // wave balance: before the first step launch every env is filed under contact count 0, in env order
    std::vector<int32_t> bal(TREX_BAL_LISTS + n, 0);
    bal[TREX_BAL_COUNTS + 0] = (int32_t)n;
    for (size_t i = 0; i < n; i++) bal[TREX_BAL_LISTS + i] = (int32_t)i;
    if (r == hipSuccess) r = hipMemcpy(b->arr.balance, bal.data(), bal.size() * sizeof(int32_t), hipMemcpyHostToDevice);
  }
  if (r == hipSuccess) r = trex_launch_fill(b->arr.mass_scale, 1.0f, num_envs * TREX_TL, nullptr);
  if (r == hipSuccess) r = trex_launch_fill(b->arr.friction, (float)model->host.prm.friction, num_envs, nullptr);
  if (r == hipSuccess) r = hipDeviceSynchronize();
  if (r != hipSuccess) { cleanup(); return hip_fail(r, "batch initialisation"); }
  *out = b.release();
  return TREX_OK;
}

void trex_batch_destroy(TrexBatch *b) {
  if (!b) return;
  DeviceGuard guard(b->device);
  (void)hipDeviceSynchronize();
  for (void *p : b->allocs) (void)hipFree(p);
  delete b;
}

int trex_batch_num_envs(const TrexBatch *b) { return b ? b->n : fail(TREX_E_INVALID, "null batch"); }

int trex_batch_set_reward_weights(TrexBatch *b, float distance, float energy, float drift) {
  if (check_batch(b)) return TREX_E_INVALID;
  b->wd = distance; b->we = energy; b->wk = drift;
  return TREX_OK;
}

int trex_batch_set_wave_balance(TrexBatch *b, int mode) {
  if (check_batch(b)) return TREX_E_INVALID;
  if (mode < -1 || mode > 1) return fail(TREX_E_INVALID, "trex_batch_set_wave_balance: mode must be -1 (auto), 0 (off) or 1 (on)");
  b->balance_mode = mode;
  return TREX_OK;
}

int trex_batch_set_penalties_in_rows(TrexBatch *b, int enabled) {
  if (check_batch(b)) return TREX_E_INVALID;
  b->pen_in_rows = enabled != 0;
  return TREX_OK;
}

int trex_batch_forget_buffers(TrexBatch *b) {
  if (check_batch(b)) return TREX_E_INVALID;
  b->seen.clear();
  return TREX_OK;
}

int trex_batch_reset(TrexBatch *b, const uint8_t *mask_dev, float *obs_out_dev, void *stream) {
  if (check_batch(b)) return TREX_E_INVALID;
  DeviceGuard guard(b->device);
  const size_t n = (size_t)b->n;
  BUF_TRY(mask_dev, n, "trex_batch_reset: mask");
  BUF_TRY(obs_out_dev, n * 3 * b->nj * sizeof(float), "trex_batch_reset: obs_out");
  HIP_TRY(trex_launch_reset(b->dmodel, b->arr, b->n, mask_dev, obs_out_dev, b->wd, b->we, b->wk, nullptr,
                            (hipStream_t)stream, 3 * b->nj, nullptr, nullptr, 1, b->nj, 0));
  return TREX_OK;
}

int trex_batch_reset_rows(TrexBatch *b, const uint8_t *mask_dev, float *rows_dev, int row_stride, void *stream) {
  if (check_batch(b)) return TREX_E_INVALID;
  if (!rows_dev) return fail(TREX_E_INVALID, "trex_batch_reset_rows: rows is null");
  if (row_stride < 3 * b->nj + (b->pen_in_rows ? 5 : 2)) return fail(TREX_E_INVALID, "trex_batch_reset_rows: row_stride < 3J + 2 (3J + 5 with penalties in rows)");
  DeviceGuard guard(b->device);
  const size_t n = (size_t)b->n;
  BUF_TRY(mask_dev, n, "trex_batch_reset_rows: mask");
  BUF_TRY(rows_dev, ((n - 1) * row_stride + 3 * b->nj + (b->pen_in_rows ? 5 : 2)) * sizeof(float), "trex_batch_reset_rows: rows");
  HIP_TRY(trex_launch_reset(b->dmodel, b->arr, b->n, mask_dev, rows_dev, b->wd, b->we, b->wk, nullptr,
                            (hipStream_t)stream, row_stride, rows_dev + 3 * b->nj, rows_dev + 3 * b->nj + 1, row_stride, b->nj,
                            b->pen_in_rows ? 1 : 0));
  return TREX_OK;
}

int trex_batch_step(TrexBatch *b, const float *actions_dev, float *obs_dev, float *reward_dev, uint8_t *done_dev,
                    float *penalties_dev, void *stream) {
  if (check_batch(b)) return TREX_E_INVALID;
  if (!actions_dev) return fail(TREX_E_INVALID, "trex_batch_step: actions is null");
  DeviceGuard guard(b->device);
  const size_t n = (size_t)b->n;
  BUF_TRY(actions_dev, n * b->nj * sizeof(float), "trex_batch_step: actions");
  BUF_TRY(obs_dev, n * 3 * b->nj * sizeof(float), "trex_batch_step: obs");
  BUF_TRY(reward_dev, n * sizeof(float), "trex_batch_step: reward");
  BUF_TRY(done_dev, n, "trex_batch_step: done");
  BUF_TRY(penalties_dev, n * 3 * sizeof(float), "trex_batch_step: penalties");
  HIP_TRY(trex_launch_step(b->dmodel, b->arr, b->n, actions_dev, obs_dev, reward_dev, done_dev, penalties_dev, b->wd,
                           b->we, b->wk, nullptr, (hipStream_t)stream, nullptr, 3 * b->nj, 1, b->balance(), 0));
  return TREX_OK;
}

int trex_batch_step_rows(TrexBatch *b, const float *actions_dev, float *rows_dev, int row_stride, float *penalties_dev,
                         uint8_t *done_dev, void *stream) {
  if (check_batch(b)) return TREX_E_INVALID;
  if (!actions_dev || !rows_dev) return fail(TREX_E_INVALID, "trex_batch_step_rows: null argument");
  if (row_stride < 3 * b->nj + (b->pen_in_rows ? 5 : 2)) return fail(TREX_E_INVALID, "trex_batch_step_rows: row_stride < 3J + 2 (3J + 5 with penalties in rows)");
  DeviceGuard guard(b->device);
  const size_t n = (size_t)b->n;
  BUF_TRY(actions_dev, n * b->nj * sizeof(float), "trex_batch_step_rows: actions");
  BUF_TRY(rows_dev, ((n - 1) * row_stride + 3 * b->nj + (b->pen_in_rows ? 5 : 2)) * sizeof(float), "trex_batch_step_rows: rows");
  BUF_TRY(penalties_dev, n * 3 * sizeof(float), "trex_batch_step_rows: penalties");
  BUF_TRY(done_dev, n, "trex_batch_step_rows: done");
  float *rew = rows_dev + 3 * b->nj;
  HIP_TRY(trex_launch_step(b->dmodel, b->arr, b->n, actions_dev, rows_dev, rew, done_dev, penalties_dev, b->wd, b->we,
                           b->wk, nullptr, (hipStream_t)stream, rew + 1, row_stride, row_stride, b->balance(), b->pen_in_rows ? 1 : 0));
  return TREX_OK;
}

int trex_batch_step_many(TrexBatch *b, const float *actions_dev, float *rows_dev, int row_stride, int num_steps,
                         float *penalties_dev, uint8_t *done_dev, void *stream) {
  if (check_batch(b)) return TREX_E_INVALID;
  if (!actions_dev || !rows_dev) return fail(TREX_E_INVALID, "trex_batch_step_many: null argument");
  if (num_steps < 1) return fail(TREX_E_INVALID, "trex_batch_step_many: num_steps must be >= 1");
  if (row_stride < 3 * b->nj + (b->pen_in_rows ? 5 : 2)) return fail(TREX_E_INVALID, "trex_batch_step_many: row_stride < 3J + 2 (3J + 5 with penalties in rows)");
  DeviceGuard guard(b->device);
  const size_t n = (size_t)b->n, S = (size_t)num_steps;
  BUF_TRY(actions_dev, S * n * b->nj * sizeof(float), "trex_batch_step_many: actions");
  BUF_TRY(rows_dev, ((S * n - 1) * row_stride + 3 * b->nj + (b->pen_in_rows ? 5 : 2)) * sizeof(float), "trex_batch_step_many: rows");
  BUF_TRY(penalties_dev, S * n * 3 * sizeof(float), "trex_batch_step_many: penalties");
  BUF_TRY(done_dev, S * n, "trex_batch_step_many: done");
  HIP_TRY(trex_launch_step_many(b->dmodel, b->arr, b->n, actions_dev, rows_dev, row_stride, num_steps, penalties_dev, done_dev,
                                b->wd, b->we, b->wk, (hipStream_t)stream, b->balance(), b->nj, b->pen_in_rows ? 1 : 0));
  return TREX_OK;
}

int trex_batch_debug_step(TrexBatch *b, const float *actions_dev, float *obs_dev, float *debug_dev, void *stream) {
  if (check_batch(b)) return TREX_E_INVALID;
  if (!actions_dev || !debug_dev) return fail(TREX_E_INVALID, "trex_batch_debug_step: null argument");
  DeviceGuard guard(b->device);
  BUF_TRY(actions_dev, (size_t)b->n * b->nj * sizeof(float), "trex_batch_debug_step: actions");
  BUF_TRY(obs_dev, (size_t)b->n * 3 * b->nj * sizeof(float), "trex_batch_debug_step: obs");
  BUF_TRY(debug_dev, 4096 * sizeof(float), "trex_batch_debug_step: debug");   // (diagnostic builds: 4096 + 16 N)
  HIP_TRY(trex_launch_step(b->dmodel, b->arr, b->n, actions_dev, obs_dev, nullptr, nullptr, nullptr, b->wd, b->we, b->wk,
                           debug_dev, (hipStream_t)stream, nullptr, 3 * b->nj, 1, b->balance(), 0));
  return TREX_OK;
}

int trex_batch_set_episode_limit(TrexBatch *b, int max_episode_steps, const int32_t *episode_steps_dev, void *stream) {
  if (check_batch(b)) return TREX_E_INVALID;
  if (max_episode_steps < 0) return fail(TREX_E_INVALID, "max_episode_steps must be >= 0 (0 = no limit)");
  DeviceGuard guard(b->device);
  BUF_TRY(episode_steps_dev, (size_t)b->n * sizeof(int32_t), "trex_batch_set_episode_limit: episode_steps");
  b->arr.max_episode_steps = max_episode_steps;
  HIP_TRY(trex_launch_scalars_set(b->arr, b->n, episode_steps_dev, 1, -1, (hipStream_t)stream));
  return TREX_OK;
}
int trex_batch_get_episode_steps(TrexBatch *b, int32_t *episode_steps_dev, void *stream) {
  if (check_batch(b)) return TREX_E_INVALID;
  if (!episode_steps_dev) return fail(TREX_E_INVALID, "episode_steps is null");
  DeviceGuard guard(b->device);
  BUF_TRY(episode_steps_dev, (size_t)b->n * sizeof(int32_t), "trex_batch_get_episode_steps: episode_steps");
  HIP_TRY(trex_launch_scalars_get(b->arr, b->n, nullptr, nullptr, episode_steps_dev, (hipStream_t)stream));
  return TREX_OK;
}

int trex_batch_get_state(TrexBatch *b, float *state_dev, void *stream) {
  if (check_batch(b)) return TREX_E_INVALID;
  if (!state_dev) return fail(TREX_E_INVALID, "state is null");
  DeviceGuard guard(b->device);
  BUF_TRY(state_dev, (size_t)b->n * (13 + 2 * b->nj) * sizeof(float), "trex_batch_get_state: state");
  HIP_TRY(trex_launch_pack_state(b->dmodel, b->arr, b->n, state_dev, 1, (hipStream_t)stream));
  return TREX_OK;
}
int trex_batch_set_state(TrexBatch *b, const float *state_dev, void *stream) {
  if (check_batch(b)) return TREX_E_INVALID;
  if (!state_dev) return fail(TREX_E_INVALID, "state is null");
  DeviceGuard guard(b->device);
  BUF_TRY(state_dev, (size_t)b->n * (13 + 2 * b->nj) * sizeof(float), "trex_batch_set_state: state");
  HIP_TRY(trex_launch_pack_state(b->dmodel, b->arr, b->n, const_cast<float *>(state_dev), 0, (hipStream_t)stream));
  return TREX_OK;
}
int trex_batch_set_motors_enabled(TrexBatch *b, int enabled, void *stream) {
  if (check_batch(b)) return TREX_E_INVALID;
  DeviceGuard guard(b->device);
  HIP_TRY(trex_launch_scalars_set(b->arr, b->n, nullptr, 0, enabled ? 1 : 0, (hipStream_t)stream));
  return TREX_OK;
}

int trex_batch_head_position(TrexBatch *b, float *out_dev, void *stream) {
  if (check_batch(b)) return TREX_E_INVALID;
  if (!out_dev) return fail(TREX_E_INVALID, "out is null");
  DeviceGuard guard(b->device);
  BUF_TRY(out_dev, (size_t)b->n * 3 * sizeof(float), "trex_batch_head_position: out");
  HIP_TRY(trex_launch_head(b->dmodel, b->arr, b->n, out_dev, (hipStream_t)stream));
  return TREX_OK;
}

int trex_model_num_links(const TrexModel *m) { return m ? (int)m->host.link_names.size() : fail(TREX_E_INVALID, "null model"); }
int trex_model_link_info(const TrexModel *m, int link, const char **name, int *body) {
  if (!m) return fail(TREX_E_INVALID, "null model");
  if (link < 0 || link >= (int)m->host.link_names.size()) return fail(TREX_E_INVALID, "link index out of range");
  if (name) *name = m->host.link_names[link].c_str();
  if (body) *body = m->host.link_body[link];
  return TREX_OK;
}
int trex_batch_link_transforms(TrexBatch *b, float *out_dev, void *stream) {
  if (check_batch(b)) return TREX_E_INVALID;
  if (!out_dev) return fail(TREX_E_INVALID, "out is null");
  DeviceGuard guard(b->device);
  BUF_TRY(out_dev, (size_t)b->n * b->arr.num_links * 7 * sizeof(float), "trex_batch_link_transforms: out");
  HIP_TRY(trex_launch_link_transforms(b->dmodel, b->arr, b->n, out_dev, (hipStream_t)stream, 0));
  return TREX_OK;
}

int trex_model_num_visuals(const TrexModel *m) { return m ? (int)m->host.visual_file.size() : fail(TREX_E_INVALID, "null model"); }
int trex_model_visual_info(const TrexModel *m, int visual, const char **mesh_file, int *link, double xyz[3], double quat_xyzw[4]) {
  if (!m) return fail(TREX_E_INVALID, "null model");
  if (visual < 0 || visual >= (int)m->host.visual_file.size()) return fail(TREX_E_INVALID, "visual index out of range");
  if (mesh_file) *mesh_file = m->host.visual_file[visual].c_str();
  if (link) *link = m->host.visual_link[visual];
  const trex::Tf &t = m->host.visual_origin[visual];
  if (xyz) { xyz[0] = t.t.x; xyz[1] = t.t.y; xyz[2] = t.t.z; }
  if (quat_xyzw) trex::matrix_to_quat(t.R, quat_xyzw);
  return TREX_OK;
}
int trex_batch_visual_transforms(TrexBatch *b, float *out_dev, void *stream) {
  if (check_batch(b)) return TREX_E_INVALID;
  if (!out_dev) return fail(TREX_E_INVALID, "out is null");
  if (b->arr.num_visuals == 0) return fail(TREX_E_INVALID, "the model has no <visual> meshes");
  DeviceGuard guard(b->device);
  BUF_TRY(out_dev, (size_t)b->n * b->arr.num_visuals * 7 * sizeof(float), "trex_batch_visual_transforms: out");
  HIP_TRY(trex_launch_link_transforms(b->dmodel, b->arr, b->n, out_dev, (hipStream_t)stream, 1));
  return TREX_OK;
}

int trex_batch_set_domain(TrexBatch *b, const float *mass_scale_dev, const float *friction_dev, void *stream) {
  if (check_batch(b)) return TREX_E_INVALID;
  DeviceGuard guard(b->device);
  BUF_TRY(mass_scale_dev, (size_t)b->n * b->nb * sizeof(float), "trex_batch_set_domain: mass_scale");
  BUF_TRY(friction_dev, (size_t)b->n * sizeof(float), "trex_batch_set_domain: friction");
  if (mass_scale_dev) HIP_TRY(trex_launch_copy_mass_scale(mass_scale_dev, b->arr.mass_scale, b->n, b->nb, (hipStream_t)stream));
  if (friction_dev) HIP_TRY(hipMemcpyAsync(b->arr.friction, friction_dev, b->n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (mass_scale_dev || friction_dev) b->arr.domain = 1;   // from now on the step launches read the per-env arrays
  return TREX_OK;
}

int trex_batch_contact_stats(TrexBatch *b, int32_t *count_dev, float *normal_impulse_dev, void *stream) {
  if (check_batch(b)) return TREX_E_INVALID;
  DeviceGuard guard(b->device);
  BUF_TRY(count_dev, (size_t)b->n * sizeof(int32_t), "trex_batch_contact_stats: count");
  BUF_TRY(normal_impulse_dev, (size_t)b->n * sizeof(float), "trex_batch_contact_stats: normal_impulse");
  if (count_dev || normal_impulse_dev)
    HIP_TRY(trex_launch_scalars_get(b->arr, b->n, count_dev, normal_impulse_dev, nullptr, (hipStream_t)stream));
  return TREX_OK;
}

int trex_batch_launch_info(const TrexBatch *b, int *grid, int *block, int *lds_bytes, int *alg_bytes_per_env_step) {
  if (!b) return fail(TREX_E_INVALID, "null batch");
  const int epw = trex_step_envs_per_workgroup(b->n);
  if (grid) *grid = (b->n + epw - 1) / epw;
  if (block) *block = 64 * epw;       // one wavefront per env
  if (lds_bytes) *lds_bytes = trex_step_lds_bytes(b->n);
  // state in + out (13 + 2J floats each), action in (J), obs out (3J), reward (4 B), done (padded 4 B): SURVEY 8d
  if (alg_bytes_per_env_step) *alg_bytes_per_env_step = 4 * (2 * (13 + 2 * b->nj) + b->nj + 3 * b->nj + 1 + 1);
  return TREX_OK;
}

int trex_batch_time_steps(TrexBatch *b, const float *actions_dev, float *obs_dev, float *reward_dev, uint8_t *done_dev,
                          int steps, void *stream, float *avg_ms_out) {
  if (check_batch(b)) return TREX_E_INVALID;
  if (!actions_dev || !avg_ms_out || steps <= 0) return fail(TREX_E_INVALID, "trex_batch_time_steps: bad argument");
  DeviceGuard guard(b->device);
  BUF_TRY(actions_dev, (size_t)b->n * b->nj * sizeof(float), "trex_batch_time_steps: actions");
  BUF_TRY(obs_dev, (size_t)b->n * 3 * b->nj * sizeof(float), "trex_batch_time_steps: obs");
  BUF_TRY(reward_dev, (size_t)b->n * sizeof(float), "trex_batch_time_steps: reward");
  BUF_TRY(done_dev, (size_t)b->n, "trex_batch_time_steps: done");
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipEventRecord(e0, s));
  for (int i = 0; i < steps; i++)
    HIP_TRY(trex_launch_step(b->dmodel, b->arr, b->n, actions_dev, obs_dev, reward_dev, done_dev, nullptr, b->wd, b->we,
                             b->wk, nullptr, s, nullptr, 3 * b->nj, 1, b->balance(), 0));
  HIP_TRY(hipEventRecord(e1, s));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_ms_out = ms / steps;
  return TREX_OK;
}

}  // extern "C"
