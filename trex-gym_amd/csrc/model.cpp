// Host-side model compiler (see model.hpp). URDF conventions follow the reference's own parser:
// rpy = extrinsic x-y-z (tools/urdf_parsing.py:267-269), symmetric inertia (:282-296); mass is read
// from <mass value=...> (the reference parser's bug, SURVEY F8, is not reproduced).
#include "model.hpp"

#include <dirent.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <set>
#include <sstream>
#include <stdexcept>

#include "../../include/trex_batch.h"
#include "xml_min.hpp"

namespace trex {

namespace {

struct LoadError : std::runtime_error {
  int code;
  LoadError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
Vec3 operator*(double s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

Mat3 mul(const Mat3 &a, const Mat3 &b) {
  Mat3 o;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      o.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return o;
}
Mat3 transpose(const Mat3 &a) {
  Mat3 o;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) o.m[3 * i + j] = a.m[3 * j + i];
  return o;
}
Vec3 mul(const Mat3 &a, Vec3 v) {
  return {a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
          a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z};
}
Tf compose(const Tf &a, const Tf &b) { return {mul(a.R, b.R), mul(a.R, b.t) + a.t}; }
Tf inverse(const Tf &a) {
  Mat3 rt = transpose(a.R);
  return {rt, -1.0 * mul(rt, a.t)};
}
Vec3 apply(const Tf &a, Vec3 v) { return mul(a.R, v) + a.t; }

std::string read_file(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw LoadError(TREX_E_IO, "cannot open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

std::vector<double> parse_doubles(const std::string &s) {
  std::vector<double> out;
  const char *p = s.c_str();
  char *e = nullptr;
  for (;;) {
    double v = std::strtod(p, &e);
    if (e == p) break;
    // strtod reads "nan", "inf", "1e999": a non-finite vertex, mass, inertia, origin or limit would put every env of the
    // batch on the containment path - the file is refused instead
    if (!std::isfinite(v)) throw LoadError(TREX_E_PARSE, "non-finite number '" + std::string(p, (size_t)(e - p)) + "'");
    out.push_back(v);
    p = e;
  }
  return out;
}
Vec3 parse_vec3(const std::string &s, const char *what) {
  auto v = parse_doubles(s);
  if (v.size() < 3) throw LoadError(TREX_E_PARSE, std::string("expected 3 numbers in ") + what);
  return {v[0], v[1], v[2]};
}
double attr_double(const xmlmin::Node *n, const char *name, double dflt) {
  const std::string *a = n ? n->attr(name) : nullptr;
  if (!a) return dflt;
  char *e = nullptr;
  const double v = std::strtod(a->c_str(), &e);
  if (e == a->c_str()) throw LoadError(TREX_E_PARSE, std::string("attribute ") + name + "='" + *a + "' is not a number");
  if (!std::isfinite(v)) throw LoadError(TREX_E_PARSE, std::string("attribute ") + name + "='" + *a + "' is not finite");
  return v;
}
Tf parse_origin(const xmlmin::Node *parent) {
  Tf t;
  const xmlmin::Node *o = parent ? parent->child("origin") : nullptr;
  if (!o) return t;
  Vec3 rpy = parse_vec3(o->attr_or("rpy", "0 0 0"), "origin rpy");
  t.R = rpy_to_matrix(rpy.x, rpy.y, rpy.z);
  t.t = parse_vec3(o->attr_or("xyz", "0 0 0"), "origin xyz");
  return t;
}

struct UJoint {
  std::string name, type, parent, child;
  Tf origin;
  Vec3 axis{0, 0, 1};
  double lower = 0, upper = 0, damping = 0;
};
struct UMesh { std::string file; Tf origin; };
struct ULink {
  std::string name;
  double mass = 0;
  Mat3 inertia;
  Tf inertial_origin;
  std::vector<UMesh> visuals, collisions;
};

std::vector<Vec3> read_obj_vertices(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw LoadError(TREX_E_IO, "cannot open " + path);
  std::vector<Vec3> out;
  std::string line;
  while (std::getline(f, line)) {
    if (line.size() > 2 && line[0] == 'v' && line[1] == ' ') out.push_back(parse_vec3(line.substr(2), path.c_str()));
  }
  if (out.empty()) throw LoadError(TREX_E_PARSE, "no vertices in " + path);
  return out;
}

std::vector<Vec3> read_dae_vertices(const std::string &path) {
  auto root = xmlmin::parse(read_file(path));
  std::vector<const xmlmin::Node *> sources;
  root->find_all("source", sources);
  for (auto *s : sources) {
    const std::string *id = s->attr("id");
    if (!id || id->size() < 9 || id->compare(id->size() - 9, 9, "positions") != 0) continue;
    const xmlmin::Node *fa = s->child("float_array");
    if (!fa) break;
    auto v = parse_doubles(fa->text);
    std::vector<Vec3> out;
    for (size_t i = 0; i + 2 < v.size(); i += 3) out.push_back({v[i], v[i + 1], v[i + 2]});
    return out;
  }
  throw LoadError(TREX_E_PARSE, "no positions source in " + path);
}

std::string dirname_of(const std::string &p) {
  size_t s = p.find_last_of('/');
  return s == std::string::npos ? std::string(".") : p.substr(0, s);
}
std::string basename_noext(const std::string &p) {
  size_t s = p.find_last_of('/');
  std::string b = s == std::string::npos ? p : p.substr(s + 1);
  size_t d = b.find_last_of('.');
  return d == std::string::npos ? b : b.substr(0, d);
}
bool ends_with(const std::string &s, const std::string &suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}
// COL_tibia_L_convex_hull.dae -> tibia_left
std::string hull_mesh_name(const std::string &file) {
  std::string n = file.substr(4, file.size() - 4 - std::string("_convex_hull.dae").size());
  auto sub = [&](const std::string &from, const std::string &to) {
    for (size_t p = 0; (p = n.find(from, p)) != std::string::npos;) {
      size_t e = p + from.size();
      if (e == n.size() || n[e] == '_') { n.replace(p, from.size(), to); p += to.size(); }
      else p = e;
    }
  };
  sub("_L", "_left");
  sub("_R", "_right");
  return n;
}

}  // namespace

double *Params::find(const std::string &n) {
  static const std::map<std::string, double Params::*> tab = {
      {"dt", &Params::dt}, {"substeps", &Params::substeps}, {"iterations", &Params::iterations},
      {"gravity", &Params::gravity}, {"motor_kp", &Params::motor_kp}, {"motor_kd", &Params::motor_kd},
      {"motor_max_force", &Params::motor_max_force}, {"floor_z", &Params::floor_z},
      {"friction", &Params::friction}, {"erp", &Params::erp}, {"contact_erp", &Params::contact_erp},
      {"contact_margin", &Params::contact_margin}, {"link_damping", &Params::link_damping},
      {"max_coordinate_velocity", &Params::max_coordinate_velocity}, {"max_contacts", &Params::max_contacts}};
  auto it = tab.find(n);
  return it == tab.end() ? nullptr : &(this->*(it->second));
}

Mat3 rpy_to_matrix(double r, double p, double y) {
  double cr = std::cos(r), sr = std::sin(r), cp = std::cos(p), sp = std::sin(p), cy = std::cos(y), sy = std::sin(y);
  Mat3 rx{{1, 0, 0, 0, cr, -sr, 0, sr, cr}}, ry{{cp, 0, sp, 0, 1, 0, -sp, 0, cp}}, rz{{cy, -sy, 0, sy, cy, 0, 0, 0, 1}};
  return mul(mul(rz, ry), rx);
}

void matrix_to_quat(const Mat3 &a, double q[4]) {
  const double *m = a.m;
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    double s = std::sqrt(t + 1.0) * 2;
    q[0] = (m[7] - m[5]) / s; q[1] = (m[2] - m[6]) / s; q[2] = (m[3] - m[1]) / s; q[3] = 0.25 * s;
  } else {
    int i = (m[0] >= m[4] && m[0] >= m[8]) ? 0 : (m[4] >= m[8] ? 1 : 2);
    int j = (i + 1) % 3, k = (i + 2) % 3;
    double s = std::sqrt(1.0 + m[4 * i] - m[4 * j] - m[4 * k]) * 2;
    q[i] = 0.25 * s;
    q[j] = (m[3 * j + i] + m[3 * i + j]) / s;
    q[k] = (m[3 * k + i] + m[3 * i + k]) / s;
    q[3] = (m[3 * k + j] - m[3 * j + k]) / s;
  }
  if (q[3] < 0) for (int c = 0; c < 4; c++) q[c] = -q[c];
}

std::string rename_v0_name(const std::string &name) {
  for (const char *kind : {"joint", "link"}) {
    std::string suf = std::string("_") + kind;
    if (!ends_with(name, suf)) continue;
    std::string stem = name.substr(0, name.size() - suf.size());
    if (ends_with(stem, "_L")) stem = stem.substr(0, stem.size() - 2) + "_left";
    else if (ends_with(stem, "_R")) stem = stem.substr(0, stem.size() - 2) + "_right";
    return std::string(kind) + "_" + stem;
  }
  return name;
}

HostModel load_model(const std::string &urdf_path, const char *collisions_dir, int *code) {
  *code = TREX_OK;
  try {
    std::unique_ptr<xmlmin::Node> root;
    try {
      root = xmlmin::parse(read_file(urdf_path));
    } catch (const LoadError &) {
      throw;
    } catch (const std::exception &e) {
      throw LoadError(TREX_E_PARSE, urdf_path + ": " + e.what());
    }
    if (root->tag != "robot") throw LoadError(TREX_E_PARSE, urdf_path + ": root element is not <robot>");

    std::vector<UJoint> joints;
    std::vector<ULink> links;
    for (auto *j : root->all("joint")) {
      UJoint u;
      u.name = j->attr_or("name", "");
      u.type = j->attr_or("type", "fixed");
      const xmlmin::Node *p = j->child("parent"), *c = j->child("child");
      if (!p || !c) throw LoadError(TREX_E_PARSE, "joint " + u.name + " lacks parent/child");
      u.parent = p->attr_or("link", "");
      u.child = c->attr_or("link", "");
      u.origin = parse_origin(j);
      if (const xmlmin::Node *ax = j->child("axis")) u.axis = parse_vec3(ax->attr_or("xyz", "0 0 1"), "axis");
      u.lower = attr_double(j->child("limit"), "lower", 0);
      u.upper = attr_double(j->child("limit"), "upper", 0);
      u.damping = attr_double(j->child("dynamics"), "damping", 0);
      if (u.type != "fixed" && u.type != "revolute")
        throw LoadError(TREX_E_UNSUPPORTED, "joint " + u.name + ": type '" + u.type + "' not supported");
      joints.push_back(u);
    }
    for (auto *l : root->all("link")) {
      ULink u;
      u.name = l->attr_or("name", "");
      u.inertia = Mat3{{0, 0, 0, 0, 0, 0, 0, 0, 0}};
      if (const xmlmin::Node *in = l->child("inertial")) {
        u.inertial_origin = parse_origin(in);
        u.mass = attr_double(in->child("mass"), "value", 0);
        const xmlmin::Node *i = in->child("inertia");
        double xx = attr_double(i, "ixx", 0), xy = attr_double(i, "ixy", 0), xz = attr_double(i, "ixz", 0);
        double yy = attr_double(i, "iyy", 0), yz = attr_double(i, "iyz", 0), zz = attr_double(i, "izz", 0);
        u.inertia = Mat3{{xx, xy, xz, xy, yy, yz, xz, yz, zz}};
      }
      for (auto *v : l->all("visual"))
        if (const xmlmin::Node *g = v->child("geometry"))
          if (const xmlmin::Node *m = g->child("mesh")) {
            // (the origin first: if parse_origin throws INSIDE the braced initialiser, g++ 11 leaks the string member
            // that was already built - found by tests/test_host_sanitize.py)
            const Tf o = parse_origin(v);
            u.visuals.push_back({m->attr_or("filename", ""), o});
          }
      for (auto *v : l->all("collision"))
        if (const xmlmin::Node *g = v->child("geometry"))
          if (const xmlmin::Node *m = g->child("mesh")) {
            const Tf o = parse_origin(v);
            u.collisions.push_back({m->attr_or("filename", ""), o});
          }
      links.push_back(u);
    }
    if (links.empty()) throw LoadError(TREX_E_PARSE, urdf_path + ": no links");

    std::map<std::string, int> link_index;
    for (size_t i = 0; i < links.size(); i++) link_index[links[i].name] = (int)i;
    std::map<std::string, std::vector<int>> children;  // parent link -> joint indices, document order
    std::map<std::string, bool> is_child;
    for (size_t k = 0; k < joints.size(); k++) {
      if (!link_index.count(joints[k].parent) || !link_index.count(joints[k].child))
        throw LoadError(TREX_E_PARSE, "joint " + joints[k].name + " references an unknown link");
      children[joints[k].parent].push_back((int)k);
      is_child[joints[k].child] = true;
    }
    std::string root_name;
    for (auto &l : links)
      if (!is_child.count(l.name)) {
        if (!root_name.empty()) throw LoadError(TREX_E_UNSUPPORTED, "more than one root link");
        root_name = l.name;
      }
    if (root_name.empty()) throw LoadError(TREX_E_PARSE, "no root link (cycle)");

    // hull vertices per link, in link frame
    std::map<std::string, std::vector<std::vector<Vec3>>> hull_groups;   // link -> one vertex list per hull
    bool have_collision = false;
    for (auto &l : links) have_collision |= !l.collisions.empty();
    std::string base_dir = dirname_of(urdf_path);
    if (have_collision) {
      for (auto &l : links)
        for (auto &c : l.collisions) {
          auto v = read_obj_vertices(base_dir + "/" + c.file);
          for (auto &p : v) p = apply(c.origin, p);
          hull_groups[l.name].push_back(v);
        }
    } else if (collisions_dir) {
      std::map<std::string, std::pair<std::string, Tf>> vis;
      for (auto &l : links)
        for (auto &v : l.visuals) vis[basename_noext(v.file)] = {l.name, v.origin};
      std::vector<std::string> files;
      if (DIR *d = opendir(collisions_dir)) {
        while (dirent *e = readdir(d))
          if (ends_with(e->d_name, "_convex_hull.dae") && std::string(e->d_name).compare(0, 4, "COL_") == 0)
            files.push_back(e->d_name);
        closedir(d);
      } else {
        throw LoadError(TREX_E_IO, std::string("cannot open directory ") + collisions_dir);
      }
      std::sort(files.begin(), files.end());
      for (auto &f : files) {
        auto it = vis.find(hull_mesh_name(f));
        if (it == vis.end()) throw LoadError(TREX_E_PARSE, "no visual mesh matches hull " + f);
        auto v = read_dae_vertices(std::string(collisions_dir) + "/" + f);
        // COL_ilium_L is point-reflected relative to ilium_left.obj (scripts/make_assets.py)
        double sgn = (f == "COL_ilium_L_convex_hull.dae") ? -1.0 : 1.0;
        for (auto &p : v) p = apply(it->second.second, sgn * p);
        hull_groups[it->second.first].push_back(v);
      }
    }

    // ---- merge fixed joints: bodies in depth-first document order (parent index < child index)
    struct Body { std::string head; int parent; int joint; Tf tf_parent; std::vector<std::pair<int, Tf>> members; };
    std::vector<Body> bodies;
    std::set<std::string> visited;
    std::function<void(const std::string &, int, int, const Tf &, const Tf &)> build =
        [&](const std::string &head, int parent, int joint, const Tf &tf_head, const Tf &tf_parent) {
          int idx = (int)bodies.size();
          // (refused HERE, not after the walk: `build` recurses per moving body, and a caller-named file may chain any number)
          if (idx >= kMaxBodies)
            throw LoadError(TREX_E_UNSUPPORTED, "model has more than " + std::to_string(kMaxBodies) + " moving bodies, kernels support " + std::to_string(kMaxBodies));
          bodies.push_back({head, parent, joint, tf_parent, {}});
          std::vector<std::pair<int, Tf>> pending;  // revolute joints leaving this body
          std::vector<std::pair<std::string, Tf>> queue{{head, tf_head}};
          for (size_t qi = 0; qi < queue.size(); qi++) {
            auto [ln, tf] = queue[qi];
            // a link reached twice has two parent joints or lies on a cycle below the root: the walk would never end
            if (!visited.insert(ln).second) throw LoadError(TREX_E_PARSE, "link " + ln + " is the child of two joints (or lies on a cycle)");
            bodies[idx].members.push_back({link_index[ln], tf});
            for (int k : children[ln]) {
              if (joints[k].type == "fixed") queue.push_back({joints[k].child, compose(tf, joints[k].origin)});
              else pending.push_back({k, tf});
            }
          }
          std::sort(pending.begin(), pending.end(), [](auto &a, auto &b) { return a.first < b.first; });
          for (auto &[k, tf] : pending) build(joints[k].child, idx, k, Tf{}, compose(tf, joints[k].origin));
        };
    build(root_name, -1, -1, inverse(links[link_index[root_name]].inertial_origin), Tf{});

    HostModel m;
    m.nb = (int)bodies.size();
    m.num_urdf_joints = (int)joints.size();
    if (m.nb > kMaxBodies)
      throw LoadError(TREX_E_UNSUPPORTED, "model has " + std::to_string(m.nb) + " moving bodies, kernels support " + std::to_string(kMaxBodies));
    int nb = m.nb;
    m.parent.resize(nb); m.depth.assign(nb, 0);
    m.joint_axis.resize(nb); m.joint_pos.resize(nb); m.joint_rot.resize(nb); m.com.resize(nb);
    m.q_lower.assign(nb, 0); m.q_upper.assign(nb, 0); m.joint_damping.assign(nb, 0); m.mass.assign(nb, 0);
    m.inertia.resize(nb); m.sphere_center.resize(nb); m.box_half.resize(nb); m.sphere_radius.assign(nb, 0); m.q_start.assign(nb, 0);
    m.hull_start.push_back(0);
    std::vector<int> n_children(nb, 0);
    for (int i = 0; i < nb; i++) {
      const Body &b = bodies[i];
      m.body_names.push_back(b.head);
      m.joint_names.push_back(b.joint >= 0 ? joints[b.joint].name : "");
      m.parent[i] = b.parent;
      if (i > 0) {
        m.depth[i] = m.depth[b.parent] + 1;
        if (m.depth[i] > kMaxDepth) throw LoadError(TREX_E_UNSUPPORTED, "tree deeper than " + std::to_string(kMaxDepth));
        if (++n_children[b.parent] > kMaxChildren)
          throw LoadError(TREX_E_UNSUPPORTED, "body " + bodies[b.parent].head + " has more than " + std::to_string(kMaxChildren) + " moving children");
        const UJoint &j = joints[b.joint];
        double n = std::sqrt(dot(j.axis, j.axis));
        if (n == 0) throw LoadError(TREX_E_PARSE, "joint " + j.name + " has a zero axis");
        m.joint_axis[i] = (1.0 / n) * j.axis;
        m.joint_pos[i] = b.tf_parent.t;
        m.joint_rot[i] = b.tf_parent.R;
        m.q_lower[i] = j.lower; m.q_upper[i] = j.upper; m.joint_damping[i] = j.damping;
      }
      double mass = 0;
      Vec3 mc;
      struct Part { double m; Vec3 c; Mat3 I; };
      std::vector<Part> parts;
      for (auto &[li, tf] : b.members) {
        const ULink &l = links[li];
        Vec3 c = apply(tf, l.inertial_origin.t);
        Mat3 R = mul(tf.R, l.inertial_origin.R);
        parts.push_back({l.mass, c, mul(mul(R, l.inertia), transpose(R))});
        mass += l.mass;
        mc = mc + l.mass * c;
        if (l.name == "link_atlas_axis") { m.head_body = i; m.head_point = c; }  // trex_robot.py:316
      }
      if (!(mass > 0)) throw LoadError(TREX_E_UNSUPPORTED, "body " + b.head + " has no mass");
      Vec3 com = (1.0 / mass) * mc;
      double I[9] = {0};
      for (auto &p : parts) {
        Vec3 d = p.c - com;
        double dd = dot(d, d), dv[3] = {d.x, d.y, d.z};
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) I[3 * r + c] += p.I.m[3 * r + c] + p.m * ((r == c ? dd : 0) - dv[r] * dv[c]);
      }
      m.mass[i] = mass; m.com[i] = com;
      m.inertia[i] = {I[0], I[1], I[2], I[4], I[5], I[8]};
      size_t before = m.hull_xyz.size();
      if (m.hull_group_start.empty()) m.hull_group_start.push_back(0);
      for (auto &[li, tf] : b.members) {
        auto it = hull_groups.find(links[li].name);
        if (it == hull_groups.end()) continue;
        for (auto &group : it->second) {
          for (auto &p : group) m.hull_xyz.push_back(apply(tf, p));
          m.hull_group_start.push_back((int)m.hull_xyz.size());
        }
      }
      if (m.hull_xyz.size() > before) {
        Vec3 lo = m.hull_xyz[before], hi = lo;
        for (size_t v = before; v < m.hull_xyz.size(); v++) {
          const Vec3 &p = m.hull_xyz[v];
          lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)};
          hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)};
        }
        Vec3 c = 0.5 * (lo + hi);
        double r2 = 0;
        for (size_t v = before; v < m.hull_xyz.size(); v++) r2 = std::max(r2, dot(m.hull_xyz[v] - c, m.hull_xyz[v] - c));
        m.sphere_center[i] = c; m.sphere_radius[i] = std::sqrt(r2);
        m.box_half[i] = 0.5 * (hi - lo);
      }
      m.hull_start.push_back((int)m.hull_xyz.size());
    }
    if (m.hull_group_start.empty()) m.hull_group_start.push_back(0);
    m.hull_radius.assign(m.hull_xyz.size(), 0.0);
    if (m.head_body < 0) { m.head_body = 0; m.head_point = m.com[0]; }  // generic URDF: reward point = base COM
    m.link_names.resize(links.size()); m.link_body.assign(links.size(), 0); m.link_tf.resize(links.size());
    for (size_t l = 0; l < links.size(); l++) m.link_names[l] = links[l].name;
    for (int i = 0; i < nb; i++)
      for (auto &[li, tf] : bodies[i].members) { m.link_body[li] = i; m.link_tf[li] = tf; }
    for (size_t l = 0; l < links.size(); l++)
      for (auto &v : links[l].visuals) {
        m.visual_file.push_back(v.file); m.visual_link.push_back((int)l);
        m.visual_origin.push_back(v.origin); m.visual_body_tf.push_back(compose(m.link_tf[l], v.origin));
      }

    // observation order = revolute joint names sorted (trex_robot.py:311-314)
    std::vector<std::pair<std::string, int>> named;
    for (int i = 1; i < nb; i++) named.push_back({m.joint_names[i], i});
    std::sort(named.begin(), named.end());
    std::map<std::string, int> doc_index;
    for (size_t k = 0; k < joints.size(); k++) doc_index[joints[k].name] = (int)k;
    for (auto &[n, b] : named) {
      m.obs_joint_names.push_back(n);
      m.obs_order.push_back(b);
      m.revolute_joint_indices.push_back(doc_index[n]);
    }
    for (auto &l : links) {
      m.total_mass += l.mass;
      if (l.name != root_name) m.total_mass_excluding_base += l.mass;
    }
    // default start configuration (trex_env.py:81-87 after the F2 rename), where those joints exist
    static const std::pair<const char *, double> start[] = {
        {"joint_femur_left", -0.6}, {"joint_tibia_left", 0.4}, {"joint_tarsometatarsus_left", -1.2},
        {"joint_femur_right", -0.6}, {"joint_tibia_right", 0.4}, {"joint_tarsometatarsus_right", -1.2}};
    for (auto &[n, v] : start)
      for (int i = 1; i < nb; i++)
        if (m.joint_names[i] == n) m.q_start[i] = v;
    return m;
  } catch (const LoadError &e) {
    *code = e.code;
    throw std::runtime_error(e.what());
  } catch (const std::exception &e) {
    *code = TREX_E_PARSE;
    throw std::runtime_error(std::string("model load failed: ") + e.what());
  }
}

// ------------------------------------------------------------------ collision primitives
namespace {

// eigenvectors of a symmetric 3x3 (cyclic Jacobi), eigenvalues descending
void eigen_sym3(const double a_in[9], double vec[9], double val[3]) {
  double a[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 9; i++) a[i] = a_in[i];
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (std::fabs(a[3 * p + q]) < 1e-300) continue;
        double theta = (a[3 * q + q] - a[3 * p + p]) / (2 * a[3 * p + q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
        double c = 1 / std::sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < 3; k++) {   // A <- A J
          double akp = a[3 * k + p], akq = a[3 * k + q];
          a[3 * k + p] = c * akp - s * akq; a[3 * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {   // A <- J^T A
          double apk = a[3 * p + k], aqk = a[3 * q + k];
          a[3 * p + k] = c * apk - s * aqk; a[3 * q + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          double vkp = v[3 * k + p], vkq = v[3 * k + q];
          v[3 * k + p] = c * vkp - s * vkq; v[3 * k + q] = s * vkp + c * vkq;
        }
      }
  }
  int idx[3] = {0, 1, 2};
  std::sort(idx, idx + 3, [&](int i, int j) { return a[4 * i] > a[4 * j]; });
  for (int c = 0; c < 3; c++) {
    val[c] = a[4 * idx[c]];
    for (int r = 0; r < 3; r++) vec[3 * r + c] = v[3 * r + idx[c]];
  }
}

void fit_rec(const std::vector<Vec3> &pts, double max_radius, int max_div, int min_points, int depth, std::vector<Primitive> &out) {
  Vec3 c;
  for (auto &p : pts) c = c + p;
  c = (1.0 / pts.size()) * c;
  double mom[9] = {0};
  for (auto &p : pts) {
    double d[3] = {p.x - c.x, p.y - c.y, p.z - c.z};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) mom[3 * i + j] += d[i] * d[j];
  }
  double vec[9], val[3];
  eigen_sym3(mom, vec, val);
  // z = dominant axis, y = second, x = y x z (tools/mesh_primitives.py:331-335); signs canonicalised
  double z[3] = {vec[0], vec[3], vec[6]}, y[3] = {vec[1], vec[4], vec[7]};
  for (double *a : {z, y}) {
    int k = std::fabs(a[0]) >= std::fabs(a[1]) ? (std::fabs(a[0]) >= std::fabs(a[2]) ? 0 : 2) : (std::fabs(a[1]) >= std::fabs(a[2]) ? 1 : 2);
    if (a[k] < 0) for (int i = 0; i < 3; i++) a[i] = -a[i];
  }
  double x[3] = {y[1] * z[2] - y[2] * z[1], y[2] * z[0] - y[0] * z[2], y[0] * z[1] - y[1] * z[0]};
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  std::vector<std::array<double, 3>> al(pts.size());
  for (size_t i = 0; i < pts.size(); i++) {
    double d[3] = {pts[i].x - c.x, pts[i].y - c.y, pts[i].z - c.z};
    al[i] = {d[0] * x[0] + d[1] * x[1] + d[2] * x[2], d[0] * y[0] + d[1] * y[1] + d[2] * y[2], d[0] * z[0] + d[1] * z[1] + d[2] * z[2]};
    for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], al[i][k]); hi[k] = std::max(hi[k], al[i][k]); }
  }
  double ctr_l[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
  double hwl[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
  Vec3 ctr{c.x + x[0] * ctr_l[0] + y[0] * ctr_l[1] + z[0] * ctr_l[2], c.y + x[1] * ctr_l[0] + y[1] * ctr_l[1] + z[1] * ctr_l[2],
           c.z + x[2] * ctr_l[0] + y[2] * ctr_l[1] + z[2] * ctr_l[2]};
  double r = 0.5 * std::max(hwl[0], hwl[1]), L = hwl[2] - 2 * r;
  if (r > max_radius && depth < max_div) {
    static const int oct[8][3] = {{1, 1, 1}, {1, -1, 1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, 1}, {-1, -1, 1}, {-1, -1, -1}, {-1, 1, -1}};
    for (auto &o : oct) {
      std::vector<Vec3> part;
      for (size_t i = 0; i < pts.size(); i++) {
        bool in = true;
        for (int k = 0; k < 3; k++) in &= ((al[i][k] - ctr_l[k] >= 0) == (o[k] > 0));
        if (in) part.push_back(pts[i]);
      }
      if ((int)part.size() > min_points) fit_rec(part, max_radius, max_div, min_points, depth + 1, out);
    }
    return;
  }
  if (L > 0) {
    Vec3 h{0.5 * L * z[0], 0.5 * L * z[1], 0.5 * L * z[2]};
    out.push_back({ctr - h, ctr + h, r});
  } else {
    out.push_back({ctr, ctr, r});
  }
}

}  // namespace

std::vector<Primitive> fit_primitives(const std::vector<Vec3> &points, double max_radius, int max_divisions, int min_points) {
  std::vector<Primitive> out;
  if (!points.empty()) fit_rec(points, max_radius, max_divisions, min_points, 0, out);
  return out;
}

void use_primitive_collision(HostModel &m, double max_radius, int max_divisions, int min_points) {
  std::vector<Vec3> xyz;
  std::vector<double> rad;
  std::vector<int> start{0};
  for (int b = 0; b < m.nb; b++) {
    for (size_t g = 0; g + 1 < m.hull_group_start.size(); g++) {
      int g0 = m.hull_group_start[g], g1 = m.hull_group_start[g + 1];
      if (!(m.hull_start[b] <= g0 && g0 < m.hull_start[b + 1])) continue;
      std::vector<Vec3> pts(m.hull_xyz.begin() + g0, m.hull_xyz.begin() + g1);
      for (auto &p : fit_primitives(pts, max_radius, max_divisions, min_points)) {
        xyz.push_back(p.p0); rad.push_back(p.radius);
        Vec3 d = p.p1 - p.p0;
        if (std::max({std::fabs(d.x), std::fabs(d.y), std::fabs(d.z)}) > 0) { xyz.push_back(p.p1); rad.push_back(p.radius); }
      }
    }
    start.push_back((int)xyz.size());
  }
  m.hull_xyz = xyz; m.hull_radius = rad; m.hull_start = start;
  m.hull_group_start = {0, (int)xyz.size()};
  for (int b = 0; b < m.nb; b++) {
    m.sphere_center[b] = Vec3{}; m.sphere_radius[b] = 0; m.box_half[b] = Vec3{};
    if (start[b + 1] == start[b]) continue;
    Vec3 lo{1e300, 1e300, 1e300}, hi{-1e300, -1e300, -1e300};
    for (int v = start[b]; v < start[b + 1]; v++) {
      const Vec3 &p = xyz[v];
      double r = rad[v];
      lo = {std::min(lo.x, p.x - r), std::min(lo.y, p.y - r), std::min(lo.z, p.z - r)};
      hi = {std::max(hi.x, p.x + r), std::max(hi.y, p.y + r), std::max(hi.z, p.z + r)};
    }
    Vec3 c = 0.5 * (lo + hi);
    double rr = 0;
    for (int v = start[b]; v < start[b + 1]; v++) rr = std::max(rr, std::sqrt(dot(xyz[v] - c, xyz[v] - c)) + rad[v]);
    m.sphere_center[b] = c; m.sphere_radius[b] = rr; m.box_half[b] = 0.5 * (hi - lo);
  }
}

}  // namespace trex
