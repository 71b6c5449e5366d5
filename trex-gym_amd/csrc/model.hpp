// Host-side model compiler: URDF (+ collision hulls) -> reduced articulated tree.
// Replaces pybullet's loadURDF + the introspection calls of trex_robot.py (see include/trex_batch.h).
#pragma once
#include <array>
#include <map>
#include <string>
#include <vector>

namespace trex {

constexpr int kLanes = 32;      // lanes per env team (half a wavefront)
constexpr int kMaxBodies = 26;  // bodies + 6 base dofs must fit the 32 team lanes
constexpr int kMaxDepth = 6;
constexpr int kMaxChildren = 4;

struct Vec3 { double x = 0, y = 0, z = 0; };
struct Mat3 { double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; };
struct Tf { Mat3 R; Vec3 t; };

struct Params {
  double dt = 0.01 / 5;             // trex_env.py:54,71
  double substeps = 5;              // trex_env.py:18,73
  double iterations = 60;           // trex_env.py:57,72,115
  double gravity = 9.81;            // trex_env.py:20,117
  double motor_kp = 5e-3;           // trex_robot.py:421
  double motor_kd = 0.1;            // trex_robot.py:401
  double motor_max_force = 3e5;     // trex_robot.py:260
  double floor_z = 0.0005;          // floor.urdf:21
  double friction = 0.25;           // [EXT] 0.5 * 0.5
  double erp = 0.2;                 // [EXT]
  double contact_erp = 0.2;         // [EXT]
  double contact_margin = 0.02;     // [EXT]
  double link_damping = 0.04;       // [EXT]
  double max_coordinate_velocity = 100.0;  // [EXT]
  double max_contacts = 13;   // 25 motor rows + 3 x 13 contact rows = one row per lane of a wavefront
  double *find(const std::string &name);
};

struct HostModel {
  int nb = 0;
  int num_urdf_joints = 0;
  std::vector<std::string> body_names, joint_names;  // per body (joint_names[0] empty)
  std::vector<int> parent, depth;
  std::vector<Vec3> joint_axis, joint_pos, com, sphere_center, box_half;  // box: AABB of the hull in the body frame (centre = sphere_center)
  std::vector<Mat3> joint_rot;
  std::vector<double> q_lower, q_upper, joint_damping, mass, sphere_radius, q_start;
  std::vector<std::array<double, 6>> inertia;  // xx xy xz yy yz zz about COM, body axes
  std::vector<int> obs_order;               // k-th sorted joint -> body
  std::vector<int> revolute_joint_indices;  // k-th sorted joint -> URDF joint index
  std::vector<std::string> obs_joint_names;
  int head_body = -1;
  Vec3 head_point;
  // every URDF link (document order): the body it was merged into and its frame in that body's frame
  std::vector<std::string> link_names;
  std::vector<int> link_body;
  std::vector<Tf> link_tf;
  // every <visual> mesh (document order): file name as written in the URDF, its link, its <origin> in the link
  // frame (what tools/urdf_parsing.py:93-120 calls visual_shapes) and the same in the frame of the link's body
  std::vector<std::string> visual_file;
  std::vector<int> visual_link;
  std::vector<Tf> visual_origin, visual_body_tf;
  std::vector<Vec3> hull_xyz;
  std::vector<double> hull_radius;     // support radius per point: 0 for hull vertices, > 0 for fitted spheres
  std::vector<int> hull_start;
  std::vector<int> hull_group_start;   // one group per original convex hull (CSR over hull_xyz)
  double total_mass = 0, total_mass_excluding_base = 0;
  Vec3 base_start_pos{0, 0, 3};           // trex_env.py:105
  double base_start_quat[4] = {0, 0, 0, 1};  // trex_env.py:106 rpy = 0
  Params prm;
};

// throws std::runtime_error with a message; `code` is set to the TREX_E_* value
HostModel load_model(const std::string &urdf_path, const char *collisions_dir, int *code);

std::string rename_v0_name(const std::string &name);  // femur_L_joint -> joint_femur_left

// Collision primitives: the capsule / sphere fitting of the reference's
// tools/mesh_primitives.py:323-402, restated. Returns capsules as (p0, p1, radius); p0 == p1 = sphere.
struct Primitive { Vec3 p0, p1; double radius; };
std::vector<Primitive> fit_primitives(const std::vector<Vec3> &points, double max_radius, int max_divisions, int min_points);
// Replace every hull of the model by the end spheres of its fitted primitives.
void use_primitive_collision(HostModel &m, double max_radius, int max_divisions, int min_points);
Mat3 rpy_to_matrix(double r, double p, double y);
void matrix_to_quat(const Mat3 &m, double q[4]);

}  // namespace trex
