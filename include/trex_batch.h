/*
 * trex_batch.h - C-ABI of the MI355X-native batched physics step for the T-rex gym env.
 *
 * This is the drop-in boundary for ONE path of bingjeff/trex-gym: everything that
 * TrexBulletEnv.step()/reset() delegates to the physics engine and the robot adapter
 * (boundary 3).  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative TREX_E_* code on failure;
 *     trex_last_error() returns a thread-local message for the last failure
 *     (pybullet raises pybullet.error at the same call sites [EXT]).
 *   - a TrexModel is immutable after trex_batch_create() has consumed it and may be
 *     shared by several batches; a TrexBatch is bound to one HIP device, is not re-entrant,
 *     and all of its calls are asynchronous and ordered on the hipStream_t given
 *     (passed as void* so the header needs no HIP include; NULL = the default stream).
 *   - "device" pointers are caller-owned HIP device buffers (e.g. torch tensors' data_ptr()) on the
 *     batch's device. Every batch call validates each distinct ALLOCATION once (hipPointerGetAttributes +
 *     its address range, cached per batch; pointers into a validated allocation are range-checked against it): host memory, another device's memory or a buffer shorter than
 *     the call needs returns TREX_E_INVALID instead of faulting the GPU. The cache is keyed by address: a
 *     buffer that was validated must stay allocated for as long as it is passed to the batch; a caller that
 *     FREES buffers it has passed (and may get the address back for a shorter or foreign allocation) calls
 *     trex_batch_forget_buffers() after freeing.
 *   - joints are always exposed in the reference's observation order: revolute joint names
 *     sorted (trex_robot.py:311-314); J = trex_model_num_joints() (25 for trex.urdf).
 */
#ifndef TREX_BATCH_H
#define TREX_BATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

#define TREX_OK 0
#define TREX_E_INVALID (-1)   /* bad argument (null pointer, unknown name, size mismatch) */
#define TREX_E_IO (-2)        /* file missing / unreadable */
#define TREX_E_PARSE (-3)     /* malformed URDF / mesh */
#define TREX_E_UNSUPPORTED (-4) /* model outside what the kernels handle (>26 bodies, joint type) */
#define TREX_E_HIP (-5)       /* HIP runtime error, no device */

typedef struct TrexModel TrexModel;
typedef struct TrexBatch TrexBatch;

const char *trex_last_error(void);
/* Identifies the kernel build (hash of the kernel sources, set by the Makefile): profiles/ records the
 * build its counters were collected on, bench.py quotes them only for a matching build. */
const char *trex_build_id(void);

/* ---- model: replaces loadURDF + the getJointInfo/getDynamicsInfo/getNumJoints introspection
 *      of trex_robot.py:47-56,98-117,158,175-187,294-320 and the floor of trex_env.py:103 ---- */

/* Parse a URDF (fixed + revolute joints), merge fixed joints, attach collision hulls.
 * collisions_dir: NULL -> use the URDF's own <collision><mesh .obj>; otherwise a directory of
 * COL_*_convex_hull.dae hulls placed with the <visual><origin> of the same-named mesh (this is
 * how the reference's own assets/trex.urdf, which has no <collision>, is loaded). */
int trex_model_load(const char *urdf_path, const char *collisions_dir, TrexModel **out);
void trex_model_destroy(TrexModel *model);

int trex_model_num_bodies(const TrexModel *model);      /* 26: moving bodies after the merge */
int trex_model_num_joints(const TrexModel *model);      /* 25: actuated revolute joints */
int trex_model_num_urdf_joints(const TrexModel *model); /* 132 = pybullet getNumJoints (trex_robot.py:158) */
int trex_model_num_hull_vertices(const TrexModel *model);
double trex_model_total_mass(const TrexModel *model, int include_base_link); /* trex_robot.py:318-320 sums without the base link */

/* k-th joint in observation order: name, pybullet joint index (trex_robot.py:314), limits
 * (trex_robot.py:337-346). Any out pointer may be NULL. */
int trex_model_joint_info(const TrexModel *model, int k, const char **name, int *urdf_joint_index,
                          double *lower, double *upper);

/* reset configuration (trex_env.py:81-87, trex_robot.py:305-308). Accepts the URDF joint name
 * or the pre-rename spelling ("femur_L_joint"). Unknown joint -> TREX_E_INVALID (the reference
 * raises KeyError there). Must be called before trex_batch_create. */
int trex_model_set_start_angle(TrexModel *model, const char *joint_name, double angle);
int trex_model_set_start_pose(TrexModel *model, const double xyz[3], const double rpy[3]); /* trex_env.py:105-106 */

/* Collision primitives (the step before the path): replace every convex hull by capsules /
 * spheres fitted as the reference's tools/mesh_primitives.py:323-402 does (PCA box -> capsule, octant
 * subdivision while radius > max_radius). Contacts are then generated from the capsule end spheres
 * (148 points instead of 2 181 vertices for trex.urdf at max_radius 0.2). Call before trex_batch_create. */
int trex_model_use_primitive_collision(TrexModel *model, double max_radius, int max_divisions, int min_points);
/* The fit of ONE convex hull (group g of "hull_group_start", body-frame coordinates) without changing the
 * model: writes up to `capacity` primitives as 7 doubles each (p0 xyz, p1 xyz, radius; p0 == p1 = sphere)
 * and returns their number (MJCF export, tests). */
int trex_model_fit_hull_primitives(const TrexModel *model, int group, double max_radius, int max_divisions,
                                   int min_points, double *out, int capacity);

/* engine parameters: "dt" "substeps" "iterations" "gravity" "motor_kp" "motor_kd" "motor_max_force"
 * "floor_z" "friction" "erp" "contact_erp" "contact_margin" "link_damping"
 * "max_coordinate_velocity" "max_contacts"  (setTimeStep / setPhysicsEngineParameter / setGravity,
 * trex_env.py:115-117; motor gains trex_robot.py:260,401,421). "max_contacts" is the contact-point budget
 * per env: default and upper limit 13 (25 motor rows + 3 x 13 contact rows = the 64 lanes of a wavefront;
 * larger values are clamped by the kernel). */
int trex_model_set_param(TrexModel *model, const char *name, double value);
int trex_model_get_param(const TrexModel *model, const char *name, double *value);

/* Introspection of the compiled model for tests: copies the named array as doubles, returns the
 * element count (or a negative error). Names: "parent" "depth" "joint_axis" "joint_pos" "joint_rot"
 * "q_lower" "q_upper" "joint_damping" "mass" "com" "inertia" "obs_order" "head_body" "head_point"
 * "hull_xyz" "hull_radius" "hull_group_start" "hull_start" "sphere_center" "sphere_radius" "q_start" "base_start_pos"
 * "base_start_quat" "revolute_joint_indices" "link_body" "link_tf" (12 per link: R row-major, t). */
int trex_model_get_array(const TrexModel *model, const char *name, double *out, int capacity);

/* ---- batch: N independent env copies resident on one GPU ---- */

int trex_batch_create(const TrexModel *model, int num_envs, int device, TrexBatch **out);
void trex_batch_destroy(TrexBatch *batch);
int trex_batch_num_envs(const TrexBatch *batch);

/* Forget the validated caller pointers (see the conventions above): the next call validates them afresh. */
int trex_batch_forget_buffers(TrexBatch *batch);

/* Row-block calls (trex_batch_step_rows / _reset_rows / _step_many): enabled != 0 -> columns [3J+2, 3J+5) of every row
 * receive the three penalties; 0 (default) -> nothing beyond column 3J + 1 is written. */
int trex_batch_set_penalties_in_rows(TrexBatch *batch, int enabled);

/* Which wave runs which env. All waves of a launch of <= 4096 envs are resident at once and a SIMD is done when
 * its slowest wave is, so the step kernel can rank the envs by the contact count of their previous step and deal
 * them to the SIMDs heaviest-with-lightest (device-side state only; results are bitwise independent of it).
 * mode -1 (default): on for batches of 2048 envs or more - below that most SIMDs hold at most two waves and there is
 * nothing to level -, 0: off (workgroup k runs env k), 1: on for any size. */
int trex_batch_set_wave_balance(TrexBatch *batch, int mode);

/* reward weights (trex_env.py:42-44): distance, energy, drift. Defaults 1.0, 0.005, 0.002. */
int trex_batch_set_reward_weights(TrexBatch *batch, float distance, float energy, float drift);

/* TrexBulletEnv.reset (trex_env.py:98-122): envs with mask[n] != 0 (all if mask == NULL) go to the
 * start pose with motors disabled and take ONE un-actuated substep. obs_out (device, [N, 3J],
 * nullable) receives the observation of every env (reset or not). */
int trex_batch_reset(TrexBatch *batch, const uint8_t *mask_dev, float *obs_out_dev, void *stream);

/* TrexBulletEnv.step (trex_env.py:128-154) for all N envs in ONE kernel launch:
 * clip(actions) -> substeps x [position motors + physics substep] -> obs, reward, done.
 *   actions_dev   [N, J]  f32 device, joint targets in observation order
 *   obs_dev       [N, 3J] f32 device: q, qd, appliedJointMotorTorque (trex_robot.py:365)
 *   reward_dev    [N]     f32 device (trex_env.py:192)
 *   done_dev      [N]     u8 device, 0 (trex_env.py:183-184) - except 1 for an env whose state became
 *                         non-finite (it is put back on the start pose and reports reward 0: containment) and
 *                         for an env that reached the episode limit (trex_batch_set_episode_limit)
 *   penalties_dev [N, 3]  f32 device, nullable: lifting_com, station_keeping, energy
 *                         (the three values logged at trex_env.py:193-195) */
int trex_batch_step(TrexBatch *batch, const float *actions_dev, float *obs_dev, float *reward_dev,
                    uint8_t *done_dev, float *penalties_dev, void *stream);

/* The same two calls writing ONE row block (what the multi-GPU exchange gathers):
 *   rows_dev [N, row_stride] f32 device, row_stride >= 3J + 2:
 *     [0, 3J) observation, [3J] reward, [3J+1] done as 0.0 / 1.0. Columns beyond 3J + 2 are NOT touched, whatever the
 *     stride (rows padded for alignment, or embedded in a wider tensor with columns of the caller's own) - unless the
 *     batch was told to carry the three penalties (lifting_com, station_keeping, energy: trex_env.py:193-195) in the
 *     rows: after trex_batch_set_penalties_in_rows(batch, 1) the row calls need row_stride >= 3J + 5 and write
 *     [3J+2, 3J+5) too (zeros from a reset) - one message for a consumer that wants them. (Until round 3 a stride of
 *     3J + 5 or more switched this on implicitly.)
 * done_dev [N] u8, nullable: the done flags once more as bytes (what a consumer masks with - saves it a
 *   conversion pass over the column). trex_batch_reset_rows writes the observation columns of every env (reset or
 *   not) and, for the envs it resets, reward = 0 and done = 0: the row of a new episode. */
int trex_batch_step_rows(TrexBatch *batch, const float *actions_dev, float *rows_dev, int row_stride,
                         float *penalties_dev, uint8_t *done_dev, void *stream);
int trex_batch_reset_rows(TrexBatch *batch, const uint8_t *mask_dev, float *rows_dev, int row_stride, void *stream);

/* num_steps env-steps of every env in ONE launch, for action sequences that are known in advance (open-loop rollouts:
 * random-action benchmarks, sampling-based planners, replaying recorded actions): step s reads actions_dev[s] and writes
 * the row block rows_dev[s] -
 *   actions_dev [S, N, J] f32, rows_dev [S, N, row_stride] f32 (obs | reward | done per row, as trex_batch_step_rows),
 *   penalties_dev [S, N, 3] and done_dev [S, N] u8 nullable.
 * Results are BITWISE those of num_steps trex_batch_step_rows calls (episode limit and containment included: tested).
 * Why it exists: with one step per launch every SIMD waits for the launch's slowest wave - a fifth of the launch at 4 096
 * envs -, here a wave goes straight on to its env's next step and the env's state stays on the chip between steps. A
 * closed loop (a policy that needs step s's observations for step s + 1's actions) cannot use it. */
int trex_batch_step_many(TrexBatch *batch, const float *actions_dev, float *rows_dev, int row_stride, int num_steps,
                         float *penalties_dev, uint8_t *done_dev, void *stream);

/* Episode limit of the harness. The reference env never terminates (should_terminate() is constant False,
 * trex_env.py:183-184); a training harness cuts episodes (gym's TimeLimit; baselines' VecEnv then resets the env
 * and returns the first observation of the new episode with done = True). With max_episode_steps > 0 the STEP
 * LAUNCH itself does that for the envs whose step count reaches the limit: reward of the finished step, done = 1,
 * then start pose + the reset's settle substep, observation of the new episode - no separate reset launch.
 * episode_steps_dev ([N] i32, nullable = zeros) sets the counts (e.g. to stagger the episodes);
 * max_episode_steps = 0 switches the limit off. trex_batch_reset zeroes the count of the envs it resets. */
int trex_batch_set_episode_limit(TrexBatch *batch, int max_episode_steps, const int32_t *episode_steps_dev, void *stream);
int trex_batch_get_episode_steps(TrexBatch *batch, int32_t *episode_steps_dev, void *stream);

/* env state [N, 13 + 2J] f32 device: base position(3), base orientation quaternion xyzw(4) - both
 * of the base INERTIAL frame as resetBasePositionAndOrientation/getBasePositionAndOrientation
 * (trex_robot.py:63,327) - base linear(3) and angular(3) world velocity, q(J), qd(J). */
int trex_batch_get_state(TrexBatch *batch, float *state_dev, void *stream);
int trex_batch_set_state(TrexBatch *batch, const float *state_dev, void *stream);
/* motors stay disabled after reset until the first step (trex_robot.py:309); set_state keeps the
 * flag, this call forces it (tests). */
int trex_batch_set_motors_enabled(TrexBatch *batch, int enabled, void *stream);

/* world position of the head link COM, [N,3] (trex_robot.py:330-335). */
int trex_batch_head_position(TrexBatch *batch, float *out_dev, void *stream);

/* Rollout export for rendering (the step after the path: trex_env.py:156-181, trex_train.py:126-136):
 * world pose of EVERY URDF link frame (133 for trex.urdf, document order) as [N, L, 7] f32 device =
 * position xyz + quaternion xyzw - what getLinkState(...)[4:6] returns per link [EXT]. A renderer
 * composes it with the <visual><origin> of each mesh. */
int trex_model_num_links(const TrexModel *model);
int trex_model_link_info(const TrexModel *model, int link, const char **name, int *body);
int trex_batch_link_transforms(TrexBatch *batch, float *out_dev, void *stream);

/* The table a renderer needs to place the meshes (trex_env.py:156-181 draws them through pybullet; the reference's
 * parser holds them as UrdfLink.visual_shapes, tools/urdf_parsing.py:93-120,299-307): every <visual> mesh of the URDF
 * in document order (252 for trex.urdf) - mesh file name as written in the URDF, index of its link
 * (trex_model_link_info), and its <origin> in the link frame as position + quaternion xyzw. Any out pointer may be
 * NULL. trex_batch_visual_transforms: world pose of every mesh, [N, V, 7] f32 device = link pose x <origin> -
 * no URDF re-parsing, no composition left to the caller. The mesh FILES are the caller's (not loaded here). */
int trex_model_num_visuals(const TrexModel *model);
int trex_model_visual_info(const TrexModel *model, int visual, const char **mesh_file, int *link, double xyz[3],
                           double quat_xyzw[4]);
int trex_batch_visual_transforms(TrexBatch *batch, float *out_dev, void *stream);

/* domain randomisation (BASELINE config 5; no reference counterpart): per-env mass scale of each
 * moving body [N, num_bodies] and per-env friction coefficient [N]; either may be NULL. */
int trex_batch_set_domain(TrexBatch *batch, const float *mass_scale_dev, const float *friction_dev,
                          void *stream);

/* diagnostics of the last substep: contact count per env [N] i32 (nullable), summed normal
 * impulse per env [N] f32 (nullable). */
int trex_batch_contact_stats(TrexBatch *batch, int32_t *count_dev, float *normal_impulse_dev, void *stream);

/* Diagnostics for the parity tests: one step like trex_batch_step, additionally dumping env 0's
 * intermediates of its LAST substep into debug_dev (4096 f32 device): [0,32) qdd per body lane,
 * [32,38) base spatial acceleration, [64,96) generalised velocity before the constraint solve
 * per dof lane, [96,128) its PGS correction, [128] contact count, [129] limit-row mask,
 * [160,960) the 25 joint columns of M^-1, [960+16c ..) per contact body,x,y,z,dist,1/diag(3),rhs(3),
 * lambda(3), [1216+32(3c+a) ..) the contact rows' response vectors. Not part of the product path. */
int trex_batch_debug_step(TrexBatch *batch, const float *actions_dev, float *obs_dev, float *debug_dev, void *stream);

/* Launch geometry + bytes, for bench.py: fills grid, block, lds bytes, algorithmic bytes/env-step. block = 128 means the
 * pair form of the step launch (two envs = two wavefronts per workgroup: even batches of at most 4096 envs), 64 the
 * single-env form. */
int trex_batch_launch_info(const TrexBatch *batch, int *grid, int *block, int *lds_bytes,
                           int *alg_bytes_per_env_step);

/* Times `steps` trex_batch_step launches with hipEvents on `stream` (the stream the kernels run on)
 * and returns the average per-launch duration in milliseconds. */
int trex_batch_time_steps(TrexBatch *batch, const float *actions_dev, float *obs_dev, float *reward_dev,
                          uint8_t *done_dev, int steps, void *stream, float *avg_ms_out);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* TREX_BATCH_H */
