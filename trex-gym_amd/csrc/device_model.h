// Device-side layout of the compiled model and of the per-env state (shared by host and kernels).
#pragma once
#include <stdint.h>

#define TREX_TL 32        /* lanes per env team = half a wavefront */
#define TREX_MAXD 6       /* tree depth supported */
#define TREX_MAXCH 4      /* moving children per body */
#define TREX_CM_WORDS 320  /* mask words per env: 32 for one big body + 8 for every other (T-rex: 216) */
#define TREX_MAXC 13      /* contact points kept per env: 25 motor rows + 3 x 13 contact rows = 64 lanes */

enum TrexParam {
  TP_DT, TP_SUBSTEPS, TP_ITERATIONS, TP_GRAVITY, TP_MOTOR_KP, TP_MOTOR_KD, TP_MOTOR_MAX_FORCE,
  TP_FLOOR_Z, TP_FRICTION, TP_ERP, TP_CONTACT_ERP, TP_CONTACT_MARGIN, TP_LINK_DAMPING,
  TP_MAX_COORD_VEL, TP_MAX_CONTACTS, TP_COUNT
};

/* Model constants, one copy in HBM (about 5 KB + hull vertices), L2-resident for every wave.
 * Per-body arrays are [component][lane] so that lane b of a team reads body b: coalesced 128 B. */
struct TrexDeviceModel {
  int nb, maxdepth, head_body, nv;
  float prm[16];
  /* the same, ready for scalar loads: integer parameters as integers and the derived constants, so that no
   * wave-uniform conversion / division is done (and kept) in vector registers */
  int n_substeps, n_iterations, max_contacts, pad0;
  float inv_dt, motor_max_impulse, pad1, pad2;
  float head_point[4];
  float base_pos0[4], base_quat0[4];
  int parent[TREX_TL], depth[TREX_TL];
  int obs_slot[TREX_TL];              /* body -> index in the sorted-joint observation order, -1 for base */
  int anc[TREX_MAXD][TREX_TL];        /* anc[d-1][b] = ancestor of body b at depth d (b itself at its own depth), -1 beyond */
  int child[TREX_MAXCH][TREX_TL];     /* moving children of body b, -1 = none */
  unsigned desc_mask[TREX_TL];        /* per DOF LANE: bit b set if body b's chain contains this dof (base dof lanes: all bodies) */
  int hull_start[TREX_TL + 1];
  /* In-margin vertex masks of the contact generation (LDS): body b owns 2^log words from word `off` on, word w holding the
   * vertices w, w + 2^log, ... of the body (bit j <-> vertex 2^log j + w): log = 3 for a body of at most 256 hull vertices,
   * 5 up to 1024, 0 = no mask (a larger body, or no room left: swept instead). cm_pack[b] = off << 8 | log. */
  int cm_pack[TREX_TL];
  float axis[3][TREX_TL], jpos[3][TREX_TL], jrot[9][TREX_TL], com[3][TREX_TL], inertia[6][TREX_TL];
  float mass[TREX_TL], lower[TREX_TL], upper[TREX_TL], damp[TREX_TL], q_start[TREX_TL];
  float sphere[4][TREX_TL];           /* bounding sphere of the body's hull vertices: cx cy cz r */
  float box_half[3][TREX_TL];         /* half extents of their body-frame AABB (same centre): the broad-phase bound */
  /* Scan units of the contact generation: one per convex hull (28 for trex.urdf; the pelvis body carries 6), so that
   * the broad phase tests - and the narrow phase scans - a hull, not everything merged into its body. At most 32
   * (one lane each); a model with more hulls gets one unit per body. Vertices [v0, v1) of `hull`, body-frame AABB
   * centre and half extents (support radius included). */
  int nchunk, pad3[3];
  int chunk_body[TREX_TL], chunk_v0[TREX_TL], chunk_v1[TREX_TL];
  float chunk_c[3][TREX_TL], chunk_h[3][TREX_TL];
};

/* Per-env state in HBM. One row per env, padded so that a 32-lane team reads whole 128-B segments:
 *   base  [N][16]  pos(3) quat xyzw(4) v(3) w(3) | [13] contact count of the last substep (bits 0..7) + motors-on flag
 *                  (bit 8), as an int | [14] summed normal impulse of the last substep | [15] env-steps since the last
 *                  reset, as an int (the harness's episode limit). ONE 64-byte line carries everything the step launch
 *                  reads and writes per env besides the joint rows (round 2 kept the four scalars in arrays of their
 *                  own: four more partial-line reads and writes per env-step).
 *   q, qd  [N][32]  indexed by BODY lane (lane 0 unused). (The last motor torques are NOT kept: they are an output - the
 *                  observation's third block - and no step reads them back.)
 *   mass_scale [N][32], friction [N]: domain randomisation; read only when `domain` is set (trex_batch_set_domain) */
#define TREX_BASE_FLAGS 13
#define TREX_BASE_IMPULSE 14
#define TREX_BASE_STEPS 15
#define TREX_MOTORS_BIT 256
#define TREX_BAL_PHASE 0
#define TREX_BAL_FINISHED 1
#define TREX_BAL_COUNTS 16
#define TREX_BAL_BINS 16
#define TREX_BAL_LISTS 48
#define TREX_BAL_WORDS(n) (TREX_BAL_LISTS + 2 * TREX_BAL_BINS * (size_t)(n))

struct TrexBatchArrays {
  float *base, *q, *qd, *mass_scale, *friction;
  int32_t domain;         /* != 0: per-env mass_scale / friction are in force (else the model's values: no loads) */
  int32_t pad0_;
  int32_t *balance;       /* wave balance, device-side state only: [TREX_BAL_PHASE] which of the two list sets the next
                             step launch reads, [TREX_BAL_FINISHED] waves of the running launch that have ended,
                             [TREX_BAL_COUNTS + 16 p + c] envs filed under contact count c in set p,
                             [TREX_BAL_LISTS + (16 p + c) N + i] the i-th of them */
  int32_t max_episode_steps;  /* 0 = no limit; > 0: an env whose count reaches it is reset INSIDE the step launch */
  int32_t pad_;
  float4 *hull;  /* [nv] body-frame collision points: xyz + support radius (0 for hull vertices) */
  int num_links;
  const int *link_body;   /* [L] body of each URDF link */
  const float *link_tf;   /* [L][12] body<-link transform: rotation row-major (9) + translation (3) */
  int num_visuals;
  const int *visual_body; /* [V] body of each <visual> mesh */
  const float *visual_tf; /* [V][12] body<-mesh transform (link frame in its body x the visual's <origin>) */
};
