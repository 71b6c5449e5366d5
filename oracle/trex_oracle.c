/*
 * ORACLE - TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the hot path of bingjeff/trex-gym: one TrexBulletEnv.step()
 * (trex_env.py:128-154) = clip action, 5 x [set position motors (trex_robot.py:397-422) +
 * pybullet stepSimulation (trex_env.py:150)], observations (trex_robot.py:359-365) and reward
 * (trex_env.py:186-196); and reset() (trex_env.py:98-122, trex_robot.py:39-65,300-320).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY UNPINNED: stepSimulation lives in pybullet (third-party Bullet3, un-pinned in
 * setup.py:12, absent from /root/reference and from this image), and the reference holds no
 * test, fixture or golden vector that touches it.  What is restated here is the
 * published algorithm of Bullet's btMultiBody pipeline as recollected in SURVEY Appendix C:
 *   joint damping torque -> gravity -> articulated-body algorithm -> qd += qdd*dt ->
 *   projected Gauss-Seidel over {joint-limit rows, position-motor rows, contact + friction rows}
 *   on the velocity level -> q += qd*dt.
 * The dynamics follow Featherstone, "Rigid Body Dynamics Algorithms" (RBDA) ch.7 (ABA) and
 * ch.9 (floating base).  Its correctness is pinned by physics invariants and by an independent
 * numpy RNEA/CRBA formulation in tests/, not by pybullet output.
 *
 * Formulation.  All spatial quantities of one env are expressed in ONE frame: world-aligned
 * axes, origin O at the base-frame origin.  Motion vectors are [omega; v_O], force vectors
 * [n_O; f].  Generalised velocity = [omega_base(3), v_base(3), qd_1..qd_{nb-1}] where v_base is
 * the world velocity of the base-frame origin.  Because every body uses the same frame, the
 * ABA needs no frame transforms between parent and child.
 *
 * Build:  gcc -O2 -shared -fPIC [-DORACLE_FLOAT] trex_oracle.c -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORACLE_FLOAT
typedef float real;
#else
typedef double real;
#endif

#define NBMAX 32
#define MAXC 64            /* hard cap on contact points kept by the oracle */
#define MAXROWS (2 * NBMAX + 3 * MAXC)
#define NDOF_MAX (6 + NBMAX)

enum { /* params[] indices, shared with tests/oracle_binding.py */
  P_DT, P_SUBSTEPS, P_ITERATIONS, P_GRAVITY, P_MOTOR_KP, P_MOTOR_KD, P_MOTOR_MAX_FORCE,
  P_FLOOR_Z, P_FRICTION, P_ERP, P_CONTACT_ERP, P_CONTACT_MARGIN, P_LINK_DAMPING,
  P_MAX_COORD_VEL, P_MAX_CONTACTS, P_COUNT
};

typedef struct {
  int nb;
  int parent[NBMAX], depth[NBMAX];
  real axis[NBMAX][3], jpos[NBMAX][3], jrot[NBMAX][9];
  real q_lower[NBMAX], q_upper[NBMAX], jdamp[NBMAX];
  real mass[NBMAX], com[NBMAX][3], inertia[NBMAX][6];
  int obs_order[NBMAX];
  int head_body;
  real head_point[3];
  int nv;
  real *hull; /* [nv][3] body frame */
  real *hull_r; /* [nv] support radius: 0 for hull vertices, > 0 for fitted sphere / capsule ends */
  int hull_start[NBMAX + 1];
  real sphere_c[NBMAX][3], sphere_r[NBMAX];
  real q_start[NBMAX], base_pos0[3], base_quat0[4];
  real prm[P_COUNT];
} Model;

typedef struct {
  real pos[3], quat[4]; /* base frame pose, quat = (x,y,z,w) */
  real v[3], w[3];      /* world linear velocity of base origin, world angular velocity */
  real q[NBMAX], qd[NBMAX];
  real motor_tau[NBMAX]; /* appliedJointMotorTorque of the last substep */
  real mass_scale[NBMAX]; /* domain randomisation, 1.0 default */
  real friction;          /* per-env mu */
  int motors_on;          /* 0 after reset until the first step (trex_robot.py:309) */
  /* diagnostics of the last substep */
  int n_contacts, n_limit_rows;
  real contact_body[MAXC], contact_lambda[MAXC][3], contact_pos[MAXC][3], contact_dist[MAXC];
} State;

/* ------------------------------------------------------------------ small algebra */
static void cross3(const real *a, const real *b, real *o) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static real dot3(const real *a, const real *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static real dot6(const real *a, const real *b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
static void matvec3(const real *m, const real *v, real *o) {
  real x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  real y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  real z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void matmul3(const real *a, const real *b, real *o) {
  real t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  memcpy(o, t, sizeof t);
}
static void quat_to_mat(const real *q, real *m) {
  real x = q[0], y = q[1], z = q[2], w = q[3];
  m[0] = 1 - 2 * (y * y + z * z); m[1] = 2 * (x * y - z * w); m[2] = 2 * (x * z + y * w);
  m[3] = 2 * (x * y + z * w); m[4] = 1 - 2 * (x * x + z * z); m[5] = 2 * (y * z - x * w);
  m[6] = 2 * (x * z - y * w); m[7] = 2 * (y * z + x * w); m[8] = 1 - 2 * (x * x + y * y);
}
static void axis_angle_mat(const real *a, real q, real *m) {
  real c = (real)cos(q), s = (real)sin(q), t = 1 - c;
  m[0] = t * a[0] * a[0] + c;        m[1] = t * a[0] * a[1] - s * a[2]; m[2] = t * a[0] * a[2] + s * a[1];
  m[3] = t * a[0] * a[1] + s * a[2]; m[4] = t * a[1] * a[1] + c;        m[5] = t * a[1] * a[2] - s * a[0];
  m[6] = t * a[0] * a[2] - s * a[1]; m[7] = t * a[1] * a[2] + s * a[0]; m[8] = t * a[2] * a[2] + c;
}
/* spatial cross products (RBDA 2.31, 2.32) */
static void crm(const real *v, const real *m, real *o) { /* v x m (motion) */
  real a[3], b[3], c[3];
  cross3(v, m, a); cross3(v, m + 3, b); cross3(v + 3, m, c);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
  o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
static void crf(const real *v, const real *f, real *o) { /* v x* f (force) */
  real a[3], b[3], c[3];
  cross3(v, f, a); cross3(v + 3, f + 3, b); cross3(v, f + 3, c);
  o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2];
  o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}
static void mat6vec(const real *m, const real *v, real *o) {
  real t[6];
  for (int i = 0; i < 6; i++) t[i] = dot6(m + 6 * i, v);
  memcpy(o, t, sizeof t);
}
/* symmetric positive-definite 6x6 inverse by Cholesky */
static void spd6_inverse(const real *a, real *inv) {
  real l[36] = {0};
  for (int i = 0; i < 6; i++)
    for (int j = 0; j <= i; j++) {
      real s = a[6 * i + j];
      for (int k = 0; k < j; k++) s -= l[6 * i + k] * l[6 * j + k];
      l[6 * i + j] = (i == j) ? (real)sqrt(s) : s / l[6 * j + j];
    }
  for (int c = 0; c < 6; c++) {
    real y[6], x[6];
    for (int i = 0; i < 6; i++) {
      real s = (i == c) ? 1 : 0;
      for (int k = 0; k < i; k++) s -= l[6 * i + k] * y[k];
      y[i] = s / l[6 * i + i];
    }
    for (int i = 5; i >= 0; i--) {
      real s = y[i];
      for (int k = i + 1; k < 6; k++) s -= l[6 * k + i] * x[k];
      x[i] = s / l[6 * i + i];
    }
    for (int i = 0; i < 6; i++) inv[6 * i + c] = x[i];
  }
}

/* ------------------------------------------------------------------ per-substep workspace */
typedef struct {
  real R[NBMAX][9];   /* body rotation, world <- body */
  real r[NBMAX][3];   /* body frame origin relative to O (world axes) */
  real S[NBMAX][6];   /* joint motion subspace */
  real vel[NBMAX][6]; /* body spatial velocity */
  real I[NBMAX][36];  /* rigid-body spatial inertia about O */
  real IA[NBMAX][36]; /* articulated-body inertia */
  real U[NBMAX][6], D[NBMAX], u[NBMAX];
  real I0inv[36];
  real comw[NBMAX][3]; /* COM relative to O */
} Work;

static void kinematics(const Model *m, const State *s, Work *k) {
  quat_to_mat(s->quat, k->R[0]);
  k->r[0][0] = k->r[0][1] = k->r[0][2] = 0;
  for (int i = 1; i < m->nb; i++) {
    int p = m->parent[i];
    real rq[9], t[9], d[3];
    axis_angle_mat(m->axis[i], s->q[i], rq);
    matmul3(k->R[p], m->jrot[i], t);
    matmul3(t, rq, k->R[i]);
    matvec3(k->R[p], m->jpos[i], d);
    for (int c = 0; c < 3; c++) k->r[i][c] = k->r[p][c] + d[c];
    real a[3];
    matvec3(k->R[i], m->axis[i], a);
    k->S[i][0] = a[0]; k->S[i][1] = a[1]; k->S[i][2] = a[2];
    cross3(k->r[i], a, k->S[i] + 3);
  }
  for (int i = 0; i < m->nb; i++) {
    real c[3];
    matvec3(k->R[i], m->com[i], c);
    for (int x = 0; x < 3; x++) k->comw[i][x] = k->r[i][x] + c[x];
  }
}

static void velocities(const Model *m, const State *s, Work *k) {
  for (int c = 0; c < 3; c++) { k->vel[0][c] = s->w[c]; k->vel[0][3 + c] = s->v[c]; }
  for (int i = 1; i < m->nb; i++) {
    int p = m->parent[i];
    for (int c = 0; c < 6; c++) k->vel[i][c] = k->vel[p][c] + k->S[i][c] * s->qd[i];
  }
}

static void body_inertia_world(const Model *m, const State *s, const Work *k, int i, real *Icw, real *mass) {
  const real *a = m->inertia[i];
  real Ib[9] = {a[0], a[1], a[2], a[1], a[3], a[4], a[2], a[4], a[5]};
  real t[9], Rt[9];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rt[3 * r + c] = k->R[i][3 * c + r];
  matmul3(k->R[i], Ib, t);
  matmul3(t, Rt, Icw);
  real sc = s->mass_scale[i];
  for (int c = 0; c < 9; c++) Icw[c] *= sc;
  *mass = m->mass[i] * sc;
}

static void spatial_inertias(const Model *m, const State *s, Work *k) {
  for (int i = 0; i < m->nb; i++) {
    real Ic[9], ms;
    body_inertia_world(m, s, k, i, Ic, &ms);
    const real *c = k->comw[i];
    real cx[9] = {0, -c[2], c[1], c[2], 0, -c[0], -c[1], c[0], 0};
    real cc[9];
    matmul3(cx, cx, cc); /* cx*cx ; I_O = Ic - m cx cx */
    real *I = k->I[i];
    for (int r = 0; r < 3; r++)
      for (int q = 0; q < 3; q++) {
        I[6 * r + q] = Ic[3 * r + q] - ms * cc[3 * r + q];
        I[6 * r + 3 + q] = ms * cx[3 * r + q];
        I[6 * (3 + r) + q] = -ms * cx[3 * r + q];
        I[6 * (3 + r) + 3 + q] = (r == q) ? ms : 0;
      }
  }
}

/* Articulated-body inertias and the U, D of every joint (RBDA table 7.1 pass 2, inertia part). */
static void articulated_inertias(const Model *m, Work *k) {
  memcpy(k->IA, k->I, sizeof(real) * 36 * m->nb);
  for (int i = m->nb - 1; i >= 1; i--) {
    int p = m->parent[i];
    mat6vec(k->IA[i], k->S[i], k->U[i]);
    k->D[i] = dot6(k->S[i], k->U[i]);
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 6; c++)
        k->IA[p][6 * r + c] += k->IA[i][6 * r + c] - k->U[i][r] * k->U[i][c] / k->D[i];
  }
  spd6_inverse(k->IA[0], k->I0inv);
}

/* Forward dynamics for bias forces pA (articulated), joint torques tau: fills qdd[1..], a0[6]. */
static void aba_solve(const Model *m, Work *k, real pA[][6], const real *tau, const real cvec[][6],
                      real *qdd, real *a0) {
  for (int i = m->nb - 1; i >= 1; i--) {
    int p = m->parent[i];
    k->u[i] = tau[i] - dot6(k->S[i], pA[i]);
    /* pa = pA + Ia c + U u / D, with Ia c = IA c - U (U.c)/D */
    real Ic[6];
    mat6vec(k->IA[i], cvec[i], Ic);
    real uc = dot6(k->U[i], cvec[i]);
    for (int c = 0; c < 6; c++)
      pA[p][c] += pA[i][c] + Ic[c] + k->U[i][c] * (k->u[i] - uc) / k->D[i];
  }
  for (int r = 0; r < 6; r++) {
    real s = 0;
    for (int c = 0; c < 6; c++) s -= k->I0inv[6 * r + c] * pA[0][c];
    a0[r] = s;
  }
  real acc[NBMAX][6];
  memcpy(acc[0], a0, sizeof(real) * 6);
  for (int i = 1; i < m->nb; i++) {
    int p = m->parent[i];
    real ap[6];
    for (int c = 0; c < 6; c++) ap[c] = acc[p][c] + cvec[i][c];
    qdd[i] = (k->u[i] - dot6(k->U[i], ap)) / k->D[i];
    for (int c = 0; c < 6; c++) acc[i][c] = ap[c] + k->S[i][c] * qdd[i];
  }
}

/* dv = M^-1 f for a generalised force f = [base wrench(6), tau_1..]: the ABA delta sweeps
 * (the arithmetic of Bullet's calcAccelerationDeltasMultiDof, in the common frame). */
static void apply_minv(const Model *m, Work *k, const real *f, real *dv) {
  real pA[NBMAX][6];
  real u[NBMAX];
  memset(pA, 0, sizeof pA);
  for (int i = m->nb - 1; i >= 1; i--) {
    int p = m->parent[i];
    u[i] = f[6 + i - 1] - dot6(k->S[i], pA[i]);
    for (int c = 0; c < 6; c++) pA[p][c] += pA[i][c] + k->U[i][c] * u[i] / k->D[i];
  }
  real rhs[6], acc[NBMAX][6];
  for (int c = 0; c < 6; c++) rhs[c] = f[c] - pA[0][c];
  mat6vec(k->I0inv, rhs, acc[0]);
  for (int c = 0; c < 6; c++) dv[c] = acc[0][c];
  for (int i = 1; i < m->nb; i++) {
    int p = m->parent[i];
    real qdd = (u[i] - dot6(k->U[i], acc[p])) / k->D[i];
    dv[6 + i - 1] = qdd;
    for (int c = 0; c < 6; c++) acc[i][c] = acc[p][c] + k->S[i][c] * qdd;
  }
}

/* ------------------------------------------------------------------ contacts */
typedef struct { int body; int vert; real x[3]; real dist; } Contact;

/* Hull vertices against the half-space z <= floor_z.  A vertex is a candidate when its distance
 * to the plane is below contact_margin.  Per body at most K points are kept, K =
 * clamp(max_contacts / (#bodies with candidates), 1, 4): deepest; farthest from it in the plane;
 * farthest from that line; farthest on the other side of the line.  Ties -> lowest vertex index.
 * With MORE touching bodies than max_contacts the budget goes to the bodies whose deepest vertex is
 * deepest (ties -> lower body index), one point each, emitted in body order - not to the lowest body
 * indices, which would leave a whole leg without contact rows (no counterpart in Bullet, which has no
 * such budget). */
static int generate_contacts(const Model *m, const State *s, const Work *k, Contact *out) {
  const real margin = m->prm[P_CONTACT_MARGIN], fz = m->prm[P_FLOOR_Z];
  int maxc = (int)m->prm[P_MAX_CONTACTS];
  if (maxc > MAXC) maxc = MAXC;
  int active[NBMAX], n_active = 0;
  real deepest[NBMAX];
  for (int b = 0; b < m->nb; b++) {
    active[b] = 0;
    deepest[b] = 0;
    int n = m->hull_start[b + 1] - m->hull_start[b];
    if (!n) continue;
    real c[3];
    matvec3(k->R[b], m->sphere_c[b], c);
    real cz = s->pos[2] + k->r[b][2] + c[2];
    if (cz - m->sphere_r[b] - fz >= margin) continue;
    const real *Rz = k->R[b] + 6;
    real z0 = s->pos[2] + k->r[b][2] - fz;
    for (int v = m->hull_start[b]; v < m->hull_start[b + 1]; v++) {
      real d = z0 + dot3(Rz, m->hull + 3 * v) - m->hull_r[v];
      if (d < margin && (!active[b] || d < deepest[b])) { active[b] = 1; deepest[b] = d; }
    }
    n_active += active[b];
  }
  if (!n_active) return 0;
  int K = maxc / n_active;
  if (K > 4) K = 4;
  if (K < 1) K = 1;
  if (n_active > maxc) {
    for (int b = 0; b < m->nb; b++) {
      if (!active[b]) continue;
      int rank = 0;
      for (int b2 = 0; b2 < m->nb; b2++)
        if (active[b2] && (deepest[b2] < deepest[b] || (deepest[b2] == deepest[b] && b2 < b))) rank++;
      if (rank >= maxc) active[b] = 2; /* over budget */
    }
    for (int b = 0; b < m->nb; b++) if (active[b] == 2) active[b] = 0;
  }
  int nc = 0;
  for (int b = 0; b < m->nb && nc < maxc; b++) {
    if (!active[b]) continue;
    int v0 = m->hull_start[b], v1 = m->hull_start[b + 1];
    int sel[4], nsel = 0;
    real px[4][3], pd[4];
    /* pass 1: deepest */
    int best = -1; real bd = 0;
    int ncand = 0;
    for (int v = v0; v < v1; v++) {
      real w[3];
      matvec3(k->R[b], m->hull + 3 * v, w);
      real d = s->pos[2] + k->r[b][2] + w[2] - m->hull_r[v] - fz;
      if (d < margin) { ncand++; if (best < 0 || d < bd) { best = v; bd = d; } }
    }
    (void)ncand;
    for (int pass = 0; pass < K && pass < 4; pass++) {
      int bi = -1; real bs = 0, bx[3] = {0, 0, 0}, bdist = 0;
      for (int v = v0; v < v1; v++) {
        real w[3];
        matvec3(k->R[b], m->hull + 3 * v, w);
        /* contact point = lowest point of the sphere around the (transformed) vertex */
        real x[3] = {k->r[b][0] + w[0], k->r[b][1] + w[1], k->r[b][2] + w[2] - m->hull_r[v]};
        real d = s->pos[2] + x[2] - fz;
        if (!(d < margin)) continue;
        int dup = 0;
        for (int q = 0; q < nsel; q++) dup |= (sel[q] == v);
        if (dup) continue;
        real score;
        if (pass == 0) score = -d;
        else if (pass == 1) {
          real dx = x[0] - px[0][0], dy = x[1] - px[0][1];
          score = dx * dx + dy * dy;
        } else {
          real ex = px[1][0] - px[0][0], ey = px[1][1] - px[0][1];
          real dx = x[0] - px[0][0], dy = x[1] - px[0][1];
          real cr = ex * dy - ey * dx;
          if (pass == 2) score = (real)fabs(cr);
          else {
            real e3x = px[2][0] - px[0][0], e3y = px[2][1] - px[0][1];
            real c3 = ex * e3y - ey * e3x;
            score = (c3 > 0) ? -cr : cr;
          }
        }
        if (bi < 0 || score > bs) { bi = v; bs = score; bx[0] = x[0]; bx[1] = x[1]; bx[2] = x[2]; bdist = d; }
      }
      if (bi < 0) break;
      if (pass >= 1 && !(bs > 0)) break; /* degenerate: nothing farther / nothing on the other side */
      sel[nsel] = bi; px[nsel][0] = bx[0]; px[nsel][1] = bx[1]; px[nsel][2] = bx[2]; pd[nsel] = bdist;
      nsel++;
    }
    for (int q = 0; q < nsel && nc < maxc; q++) {
      out[nc].body = b; out[nc].vert = sel[q]; out[nc].dist = pd[q];
      out[nc].x[0] = px[q][0]; out[nc].x[1] = px[q][1]; out[nc].x[2] = px[q][2];
      nc++;
    }
  }
  return nc;
}

/* ------------------------------------------------------------------ constraint rows + PGS */
typedef struct {
  real J[NDOF_MAX], W[NDOF_MAX];
  real inv_diag, rhs, lo, hi, lambda;
  int friction_of; /* row index of the normal row that bounds this friction row, else -1 */
  int kind, joint;  /* 0 limit, 1 motor, 2 normal, 3 friction */
} Row;

static void contact_jacobian(const Model *m, const Work *k, int body, const real *x, const real *d, real *J) {
  int nd = 6 + m->nb - 1;
  for (int c = 0; c < nd; c++) J[c] = 0;
  real F[6];
  cross3(x, d, F); F[3] = d[0]; F[4] = d[1]; F[5] = d[2];
  for (int c = 0; c < 6; c++) J[c] = F[c];
  for (int i = body; i >= 1; i = m->parent[i]) J[6 + i - 1] = dot6(k->S[i], F);
}

static void finish_row(const Model *m, Work *k, Row *r, const real *vgen, real target_vel) {
  int nd = 6 + m->nb - 1;
  apply_minv(m, k, r->J, r->W);
  real diag = 0, jv = 0;
  for (int c = 0; c < nd; c++) { diag += r->J[c] * r->W[c]; jv += r->J[c] * vgen[c]; }
  r->inv_diag = 1 / diag;
  r->rhs = (target_vel - jv) * r->inv_diag;
  r->lambda = 0;
}

#ifdef ORACLE_ROWS_HOOK
typedef void (*oracle_rows_hook_t)(int nr, const double *B, const double *rowdata, double friction);
static oracle_rows_hook_t oracle_rows_hook = 0;
__attribute__((visibility("default"))) void oracle_set_rows_hook(oracle_rows_hook_t f) { oracle_rows_hook = f; }
#endif

/* One physics substep = one pybullet stepSimulation() at dt (trex_env.py:150). target[] is the
 * position-motor target per BODY index (ignored when !motors_on). */
static void substep(const Model *m, State *s, const real *target) {
  static Work wk; /* oracle is single-threaded */
  static Row rows[MAXROWS];
  Work *k = &wk;
  const real dt = m->prm[P_DT];
  const int nb = m->nb, nd = 6 + nb - 1;
  kinematics(m, s, k);
  velocities(m, s, k);
  spatial_inertias(m, s, k);
  articulated_inertias(m, k);

  /* bias forces: velocity-product terms, gravity, link damping; joint torques: joint damping */
  real pA[NBMAX][6], cvec[NBMAX][6], tau[NBMAX];
  const real kd = m->prm[P_LINK_DAMPING];
  for (int i = 0; i < nb; i++) {
    real h[6];
    mat6vec(k->I[i], k->vel[i], h);
    crf(k->vel[i], h, pA[i]);
    real Ic[9], ms;
    body_inertia_world(m, s, k, i, Ic, &ms);
    real f[3] = {0, 0, -ms * m->prm[P_GRAVITY]}, n[3] = {0, 0, 0};
    if (kd > 0) {
      real vc[3], wxc[3];
      cross3(k->vel[i], k->comw[i], wxc);
      for (int c = 0; c < 3; c++) vc[c] = k->vel[i][3 + c] + wxc[c];
      real sv = (real)sqrt(dot3(vc, vc)), sw = (real)sqrt(dot3(k->vel[i], k->vel[i]));
      real Iw[3];
      matvec3(Ic, k->vel[i], Iw);
      for (int c = 0; c < 3; c++) {
        f[c] -= ms * vc[c] * (kd + kd * sv);
        n[c] -= Iw[c] * (kd + kd * sw);
      }
    }
    real cxf[3];
    cross3(k->comw[i], f, cxf);
    for (int c = 0; c < 3; c++) { pA[i][c] -= n[c] + cxf[c]; pA[i][3 + c] -= f[c]; }
    if (i >= 1) {
      real sq[6];
      for (int c = 0; c < 6; c++) sq[c] = k->S[i][c] * s->qd[i];
      crm(k->vel[i], sq, cvec[i]);
      tau[i] = -m->jdamp[i] * s->qd[i];
    }
  }
  real qdd[NBMAX], a0[6];
  aba_solve(m, k, pA, tau, (const real(*)[6])cvec, qdd, a0);

  /* unconstrained velocity update (classical base acceleration = spatial + w x v) */
  const real vmax = m->prm[P_MAX_COORD_VEL];
  real wxv[3];
  cross3(s->w, s->v, wxv);
  real vgen[NDOF_MAX];
  for (int c = 0; c < 3; c++) {
    vgen[c] = s->w[c] + a0[c] * dt;
    vgen[3 + c] = s->v[c] + (a0[3 + c] + wxv[c]) * dt;
  }
  for (int i = 1; i < nb; i++) vgen[6 + i - 1] = s->qd[i] + qdd[i] * dt;
  for (int c = 0; c < nd; c++) {
    if (vgen[c] > vmax) vgen[c] = vmax;
    if (vgen[c] < -vmax) vgen[c] = -vmax;
  }

  /* ---- rows: limits, motors, contacts (normal, friction x, friction y per point) */
  int nr = 0;
  int motor_row[NBMAX];
  s->n_limit_rows = 0;
  for (int i = 1; i < nb; i++) {
    real pen, dir;
    if (s->q[i] - m->q_lower[i] <= 0) { pen = s->q[i] - m->q_lower[i]; dir = 1; }
    else if (m->q_upper[i] - s->q[i] <= 0) { pen = m->q_upper[i] - s->q[i]; dir = -1; }
    else continue;
    Row *r = &rows[nr++];
    for (int c = 0; c < nd; c++) r->J[c] = 0;
    r->J[6 + i - 1] = dir;
    finish_row(m, k, r, vgen, -pen * m->prm[P_ERP] / dt);
    r->lo = 0; r->hi = (real)1e30; r->friction_of = -1; r->kind = 0; r->joint = i;
    s->n_limit_rows++;
  }
  for (int i = 1; i < nb; i++) {
    motor_row[i] = -1;
    if (!s->motors_on) continue;
    Row *r = &rows[nr];
    motor_row[i] = nr++;
    for (int c = 0; c < nd; c++) r->J[c] = 0;
    r->J[6 + i - 1] = 1;
    real qdi = vgen[6 + i - 1];
    /* btMultiBodyJointMotor: kp*erp(1)*(target-q)/dt + qd + kd*(0-qd)   */
    real tv = m->prm[P_MOTOR_KP] * (target[i] - s->q[i]) / dt + qdi + m->prm[P_MOTOR_KD] * (0 - qdi);
    finish_row(m, k, r, vgen, tv);
    r->hi = m->prm[P_MOTOR_MAX_FORCE] * dt; r->lo = -r->hi; r->friction_of = -1; r->kind = 1; r->joint = i;
  }
  Contact cts[MAXC];
  int nc = generate_contacts(m, s, k, cts);
  int contact_row[MAXC];
  for (int c = 0; c < nc; c++) {
    static const real dirs[3][3] = {{0, 0, 1}, {1, 0, 0}, {0, 1, 0}};
    contact_row[c] = nr;
    for (int a = 0; a < 3; a++) {
      Row *r = &rows[nr];
      contact_jacobian(m, k, cts[c].body, cts[c].x, dirs[a], r->J);
      real tv = 0;
      if (a == 0) { /* Bullet: penetration>0 -> allow approach dist/dt, else ERP push-out */
        real pen = cts[c].dist;
        tv = (pen > 0) ? -pen / dt : -pen * m->prm[P_CONTACT_ERP] / dt;
      }
      finish_row(m, k, r, vgen, tv);
      if (a == 0) { r->lo = 0; r->hi = (real)1e30; r->friction_of = -1; r->kind = 2; }
      else { r->lo = r->hi = 0; r->friction_of = contact_row[c]; r->kind = 3; }
      r->joint = cts[c].body;
      nr++;
    }
  }

#ifdef ORACLE_ROWS_HOOK
  /* study builds only (scripts/matrix_sweep_proto.py): hands the substep's constraint system to a callback - the Delassus
   * operator in the residual form of the GPU kernel, B[s][r] = -(J_s . W_r) / diag_s, and per row rhs, bounds, kind, friction_of */
  if (oracle_rows_hook) {
    static double hB[MAXROWS * MAXROWS], hrow[MAXROWS * 6];
    for (int a = 0; a < nr; a++) {
      for (int b = 0; b < nr; b++) {
        double t = 0;
        for (int c = 0; c < nd; c++) t += (double)rows[a].J[c] * (double)rows[b].W[c];
        hB[a * nr + b] = -t * (double)rows[a].inv_diag;
      }
      hrow[6 * a] = (double)rows[a].rhs; hrow[6 * a + 1] = (double)rows[a].lo; hrow[6 * a + 2] = (double)rows[a].hi;
      hrow[6 * a + 3] = rows[a].kind; hrow[6 * a + 4] = rows[a].friction_of; hrow[6 * a + 5] = rows[a].joint;
    }
    oracle_rows_hook(nr, hB, hrow, (double)s->friction);
  }
#endif
  /* ---- projected Gauss-Seidel on the velocity level */
  real dv[NDOF_MAX];
  for (int c = 0; c < nd; c++) dv[c] = 0;
  const int iters = (int)m->prm[P_ITERATIONS];
  for (int it = 0; it < iters; it++) {
    for (int ri = 0; ri < nr; ri++) {
      Row *r = &rows[ri];
      real jdv = 0;
      for (int c = 0; c < nd; c++) jdv += r->J[c] * dv[c];
      real lo = r->lo, hi = r->hi;
      if (r->friction_of >= 0) { hi = s->friction * rows[r->friction_of].lambda; lo = -hi; }
      real nl = r->lambda + (r->rhs - jdv * r->inv_diag);
      if (nl < lo) nl = lo;
      if (nl > hi) nl = hi;
      real d = nl - r->lambda;
      r->lambda = nl;
      for (int c = 0; c < nd; c++) dv[c] += d * r->W[c];
    }
  }
  for (int c = 0; c < nd; c++) vgen[c] += dv[c];
  for (int c = 0; c < 3; c++) { s->w[c] = vgen[c]; s->v[c] = vgen[3 + c]; }
  for (int i = 1; i < nb; i++) {
    s->qd[i] = vgen[6 + i - 1];
    s->motor_tau[i] = (motor_row[i] >= 0) ? rows[motor_row[i]].lambda / dt : 0;
  }
  s->n_contacts = nc;
  for (int c = 0; c < nc; c++) {
    s->contact_body[c] = (real)cts[c].body;
    s->contact_dist[c] = cts[c].dist;
    for (int a = 0; a < 3; a++) {
      s->contact_lambda[c][a] = rows[contact_row[c] + a].lambda;
      s->contact_pos[c][a] = cts[c].x[a] + s->pos[a];
    }
  }

  /* ---- integrate positions with the new velocities */
  for (int i = 1; i < nb; i++) s->q[i] += s->qd[i] * dt;
  for (int c = 0; c < 3; c++) s->pos[c] += s->v[c] * dt;
  real wn = (real)sqrt(dot3(s->w, s->w)), th = wn * dt;
  real dq[4] = {0, 0, 0, 1};
  if (th > (real)1e-12) {
    real sh = (real)sin(th / 2) / wn;
    dq[0] = s->w[0] * sh; dq[1] = s->w[1] * sh; dq[2] = s->w[2] * sh; dq[3] = (real)cos(th / 2);
  }
  /* world-frame angular velocity: q <- dq (x) q */
  real *q = s->quat, o[4];
  o[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
  o[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
  o[1] = dq[3] * q[1] - dq[0] * q[2] + dq[1] * q[3] + dq[2] * q[0];
  o[2] = dq[3] * q[2] + dq[0] * q[1] - dq[1] * q[0] + dq[2] * q[3];
  real qn = (real)sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
  for (int c = 0; c < 4; c++) q[c] = o[c] / qn;
}

/* ------------------------------------------------------------------ public (ctypes) API */
#define API __attribute__((visibility("default")))

API int oracle_real_size(void) { return (int)sizeof(real); }
API int oracle_param_count(void) { return P_COUNT; }
API int oracle_state_size(void) { return (int)sizeof(State); }

API Model *oracle_model_create(int nb, const int *parent, const double *axis, const double *jpos,
                               const double *jrot, const double *q_lower, const double *q_upper,
                               const double *jdamp, const double *mass, const double *com,
                               const double *inertia, const int *obs_order, int head_body,
                               const double *head_point, int nv, const double *hull, const double *hull_radius,
                               const int *hull_start, const double *sphere_c, const double *sphere_r,
                               const double *q_start, const double *base_pos0, const double *base_quat0,
                               const double *params) {
  if (nb > NBMAX) return NULL;
  Model *m = (Model *)calloc(1, sizeof(Model));
  m->nb = nb;
  for (int i = 0; i < nb; i++) {
    m->parent[i] = parent[i];
    m->depth[i] = (i == 0) ? 0 : m->depth[parent[i]] + 1;
    for (int c = 0; c < 3; c++) {
      m->axis[i][c] = (real)axis[3 * i + c]; m->jpos[i][c] = (real)jpos[3 * i + c];
      m->com[i][c] = (real)com[3 * i + c]; m->sphere_c[i][c] = (real)sphere_c[3 * i + c];
    }
    for (int c = 0; c < 9; c++) m->jrot[i][c] = (real)jrot[9 * i + c];
    for (int c = 0; c < 6; c++) m->inertia[i][c] = (real)inertia[6 * i + c];
    m->q_lower[i] = (real)q_lower[i]; m->q_upper[i] = (real)q_upper[i]; m->jdamp[i] = (real)jdamp[i];
    m->mass[i] = (real)mass[i]; m->sphere_r[i] = (real)sphere_r[i]; m->q_start[i] = (real)q_start[i];
  }
  for (int i = 0; i < nb - 1; i++) m->obs_order[i] = obs_order[i];
  for (int i = 0; i <= nb; i++) m->hull_start[i] = hull_start[i];
  m->head_body = head_body;
  for (int c = 0; c < 3; c++) { m->head_point[c] = (real)head_point[c]; m->base_pos0[c] = (real)base_pos0[c]; }
  for (int c = 0; c < 4; c++) m->base_quat0[c] = (real)base_quat0[c];
  m->nv = nv;
  m->hull = (real *)malloc(sizeof(real) * 3 * (nv ? nv : 1));
  for (int i = 0; i < 3 * nv; i++) m->hull[i] = (real)hull[i];
  m->hull_r = (real *)calloc(nv ? nv : 1, sizeof(real));
  if (hull_radius) for (int i = 0; i < nv; i++) m->hull_r[i] = (real)hull_radius[i];
  for (int i = 0; i < P_COUNT; i++) m->prm[i] = (real)params[i];
  return m;
}
API void oracle_model_destroy(Model *m) { if (m) { free(m->hull); free(m->hull_r); free(m); } }
API void oracle_model_set_param(Model *m, int idx, double v) { m->prm[idx] = (real)v; }

API State *oracle_state_create(const Model *m) {
  State *s = (State *)calloc(1, sizeof(State));
  for (int i = 0; i < NBMAX; i++) s->mass_scale[i] = 1;
  s->friction = m->prm[P_FRICTION];
  s->quat[3] = 1;
  return s;
}
API void oracle_state_destroy(State *s) { free(s); }
API void oracle_state_copy(State *dst, const State *src) { memcpy(dst, src, sizeof(State)); }

API void oracle_set_domain(const Model *m, State *s, const double *mass_scale, double friction) {
  for (int i = 0; i < m->nb; i++) s->mass_scale[i] = mass_scale ? (real)mass_scale[i] : 1;
  s->friction = (real)friction;
}

/* state vector layout (C-ABI order): pos3 quat4(xyzw) v3 w3 q[nj] qd[nj], joints in obs order */
API void oracle_get_state(const Model *m, const State *s, double *out) {
  int nj = m->nb - 1;
  for (int c = 0; c < 3; c++) { out[c] = s->pos[c]; out[7 + c] = s->v[c]; out[10 + c] = s->w[c]; }
  for (int c = 0; c < 4; c++) out[3 + c] = s->quat[c];
  for (int k = 0; k < nj; k++) { out[13 + k] = s->q[m->obs_order[k]]; out[13 + nj + k] = s->qd[m->obs_order[k]]; }
}
API void oracle_set_state(const Model *m, State *s, const double *in) {
  int nj = m->nb - 1;
  for (int c = 0; c < 3; c++) { s->pos[c] = (real)in[c]; s->v[c] = (real)in[7 + c]; s->w[c] = (real)in[10 + c]; }
  for (int c = 0; c < 4; c++) s->quat[c] = (real)in[3 + c];
  for (int k = 0; k < nj; k++) { s->q[m->obs_order[k]] = (real)in[13 + k]; s->qd[m->obs_order[k]] = (real)in[13 + nj + k]; }
}
API void oracle_set_motors_on(State *s, int on) { s->motors_on = on; }

API void oracle_observe(const Model *m, const State *s, double *obs) {
  int nj = m->nb - 1;
  for (int k = 0; k < nj; k++) {
    int b = m->obs_order[k];
    obs[k] = s->q[b]; obs[nj + k] = s->qd[b]; obs[2 * nj + k] = s->motor_tau[b];
  }
}

/* head COM in world coordinates (trex_robot.py:330-335) */
API void oracle_head_position(const Model *m, const State *s, double *out) {
  static Work wk;
  kinematics(m, s, &wk);
  real p[3];
  matvec3(wk.R[m->head_body], m->head_point, p);
  for (int c = 0; c < 3; c++) out[c] = s->pos[c] + wk.r[m->head_body][c] + p[c];
}

/* reward (trex_env.py:186-196); weights = {distance, energy, drift}; penalties out[3] =
 * {lifting_com, station_keeping, energy} as logged at trex_env.py:193-195 */
API double oracle_reward(const Model *m, const State *s, const double *weights, double *penalties) {
  double h[3];
  oracle_head_position(m, s, h);
  double power = 0;
  for (int i = 1; i < m->nb; i++) power += fabs((double)s->qd[i] * (double)s->motor_tau[i]);
  double lift = weights[0] * (2.5 - h[2]) * (2.5 - h[2]);
  double drift = weights[2] * (h[0] * h[0] + h[1] * h[1]);
  double energy = weights[1] * power;
  if (penalties) { penalties[0] = lift; penalties[1] = drift; penalties[2] = energy; }
  return -lift - drift - energy;
}

API void oracle_reset(const Model *m, State *s) {
  for (int c = 0; c < 3; c++) { s->pos[c] = m->base_pos0[c]; s->v[c] = 0; s->w[c] = 0; }
  for (int c = 0; c < 4; c++) s->quat[c] = m->base_quat0[c];
  for (int i = 0; i < NBMAX; i++) { s->q[i] = (i < m->nb) ? m->q_start[i] : 0; s->qd[i] = 0; s->motor_tau[i] = 0; }
  s->motors_on = 0;              /* remove_joint_control, trex_robot.py:309 */
  real dummy[NBMAX] = {0};
  substep(m, s, dummy);          /* trex_env.py:120 */
}

API void oracle_substep(const Model *m, State *s, const double *target_obs_order) {
  real tgt[NBMAX] = {0};
  if (target_obs_order)
    for (int k = 0; k < m->nb - 1; k++) tgt[m->obs_order[k]] = (real)target_obs_order[k];
  substep(m, s, tgt);
}

/* TrexBulletEnv.step (trex_env.py:128-154) */
API void oracle_step(const Model *m, State *s, const double *action, const double *weights,
                     double *obs, double *reward, double *penalties) {
  real tgt[NBMAX] = {0};
  for (int k = 0; k < m->nb - 1; k++) {
    int b = m->obs_order[k];
    double a = action[k];
    if (a < m->q_lower[b]) a = m->q_lower[b]; /* np.clip to action_space, trex_env.py:147 */
    if (a > m->q_upper[b]) a = m->q_upper[b];
    tgt[b] = (real)a;
  }
  s->motors_on = 1;
  int n = (int)m->prm[P_SUBSTEPS];
  for (int i = 0; i < n; i++) substep(m, s, tgt);
  if (obs) oracle_observe(m, s, obs);
  double r = oracle_reward(m, s, weights, penalties);
  if (reward) *reward = r;
}

/* ---- diagnostics for the invariant tests ---- */
/* forward dynamics only: qdd (obs order) and classical base acceleration for given tau (obs order) */
API void oracle_forward_dynamics(const Model *m, const State *s, const double *tau_obs, int with_damping,
                                 double *qdd_out, double *base_acc_out) {
  static Work wk;
  Work *k = &wk;
  kinematics(m, s, k); velocities(m, s, k); spatial_inertias(m, s, k); articulated_inertias(m, k);
  real pA[NBMAX][6], cvec[NBMAX][6], tau[NBMAX] = {0};
  for (int i = 0; i < m->nb; i++) {
    real h[6];
    mat6vec(k->I[i], k->vel[i], h);
    crf(k->vel[i], h, pA[i]);
    real Ic[9], ms;
    body_inertia_world(m, s, k, i, Ic, &ms);
    real f[3] = {0, 0, -ms * m->prm[P_GRAVITY]}, cxf[3];
    cross3(k->comw[i], f, cxf);
    for (int c = 0; c < 3; c++) { pA[i][c] -= cxf[c]; pA[i][3 + c] -= f[c]; }
    if (i >= 1) {
      real sq[6];
      for (int c = 0; c < 6; c++) sq[c] = k->S[i][c] * s->qd[i];
      crm(k->vel[i], sq, cvec[i]);
    }
  }
  for (int j = 0; j < m->nb - 1; j++) {
    int b = m->obs_order[j];
    tau[b] = (real)(tau_obs ? tau_obs[j] : 0) - (with_damping ? m->jdamp[b] * s->qd[b] : 0);
  }
  real qdd[NBMAX], a0[6], wxv[3];
  aba_solve(m, k, pA, tau, (const real(*)[6])cvec, qdd, a0);
  cross3(s->w, s->v, wxv);
  for (int j = 0; j < m->nb - 1; j++) qdd_out[j] = qdd[m->obs_order[j]];
  for (int c = 0; c < 3; c++) { base_acc_out[c] = a0[c]; base_acc_out[3 + c] = a0[3 + c] + wxv[c]; }
}

/* dense M^-1 in generalised coordinates [w(3), v(3), joints in BODY order 1..nb-1] */
API void oracle_minv(const Model *m, const State *s, double *out) {
  static Work wk;
  Work *k = &wk;
  int nd = 6 + m->nb - 1;
  kinematics(m, s, k); velocities(m, s, k); spatial_inertias(m, s, k); articulated_inertias(m, k);
  for (int c = 0; c < nd; c++) {
    real f[NDOF_MAX] = {0}, dv[NDOF_MAX];
    f[c] = 1;
    apply_minv(m, k, f, dv);
    for (int r = 0; r < nd; r++) out[r * nd + c] = dv[r];
  }
}

/* world pose of every body frame: pos[nb][3], rot[nb][9] */
API void oracle_body_poses(const Model *m, const State *s, double *pos, double *rot) {
  static Work wk;
  kinematics(m, s, &wk);
  for (int i = 0; i < m->nb; i++) {
    for (int c = 0; c < 3; c++) pos[3 * i + c] = s->pos[c] + wk.r[i][c];
    for (int c = 0; c < 9; c++) rot[9 * i + c] = wk.R[i][c];
  }
}

API int oracle_contacts(const State *s, double *body, double *lambda, double *pos, double *dist) {
  for (int c = 0; c < s->n_contacts; c++) {
    body[c] = s->contact_body[c]; dist[c] = s->contact_dist[c];
    for (int a = 0; a < 3; a++) { lambda[3 * c + a] = s->contact_lambda[c][a]; pos[3 * c + a] = s->contact_pos[c][a]; }
  }
  return s->n_contacts;
}
API int oracle_limit_rows(const State *s) { return s->n_limit_rows; }

/* total kinetic energy, potential energy, linear momentum (for invariant tests) */
API void oracle_energy(const Model *m, const State *s, double *out) {
  static Work wk;
  Work *k = &wk;
  kinematics(m, s, k); velocities(m, s, k); spatial_inertias(m, s, k);
  double ke = 0, pe = 0, mom[6] = {0};
  for (int i = 0; i < m->nb; i++) {
    real h[6];
    mat6vec(k->I[i], k->vel[i], h);
    ke += 0.5 * dot6(h, k->vel[i]);
    pe += m->mass[i] * s->mass_scale[i] * m->prm[P_GRAVITY] * (s->pos[2] + k->comw[i][2]);
    for (int c = 0; c < 6; c++) mom[c] += h[c];
  }
  out[0] = ke; out[1] = pe;
  for (int c = 0; c < 6; c++) out[2 + c] = mom[c]; /* spatial momentum about O */
}
