/* mjoracle.c -- CPU restatement of the physics step.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (mjlab_amd/csrc) never links or calls it.
 *
 * PARITY UNPINNED: the arithmetic of the reference's hot path lives in third-party
 * packages that are absent from /root/reference and from this image
 * (mujoco_warp @ 486642c3fa262a989b482e0e506716d5793d61a9, mujoco 3.3.7.dev811775910;
 * reference pyproject.toml:94-96, call sites src/mjlab/sim/sim.py:136,139,187,195).
 * This file restates MuJoCo's published forward-dynamics pipeline (mj_step:
 * engine_forward.c / engine_core_smooth.c / engine_core_constraint.c /
 * engine_collision_primitive.c / engine_solver.c of the mujoco 3.3 line) from its
 * documentation and public algorithm descriptions; it is validated against analytic
 * cases and physical invariants (tests/test_oracle_physics.py), not against golden
 * vectors of the reference, which has none.
 *
 * Stage names follow the reference's stub docstrings
 * (typings/mujoco/_functions.pyi: mj_kinematics :803, mj_comPos :358, mj_crb :399,
 * mj_factorM :449, mj_collision :353, mj_makeConstraint :835, mj_comVel :363,
 * mj_passive :957, mj_rne :1070, mj_fwdActuation :493, mj_fwdAcceleration :488,
 * mj_fwdConstraint :498, mj_sensorAcc :1104, mj_implicit :555, mj_Euler :919).
 *
 * Build: gcc -O2 -shared -fPIC [-DMJO_FLOAT] (see oracle/Makefile).
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifdef MJO_FLOAT
#define MJLAB_REAL float
#else
#define MJLAB_REAL double
#endif
#define MJLAB_MODEL_T mjo_model_t
#define MJLAB_DATA_T mjo_data_t
#include "../include/mjlab_fields.h"

typedef MJLAB_REAL real;
#define MINVAL ((real)1e-15)
#define MINIMP ((real)0.0001)
#define MAXIMP ((real)0.9999)

#define MF(name, w) (m->name + (size_t)(w) * (size_t)m->name##_ws)

const char* mjo_model_layout(void) { return MJLAB_MODEL_LAYOUT_STRING; }
const char* mjo_data_layout(void) { return MJLAB_DATA_LAYOUT_STRING; }
int mjo_sizeof_real(void) { return (int)sizeof(real); }
int mjo_sizeof_model(void) { return (int)sizeof(mjo_model_t); }
int mjo_sizeof_data(void) { return (int)sizeof(mjo_data_t); }

/* ------------------------------------------------------------------ small math */
static inline real dot3(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(real* r, const real* a, const real* b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline real normalize3(real* v) {
  real n = sqrt(dot3(v, v));
  if (n < MINVAL) { v[0] = 1; v[1] = 0; v[2] = 0; return 0; }
  v[0] /= n; v[1] /= n; v[2] /= n;
  return n;
}
static inline void normalize4(real* q) {
  real n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static inline void mul_quat(real* r, const real* a, const real* b) {
  real w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  real x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  real y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  real z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static inline void quat2mat(real* R, const real* q) {
  real q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  real q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
  real q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  R[0] = q00 + q11 - q22 - q33; R[4] = q00 - q11 + q22 - q33; R[8] = q00 - q11 - q22 + q33;
  R[1] = 2 * (q12 - q03); R[2] = 2 * (q13 + q02);
  R[3] = 2 * (q12 + q03); R[5] = 2 * (q23 - q01);
  R[6] = 2 * (q13 - q02); R[7] = 2 * (q23 + q01);
}
static inline void rot_vec_quat(real* r, const real* v, const real* q) {
  real R[9];
  quat2mat(R, q);
  real x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  real y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  real z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void mul_mat_vec3(real* r, const real* R, const real* v) {
  real x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  real y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  real z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void axis_angle2quat(real* q, const real* axis, real angle) {
  real s = sin(angle * (real)0.5);
  q[0] = cos(angle * (real)0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
static inline real clipr(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* spatial helpers; motion vectors are [angular(3), linear(3)] about the com-frame origin */
static void mul_inert_vec(real* r, const real* i, const real* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
static void cross_motion(real* r, const real* vel, const real* v) {
  r[0] = -vel[2] * v[1] + vel[1] * v[2];
  r[1] = vel[2] * v[0] - vel[0] * v[2];
  r[2] = -vel[1] * v[0] + vel[0] * v[1];
  r[3] = -vel[2] * v[4] + vel[1] * v[5] - vel[5] * v[1] + vel[4] * v[2];
  r[4] = vel[2] * v[3] - vel[0] * v[5] + vel[5] * v[0] - vel[3] * v[2];
  r[5] = -vel[1] * v[3] + vel[0] * v[4] - vel[4] * v[0] + vel[3] * v[1];
}
static void cross_force(real* r, const real* vel, const real* f) {
  r[0] = -vel[2] * f[1] + vel[1] * f[2] - vel[5] * f[4] + vel[4] * f[5];
  r[1] = vel[2] * f[0] - vel[0] * f[2] + vel[5] * f[3] - vel[3] * f[5];
  r[2] = -vel[1] * f[0] + vel[0] * f[1] - vel[4] * f[3] + vel[3] * f[4];
  r[3] = -vel[2] * f[4] + vel[1] * f[5];
  r[4] = vel[2] * f[3] - vel[0] * f[5];
  r[5] = -vel[1] * f[3] + vel[0] * f[4];
}

/* per-world views */
#define D(name, n) (d->name + (size_t)w * (size_t)(n))

/* ------------------------------------------------------------------ position stage */
static void local2global(real* xp, real* xm, const real* bpos, const real* bquat, const real* bmat,
                         const real* pos, const real* quat) {
  real q[4], t[3];
  mul_mat_vec3(t, bmat, pos);
  xp[0] = bpos[0] + t[0]; xp[1] = bpos[1] + t[1]; xp[2] = bpos[2] + t[2];
  mul_quat(q, bquat, quat);
  quat2mat(xm, q);
}

static void kinematics(const mjo_model_t* m, mjo_data_t* d, int w) {
  const mjlab_sizes_t* s = &m->size;
  real* qpos = D(qpos, s->nq);
  real *xpos = D(xpos, 3 * s->nbody), *xquat = D(xquat, 4 * s->nbody), *xmat = D(xmat, 9 * s->nbody);
  real *xipos = D(xipos, 3 * s->nbody), *ximat = D(ximat, 9 * s->nbody);
  real *xanchor = D(xanchor, 3 * s->njnt), *xaxis = D(xaxis, 3 * s->njnt);
  const real* qpos0 = MF(qpos0, w);
  /* world */
  xpos[0] = xpos[1] = xpos[2] = 0;
  xquat[0] = 1; xquat[1] = xquat[2] = xquat[3] = 0;
  quat2mat(xmat, xquat);
  for (int i = 1; i < s->nbody; i++) {
    int pid = m->body_parentid[i], ja = m->body_jntadr[i], jn = m->body_jntnum[i];
    real pos[3], quat[4];
    if (jn == 1 && m->jnt_type[ja] == MJLAB_JNT_FREE) {
      int qa = m->jnt_qposadr[ja];
      memcpy(pos, qpos + qa, 3 * sizeof(real));
      memcpy(quat, qpos + qa + 3, 4 * sizeof(real));
      normalize4(quat);
      memcpy(xanchor + 3 * ja, pos, 3 * sizeof(real));
      memcpy(xaxis + 3 * ja, MF(jnt_axis, w) + 3 * ja, 3 * sizeof(real));
    } else {
      real t[3];
      mul_mat_vec3(t, xmat + 9 * pid, MF(body_pos, w) + 3 * i);
      pos[0] = xpos[3 * pid] + t[0]; pos[1] = xpos[3 * pid + 1] + t[1]; pos[2] = xpos[3 * pid + 2] + t[2];
      mul_quat(quat, xquat + 4 * pid, MF(body_quat, w) + 4 * i);
      for (int j = ja; j < ja + jn; j++) {
        int qa = m->jnt_qposadr[j];
        const real *jaxis = MF(jnt_axis, w) + 3 * j, *jpos = MF(jnt_pos, w) + 3 * j;
        rot_vec_quat(xaxis + 3 * j, jaxis, quat);
        rot_vec_quat(t, jpos, quat);
        xanchor[3 * j] = t[0] + pos[0]; xanchor[3 * j + 1] = t[1] + pos[1]; xanchor[3 * j + 2] = t[2] + pos[2];
        if (m->jnt_type[j] == MJLAB_JNT_SLIDE) {
          real dq = qpos[qa] - qpos0[qa];
          pos[0] += xaxis[3 * j] * dq; pos[1] += xaxis[3 * j + 1] * dq; pos[2] += xaxis[3 * j + 2] * dq;
        } else { /* hinge */
          real ql[4], qn[4];
          axis_angle2quat(ql, jaxis, qpos[qa] - qpos0[qa]);
          mul_quat(qn, quat, ql);
          memcpy(quat, qn, sizeof(qn));
          rot_vec_quat(t, jpos, quat);
          pos[0] = xanchor[3 * j] - t[0]; pos[1] = xanchor[3 * j + 1] - t[1]; pos[2] = xanchor[3 * j + 2] - t[2];
        }
      }
    }
    normalize4(quat);
    memcpy(xpos + 3 * i, pos, sizeof(pos));
    memcpy(xquat + 4 * i, quat, sizeof(quat));
    quat2mat(xmat + 9 * i, quat);
  }
  for (int i = 0; i < s->nbody; i++)
    local2global(xipos + 3 * i, ximat + 9 * i, xpos + 3 * i, xquat + 4 * i, xmat + 9 * i,
                 MF(body_ipos, w) + 3 * i, MF(body_iquat, w) + 4 * i);
  /* geoms of static bodies (the first nstaticgeom) are posed once, by mjo_static_geoms() */
  real *gx = D(geom_xpos, 3 * s->ngeom), *gm = D(geom_xmat, 9 * s->ngeom);
  for (int g = s->nstaticgeom; g < s->ngeom; g++) {
    int b = m->geom_bodyid[g];
    local2global(gx + 3 * g, gm + 9 * g, xpos + 3 * b, xquat + 4 * b, xmat + 9 * b, MF(geom_pos, w) + 3 * g,
                 MF(geom_quat, w) + 4 * g);
  }
  real *sx = D(site_xpos, 3 * s->nsite), *sm = D(site_xmat, 9 * s->nsite);
  for (int g = s->nstaticsite; g < s->nsite; g++) { /* static sites: posed once, like the static geoms */
    int b = m->site_bodyid[g];
    local2global(sx + 3 * g, sm + 9 * g, xpos + 3 * b, xquat + 4 * b, xmat + 9 * b, MF(site_pos, w) + 3 * g,
                 MF(site_quat, w) + 4 * g);
  }
}

static void com_pos(const mjo_model_t* m, mjo_data_t* d, int w) {
  const mjlab_sizes_t* s = &m->size;
  int nb = s->nbody;
  real *xipos = D(xipos, 3 * nb), *ximat = D(ximat, 9 * nb), *sub = D(subtree_com, 3 * nb);
  real *cinert = D(cinert, 10 * nb), *cdof = D(cdof, 6 * s->nv);
  real *xanchor = D(xanchor, 3 * s->njnt), *xaxis = D(xaxis, 3 * s->njnt), *xmat = D(xmat, 9 * nb);
  const real *mass = MF(body_mass, w), *stm = MF(body_subtreemass, w), *inertia = MF(body_inertia, w);
  for (int i = 0; i < nb; i++)
    for (int k = 0; k < 3; k++) sub[3 * i + k] = mass[i] * xipos[3 * i + k];
  for (int i = nb - 1; i > 0; i--)
    for (int k = 0; k < 3; k++) sub[3 * m->body_parentid[i] + k] += sub[3 * i + k];
  for (int i = 0; i < nb; i++) {
    if (stm[i] < MINVAL) memcpy(sub + 3 * i, xipos + 3 * i, 3 * sizeof(real));
    else for (int k = 0; k < 3; k++) sub[3 * i + k] /= stm[i];
  }
  memset(cinert, 0, 10 * sizeof(real));
  for (int i = 1; i < nb; i++) {
    const real *mat = ximat + 9 * i, *in = inertia + 3 * i;
    real dif[3], tmp[9], *res = cinert + 10 * i, ms = mass[i];
    for (int k = 0; k < 3; k++) dif[k] = xipos[3 * i + k] - sub[3 * m->body_rootid[i] + k];
    tmp[0] = mat[0] * in[0]; tmp[1] = mat[3] * in[0]; tmp[2] = mat[6] * in[0];
    tmp[3] = mat[1] * in[1]; tmp[4] = mat[4] * in[1]; tmp[5] = mat[7] * in[1];
    tmp[6] = mat[2] * in[2]; tmp[7] = mat[5] * in[2]; tmp[8] = mat[8] * in[2];
    res[0] = mat[0] * tmp[0] + mat[1] * tmp[3] + mat[2] * tmp[6];
    res[1] = mat[3] * tmp[1] + mat[4] * tmp[4] + mat[5] * tmp[7];
    res[2] = mat[6] * tmp[2] + mat[7] * tmp[5] + mat[8] * tmp[8];
    res[3] = mat[0] * tmp[1] + mat[1] * tmp[4] + mat[2] * tmp[7];
    res[4] = mat[0] * tmp[2] + mat[1] * tmp[5] + mat[2] * tmp[8];
    res[5] = mat[3] * tmp[2] + mat[4] * tmp[5] + mat[5] * tmp[8];
    res[0] += ms * (dif[1] * dif[1] + dif[2] * dif[2]);
    res[1] += ms * (dif[0] * dif[0] + dif[2] * dif[2]);
    res[2] += ms * (dif[0] * dif[0] + dif[1] * dif[1]);
    res[3] -= ms * dif[0] * dif[1];
    res[4] -= ms * dif[0] * dif[2];
    res[5] -= ms * dif[1] * dif[2];
    res[6] = ms * dif[0]; res[7] = ms * dif[1]; res[8] = ms * dif[2];
    res[9] = ms;
  }
  for (int j = 0; j < s->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    real off[3];
    for (int k = 0; k < 3; k++) off[k] = sub[3 * m->body_rootid[b] + k] - xanchor[3 * j + k];
    switch (m->jnt_type[j]) {
      case MJLAB_JNT_FREE:
        memset(cdof + 6 * da, 0, 18 * sizeof(real));
        for (int k = 0; k < 3; k++) cdof[6 * (da + k) + 3 + k] = 1;
        for (int k = 0; k < 3; k++) {
          real ax[3] = {xmat[9 * b + k], xmat[9 * b + 3 + k], xmat[9 * b + 6 + k]};
          real* c = cdof + 6 * (da + 3 + k);
          memcpy(c, ax, sizeof(ax));
          cross3(c + 3, ax, off);
        }
        break;
      case MJLAB_JNT_SLIDE:
        cdof[6 * da] = cdof[6 * da + 1] = cdof[6 * da + 2] = 0;
        memcpy(cdof + 6 * da + 3, xaxis + 3 * j, 3 * sizeof(real));
        break;
      default: /* hinge */
        memcpy(cdof + 6 * da, xaxis + 3 * j, 3 * sizeof(real));
        cross3(cdof + 6 * da + 3, xaxis + 3 * j, off);
    }
  }
}

/* dense Cholesky A = L L^T in place (lower), returns rank deficiency count */
static int chol_factor(real* A, int n) {
  int bad = 0;
  for (int j = 0; j < n; j++) {
    real t = A[j * n + j];
    for (int k = 0; k < j; k++) t -= A[j * n + k] * A[j * n + k];
    if (t < MINVAL) { t = MINVAL; bad++; }
    t = sqrt(t);
    A[j * n + j] = t;
    for (int i = j + 1; i < n; i++) {
      real v = A[i * n + j];
      for (int k = 0; k < j; k++) v -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = v / t;
    }
  }
  return bad;
}
static void chol_solve(const real* L, int n, real* x) {
  for (int i = 0; i < n; i++) {
    real t = x[i];
    for (int k = 0; k < i; k++) t -= L[i * n + k] * x[k];
    x[i] = t / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    real t = x[i];
    for (int k = i + 1; k < n; k++) t -= L[k * n + i] * x[k];
    x[i] = t / L[i * n + i];
  }
}

static void crb_factor(const mjo_model_t* m, mjo_data_t* d, int w) {
  const mjlab_sizes_t* s = &m->size;
  int nb = s->nbody, nv = s->nv;
  real *cinert = D(cinert, 10 * nb), *cdof = D(cdof, 6 * nv), *M = D(qM, nv * nv), *L = D(qLD, nv * nv);
  real* crb = (real*)malloc(sizeof(real) * 10 * nb);
  memcpy(crb, cinert, sizeof(real) * 10 * nb);
  for (int i = nb - 1; i > 0; i--) {
    int p = m->body_parentid[i];
    if (p > 0) for (int k = 0; k < 10; k++) crb[10 * p + k] += crb[10 * i + k];
  }
  memset(M, 0, sizeof(real) * nv * nv);
  const real* arm = MF(dof_armature, w);
  for (int i = 0; i < nv; i++) {
    real buf[6];
    mul_inert_vec(buf, crb + 10 * m->dof_bodyid[i], cdof + 6 * i);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      real v = 0;
      for (int k = 0; k < 6; k++) v += cdof[6 * j + k] * buf[k];
      M[i * nv + j] = v;
      M[j * nv + i] = v;
    }
    M[i * nv + i] += arm[i];
  }
  free(crb);
  memcpy(L, M, sizeof(real) * nv * nv);
  chol_factor(L, nv);
  for (int i = 0; i < nv; i++) for (int j = i + 1; j < nv; j++) L[i * nv + j] = 0;
}

/* ------------------------------------------------------------------ collision */
typedef struct { real dist, pos[3], frame[9]; } rawcon_t;

static void make_frame(real* f) {
  /* f[0:3] = normal (unit). Build tangents as mju_makeFrame does. */
  real y[3] = {f[3], f[4], f[5]};
  if (sqrt(dot3(y, y)) < (real)0.5) {
    y[0] = y[1] = y[2] = 0;
    if (f[1] < (real)0.5 && f[1] > (real)-0.5) y[1] = 1; else y[2] = 1;
  }
  real t = dot3(f, y);
  y[0] -= t * f[0]; y[1] -= t * f[1]; y[2] -= t * f[2];
  normalize3(y);
  f[3] = y[0]; f[4] = y[1]; f[5] = y[2];
  cross3(f + 6, f, y);
}

static int plane_sphere(rawcon_t* c, real margin, const real* ppos, const real* pmat, const real* spos, real r) {
  real n[3] = {pmat[2], pmat[5], pmat[8]}, dif[3] = {spos[0] - ppos[0], spos[1] - ppos[1], spos[2] - ppos[2]};
  real cdist = dot3(dif, n);
  if (cdist > margin + r) return 0;
  c->dist = cdist - r;
  for (int k = 0; k < 3; k++) c->pos[k] = spos[k] + n[k] * (-c->dist * (real)0.5 - r);
  memcpy(c->frame, n, sizeof(n));
  c->frame[3] = c->frame[4] = c->frame[5] = 0;
  return 1;
}
static int sphere_sphere(rawcon_t* c, real margin, const real* p1, real r1, const real* p2, real r2) {
  real dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  real cd2 = dot3(dif, dif), mn = margin + r1 + r2;
  if (cd2 > mn * mn) return 0;
  real len = sqrt(cd2);
  if (len < MINVAL) { dif[0] = 1; dif[1] = dif[2] = 0; }
  else { dif[0] /= len; dif[1] /= len; dif[2] /= len; }
  c->dist = len - r1 - r2;
  for (int k = 0; k < 3; k++) c->pos[k] = p1[k] + dif[k] * (r1 + c->dist * (real)0.5);
  memcpy(c->frame, dif, sizeof(dif));
  c->frame[3] = c->frame[4] = c->frame[5] = 0;
  return 1;
}
static int plane_capsule(rawcon_t* c, real margin, const real* ppos, const real* pmat, const real* cpos,
                         const real* cmat, const real* size) {
  real axis[3] = {cmat[2], cmat[5], cmat[8]}, p[3];
  int n = 0;
  for (int k = 0; k < 3; k++) p[k] = cpos[k] + axis[k] * size[1];
  int n1 = plane_sphere(c + n, margin, ppos, pmat, p, size[0]);
  if (n1) { memcpy(c[n].frame + 3, axis, sizeof(axis)); n++; }
  for (int k = 0; k < 3; k++) p[k] = cpos[k] - axis[k] * size[1];
  int n2 = plane_sphere(c + n, margin, ppos, pmat, p, size[0]);
  if (n2) { memcpy(c[n].frame + 3, axis, sizeof(axis)); n++; }
  return n;
}
static int plane_box(rawcon_t* c, real margin, const real* ppos, const real* pmat, const real* bpos, const real* bmat,
                     const real* size) {
  real n[3] = {pmat[2], pmat[5], pmat[8]}, dif[3] = {bpos[0] - ppos[0], bpos[1] - ppos[1], bpos[2] - ppos[2]};
  real dist = dot3(dif, n);
  int cnt = 0;
  for (int i = 0; i < 8; i++) {
    real vec[3] = {(i & 1) ? size[0] : -size[0], (i & 2) ? size[1] : -size[1], (i & 4) ? size[2] : -size[2]}, corner[3];
    mul_mat_vec3(corner, bmat, vec);
    real ldist = dot3(n, corner);
    if (dist + ldist > margin || ldist > 0) continue;
    c[cnt].dist = dist + ldist;
    for (int k = 0; k < 3; k++) c[cnt].pos[k] = corner[k] + bpos[k] + n[k] * (-c[cnt].dist * (real)0.5);
    memcpy(c[cnt].frame, n, sizeof(n));
    c[cnt].frame[3] = c[cnt].frame[4] = c[cnt].frame[5] = 0;
    if (++cnt >= 4) return 4;
  }
  return cnt;
}
static int sphere_capsule(rawcon_t* c, real margin, const real* spos, real r, const real* cpos, const real* cmat,
                          const real* size) {
  real axis[3] = {cmat[2], cmat[5], cmat[8]}, vec[3] = {spos[0] - cpos[0], spos[1] - cpos[1], spos[2] - cpos[2]};
  real x = clipr(dot3(axis, vec), -size[1], size[1]);
  for (int k = 0; k < 3; k++) vec[k] = cpos[k] + axis[k] * x;
  return sphere_sphere(c, margin, spos, r, vec, size[0]);
}
static int capsule_capsule(rawcon_t* c, real margin, const real* pos1, const real* mat1, const real* size1,
                           const real* pos2, const real* mat2, const real* size2) {
  real axis1[3] = {mat1[2], mat1[5], mat1[8]}, axis2[3] = {mat2[2], mat2[5], mat2[8]};
  real dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  real ma = dot3(axis1, axis1), mb = -dot3(axis1, axis2), mc = dot3(axis2, axis2);
  real u = -dot3(axis1, dif), v = dot3(axis2, dif), det = ma * mc - mb * mb;
  real vec1[3], vec2[3];
  if (fabs(det) >= MINVAL) {
    real x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > size1[1]) { x1 = size1[1]; x2 = (v - mb * size1[1]) / mc; }
    else if (x1 < -size1[1]) { x1 = -size1[1]; x2 = (v + mb * size1[1]) / mc; }
    if (x2 > size2[1]) { x2 = size2[1]; x1 = clipr((u - mb * size2[1]) / ma, -size1[1], size1[1]); }
    else if (x2 < -size2[1]) { x2 = -size2[1]; x1 = clipr((u + mb * size2[1]) / ma, -size1[1], size1[1]); }
    for (int k = 0; k < 3; k++) { vec1[k] = pos1[k] + axis1[k] * x1; vec2[k] = pos2[k] + axis2[k] * x2; }
    return sphere_sphere(c, margin, vec1, size1[0], vec2, size2[0]);
  }
  /* parallel axes: test the segment ends, up to two contacts */
  int n = 0;
  real x2;
  for (int k = 0; k < 3; k++) vec1[k] = pos1[k] + axis1[k] * size1[1];
  x2 = clipr((v - mb * size1[1]) / mc, -size2[1], size2[1]);
  for (int k = 0; k < 3; k++) vec2[k] = pos2[k] + axis2[k] * x2;
  n += sphere_sphere(c + n, margin, vec1, size1[0], vec2, size2[0]);
  for (int k = 0; k < 3; k++) vec1[k] = pos1[k] - axis1[k] * size1[1];
  x2 = clipr((v + mb * size1[1]) / mc, -size2[1], size2[1]);
  for (int k = 0; k < 3; k++) vec2[k] = pos2[k] + axis2[k] * x2;
  n += sphere_sphere(c + n, margin, vec1, size1[0], vec2, size2[0]);
  if (n == 2) return n;
  real x1;
  for (int k = 0; k < 3; k++) vec2[k] = pos2[k] + axis2[k] * size2[1];
  x1 = clipr((u - mb * size2[1]) / ma, -size1[1], size1[1]);
  for (int k = 0; k < 3; k++) vec1[k] = pos1[k] + axis1[k] * x1;
  n += sphere_sphere(c + n, margin, vec1, size1[0], vec2, size2[0]);
  if (n == 2) return n;
  for (int k = 0; k < 3; k++) vec2[k] = pos2[k] - axis2[k] * size2[1];
  x1 = clipr((u + mb * size2[1]) / ma, -size1[1], size1[1]);
  for (int k = 0; k < 3; k++) vec1[k] = pos1[k] + axis1[k] * x1;
  n += sphere_sphere(c + n, margin, vec1, size1[0], vec2, size2[0]);
  return n;
}

/* Sphere vs box (MuJoCo's mjc_SphereBox semantics: engine_collision_box.c).  The sphere centre is
 * taken into the box frame and clamped to the box; outside, the normal runs along the
 * clamped-point -> centre direction, inside, through the nearest face.  The contact normal points
 * from the sphere (geom1) into the box (geom2); pos is midway between the two surfaces. */
static int sphere_box(rawcon_t* c, real margin, const real* spos, real r, const real* bpos, const real* bmat, const real* bsize) {
  real dif[3] = {spos[0] - bpos[0], spos[1] - bpos[1], spos[2] - bpos[2]}, loc[3], dv[3], nl[3], pl[3];
  for (int i = 0; i < 3; i++) loc[i] = bmat[i] * dif[0] + bmat[3 + i] * dif[1] + bmat[6 + i] * dif[2];
  for (int i = 0; i < 3; i++) dv[i] = loc[i] - clipr(loc[i], -bsize[i], bsize[i]);
  real d2 = dot3(dv, dv), mn = margin + r;
  if (d2 > mn * mn) return 0;
  if (d2 > 0) {
    real len = sqrt(d2);
    c->dist = len - r;
    for (int i = 0; i < 3; i++) { nl[i] = dv[i] / len; pl[i] = (loc[i] - dv[i]) + nl[i] * (c->dist * (real)0.5); }
  } else {
    /* centre inside the box: leave through the nearest face (first one on ties) */
    int k = 0;
    real depth = bsize[0] - fabs(loc[0]);
    for (int i = 1; i < 3; i++) { real di = bsize[i] - fabs(loc[i]); if (di < depth) { depth = di; k = i; } }
    nl[0] = nl[1] = nl[2] = 0;
    nl[k] = loc[k] >= 0 ? 1 : -1;
    c->dist = -depth - r;
    for (int i = 0; i < 3; i++) pl[i] = loc[i] + nl[i] * ((depth - r) * (real)0.5);
  }
  for (int i = 0; i < 3; i++) {
    c->pos[i] = bpos[i] + bmat[3 * i] * pl[0] + bmat[3 * i + 1] * pl[1] + bmat[3 * i + 2] * pl[2];
    c->frame[i] = -(bmat[3 * i] * nl[0] + bmat[3 * i + 1] * nl[1] + bmat[3 * i + 2] * nl[2]);
  }
  c->frame[3] = c->frame[4] = c->frame[5] = 0;
  return 1;
}

/* d/dt of half the squared distance between the point pc + t h (box frame) and the box */
static inline real seg_box_slope(const real* pc, const real* h, const real* bsize, real t) {
  real s = 0;
  for (int i = 0; i < 3; i++) { real p = pc[i] + t * h[i]; s += (p - clipr(p, -bsize[i], bsize[i])) * h[i]; }
  return s;
}
static inline real seg_box_dist2(const real* pc, const real* h, const real* bsize, real t) {
  real s = 0;
  for (int i = 0; i < 3; i++) { real p = pc[i] + t * h[i], e = p - clipr(p, -bsize[i], bsize[i]); s += e * e; }
  return s;
}
/* Capsule vs box, up to 4 contacts, all of them sphere_box() contacts of spheres of the capsule's
 * radius centred on its axis at t in [-1, 1] (the construction mjc_CapsuleBox uses, with its own
 * choice of points -- upstream's feature-case analysis is not reproducible from its documentation,
 * so the points are defined here): the two end points (exactly plane_capsule's contacts when the
 * capsule lies over a face), plus the ends ta <= tb of the interval where the axis is closest to
 * the box -- the distance along the axis is convex, so its slope is monotone and ta / tb are found by
 * bisection -- when they are interior points OUTSIDE the box (a capsule crossing an edge or a
 * corner).  Where the axis itself enters the box, an end point inside it carries the contact; if
 * both ends are outside, the point of the inside stretch nearest to the capsule's centre does (not
 * its middle: the middle of a chord that enters through one face and leaves through the opposite
 * one is equidistant from both, and "nearest face" would be decided by rounding). */
#define MJO_CAPBOX_ITERS 24
static int capsule_box(rawcon_t* c, real margin, const real* cpos, const real* cmat, const real* csize, const real* bpos,
                       const real* bmat, const real* bsize) {
  real axis[3] = {cmat[2], cmat[5], cmat[8]}, dif[3] = {cpos[0] - bpos[0], cpos[1] - bpos[1], cpos[2] - bpos[2]};
  real pc[3], h[3];
  for (int i = 0; i < 3; i++) {
    pc[i] = bmat[i] * dif[0] + bmat[3 + i] * dif[1] + bmat[6 + i] * dif[2];
    h[i] = (bmat[i] * axis[0] + bmat[3 + i] * axis[1] + bmat[6 + i] * axis[2]) * csize[1];
  }
  real sm = seg_box_slope(pc, h, bsize, -1), sp = seg_box_slope(pc, h, bsize, 1);
  real ta, tb;
  /* ta = smallest t with slope >= 0 */
  if (sm >= 0) ta = -1;
  else if (sp < 0) ta = 1;
  else {
    real lo = -1, hi = 1;
    for (int it = 0; it < MJO_CAPBOX_ITERS; it++) {
      real mid = (real)0.5 * (lo + hi);
      if (seg_box_slope(pc, h, bsize, mid) >= 0) hi = mid; else lo = mid;
    }
    ta = hi;
  }
  /* tb = largest t with slope <= 0 */
  if (sp <= 0) tb = 1;
  else if (sm > 0) tb = -1;
  else {
    real lo = -1, hi = 1;
    for (int it = 0; it < MJO_CAPBOX_ITERS; it++) {
      real mid = (real)0.5 * (lo + hi);
      if (seg_box_slope(pc, h, bsize, mid) <= 0) lo = mid; else hi = mid;
    }
    tb = lo;
  }
  const real eps = (real)1e-6;
  const int ia = ta > -1 + eps && ta < 1 - eps, ib = tb > -1 + eps && tb < 1 - eps;
  const int oa = seg_box_dist2(pc, h, bsize, ta) > 0, ob = seg_box_dist2(pc, h, bsize, tb) > 0;
  /* the axis runs THROUGH the box with both ends outside (a thin capsule across an edge, deeper
   * than its radius): the inside point nearest to the capsule's centre carries the contact */
  const int pierce = ia && ib && !oa && !ob;
  real ts[4] = {1, -1, pierce ? clipr(0, ta, tb) : ta, tb};
  int use[4] = {1, 1, pierce || (ia && oa), ib && tb - ta > eps && ob};
  int n = 0;
  for (int q = 0; q < 4; q++) {
    if (!use[q]) continue;
    real p[3];
    for (int k = 0; k < 3; k++) p[k] = cpos[k] + axis[k] * (csize[1] * ts[q]);
    if (sphere_box(c + n, margin, p, csize[0], bpos, bmat, bsize)) { memcpy(c[n].frame + 3, axis, sizeof(axis)); n++; }
  }
  return n;
}

/* Moving box (geom1, pose pos / mat, half sizes size) vs a static terrain box (geom2).  NOT
 * mjc_BoxBox (upstream's separating-axis + face-clipping routine cannot be restated from its
 * documentation): a documented rule of this repository, built from the two exact point / segment
 * primitives above, the same on the HIP side (stage_collision.h box_box_candidate):
 *   candidates  0.. 7  corners of the moving box as points against the terrain box (sphere_box, r = 0):
 *                      on a face exactly plane_box's contacts;
 *               8..15  corners of the TERRAIN box as points against the moving box, normal flipped
 *                      (a stair corner poking into a trunk face);
 *              16..39  the 12 edges of the terrain box clipped to the inside of the moving box (slab
 *                      clipping, exact): where an edge runs through it, the points at 1/4 and 3/4 of the
 *                      inside interval, each leaving through the moving box's nearest face (a stair EDGE
 *                      across a trunk face: two support points along the edge);
 *   the first 4 hits in candidate order are the pair's contacts (when 4 corners already touch, the
 *   edge candidates add nothing new).  The contact normal points from the moving box into the terrain. */
static int box_box_candidate(rawcon_t* c, int cand, real margin, const real* pos, const real* mat, const real* size, const real* bpos,
                             const real* bmat, const real* bsize) {
  if (cand < 8) {
    real vec[3] = {(cand & 1) ? size[0] : -size[0], (cand & 2) ? size[1] : -size[1], (cand & 4) ? size[2] : -size[2]}, corner[3];
    mul_mat_vec3(corner, mat, vec);
    for (int k = 0; k < 3; k++) corner[k] += pos[k];
    return sphere_box(c, margin, corner, 0, bpos, bmat, bsize);
  }
  real pt[3];
  if (cand < 16) {
    int i = cand - 8;
    real vec[3] = {(i & 1) ? bsize[0] : -bsize[0], (i & 2) ? bsize[1] : -bsize[1], (i & 4) ? bsize[2] : -bsize[2]};
    mul_mat_vec3(pt, bmat, vec);
    for (int k = 0; k < 3; k++) pt[k] += bpos[k];
  } else {
    /* edge e = (cand - 16) / 2 of the terrain box: axis a = e / 4, the other two coordinates at their
     * +- extremes (bits of e % 4); sample s = (cand - 16) % 2 */
    int e = (cand - 16) >> 1, smp = (cand - 16) & 1, a = e >> 2, b1 = (a + 1) % 3, b2 = (a + 2) % 3;
    real v0[3], v1[3], e0[3], e1[3], q0[3], q1[3];
    v0[a] = -bsize[a]; v1[a] = bsize[a];
    v0[b1] = v1[b1] = (e & 1) ? bsize[b1] : -bsize[b1];
    v0[b2] = v1[b2] = (e & 2) ? bsize[b2] : -bsize[b2];
    mul_mat_vec3(e0, bmat, v0);
    mul_mat_vec3(e1, bmat, v1);
    for (int k = 0; k < 3; k++) { e0[k] += bpos[k] - pos[k]; e1[k] += bpos[k] - pos[k]; }
    for (int i = 0; i < 3; i++) { /* into the moving box's frame */
      q0[i] = mat[i] * e0[0] + mat[3 + i] * e0[1] + mat[6 + i] * e0[2];
      q1[i] = mat[i] * e1[0] + mat[3 + i] * e1[1] + mat[6 + i] * e1[2];
    }
    real t0 = 0, t1 = 1;
    for (int i = 0; i < 3; i++) {
      real h = q1[i] - q0[i];
      if (fabs(h) < MINVAL) { if (fabs(q0[i]) > size[i]) return 0; continue; }
      real ta = (-size[i] - q0[i]) / h, tb = (size[i] - q0[i]) / h;
      if (ta > tb) { real tmp = ta; ta = tb; tb = tmp; }
      if (ta > t0) t0 = ta;
      if (tb < t1) t1 = tb;
    }
    if (t1 - t0 <= (real)1e-6) return 0; /* the edge does not run through the moving box */
    real t = t0 + (t1 - t0) * (smp ? (real)0.75 : (real)0.25);
    for (int k = 0; k < 3; k++) pt[k] = pos[k] + e0[k] + t * (e1[k] - e0[k]);
  }
  /* a point of the terrain box against the moving box; geom order is (moving, terrain): flip the normal */
  if (!sphere_box(c, margin, pt, 0, pos, mat, size)) return 0;
  for (int k = 0; k < 3; k++) c->frame[k] = -c->frame[k];
  return 1;
}
#define MJO_BOXBOX_NCAND 40
static int box_box(rawcon_t* c, real margin, const real* pos, const real* mat, const real* size, const real* bpos, const real* bmat,
                   const real* bsize) {
  int n = 0;
  for (int cand = 0; cand < MJO_BOXBOX_NCAND && n < 4; cand++) n += box_box_candidate(c + n, cand, margin, pos, mat, size, bpos, bmat, bsize);
  return n;
}

/* contact parameters (mj_contactParam) + append n raw contacts of the pair (g1, g2) */
static void emit_contacts(const mjo_model_t* m, mjo_data_t* d, int w, int g1, int g2, real margin, real gap, rawcon_t* rc, int n,
                          int* pncon) {
  const int ncm = m->size.nconmax;
  const real *gfri = MF(geom_friction, w), *gsolref = MF(geom_solref, w), *gsolimp = MF(geom_solimp, w), *gsolmix = MF(geom_solmix, w);
  int ncon = *pncon;
  int condim;
  real fri[3], solref[2], solimp[5];
  int pr1 = m->geom_priority[g1], pr2 = m->geom_priority[g2];
  if (pr1 != pr2) {
    int gi = pr1 > pr2 ? g1 : g2;
    condim = m->geom_condim[gi];
    memcpy(fri, gfri + 3 * gi, sizeof(fri));
    memcpy(solref, gsolref + 2 * gi, sizeof(solref));
    memcpy(solimp, gsolimp + 5 * gi, sizeof(solimp));
  } else {
    condim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
    for (int k = 0; k < 3; k++) fri[k] = gfri[3 * g1 + k] > gfri[3 * g2 + k] ? gfri[3 * g1 + k] : gfri[3 * g2 + k];
    real mix;
    real sm1 = gsolmix[g1], sm2 = gsolmix[g2];
    if (sm1 >= MINVAL && sm2 >= MINVAL) mix = sm1 / (sm1 + sm2);
    else if (sm1 < MINVAL && sm2 < MINVAL) mix = (real)0.5;
    else if (sm1 < MINVAL) mix = 0;
    else mix = 1;
    if (gsolref[2 * g1] > 0 && gsolref[2 * g2] > 0)
      for (int k = 0; k < 2; k++) solref[k] = mix * gsolref[2 * g1 + k] + (1 - mix) * gsolref[2 * g2 + k];
    else
      for (int k = 0; k < 2; k++) solref[k] = gsolref[2 * g1 + k] < gsolref[2 * g2 + k] ? gsolref[2 * g1 + k] : gsolref[2 * g2 + k];
    for (int k = 0; k < 5; k++) solimp[k] = mix * gsolimp[5 * g1 + k] + (1 - mix) * gsolimp[5 * g2 + k];
  }
  if (ncon + n > ncm) d->overflow[w] |= MJLAB_OVF_NCONMAX;
  for (int i = 0; i < n && ncon < ncm; i++) {
    make_frame(rc[i].frame);
    D(contact_dist, ncm)[ncon] = rc[i].dist;
    memcpy(D(contact_pos, 3 * ncm) + 3 * ncon, rc[i].pos, 3 * sizeof(real));
    memcpy(D(contact_frame, 9 * ncm) + 9 * ncon, rc[i].frame, 9 * sizeof(real));
    D(contact_includemargin, ncm)[ncon] = margin - gap;
    real* f5 = D(contact_friction, 5 * ncm) + 5 * ncon;
    for (int k = 0; k < 3; k++) if (fri[k] < (real)1e-5) fri[k] = (real)1e-5;  /* mj_contactParam: fri[i] = max(mjMINMU, fri[i]) */
    f5[0] = f5[1] = fri[0]; f5[2] = fri[1]; f5[3] = f5[4] = fri[2];
    memcpy(D(contact_solref, 2 * ncm) + 2 * ncon, solref, sizeof(solref));
    memcpy(D(contact_solimp, 5 * ncm) + 5 * ncon, solimp, sizeof(solimp));
    (d->contact_dim + (size_t)w * ncm)[ncon] = condim;
    (d->contact_geom + (size_t)w * 2 * ncm)[2 * ncon] = g1;
    (d->contact_geom + (size_t)w * 2 * ncm)[2 * ncon + 1] = g2;
    (d->contact_efc_address + (size_t)w * ncm)[ncon] = -1;
    ncon++;
  }
  *pncon = ncon;
}

/* Terrain broadphase for one moving geom: walk the grid cells under its bounding sphere and keep
 * the (at most MJLAB_TCAND_MAX, smallest ids first) boxes within reach, in ascending order.  A box
 * listed in several cells is looked at once, in the lowest cell the two footprints share. */
static int terrain_candidates(const mjo_model_t* m, const real* centre, real reach, int* cand, int* dropped) {
  const mjlab_sizes_t* s = &m->size;
  const int nx = s->tgrid_nx, ny = s->tgrid_ny;
  const real x0 = (real)m->opt.tgrid_x0, y0 = (real)m->opt.tgrid_y0, inv = (real)1 / (real)m->opt.tgrid_cell;
  int ix0 = (int)floor((centre[0] - reach - x0) * inv), ix1 = (int)floor((centre[0] + reach - x0) * inv);
  int iy0 = (int)floor((centre[1] - reach - y0) * inv), iy1 = (int)floor((centre[1] + reach - y0) * inv);
  ix0 = ix0 < 0 ? 0 : (ix0 > nx - 1 ? nx - 1 : ix0); ix1 = ix1 < 0 ? 0 : (ix1 > nx - 1 ? nx - 1 : ix1);
  iy0 = iy0 < 0 ? 0 : (iy0 > ny - 1 ? ny - 1 : iy0); iy1 = iy1 < 0 ? 0 : (iy1 > ny - 1 ? ny - 1 : iy1);
  int n = 0;
  for (int ix = ix0; ix <= ix1; ix++)
    for (int iy = iy0; iy <= iy1; iy++) {
      const int c = ix * ny + iy;
      if (centre[2] - reach > m->tgrid_ztop[c]) continue; /* wholly above everything in this cell */
      for (int k = m->tgrid_start[c]; k < m->tgrid_start[c + 1]; k++) {
        const int b = m->tgrid_item[k];
        const int bx = m->tbox_cell0[2 * b], by = m->tbox_cell0[2 * b + 1];
        if (ix != (ix0 > bx ? ix0 : bx) || iy != (iy0 > by ? iy0 : by)) continue;
        const real *bpos = m->tbox_pos + 3 * b, *bmat = m->tbox_mat + 9 * b, *bsize = m->tbox_size + 3 * b;
        real dif[3] = {centre[0] - bpos[0], centre[1] - bpos[1], centre[2] - bpos[2]}, d2 = 0;
        for (int i = 0; i < 3; i++) {
          real loc = bmat[i] * dif[0] + bmat[3 + i] * dif[1] + bmat[6 + i] * dif[2];
          real dv = loc - clipr(loc, -bsize[i], bsize[i]);
          d2 += dv * dv;
        }
        if (d2 > reach * reach) continue;
        /* sorted insert, bounded: the largest id falls off the end */
        int pos = n < MJLAB_TCAND_MAX ? n : MJLAB_TCAND_MAX;
        while (pos > 0 && cand[pos - 1] > b) pos--;
        if (n >= MJLAB_TCAND_MAX) *dropped = 1; /* this box or the largest id in the list falls off */
        if (pos >= MJLAB_TCAND_MAX) continue;
        int last = n < MJLAB_TCAND_MAX ? n : MJLAB_TCAND_MAX - 1;
        for (int q = last; q > pos; q--) cand[q] = cand[q - 1];
        cand[pos] = b;
        if (n < MJLAB_TCAND_MAX) n++;
      }
    }
  return n;
}

static void collision(const mjo_model_t* m, mjo_data_t* d, int w) {
  const mjlab_sizes_t* s = &m->size;
  real *gx = D(geom_xpos, 3 * s->ngeom), *gm = D(geom_xmat, 9 * s->ngeom);
  const real *gsize = MF(geom_size, w), *rbound = MF(geom_rbound, w), *gmargin = MF(geom_margin, w), *ggap = MF(geom_gap, w);
  int ncon = 0, tdrop = 0;
  d->overflow[w] = 0;
  for (int p = 0; p < s->npair; p++) {
    int g1 = m->pair_geom[2 * p], g2 = m->pair_geom[2 * p + 1];
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    real margin = gmargin[g1] > gmargin[g2] ? gmargin[g1] : gmargin[g2];
    real gap = ggap[g1] > ggap[g2] ? ggap[g1] : ggap[g2];
    const real *p1 = gx + 3 * g1, *p2 = gx + 3 * g2, *m1 = gm + 9 * g1, *m2 = gm + 9 * g2;
    const real *s1 = gsize + 3 * g1, *s2 = gsize + 3 * g2;
    /* bounding-sphere / plane rejection */
    if (t1 == MJLAB_GEOM_PLANE) {
      real n[3] = {m1[2], m1[5], m1[8]}, dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
      if (dot3(dif, n) > margin + rbound[g2]) continue;
    } else {
      real dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, bound = margin + rbound[g1] + rbound[g2];
      if (dot3(dif, dif) > bound * bound) continue;
    }
    rawcon_t rc[4];
    int n = 0;
    if (t1 == MJLAB_GEOM_PLANE && t2 == MJLAB_GEOM_SPHERE) n = plane_sphere(rc, margin, p1, m1, p2, s2[0]);
    else if (t1 == MJLAB_GEOM_PLANE && t2 == MJLAB_GEOM_CAPSULE) n = plane_capsule(rc, margin, p1, m1, p2, m2, s2);
    else if (t1 == MJLAB_GEOM_PLANE && t2 == MJLAB_GEOM_BOX) n = plane_box(rc, margin, p1, m1, p2, m2, s2);
    else if (t1 == MJLAB_GEOM_SPHERE && t2 == MJLAB_GEOM_SPHERE) n = sphere_sphere(rc, margin, p1, s1[0], p2, s2[0]);
    else if (t1 == MJLAB_GEOM_SPHERE && t2 == MJLAB_GEOM_CAPSULE) n = sphere_capsule(rc, margin, p1, s1[0], p2, m2, s2);
    else if (t1 == MJLAB_GEOM_CAPSULE && t2 == MJLAB_GEOM_CAPSULE) n = capsule_capsule(rc, margin, p1, m1, s1, p2, m2, s2);
    else continue; /* unsupported pair types never reach here: the compiler rejects them */
    if (!n) continue;
    emit_contacts(m, d, w, g1, g2, margin, gap, rc, n, &ncon);
  }
  /* moving spheres / capsules vs the box terrain; terrain boxes carry no margin (checked at
   * model compile time), so the pair margin is the moving geom's */
  /* two passes: spheres and capsules first, then moving boxes (the HIP path runs them as separate
   * sweeps; contact order is part of the parity contract) */
  for (int pass = 0; pass < 2; pass++)
  for (int ti = 0; ti < s->ntgeom; ti++) {
    int g = m->tgeom[ti], cand[MJLAB_TCAND_MAX];
    if ((m->geom_type[g] == MJLAB_GEOM_BOX) != (pass == 1)) continue;
    real margin = gmargin[g], gap = ggap[g];
    int nc = terrain_candidates(m, gx + 3 * g, rbound[g] + margin, cand, &tdrop);
    for (int q = 0; q < nc; q++) {
      int b = cand[q];
      rawcon_t rc[4];
      int n = 0;
      if (m->geom_type[g] == MJLAB_GEOM_SPHERE)
        n = sphere_box(rc, margin, gx + 3 * g, gsize[3 * g], m->tbox_pos + 3 * b, m->tbox_mat + 9 * b, m->tbox_size + 3 * b);
      else if (m->geom_type[g] == MJLAB_GEOM_CAPSULE)
        n = capsule_box(rc, margin, gx + 3 * g, gm + 9 * g, gsize + 3 * g, m->tbox_pos + 3 * b, m->tbox_mat + 9 * b, m->tbox_size + 3 * b);
      else if (m->geom_type[g] == MJLAB_GEOM_BOX)
        n = box_box(rc, margin, gx + 3 * g, gm + 9 * g, gsize + 3 * g, m->tbox_pos + 3 * b, m->tbox_mat + 9 * b, m->tbox_size + 3 * b);
      if (n) emit_contacts(m, d, w, g, m->tbox_geom[b], margin, gap, rc, n, &ncon);
    }
  }
  d->ncon[w] = ncon;
}

/* ------------------------------------------------------------------ constraints */
static real impedance(const real* solimp, real pos, real margin) {
  real dmin = clipr(solimp[0], MINIMP, MAXIMP), dmax = clipr(solimp[1], MINIMP, MAXIMP);
  real width = solimp[2] > MINVAL ? solimp[2] : MINVAL;
  real mid = clipr(solimp[3], MINIMP, MAXIMP), power = solimp[4] > 1 ? solimp[4] : 1;
  real x = (pos - margin) / width;
  if (x < 0) x = -x;
  real y;
  if (x >= 1) y = 1;
  else if (x == 0) y = 0;
  else if (x <= mid) y = (1 / pow(mid, power - 1)) * pow(x, power);
  else y = 1 - (1 / pow(1 - mid, power - 1)) * pow(1 - x, power);
  if (solimp[2] <= MINVAL) y = (real)0.5; /* mj getimpedance: "flat function" when the width is <= mjMINVAL -- imp = 0.5 (dmin + dmax) */
  return dmin + y * (dmax - dmin);
}

/* writes row r: J given, pos/margin/solref/solimp/diagApprox; returns imp-based R */
static void finish_row(const mjo_model_t* m, mjo_data_t* d, int w, int r, real pos, real margin, const real* solref,
                       const real* solimp, real diag_approx, int type, int id) {
  const mjlab_sizes_t* s = &m->size;
  int nv = s->nv, njm = s->njmax;
  real* J = D(efc_J, njm * nv) + (size_t)r * nv;
  real* qvel = D(qvel, nv);
  real vel = 0;
  for (int i = 0; i < nv; i++) vel += J[i] * qvel[i];
  real imp = impedance(solimp, pos, margin);
  real dmax = clipr(solimp[1], MINIMP, MAXIMP);
  real k, b;
  if (solref[0] > 0) {
    real tc = solref[0], dr = solref[1];
    real h2 = 2 * (real)m->opt.timestep;
    if (tc < h2) tc = h2; /* refsafe */
    real kd = dmax * dmax * tc * tc * dr * dr, bd = dmax * tc;
    k = 1 / (kd > MINVAL ? kd : MINVAL);
    b = 2 / (bd > MINVAL ? bd : MINVAL);
  } else {
    real kd = dmax * dmax, bd = dmax;
    k = -solref[0] / (kd > MINVAL ? kd : MINVAL);
    b = -solref[1] / (bd > MINVAL ? bd : MINVAL);
  }
  real R = (1 - imp) / imp * diag_approx;
  if (R < MINVAL) R = MINVAL;
  D(efc_pos, njm)[r] = pos;
  D(efc_margin, njm)[r] = margin;
  D(efc_D, njm)[r] = 1 / R; /* pyramidal rows are rescaled by the caller */
  D(efc_aref, njm)[r] = -b * vel - k * imp * (pos - margin);
  (d->efc_type + (size_t)w * njm)[r] = type;
  (d->efc_id + (size_t)w * njm)[r] = id;
}

static void make_constraint(const mjo_model_t* m, mjo_data_t* d, int w) {
  const mjlab_sizes_t* s = &m->size;
  int nv = s->nv, njm = s->njmax, ncm = s->nconmax, nb = s->nbody;
  real* J = D(efc_J, njm * nv);
  real *qpos = D(qpos, s->nq), *cdof = D(cdof, 6 * nv), *sub = D(subtree_com, 3 * nb);
  int nefc = 0;
  const real *range = MF(jnt_range, w), *jmargin = MF(jnt_margin, w), *jsolref = MF(jnt_solref, w),
             *jsolimp = MF(jnt_solimp, w), *dinv = MF(dof_invweight0, w);
  /* friction loss (mj_instantiateFriction): one row per dof with dof_frictionloss > 0, in dof order, before
   * every other row; J = unit vector of the dof, pos = margin = 0, solref / solimp of the dof */
  const real *floss = MF(dof_frictionloss, w), *dsolref = MF(dof_solref, w), *dsolimp = MF(dof_solimp, w);
  for (int i = 0; i < nv; i++) {
    if (!(m->opt.flags & MJLAB_OPT_FRICTIONLOSS) || !(floss[i] > 0)) continue;
    if (nefc >= njm) { d->overflow[w] |= MJLAB_OVF_NJMAX; continue; }
    real* row = J + (size_t)nefc * nv;
    memset(row, 0, sizeof(real) * nv);
    row[i] = 1;
    finish_row(m, d, w, nefc, 0, 0, dsolref + 2 * i, dsolimp + 5 * i, dinv[i], MJLAB_EFC_FRICTION_DOF, i);
    D(efc_frictionloss, njm)[nefc] = floss[i];
    nefc++;
  }
  d->nf[w] = nefc;
  /* joint limits (hinge / slide) */
  for (int j = 0; j < s->njnt; j++) {
    if (!m->jnt_limited[j] || m->jnt_type[j] == MJLAB_JNT_FREE) continue;
    real value = qpos[m->jnt_qposadr[j]], mg = jmargin[j];
    for (int side = -1; side <= 1; side += 2) {
      real dist = side * (range[2 * j + (side + 1) / 2] - value);
      if (dist < mg && nefc >= njm) d->overflow[w] |= MJLAB_OVF_NJMAX;
      if (dist < mg && nefc < njm) {
        real* row = J + (size_t)nefc * nv;
        memset(row, 0, sizeof(real) * nv);
        row[m->jnt_dofadr[j]] = (real)(-side);
        finish_row(m, d, w, nefc, dist, mg, jsolref + 2 * j, jsolimp + 5 * j, dinv[m->jnt_dofadr[j]], MJLAB_EFC_LIMIT, j);
        nefc++;
      }
    }
  }
  /* contacts */
  int ncon = d->ncon[w];
  const real* binv = MF(body_invweight0, w);
  for (int c = 0; c < ncon; c++) {
    int dim = (d->contact_dim + (size_t)w * ncm)[c];
    real dist = D(contact_dist, ncm)[c], inc = D(contact_includemargin, ncm)[c];
    (d->contact_efc_address + (size_t)w * ncm)[c] = -1;
    if (dist >= inc) continue;
    const int elliptic = m->opt.cone == MJLAB_CONE_ELLIPTIC;
    int nrow = dim == 1 ? 1 : (elliptic ? dim : 2 * (dim - 1));
    if (nefc + nrow > njm) { d->overflow[w] |= MJLAB_OVF_NJMAX; continue; }
    int g1 = (d->contact_geom + (size_t)w * 2 * ncm)[2 * c], g2 = (d->contact_geom + (size_t)w * 2 * ncm)[2 * c + 1];
    int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
    const real *pos = D(contact_pos, 3 * ncm) + 3 * c, *frame = D(contact_frame, 9 * ncm) + 9 * c;
    const real* fri = D(contact_friction, 5 * ncm) + 5 * c;
    /* translational Jacobian difference (body2 - body1) at the contact point, in the contact frame */
    real jf[3][64];
    int bodies[2] = {b1, b2};
    for (int i = 0; i < nv; i++) jf[0][i] = jf[1][i] = jf[2][i] = 0;
    for (int side = 0; side < 2; side++) {
      int b = bodies[side];
      real sgn = side ? 1 : -1;
      unsigned lo = (unsigned)m->body_dofmask[2 * b], hi = (unsigned)m->body_dofmask[2 * b + 1];
      real off[3];
      for (int k = 0; k < 3; k++) off[k] = pos[k] - sub[3 * m->body_rootid[b] + k];
      for (int i = 0; i < nv; i++) {
        int on = i < 32 ? (lo >> i) & 1 : (hi >> (i - 32)) & 1;
        if (!on) continue;
        real jp[3];
        cross3(jp, cdof + 6 * i, off);
        for (int k = 0; k < 3; k++) jp[k] += cdof[6 * i + 3 + k];
        for (int a = 0; a < 3; a++) jf[a][i] += sgn * dot3(frame + 3 * a, jp);
      }
    }
    real tran = binv[2 * b1] + binv[2 * b2];
    (d->contact_efc_address + (size_t)w * ncm)[c] = nefc;
    const real *solref = D(contact_solref, 2 * ncm) + 2 * c, *solimp = D(contact_solimp, 5 * ncm) + 5 * c;
    if (dim == 1) {
      memcpy(J + (size_t)nefc * nv, jf[0], sizeof(real) * nv);
      finish_row(m, d, w, nefc, dist, inc, solref, solimp, tran, MJLAB_EFC_CONTACT_FRICTIONLESS, c);
      nefc++;
    } else if (elliptic) {
      /* ELLIPTIC cone (mj_instantiateContact + mj_makeImpedance, restated from MuJoCo's documented model; UNVERIFIED like the rest):
       * dim rows = the contact-frame components of the relative acceleration [normal, tangent 1, tangent 2]; only the normal row has
       * a position (dist) and margin, the friction rows are pure velocity constraints (pos = margin = 0: aref = -b * vel).  The normal
       * row's regulariser R comes from the impedance as for every row; the friction rows get R_1 = R_0 / impratio and
       * R_j = R_1 friction[0]^2 / friction[j-1]^2, and the cone's friction coefficient in the regularised problem is
       * mu = friction[0] sqrt(R_1 / R_0) = friction[0] / sqrt(impratio) (the solver below reads it the same way). */
      int first = nefc;
      for (int k = 0; k < dim; k++) {
        memcpy(J + (size_t)nefc * nv, jf[k], sizeof(real) * nv);
        finish_row(m, d, w, nefc, k == 0 ? dist : 0, k == 0 ? inc : 0, solref, solimp, tran, MJLAB_EFC_CONTACT_ELLIPTIC, c);
        nefc++;
      }
      real R0 = 1 / D(efc_D, njm)[first], ir = (real)m->opt.impratio;
      real R1 = R0 / (ir > MINVAL ? ir : MINVAL);
      for (int k = 1; k < dim; k++) {
        real Rk = R1 * fri[0] * fri[0] / (fri[k - 1] * fri[k - 1]);
        if (Rk < MINVAL) Rk = MINVAL;
        D(efc_D, njm)[first + k] = 1 / Rk;
      }
    } else {
      int first = nefc;
      for (int k = 1; k < dim; k++) {
        real mu = fri[k - 1];
        for (int sg = 0; sg < 2; sg++) {
          real* row = J + (size_t)nefc * nv;
          for (int i = 0; i < nv; i++) row[i] = jf[0][i] + (sg ? -mu : mu) * jf[k][i];
          finish_row(m, d, w, nefc, dist, inc, solref, solimp, tran + mu * mu * tran, MJLAB_EFC_CONTACT_PYRAMIDAL, c);
          nefc++;
        }
      }
      /* pyramid rows share R = 2 mu^2 R_first, mu = friction[0] / sqrt(impratio) */
      real mu = fri[0] * sqrt(1 / (real)m->opt.impratio);
      real Rpy = 2 * mu * mu * (1 / D(efc_D, njm)[first]);
      if (Rpy < MINVAL) Rpy = MINVAL;
      for (int r = first; r < nefc; r++) D(efc_D, njm)[r] = 1 / Rpy;
    }
  }
  d->nefc[w] = nefc;
}

/* ------------------------------------------------------------------ velocity / forces */
static void com_vel(const mjo_model_t* m, mjo_data_t* d, int w) {
  const mjlab_sizes_t* s = &m->size;
  int nb = s->nbody, nv = s->nv;
  real *cvel = D(cvel, 6 * nb), *cdof = D(cdof, 6 * nv), *cdd = D(cdof_dot, 6 * nv), *qvel = D(qvel, nv);
  memset(cvel, 0, 6 * sizeof(real));
  for (int i = 1; i < nb; i++) {
    real v[6];
    memcpy(v, cvel + 6 * m->body_parentid[i], sizeof(v));
    int ja = m->body_jntadr[i], jn = m->body_jntnum[i];
    for (int j = ja; j < ja + jn; j++) {
      int da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == MJLAB_JNT_FREE) {
        memset(cdd + 6 * da, 0, 18 * sizeof(real));
        for (int k = 0; k < 3; k++) for (int a = 0; a < 6; a++) v[a] += cdof[6 * (da + k) + a] * qvel[da + k];
        for (int k = 3; k < 6; k++) cross_motion(cdd + 6 * (da + k), v, cdof + 6 * (da + k));
        for (int k = 3; k < 6; k++) for (int a = 0; a < 6; a++) v[a] += cdof[6 * (da + k) + a] * qvel[da + k];
      } else {
        cross_motion(cdd + 6 * da, v, cdof + 6 * da);
        for (int a = 0; a < 6; a++) v[a] += cdof[6 * da + a] * qvel[da];
      }
    }
    memcpy(cvel + 6 * i, v, sizeof(v));
  }
}

static void rne_bias(const mjo_model_t* m, mjo_data_t* d, int w) {
  const mjlab_sizes_t* s = &m->size;
  int nb = s->nbody, nv = s->nv;
  real *cvel = D(cvel, 6 * nb), *cdof = D(cdof, 6 * nv), *cdd = D(cdof_dot, 6 * nv), *qvel = D(qvel, nv);
  real *cinert = D(cinert, 10 * nb), *bias = D(qfrc_bias, nv);
  real* cacc = (real*)calloc(6 * nb, sizeof(real));
  real* cfrc = (real*)calloc(6 * nb, sizeof(real));
  for (int k = 0; k < 3; k++) cacc[3 + k] = -(real)m->opt.gravity[k];
  for (int i = 1; i < nb; i++) {
    real* a = cacc + 6 * i;
    memcpy(a, cacc + 6 * m->body_parentid[i], 6 * sizeof(real));
    int da = m->body_dofadr[i];
    for (int k = 0; k < m->body_dofnum[i]; k++)
      for (int c = 0; c < 6; c++) a[c] += cdd[6 * (da + k) + c] * qvel[da + k];
    real t1[6], t2[6], t3[6];
    mul_inert_vec(t1, cinert + 10 * i, a);
    mul_inert_vec(t2, cinert + 10 * i, cvel + 6 * i);
    cross_force(t3, cvel + 6 * i, t2);
    for (int c = 0; c < 6; c++) cfrc[6 * i + c] = t1[c] + t3[c];
  }
  for (int i = nb - 1; i > 0; i--) {
    int p = m->body_parentid[i];
    if (p > 0) for (int c = 0; c < 6; c++) cfrc[6 * p + c] += cfrc[6 * i + c];
  }
  for (int i = 0; i < nv; i++) {
    real v = 0;
    for (int c = 0; c < 6; c++) v += cdof[6 * i + c] * cfrc[6 * m->dof_bodyid[i] + c];
    bias[i] = v;
  }
  free(cacc);
  free(cfrc);
}

static void smooth_forces(const mjo_model_t* m, mjo_data_t* d, int w) {
  const mjlab_sizes_t* s = &m->size;
  int nb = s->nbody, nv = s->nv, nu = s->nu;
  real *qpos = D(qpos, s->nq), *qvel = D(qvel, nv), *ctrl = D(ctrl, nu);
  real *passive = D(qfrc_passive, nv), *qact = D(qfrc_actuator, nv), *aforce = D(actuator_force, nu);
  real *smooth = D(qfrc_smooth, nv), *bias = D(qfrc_bias, nv), *applied = D(qfrc_applied, nv);
  const real *damping = MF(dof_damping, w), *stiff = MF(jnt_stiffness, w), *qspring = MF(qpos_spring, w);
  /* passive: joint springs about mjModel.qpos_spring (springref, NOT qpos0) and dof damping */
  for (int i = 0; i < nv; i++) passive[i] = -damping[i] * qvel[i];
  for (int j = 0; j < s->njnt; j++) {
    if (stiff[j] == 0 || m->jnt_type[j] == MJLAB_JNT_FREE) continue;
    int qa = m->jnt_qposadr[j];
    passive[m->jnt_dofadr[j]] -= stiff[j] * (qpos[qa] - qspring[qa]);
  }
  /* actuation: joint transmission, fixed gain, affine bias */
  memset(qact, 0, sizeof(real) * nv);
  const real *gain = MF(actuator_gainprm, w), *biasprm = MF(actuator_biasprm, w), *crange = MF(actuator_ctrlrange, w),
             *frange = MF(actuator_forcerange, w), *gear = MF(actuator_gear, w);
  for (int a = 0; a < nu; a++) {
    int j = m->actuator_trnid[2 * a], qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    real g = gear[6 * a], c = ctrl[a];
    if (m->actuator_ctrllimited[a]) c = clipr(c, crange[2 * a], crange[2 * a + 1]);
    real len = g * qpos[qa], vel = g * qvel[da];
    real f = gain[10 * a] * c + biasprm[10 * a] + biasprm[10 * a + 1] * len + biasprm[10 * a + 2] * vel;
    if (m->actuator_forcelimited[a]) f = clipr(f, frange[2 * a], frange[2 * a + 1]);
    aforce[a] = f;
    qact[da] += g * f;
  }
  for (int i = 0; i < nv; i++) smooth[i] = passive[i] - bias[i] + applied[i] + qact[i];
  /* Cartesian perturbations: xfrc_applied = [force, torque] at xipos, world frame */
  real *xfrc = D(xfrc_applied, 6 * nb), *xipos = D(xipos, 3 * nb), *sub = D(subtree_com, 3 * nb), *cdof = D(cdof, 6 * nv);
  for (int b = 1; b < nb; b++) {
    const real* f = xfrc + 6 * b;
    if (f[0] == 0 && f[1] == 0 && f[2] == 0 && f[3] == 0 && f[4] == 0 && f[5] == 0) continue;
    unsigned lo = (unsigned)m->body_dofmask[2 * b], hi = (unsigned)m->body_dofmask[2 * b + 1];
    real off[3];
    for (int k = 0; k < 3; k++) off[k] = xipos[3 * b + k] - sub[3 * m->body_rootid[b] + k];
    for (int i = 0; i < nv; i++) {
      int on = i < 32 ? (lo >> i) & 1 : (hi >> (i - 32)) & 1;
      if (!on) continue;
      real jp[3];
      cross3(jp, cdof + 6 * i, off);
      for (int k = 0; k < 3; k++) jp[k] += cdof[6 * i + 3 + k];
      smooth[i] += dot3(jp, f) + dot3(cdof + 6 * i, f + 3);
    }
  }
  real* qas = D(qacc_smooth, nv);
  memcpy(qas, smooth, sizeof(real) * nv);
  chol_solve(D(qLD, nv * nv), nv, qas);
}

/* ------------------------------------------------------------------ Newton solver */
typedef struct {
  int nv, nefc, nf, ls_iter; /* rows [0, nf) are friction-loss rows */
  int cg;                    /* mjSOL_CG: no Hessian, directions preconditioned by M (its factor: qLD) */
  const real* L;
  real *Mgrad, *gradold, *Mgradold;
  const real *J, *Dv, *aref, *floss, *M, *qfrc_smooth, *qacc_smooth;
  real *qacc, *Ma, *jar, *grad, *search, *Mv, *jv, *force, *qfrc_constraint, *H;
  real quad_gauss[3], cost, gauss;
  /* elliptic cones: row r with type[r] == MJLAB_EFC_CONTACT_ELLIPTIC that is the FIRST row of its contact (r == cadr[id[r]]) starts a
   * group of 3 rows [normal, tangent 1, tangent 2]; fri = contact_friction (5 per contact), impratio from the options */
  const int *type, *id, *cadr;
  const real* fri;
  real impratio;
} nctx_t;
static inline int cone_start(const nctx_t* c, int r) { return c->type && c->type[r] == MJLAB_EFC_CONTACT_ELLIPTIC && c->cadr[c->id[r]] == r; }

typedef struct { real alpha, cost, d0, d1; } lspnt_t;

/* Cost of one row at residual x = (J qacc - aref)_r (mj_constraintUpdate): inequality rows are
 * quadratic where x < 0 and free otherwise; friction-loss rows (r < nf) are quadratic inside
 * |x| < R f (R = 1 / D, f = efc_frictionloss) and linear with slope -+f outside (a Huber cost).
 * Returns 1 where the row is in its quadratic zone (those rows enter the Hessian). */
static inline int row_cost(const nctx_t* c, int r, real x, real* cost, real* force) {
  real Dr = c->Dv[r];
  if (r < c->nf) {
    real f = c->floss[r], rf = f / Dr;
    if (x <= -rf) { *force = f; *cost = f * ((real)-0.5 * rf - x); return 0; }
    if (x >= rf) { *force = -f; *cost = f * ((real)-0.5 * rf + x); return 0; }
    *force = -Dr * x; *cost = (real)0.5 * Dr * x * x; return 1;
  }
  if (x < 0) { *force = -Dr * x; *cost = (real)0.5 * Dr * x * x; return 1; }
  *force = 0; *cost = 0; return 0;
}

/* One elliptic contact (condim 3) at residuals x[3] = (J qacc - aref) of its rows [normal, t1, t2] (mj_constraintUpdate, elliptic branch,
 * restated from the cone's definition).  With mu = friction[0] / sqrt(impratio), U = (mu x0, f1 x1, f2 x2), N = U0, T = |(U1, U2)|:
 *   top zone     N >= mu T (or T = 0, N >= 0): satisfied, cost 0, force 0;
 *   bottom zone  mu N + T <= 0 (or T = 0, N < 0): every row quadratic, cost = 0.5 sum_k D_k x_k^2, force_k = -D_k x_k;
 *   middle zone  otherwise: cost = 0.5 Dm (N - mu T)^2 with Dm = D_0 / (mu^2 (1 + mu^2)); force = -d cost / d x.
 * The three pieces join continuously (the friction rows' D_k = D_0 friction[k-1]^2 / mu^2 -- R_k = R_0 / impratio * friction[0]^2 / friction[k-1]^2 -- is what makes them).
 * H (optional): the 3 x 3 Hessian d^2 cost / d x^2 (row-major).  Returns the zone: 0 top, 1 bottom, 2 middle. */
static int cone_eval(const nctx_t* c, int r, const real* x, real* cost, real* force, real* H) {
  const real* fr = c->fri + 5 * c->id[r];
  real mu = fr[0] / (real)sqrt(c->impratio > MINVAL ? c->impratio : MINVAL);
  real f[3] = {mu, fr[0], fr[1]};
  real U[3] = {x[0] * f[0], x[1] * f[1], x[2] * f[2]};
  real N = U[0], T = (real)sqrt(U[1] * U[1] + U[2] * U[2]);
  if (H) memset(H, 0, 9 * sizeof(real));
  if (N >= mu * T || (T <= 0 && N >= 0)) {
    *cost = 0; force[0] = force[1] = force[2] = 0;
    return 0;
  }
  if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
    *cost = 0;
    for (int k = 0; k < 3; k++) {
      real Dk = c->Dv[r + k];
      *cost += (real)0.5 * Dk * x[k] * x[k];
      force[k] = -Dk * x[k];
      if (H) H[4 * k] = Dk;
    }
    return 1;
  }
  real Dm = c->Dv[r] / (mu * mu * (1 + mu * mu)), phi = N - mu * T;
  *cost = (real)0.5 * Dm * phi * phi;
  /* g = d phi / d x = (mu, -mu f1 U1 / T, -mu f2 U2 / T) */
  real g[3] = {mu, -mu * f[1] * U[1] / T, -mu * f[2] * U[2] / T};
  for (int k = 0; k < 3; k++) force[k] = -Dm * phi * g[k];
  if (H) {
    /* d^2 cost = Dm (g g^T + phi d^2 phi), d^2 phi = -mu d^2 T, d^2 T_jk = f_j f_k (delta_jk / T - U_j U_k / T^3), j, k >= 1 */
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        real h = g[a] * g[b];
        if (a >= 1 && b >= 1) h += phi * (-mu) * f[a] * f[b] * ((a == b ? 1 / T : 0) - U[a] * U[b] / (T * T * T));
        H[3 * a + b] = Dm * h;
      }
  }
  return 2;
}

static void update_constraint(nctx_t* c) {
  int nv = c->nv;
  real cost = 0;
  memset(c->qfrc_constraint, 0, sizeof(real) * nv);
  for (int r = 0; r < c->nefc; r++) {
    real rc;
    int nr = 1;
    if (cone_start(c, r)) { cone_eval(c, r, c->jar + r, &rc, c->force + r, NULL); nr = 3; }
    else row_cost(c, r, c->jar[r], &rc, &c->force[r]);
    cost += rc;
    for (int k = 0; k < nr; k++)
      if (c->force[r + k] != 0) {
        const real* row = c->J + (size_t)(r + k) * nv;
        for (int i = 0; i < nv; i++) c->qfrc_constraint[i] += row[i] * c->force[r + k];
      }
    r += nr - 1;
  }
  real gauss = 0;
  for (int i = 0; i < nv; i++) gauss += (real)0.5 * (c->Ma[i] - c->qfrc_smooth[i]) * (c->qacc[i] - c->qacc_smooth[i]);
  c->gauss = gauss;
  c->cost = cost + gauss;
}

static void update_gradient(nctx_t* c) {
  int nv = c->nv;
  for (int i = 0; i < nv; i++) c->grad[i] = c->Ma[i] - c->qfrc_smooth[i] - c->qfrc_constraint[i];
  if (c->cg) { /* Mgrad = M^-1 grad (mj_solveM); the caller combines it with the previous direction */
    memcpy(c->Mgrad, c->grad, sizeof(real) * nv);
    chol_solve(c->L, nv, c->Mgrad);
    return;
  }
  memcpy(c->H, c->M, sizeof(real) * nv * nv);
  for (int r = 0; r < c->nefc; r++) {
    real rc, rfo;
    if (cone_start(c, r)) { /* H += Jc^T Hc Jc with the cone's 3 x 3 Hessian (diagonal D in the bottom zone, dense in the middle zone) */
      real fo[3], Hc[9];
      if (cone_eval(c, r, c->jar + r, &rc, fo, Hc)) {
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) {
            if (Hc[3 * a + b] == 0) continue;
            const real *ra = c->J + (size_t)(r + a) * nv, *rb = c->J + (size_t)(r + b) * nv;
            for (int i = 0; i < nv; i++) {
              if (ra[i] == 0) continue;
              real t = Hc[3 * a + b] * ra[i];
              for (int j = 0; j <= i; j++) c->H[i * nv + j] += t * rb[j];
            }
          }
      }
      r += 2;
      continue;
    }
    if (!row_cost(c, r, c->jar[r], &rc, &rfo)) continue;
    const real* row = c->J + (size_t)r * nv;
    real Dr = c->Dv[r];
    for (int i = 0; i < nv; i++) {
      if (row[i] == 0) continue;
      real t = Dr * row[i];
      for (int j = 0; j <= i; j++) c->H[i * nv + j] += t * row[j];
    }
  }
  chol_factor(c->H, nv);
  for (int i = 0; i < nv; i++) c->search[i] = c->grad[i];
  chol_solve(c->H, nv, c->search);
  for (int i = 0; i < nv; i++) c->search[i] = -c->search[i];
}

/* debug counters: line-search evaluations / searches of the CALLING thread (thread-local, so the
 * worker threads of mjo_run do not fight over a cache line; meaningful for nthread = 1 runs) */
static __thread long mjo_dbg_evals = 0, mjo_dbg_searches = 0;
long mjo_debug_counter(int which, int reset) { long v = which ? mjo_dbg_searches : mjo_dbg_evals; if (reset) mjo_dbg_evals = mjo_dbg_searches = 0; return v; }

static void ls_eval(nctx_t* c, lspnt_t* p, real alpha) {
  mjo_dbg_evals++;
  real cost = alpha * alpha * c->quad_gauss[2] + alpha * c->quad_gauss[1] + c->quad_gauss[0];
  real d0 = 2 * alpha * c->quad_gauss[2] + c->quad_gauss[1], d1 = 2 * c->quad_gauss[2];
  for (int r = 0; r < c->nefc; r++) {
    if (cone_start(c, r)) { /* cost, slope -force . jv and curvature jv^T Hc jv of the cone along the direction */
      real x[3], fo[3], Hc[9], rc;
      for (int k = 0; k < 3; k++) x[k] = c->jar[r + k] + alpha * c->jv[r + k];
      cone_eval(c, r, x, &rc, fo, Hc);
      cost += rc;
      for (int a = 0; a < 3; a++) {
        d0 -= fo[a] * c->jv[r + a];
        for (int b = 0; b < 3; b++) d1 += c->jv[r + a] * Hc[3 * a + b] * c->jv[r + b];
      }
      r += 2;
      continue;
    }
    real x = c->jar[r] + alpha * c->jv[r];
    if (r < c->nf) { /* friction loss: linear outside |x| < R f (mj PrimalEval) */
      real f = c->floss[r], rf = f / c->Dv[r];
      if (x <= -rf) { cost += f * ((real)-0.5 * rf - c->jar[r]) - alpha * f * c->jv[r]; d0 -= f * c->jv[r]; continue; }
      if (x >= rf) { cost += f * ((real)-0.5 * rf + c->jar[r]) + alpha * f * c->jv[r]; d0 += f * c->jv[r]; continue; }
    }
    if (x < 0 || r < c->nf) {
      real Dr = c->Dv[r], q0 = (real)0.5 * Dr * c->jar[r] * c->jar[r], q1 = Dr * c->jar[r] * c->jv[r], q2 = (real)0.5 * Dr * c->jv[r] * c->jv[r];
      cost += alpha * alpha * q2 + alpha * q1 + q0;
      d0 += 2 * alpha * q2 + q1;
      d1 += 2 * q2;
    }
  }
  if (d1 <= 0) d1 = MINVAL;
  p->alpha = alpha; p->cost = cost; p->d0 = d0; p->d1 = d1;
  c->ls_iter++;
}

static int update_bracket(nctx_t* c, lspnt_t* p, const lspnt_t* cand, lspnt_t* pnext) {
  int flag = 0;
  for (int i = 0; i < 3; i++) {
    if (p->d0 < 0 && cand[i].d0 < 0 && p->d0 < cand[i].d0) { *p = cand[i]; flag = 1; }
    else if (p->d0 > 0 && cand[i].d0 > 0 && p->d0 > cand[i].d0) { *p = cand[i]; flag = 2; }
  }
  if (flag) ls_eval(c, pnext, p->alpha - p->d0 / p->d1);
  return flag;
}

static real line_search(const mjo_model_t* m, nctx_t* c) {
  int nv = c->nv, lsmax = m->opt.ls_iterations;
  real snorm = 0;
  for (int i = 0; i < nv; i++) snorm += c->search[i] * c->search[i];
  snorm = sqrt(snorm);
  c->ls_iter = 0;
  mjo_dbg_searches++;
  if (snorm < MINVAL) return 0;
  real scale = (real)m->opt.meaninertia * (nv > 1 ? nv : 1);
  real gtol = (real)m->opt.tolerance * (real)m->opt.ls_tolerance * snorm * scale;
  /* prepare: Mv, jv, Gauss quadratic */
  for (int i = 0; i < nv; i++) {
    real t = 0;
    for (int j = 0; j < nv; j++) t += c->M[i * nv + j] * c->search[j];
    c->Mv[i] = t;
  }
  for (int r = 0; r < c->nefc; r++) {
    real t = 0;
    const real* row = c->J + (size_t)r * nv;
    for (int i = 0; i < nv; i++) t += row[i] * c->search[i];
    c->jv[r] = t;
  }
  c->quad_gauss[0] = c->gauss;
  c->quad_gauss[1] = 0;
  c->quad_gauss[2] = 0;
  for (int i = 0; i < nv; i++) {
    c->quad_gauss[1] += c->search[i] * (c->Ma[i] - c->qfrc_smooth[i]);
    c->quad_gauss[2] += (real)0.5 * c->search[i] * c->Mv[i];
  }
  /* The derivative d0(alpha) = sum_r (q1_r + 2 alpha q2_r) + Gauss terms is a sum of large terms
   * that cancel at the minimiser: it cannot be evaluated below a few ulps of them.  The tolerance on
   * it is MuJoCo's gtol, but never less than that noise band (4 ulps).  In fp64 the band is far
   * below gtol and never binds; in fp32 (MJO_FLOAT build, the HIP kernels) gtol ~ 1e-7 is not
   * resolvable and without the band the search burns its evaluations on noise. */
  real dn1 = fabs(c->quad_gauss[1]), dn2 = fabs(c->quad_gauss[2]);
  for (int r = 0; r < c->nefc; r++) {
    real dj = c->Dv[r] * c->jv[r];
    dn1 += fabs(dj * c->jar[r]);
    dn2 += fabs((real)0.5 * dj * c->jv[r]);
  }
  const real ulp4 = (m->opt.flags & MJLAB_OPT_LITERAL_TERMINATION) ? 0 : 4 * (sizeof(real) == 4 ? (real)5.9604645e-8 : (real)1.1102230246251565e-16);
  dn1 *= ulp4;
  dn2 *= 2 * ulp4;
#define LS_TOL(alpha_) (gtol > dn1 + fabs(alpha_) * dn2 ? gtol : dn1 + fabs(alpha_) * dn2)
  lspnt_t p0, p1, p2, pmid, p1next, p2next;
  if (m->opt.flags & MJLAB_OPT_LS_PARALLEL) {
    /* mujoco_warp's parallel line search (what the reference configures, src/mjlab/sim/sim.py:89,111): the cost at
     * ls_iterations log-spaced step sizes in [ls_parallel_min_step, 1], lowest cost wins, the first one on ties; the cost at
     * alpha = 0 is not a candidate.  Grid definition recalled from mujoco_warp (_log_scale / linesearch_parallel_best_alpha);
     * UNVERIFIED against the pinned source. */
    real lo = (real)log(m->opt.ls_parallel_min_step), step = (0 - lo) / (real)(lsmax > 1 ? lsmax - 1 : 1), best_alpha = 0, best_cost = 0;
    for (int i = 0; i < lsmax; i++) {
      real alpha = (real)exp(lo + (real)i * step);
      /* candidates compared by cost(alpha) - cost(0), formed row by row as a product of differences like the device kernel
       * (stage_solve.h line_search_parallel): the same argmin in exact arithmetic.  The fp32 build takes this form unless
       * MJLAB_OPT_LS_LITERAL_COST asks for the literal totals; the fp64 build is literal either way (include/mjlab_fields.h) */
      if (sizeof(real) == 4 && !(m->opt.flags & MJLAB_OPT_LS_LITERAL_COST)) {
        real acc = 0;
        for (int r = c->nf; r < c->nefc; r++) {
          if (cone_start(c, r)) { /* (no product-of-differences form for a cone: the difference of its two costs) */
            real x[3], fo[3], ca, c0;
            for (int k = 0; k < 3; k++) x[k] = c->jar[r + k] + alpha * c->jv[r + k];
            cone_eval(c, r, x, &ca, fo, NULL);
            cone_eval(c, r, c->jar + r, &c0, fo, NULL);
            acc += 2 * (ca - c0);
            r += 2;
            continue;
          }
          real x = c->jar[r] + alpha * c->jv[r], xm = x < 0 ? x : 0, xm0 = c->jar[r] < 0 ? c->jar[r] : 0;
          acc += c->Dv[r] * (xm - xm0) * (xm + xm0);
        }
        for (int r = 0; r < c->nf; r++) { /* friction loss (Huber cost), as a difference too */
          real x = c->jar[r] + alpha * c->jv[r], fl = c->floss[r], rf = fl / c->Dv[r], ax = fabs(x), a0 = fabs(c->jar[r]);
          real ha = ax >= rf ? 2 * fl * (ax - (real)0.5 * rf) : c->Dv[r] * x * x;
          real h0 = a0 >= rf ? 2 * fl * (a0 - (real)0.5 * rf) : c->Dv[r] * c->jar[r] * c->jar[r];
          acc += ha - h0;
        }
        p0.cost = (real)0.5 * acc + alpha * (alpha * c->quad_gauss[2] + c->quad_gauss[1]);
      } else {
        ls_eval(c, &p0, alpha);
      }
      if (i == 0 || p0.cost < best_cost) { best_cost = p0.cost; best_alpha = alpha; }
    }
    return best_alpha;
  }
  ls_eval(c, &p0, 0);
  ls_eval(c, &p1, p0.alpha - p0.d0 / p0.d1);
  if (p0.cost < p1.cost) p1 = p0;
  if (fabs(p1.d0) < LS_TOL(p1.alpha)) return p1.alpha;
  int dir = p1.d0 < 0 ? 1 : -1;
  int p2update = 0;
  p2 = p1;
  while (p1.d0 * dir <= -LS_TOL(p1.alpha) && c->ls_iter < lsmax) {
    p2 = p1;
    p2update = 1;
    ls_eval(c, &p1, p1.alpha - p1.d0 / p1.d1);
    if (fabs(p1.d0) < LS_TOL(p1.alpha)) return p1.alpha;
  }
  if (c->ls_iter >= lsmax) return p1.alpha;
  if (!p2update) return p1.alpha;
  p2next = p1;
  ls_eval(c, &p1next, p1.alpha - p1.d0 / p1.d1);
  while (c->ls_iter < lsmax) {
    ls_eval(c, &pmid, (real)0.5 * (p1.alpha + p2.alpha));
    lspnt_t cand[3] = {p1next, p2next, pmid};
    real bestcost = 0;
    int best = -1;
    for (int i = 0; i < 3; i++)
      if (fabs(cand[i].d0) < LS_TOL(cand[i].alpha) && (best == -1 || cand[i].cost < bestcost)) { bestcost = cand[i].cost; best = i; }
    if (best >= 0) return cand[best].alpha;
    int b1 = update_bracket(c, &p1, cand, &p1next);
    int b2 = update_bracket(c, &p2, cand, &p2next);
    if (!b1 && !b2) return pmid.cost < p0.cost ? pmid.alpha : 0;
  }
  if (p1.cost <= p2.cost && p1.cost < p0.cost) return p1.alpha;
  if (p2.cost <= p1.cost && p2.cost < p0.cost) return p2.alpha;
  return 0;
#undef LS_TOL
}

static real constraint_cost_at(nctx_t* c, const real* qacc, int with_gauss) {
  int nv = c->nv;
  real cost = 0;
  for (int r = 0; r < c->nefc; r++) {
    int nr = cone_start(c, r) ? 3 : 1;
    real x[3], rc, fo[3];
    for (int k = 0; k < nr; k++) {
      const real* row = c->J + (size_t)(r + k) * nv;
      x[k] = -c->aref[r + k];
      for (int i = 0; i < nv; i++) x[k] += row[i] * qacc[i];
    }
    if (nr == 3) cone_eval(c, r, x, &rc, fo, NULL);
    else row_cost(c, r, x[0], &rc, fo);
    cost += rc;
    r += nr - 1;
  }
  if (with_gauss)
    for (int i = 0; i < nv; i++) {
      real ma = 0;
      for (int j = 0; j < nv; j++) ma += c->M[i * nv + j] * qacc[j];
      cost += (real)0.5 * (ma - c->qfrc_smooth[i]) * (qacc[i] - c->qacc_smooth[i]);
    }
  return cost;
}

/* ------------------------------------------------------------------ PGS (dual) solver: mj_solPGS with scalar rows
 * (pyramidal / frictionless contacts, limits, friction loss: every row is its own block).  AR = J M^-1 J^T + diag(R), R = 1 / D,
 * b = J qacc_smooth - aref.  Warm start (mj_fwdConstraint's warmstart()): forces of the constraint update at qacc_warmstart,
 * kept if their dual cost 0.5 f' AR f + f' b is negative (cost of zero force = 0), else zero.  Sweep: row by row
 * f_r -= res_r / AR_rr with res_r = b_r + (AR f)_r, projected on f >= 0 (inequality rows) or |f| <= frictionloss; a row
 * update that would raise the cost by more than 1e-10 is undone (costChange); stop when the cost improvement of a sweep, scaled
 * by 1 / (meaninertia max(1, nv)), drops below tolerance.  Then qfrc_constraint = J' f, qacc = qacc_smooth + M^-1 J' f.
 * AR is never formed: B_r = M^-1 J_r' is kept per row (data.efc_B) and v = sum_r f_r B_r = M^-1 J' f is carried along. */
static void solve_pgs(const mjo_model_t* m, mjo_data_t* d, int w) {
  const mjlab_sizes_t* s = &m->size;
  int nv = s->nv, njm = s->njmax, nefc = d->nefc[w], nf = d->nf[w];
  real *qacc = D(qacc, nv), *ws = D(qacc_warmstart, nv), *qas = D(qacc_smooth, nv), *force = D(efc_force, njm), *fc = D(qfrc_constraint, nv);
  const real *J = D(efc_J, njm * nv), *Dv = D(efc_D, njm), *aref = D(efc_aref, njm), *floss = D(efc_frictionloss, njm), *L = D(qLD, nv * nv);
  real* B = D(efc_B, njm * nv);
  real* buf = (real*)calloc((size_t)nv + 2 * (size_t)nefc, sizeof(real));
  real *v = buf, *b = buf + nv, *ARinv = b + nefc;
  for (int r = 0; r < nefc; r++) {
    const real* row = J + (size_t)r * nv;
    real* Br = B + (size_t)r * nv;
    memcpy(Br, row, sizeof(real) * nv);
    chol_solve(L, nv, Br);
    real arr = 1 / Dv[r], br = -aref[r], x = -aref[r];
    for (int i = 0; i < nv; i++) { arr += row[i] * Br[i]; br += row[i] * qas[i]; x += row[i] * ws[i]; }
    ARinv[r] = 1 / arr;
    b[r] = br;
    /* constraint update at the warm-start acceleration (row_cost's force rule) */
    real f;
    if (r < nf) { real rf = floss[r] / Dv[r]; f = x <= -rf ? floss[r] : (x >= rf ? -floss[r] : -Dv[r] * x); }
    else f = x < 0 ? -Dv[r] * x : 0;
    force[r] = f;
    for (int i = 0; i < nv; i++) v[i] += f * Br[i];
  }
  real cost = 0;
  for (int r = 0; r < nefc; r++) {
    const real* row = J + (size_t)r * nv;
    real jv = 0;
    for (int i = 0; i < nv; i++) jv += row[i] * v[i];
    cost += force[r] * ((real)0.5 * (jv + force[r] / Dv[r]) + b[r]);
  }
  if (cost > 0) {
    for (int r = 0; r < nefc; r++) force[r] = 0;
    for (int i = 0; i < nv; i++) v[i] = 0;
  }
  real scale = 1 / ((real)m->opt.meaninertia * (nv > 1 ? nv : 1));
  int iter = 0;
  while (iter < m->opt.iterations) {
    real improvement = 0;
    for (int r = 0; r < nefc; r++) {
      const real *row = J + (size_t)r * nv, *Br = B + (size_t)r * nv;
      real res = b[r] + force[r] / Dv[r];
      for (int i = 0; i < nv; i++) res += row[i] * v[i];
      real old = force[r], f = old - res * ARinv[r];
      if (r < nf) { if (f < -floss[r]) f = -floss[r]; else if (f > floss[r]) f = floss[r]; }
      else if (f < 0) f = 0;
      real delta = f - old, change = (real)0.5 * delta * delta / ARinv[r] + delta * res;
      if (change > (real)1e-10) { f = old; delta = 0; change = 0; }
      if (delta != 0) for (int i = 0; i < nv; i++) v[i] += delta * Br[i];
      force[r] = f;
      improvement -= change;
    }
    iter++;
    if (scale * improvement < (real)m->opt.tolerance) break;
  }
  memset(fc, 0, sizeof(real) * nv);
  for (int r = 0; r < nefc; r++) {
    if (force[r] == 0) continue;
    const real* row = J + (size_t)r * nv;
    for (int i = 0; i < nv; i++) fc[i] += row[i] * force[r];
  }
  for (int i = 0; i < nv; i++) qacc[i] = qas[i] + v[i];
  d->solver_niter[w] = iter;
  if (!(m->opt.flags & MJLAB_OPT_WARMSTART_AT_ADVANCE)) memcpy(ws, qacc, sizeof(real) * nv);
  free(buf);
}

static void solve(const mjo_model_t* m, mjo_data_t* d, int w) {
  const mjlab_sizes_t* s = &m->size;
  int nv = s->nv, njm = s->njmax, nefc = d->nefc[w];
  real *qacc = D(qacc, nv), *ws = D(qacc_warmstart, nv), *qas = D(qacc_smooth, nv);
  real* force = D(efc_force, njm);
  if (nefc == 0) {
    memcpy(qacc, qas, sizeof(real) * nv);
    if (!(m->opt.flags & MJLAB_OPT_WARMSTART_AT_ADVANCE)) memcpy(ws, qas, sizeof(real) * nv);
    memset(D(qfrc_constraint, nv), 0, sizeof(real) * nv);
    d->solver_niter[w] = 0;
    return;
  }
  if (m->opt.solver == MJLAB_SOL_PGS) { solve_pgs(m, d, w); return; }
  nctx_t c;
  c.nv = nv; c.nefc = nefc; c.nf = d->nf[w];
  c.J = D(efc_J, njm * nv); c.Dv = D(efc_D, njm); c.aref = D(efc_aref, njm); c.floss = D(efc_frictionloss, njm); c.M = D(qM, nv * nv);
  c.qfrc_smooth = D(qfrc_smooth, nv); c.qacc_smooth = qas; c.qacc = qacc; c.force = force;
  c.qfrc_constraint = D(qfrc_constraint, nv);
  c.type = m->opt.cone == MJLAB_CONE_ELLIPTIC ? d->efc_type + (size_t)w * njm : NULL;
  c.id = d->efc_id + (size_t)w * njm;
  c.cadr = d->contact_efc_address + (size_t)w * s->nconmax;
  c.fri = D(contact_friction, 5 * s->nconmax);
  c.impratio = (real)m->opt.impratio;
  c.cg = m->opt.solver == MJLAB_SOL_CG;
  c.L = D(qLD, nv * nv);
  real* buf = (real*)calloc((size_t)8 * nv + 2 * nefc + (size_t)nv * nv, sizeof(real));
  c.Ma = buf; c.grad = buf + nv; c.search = buf + 2 * nv; c.Mv = buf + 3 * nv;
  c.Mgrad = buf + 5 * nv; c.gradold = buf + 6 * nv; c.Mgradold = buf + 7 * nv;
  c.jar = buf + 8 * nv; c.jv = c.jar + nefc; c.H = c.jv + nefc;
  /* warmstart: better of qacc_warmstart and qacc_smooth */
  real cw = constraint_cost_at(&c, ws, 1), cs = constraint_cost_at(&c, qas, 0);
  memcpy(qacc, cw > cs ? qas : ws, sizeof(real) * nv);
  for (int i = 0; i < nv; i++) {
    real t = 0;
    for (int j = 0; j < nv; j++) t += c.M[i * nv + j] * qacc[j];
    c.Ma[i] = t;
  }
  for (int r = 0; r < nefc; r++) {
    const real* row = c.J + (size_t)r * nv;
    real x = -c.aref[r];
    for (int i = 0; i < nv; i++) x += row[i] * qacc[i];
    c.jar[r] = x;
  }
  real scale = 1 / ((real)m->opt.meaninertia * (nv > 1 ? nv : 1));
  update_constraint(&c);
  update_gradient(&c);
  if (c.cg) for (int i = 0; i < nv; i++) c.search[i] = -c.Mgrad[i];
  int iter = 0;
  while (iter < m->opt.iterations) {
    real alpha = line_search(m, &c);
    if (alpha == 0) break;
    for (int i = 0; i < nv; i++) { qacc[i] += alpha * c.search[i]; c.Ma[i] += alpha * c.Mv[i]; }
    for (int r = 0; r < nefc; r++) c.jar[r] += alpha * c.jv[r];
    real oldcost = c.cost;
    if (c.cg) { memcpy(c.gradold, c.grad, sizeof(real) * nv); memcpy(c.Mgradold, c.Mgrad, sizeof(real) * nv); }
    update_constraint(&c);
    update_gradient(&c);
    if (c.cg) { /* Polak-Ribiere: beta = grad . (Mgrad - Mgradold) / (gradold . Mgradold), clamped at 0 (mj_solPrimal) */
      real num = 0, den = 0;
      for (int i = 0; i < nv; i++) { num += c.grad[i] * (c.Mgrad[i] - c.Mgradold[i]); den += c.gradold[i] * c.Mgradold[i]; }
      real beta = num / (den > MINVAL ? den : MINVAL);
      if (beta < 0) beta = 0;
      for (int i = 0; i < nv; i++) c.search[i] = -c.Mgrad[i] + beta * c.search[i];
    }
    real improvement = scale * (oldcost - c.cost), gn = 0, tn = 0;
    for (int i = 0; i < nv; i++) {
      real t = fabs(c.Ma[i]) + fabs(c.qfrc_smooth[i]) + fabs(c.qfrc_constraint[i]);
      gn += c.grad[i] * c.grad[i];
      tn += t * t;
    }
    real gradient = scale * sqrt(gn);
    /* rounding noise of the gradient in this precision (4 ulps of the terms it is the difference
     * of): below it another Newton step is noise.  Never binds before the tolerance in fp64; in
     * fp32 (this file's MJO_FLOAT build, the HIP kernels) it is what ends the iteration, because
     * neither `improvement` nor `gradient` can resolve 1e-8 there. */
    real noise = 4 * (sizeof(real) == 4 ? (real)5.9604645e-8 : (real)1.1102230246251565e-16) * scale * sqrt(tn);
    if (m->opt.flags & MJLAB_OPT_LITERAL_TERMINATION) noise = 0;
    iter++;
    if (improvement < (real)m->opt.tolerance || gradient < (real)m->opt.tolerance || gradient < noise) break;
  }
  d->solver_niter[w] = iter;
  if (!(m->opt.flags & MJLAB_OPT_WARMSTART_AT_ADVANCE)) memcpy(ws, qacc, sizeof(real) * nv);
  free(buf);
}

/* ------------------------------------------------------------------ sensors */
static int in_subtree(const mjo_model_t* m, int body, int root) {
  while (body > 0 && body != root) body = m->body_parentid[body];
  return body == root;
}
static int sensor_match(const mjo_model_t* m, int type, int id, int geom) {
  if (type < 0) return 1;
  int b = m->geom_bodyid[geom];
  if (type == MJLAB_OBJ_GEOM) return geom == id;
  if (type == MJLAB_OBJ_BODY) return b == id;
  if (type == MJLAB_OBJ_XBODY) return in_subtree(m, b, id);
  return 0;
}
static void sensors(const mjo_model_t* m, mjo_data_t* d, int w) {
  const mjlab_sizes_t* s = &m->size;
  int ncm = s->nconmax, ncon = d->ncon[w];
  real* sd = D(sensordata, s->nsensordata);
  for (int i = 0; i < s->nsensordata; i++) sd[i] = 0;
  const int* cg = d->contact_geom + (size_t)w * 2 * ncm;
  const int* cadr = d->contact_efc_address + (size_t)w * ncm;
  for (int k = 0; k < s->nsensor; k++) {
    int cnt = 0;
    for (int c = 0; c < ncon; c++) {
      if (cadr[c] < 0) continue;
      int g1 = cg[2 * c], g2 = cg[2 * c + 1];
      int ot = m->sensor_objtype[k], oi = m->sensor_objid[k], rt = m->sensor_reftype[k], ri = m->sensor_refid[k];
      if ((sensor_match(m, ot, oi, g1) && sensor_match(m, rt, ri, g2)) ||
          (sensor_match(m, ot, oi, g2) && sensor_match(m, rt, ri, g1)))
        cnt++;
    }
    /* dataspec = found only: slot 0 holds the number of matching contacts */
    sd[m->sensor_adr[k]] = (real)cnt;
  }
}

/* ------------------------------------------------------------------ integration */
static void integrate(const mjo_model_t* m, mjo_data_t* d, int w) {
  const mjlab_sizes_t* s = &m->size;
  int nv = s->nv, nu = s->nu;
  real h = (real)m->opt.timestep;
  real *qpos = D(qpos, s->nq), *qvel = D(qvel, nv), *qacc = D(qacc, nv);
  real* a = (real*)malloc(sizeof(real) * (nv + (size_t)nv * nv));
  real* A = a + nv;
  const real* damping = MF(dof_damping, w);
  int need_solve = 0;
  real* diag = (real*)calloc(nv, sizeof(real));
  for (int i = 0; i < nv; i++) { diag[i] = damping[i]; if (damping[i] > 0) need_solve = 1; }
  if (m->opt.integrator == MJLAB_INT_IMPLICITFAST) {
    /* qDeriv restricted to its diagonal: joint-transmission actuator velocity gains + dof damping */
    const real *biasprm = MF(actuator_biasprm, w), *gear = MF(actuator_gear, w), *frange = MF(actuator_forcerange, w);
    real* aforce = D(actuator_force, nu);
    for (int k = 0; k < nu; k++) {
      if (m->actuator_forcelimited[k] && (aforce[k] <= frange[2 * k] || aforce[k] >= frange[2 * k + 1])) continue;
      int da = m->jnt_dofadr[m->actuator_trnid[2 * k]];
      diag[da] -= gear[6 * k] * gear[6 * k] * biasprm[10 * k + 2];
    }
    need_solve = 1;
  }
  if (need_solve) {
    memcpy(A, D(qM, nv * nv), sizeof(real) * nv * nv);
    for (int i = 0; i < nv; i++) A[i * nv + i] += h * diag[i];
    real *smooth = D(qfrc_smooth, nv), *qc = D(qfrc_constraint, nv);
    for (int i = 0; i < nv; i++) a[i] = smooth[i] + qc[i];
    chol_factor(A, nv);
    chol_solve(A, nv, a);
  } else memcpy(a, qacc, sizeof(real) * nv);
  if (m->opt.flags & MJLAB_OPT_WARMSTART_AT_ADVANCE) memcpy(D(qacc_warmstart, nv), qacc, sizeof(real) * nv);
  for (int i = 0; i < nv; i++) qvel[i] += h * a[i];
  for (int j = 0; j < s->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == MJLAB_JNT_FREE) {
      for (int k = 0; k < 3; k++) qpos[qa + k] += h * qvel[da + k];
      real ax[3] = {qvel[da + 3], qvel[da + 4], qvel[da + 5]}, q[4], qr[4], *quat = qpos + qa + 3;
      real ang = h * normalize3(ax);
      axis_angle2quat(qr, ax, ang);
      normalize4(quat);
      mul_quat(q, quat, qr);
      memcpy(quat, q, sizeof(q));
      normalize4(quat);
    } else qpos[qa] += h * qvel[da];
  }
  d->time[w] += h;
  free(a);
  free(diag);
}

/* ------------------------------------------------------------------ public API */
void mjo_forward(const mjo_model_t* m, mjo_data_t* d, int w) {
  kinematics(m, d, w);
  com_pos(m, d, w);
  crb_factor(m, d, w);
  collision(m, d, w);
  make_constraint(m, d, w);
  com_vel(m, d, w);
  rne_bias(m, d, w);
  smooth_forces(m, d, w);
  solve(m, d, w);
  sensors(m, d, w);
}

void mjo_step(const mjo_model_t* m, mjo_data_t* d, int w) {
  mjo_forward(m, d, w);
  integrate(m, d, w);
}

typedef struct { const mjo_model_t* m; mjo_data_t* d; int w0, w1, nstep, fwd_only; } job_t;
static void* worker(void* arg) {
  job_t* j = (job_t*)arg;
  for (int w = j->w0; w < j->w1; w++)
    for (int k = 0; k < j->nstep; k++) {
      if (j->fwd_only) mjo_forward(j->m, j->d, w); else mjo_step(j->m, j->d, w);
    }
  return 0;
}
/* Poses of the static geoms (world / terrain bodies) of every world; to be called once after the
 * data arrays are (re)initialised -- the per-step kinematics skips them, like the HIP path. */
void mjo_static_geoms(const mjo_model_t* m, mjo_data_t* d) {
  const mjlab_sizes_t* s = &m->size;
  for (int w = 0; w < s->nworld; w++) {
    real *gx = D(geom_xpos, 3 * s->ngeom), *gm = D(geom_xmat, 9 * s->ngeom);
    for (int g = 0; g < s->nstaticgeom; g++) {
      /* static bodies: compose the (constant) chain up to the world */
      real pos[3] = {0, 0, 0}, quat[4] = {1, 0, 0, 0}, mat[9];
      int chain[64], n = 0;
      for (int b = m->geom_bodyid[g]; b > 0 && n < 64; b = m->body_parentid[b]) chain[n++] = b;
      for (int k = n - 1; k >= 0; k--) {
        real p[3], q[4];
        rot_vec_quat(p, MF(body_pos, w) + 3 * chain[k], quat);
        for (int i = 0; i < 3; i++) pos[i] += p[i];
        mul_quat(q, quat, MF(body_quat, w) + 4 * chain[k]);
        normalize4(q);
        memcpy(quat, q, sizeof(q));
      }
      quat2mat(mat, quat);
      local2global(gx + 3 * g, gm + 9 * g, pos, quat, mat, MF(geom_pos, w) + 3 * g, MF(geom_quat, w) + 4 * g);
    }
    real *sx = D(site_xpos, 3 * s->nsite), *sm = D(site_xmat, 9 * s->nsite);
    for (int g = 0; g < s->nstaticsite; g++) {
      real pos[3] = {0, 0, 0}, quat[4] = {1, 0, 0, 0}, mat[9];
      int chain[64], n = 0;
      for (int b = m->site_bodyid[g]; b > 0 && n < 64; b = m->body_parentid[b]) chain[n++] = b;
      for (int k = n - 1; k >= 0; k--) {
        real p[3], q[4];
        rot_vec_quat(p, MF(body_pos, w) + 3 * chain[k], quat);
        for (int i = 0; i < 3; i++) pos[i] += p[i];
        mul_quat(q, quat, MF(body_quat, w) + 4 * chain[k]);
        normalize4(q);
        memcpy(quat, q, sizeof(q));
      }
      quat2mat(mat, quat);
      local2global(sx + 3 * g, sm + 9 * g, pos, quat, mat, MF(site_pos, w) + 3 * g, MF(site_quat, w) + 4 * g);
    }
  }
}

/* runs `nstep` steps (or one forward when nstep == 0) on every world with `nthread` threads */
void mjo_run(const mjo_model_t* m, mjo_data_t* d, int nstep, int nthread) {
  int nw = m->size.nworld;
  if (nthread < 1) nthread = 1;
  if (nthread > nw) nthread = nw;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthread);
  job_t* jobs = (job_t*)malloc(sizeof(job_t) * nthread);
  for (int t = 0; t < nthread; t++) {
    jobs[t].m = m; jobs[t].d = d;
    jobs[t].w0 = (int)((long long)nw * t / nthread);
    jobs[t].w1 = (int)((long long)nw * (t + 1) / nthread);
    jobs[t].nstep = nstep > 0 ? nstep : 1;
    jobs[t].fwd_only = nstep == 0;
    if (nthread == 1) worker(&jobs[t]); else pthread_create(&th[t], 0, worker, &jobs[t]);
  }
  if (nthread > 1) for (int t = 0; t < nthread; t++) pthread_join(th[t], 0);
  free(th);
  free(jobs);
}
