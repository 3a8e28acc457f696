// mjlab_amd.hip -- C ABI of the batched physics step (include/mjlab_amd.h) and the kernels that do not depend on
// the padded dof count.  The solve / substep / control-step kernels are instantiated per padded size NVP in
// nvp_inst.hip (one translation unit per size, compiled in parallel; mjlab_amd/native.py) and reached through the
// launch functions declared below.
#define MJLAB_MAIN_TU
#include "kernels.h"
#include "nvp_launch.h"
#include "env_terms.h"

// ====================================================================================
// C ABI
// ====================================================================================
static thread_local char g_err[512] = "";
static int fail(int code, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s (code %d%s%s)", what, code, code > 0 ? ": " : "", code > 0 ? hipGetErrorString((hipError_t)code) : "");
  return code;
}

extern "C" {

int mjlab_abi_version(void) { return MJLAB_ABI_VERSION; }
const char* mjlab_last_error(void) { return g_err; }
const char* mjlab_model_layout(void) { return MJLAB_MODEL_LAYOUT_STRING; }
const char* mjlab_data_layout(void) { return MJLAB_DATA_LAYOUT_STRING; }
int mjlab_sizeof_model(void) { return (int)sizeof(mjlab_model_t); }
int mjlab_sizeof_data(void) { return (int)sizeof(mjlab_data_t); }
int mjlab_sizeof_option(void) { return (int)sizeof(mjlab_option_t); }
int mjlab_sizeof_sizes(void) { return (int)sizeof(mjlab_sizes_t); }
int mjlab_sizeof_control(void) { return (int)sizeof(mjlab_control_t); }
int mjlab_sizeof_motion_reset(void) { return (int)sizeof(mjlab_motion_reset_t); }
int mjlab_sizeof_velocity_command(void) { return (int)sizeof(mjlab_velocity_command_t); }

// the solve kernel's LDS block: the primal solvers' layout, or the dual solver's where that one is configured
static int solve_stage_lds_floats(const mjlab_model_t* m) {
  const int a = solve_lds_floats(m->size), b = m->opt.solver == MJLAB_SOL_PGS ? pgs_lds_floats(m->size) : 0;
  const int c = m->opt.cone == MJLAB_CONE_ELLIPTIC ? cone_lds_floats(m->size) : 0;
  return a > b ? (a > c ? a : c) : (b > c ? b : c);
}
int mjlab_lds_bytes(const mjlab_model_t* m, int stage) {
  switch (stage) {
    case MJLAB_STAGE_POSITION: return 4 * position_lds_floats(m->size);
    case MJLAB_STAGE_COLLISION: return 4 * collision_lds_floats(m->size);
    case MJLAB_STAGE_VELOCITY: return 4 * velocity_lds_floats(m->size);
    case MJLAB_STAGE_CONSTRAINT: return 4 * constraint_lds_floats(m->size);
    case MJLAB_STAGE_SOLVE:
    case MJLAB_STAGE_INTEGRATE: return 4 * solve_stage_lds_floats(m);
  }
  return -1;
}

static int check_model(const mjlab_model_t* m) {
  const mjlab_sizes_t& s = m->size;
  if (s.nworld < 1) return fail(-2, "nworld must be >= 1");
  if (s.nv < 1 || s.nv > 64) return fail(-3, "nv must be in [1, 64] (one dof per lane)");
  if (s.nbody < 1 || s.nbody > 64) return fail(-13, "nbody must be in [1, 64] (one body per lane in the kinematics sweep)");
  if (s.njmax < 1 || s.nconmax < 1) return fail(-4, "njmax and nconmax must be >= 1");
  if (s.ngeom > 65535) return fail(-19, "ngeom must be < 65536 (geom pairs are packed into one word)");
  if (m->opt.cone != MJLAB_CONE_PYRAMIDAL && m->opt.cone != MJLAB_CONE_ELLIPTIC) return fail(-5, "opt.cone must be MJLAB_CONE_PYRAMIDAL or MJLAB_CONE_ELLIPTIC");
  if (m->opt.cone == MJLAB_CONE_ELLIPTIC && ((m->opt.solver != MJLAB_SOL_NEWTON && m->opt.solver != MJLAB_SOL_CG) || (m->opt.flags & MJLAB_OPT_FUSE_PRESOLVE)))
    return fail(-5, "MJLAB_CONE_ELLIPTIC runs with MJLAB_SOL_NEWTON or MJLAB_SOL_CG, one kernel per stage or MJLAB_OPT_FUSE_STEP (no MJLAB_OPT_FUSE_PRESOLVE)");
  if (m->opt.cone == MJLAB_CONE_ELLIPTIC && !(m->opt.impratio > 0)) return fail(-5, "opt.impratio must be positive (the elliptic cone's friction rows are scaled by 1 / impratio)");
  if (m->opt.integrator != MJLAB_INT_EULER && m->opt.integrator != MJLAB_INT_IMPLICITFAST)
    return fail(-6, "integrator must be Euler or implicitfast");
  if (m->opt.solver != MJLAB_SOL_CG && m->opt.solver != MJLAB_SOL_NEWTON && m->opt.solver != MJLAB_SOL_PGS)
    return fail(-20, "opt.solver must be MJLAB_SOL_NEWTON, MJLAB_SOL_CG or MJLAB_SOL_PGS");
  if (m->opt.solver == MJLAB_SOL_PGS && (m->opt.flags & (MJLAB_OPT_FUSE_PRESOLVE | MJLAB_OPT_FUSE_STEP)))
    return fail(-20, "MJLAB_SOL_PGS runs with one kernel per stage only (clear MJLAB_OPT_FUSE_PRESOLVE / MJLAB_OPT_FUSE_STEP)");
  for (int st = 1; st <= 16; st <<= 1)
    if (mjlab_lds_bytes(m, st) > 160 * 1024) return fail(-7, "model too large for the LDS-resident stage kernels");
  return 0;
}

#define LAUNCH(kernel, ldsfloats, ...)                                                                      \
  do {                                                                                                      \
    hipLaunchKernelGGL(kernel, dim3(m->size.nworld), dim3(64), (size_t)4 * (ldsfloats), st, __VA_ARGS__);  \
    hipError_t e_ = hipGetLastError();                                                                      \
    if (e_ != hipSuccess) return fail((int)e_, #kernel " launch failed");                                   \
  } while (0)

static int launch_solve(const mjlab_model_t* m, const mjlab_data_t* d, int do_solve, int do_integrate, int flags, hipStream_t st) {
  const NvpLaunch* L = nvp_launch(solve_nvp(m->size.nv));
  if (!L) return fail(-3, "nv must be in [1, 64]");
  if (do_solve && m->opt.cone == MJLAB_CONE_ELLIPTIC) {  // the cone solver's own kernel, then the integrator with the solve switched off
    hipError_t e = L->cone(m, d, flags, 4 * solve_stage_lds_floats(m), st);
    if (e != hipSuccess) return fail((int)e, "k_solve_cone launch failed");
    if (!do_integrate) return 0;
    do_solve = 0;
  }
  hipError_t e = L->solve(m, d, do_solve, do_integrate, flags, 4 * solve_stage_lds_floats(m), st);
  if (e != hipSuccess) return fail((int)e, "k_solve_integrate launch failed");
  return 0;
}

static int max4(int a, int b, int c, int e) { return (a > b ? a : b) > (c > e ? c : e) ? (a > b ? a : b) : (c > e ? c : e); }
static int presolve_lds_floats(const mjlab_sizes_t& s) {
  return max4(position_lds_floats(s), collision_lds_floats(s), velocity_lds_floats(s), constraint_lds_floats(s));
}
static int launch_substep(const mjlab_model_t* m, const mjlab_data_t* d, int do_integrate, int flags, int nsub, hipStream_t st) {
  const int a = presolve_lds_floats(m->size), b = solve_stage_lds_floats(m), lds = a > b ? a : b;
  const NvpLaunch* L = nvp_launch(solve_nvp(m->size.nv));
  if (!L) return fail(-3, "nv must be in [1, 64]");
  const bool cone = m->opt.cone == MJLAB_CONE_ELLIPTIC;
  hipError_t e = do_integrate ? (cone ? L->step_cone : L->step)(m, d, flags, nsub, 4 * lds, st) : (cone ? L->forward_cone : L->forward)(m, d, flags, 1, 4 * lds, st);
  if (e != hipSuccess) return fail((int)e, "k_substep launch failed");
  return 0;
}

static int forward_stages_impl(const mjlab_model_t* m, const mjlab_data_t* d, int stages, int flags, void* stream) {
  int rc = check_model(m);
  if (rc) return rc;
  if (m->opt.solver == MJLAB_SOL_PGS && !d->efc_B) return fail(-21, "MJLAB_SOL_PGS needs mjlab_data_t.efc_B (njmax x nv reals per world); it may be NULL for the primal solvers only");
  hipStream_t st = (hipStream_t)stream;
  const int all = MJLAB_STAGE_FORWARD;
  if ((stages & all) == all && (m->opt.flags & MJLAB_OPT_FUSE_STEP)) {  // a whole forward() / step() as ONE launch
    return launch_substep(m, d, (stages & MJLAB_STAGE_INTEGRATE) != 0, flags, 1, st);  // forward(): the snapshot is taken by the same launch
  }
  const int pre = MJLAB_STAGE_POSITION | MJLAB_STAGE_COLLISION | MJLAB_STAGE_VELOCITY | MJLAB_STAGE_CONSTRAINT;
  if ((stages & pre) == pre && (m->opt.flags & MJLAB_OPT_FUSE_PRESOLVE)) {  // the four pre-solve stages as one launch
    LAUNCH(k_presolve, presolve_lds_floats(m->size), *m, *d, flags);
    stages &= ~pre;
  }
  if (stages & MJLAB_STAGE_POSITION) LAUNCH(k_position, position_lds_floats(m->size), *m, *d, flags);
  if (stages & MJLAB_STAGE_COLLISION) LAUNCH(k_collision, collision_lds_floats(m->size), *m, *d, flags);
  if (stages & MJLAB_STAGE_VELOCITY) LAUNCH(k_velocity, velocity_lds_floats(m->size), *m, *d, flags);
  if (stages & MJLAB_STAGE_CONSTRAINT) {
    if (m->opt.cone == MJLAB_CONE_ELLIPTIC) LAUNCH(k_constraint_cone, constraint_lds_floats(m->size), *m, *d, flags);
    else LAUNCH(k_constraint, constraint_lds_floats(m->size), *m, *d, flags);
  }
  if (stages & (MJLAB_STAGE_SOLVE | MJLAB_STAGE_INTEGRATE)) {
    rc = launch_solve(m, d, (stages & MJLAB_STAGE_SOLVE) != 0, (stages & MJLAB_STAGE_INTEGRATE) != 0, flags, st);
    if (rc) return rc;
  }
  if (flags & FLAG_SNAPSHOT) LAUNCH(k_fold_snapshot, 0, *m, *d, flags);
  return 0;
}

int mjlab_forward_stages(const mjlab_model_t* m, const mjlab_data_t* d, int stages, void* stream) {
  // stage-by-stage execution (testing / profiling) leaves the derived arrays in a state the fold
  // bookkeeping does not describe: drop it
  if (hipMemsetAsync(d->fold_valid, 0, sizeof(int) * (size_t)m->size.nworld, (hipStream_t)stream) != hipSuccess)
    return fail(-18, "forward_stages: memset failed");
  return forward_stages_impl(m, d, stages, 0, stream);
}

int mjlab_forward_masked(const mjlab_model_t* m, const mjlab_data_t* d, void* stream) {
  return forward_stages_impl(m, d, MJLAB_STAGE_FORWARD, FLAG_MASK | FLAG_SNAPSHOT, stream);
}

int mjlab_forward(const mjlab_model_t* m, const mjlab_data_t* d, void* stream) {
  return forward_stages_impl(m, d, MJLAB_STAGE_FORWARD, FLAG_SNAPSHOT, stream);
}

int mjlab_step(const mjlab_model_t* m, const mjlab_data_t* d, int nsubstep, void* stream) {
  if (nsubstep < 1) return fail(-8, "nsubstep must be >= 1");
  if (nsubstep > 1 && (m->opt.flags & MJLAB_OPT_FUSE_STEP)) {  // all substeps in one launch
    int rc = check_model(m);
    if (rc) return rc;
    return launch_substep(m, d, 1, (m->opt.flags & MJLAB_OPT_FOLD_FORWARD) ? FLAG_FOLD : 0, nsubstep, (hipStream_t)stream);
  }
  for (int k = 0; k < nsubstep; ++k) {
    int rc = forward_stages_impl(m, d, MJLAB_STAGE_STEP, (k == 0 && (m->opt.flags & MJLAB_OPT_FOLD_FORWARD)) ? FLAG_FOLD : 0, stream);
    if (rc) return rc;
  }
  return 0;
}

int mjlab_entity_readback(const mjlab_model_t* m, const mjlab_data_t* d, const mjlab_entity_view_t* v, void* stream) {
  int rc = check_model(m);
  if (rc) return rc;
  if (!v || v->nbody < 0 || v->njoint < 0 || v->root_body_id < 0 || v->root_body_id >= m->size.nbody)
    return fail(-14, "entity_readback: bad view");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_entity_readback, dim3(m->size.nworld), dim3(64), 0, st, *m, *d, *v);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "k_entity_readback launch failed");
  return 0;
}

int mjlab_masked_reset(const mjlab_model_t* m, const mjlab_data_t* d, const float* key_qpos, const float* rnd3,
                       int* episode_length, int max_len, float min_height, int* reset_mask, const float* env_origins,
                       float min_up_z, void* stream) {
  int rc = check_model(m);
  if (rc) return rc;
  if (!key_qpos || !rnd3 || !episode_length || !reset_mask) return fail(-15, "masked_reset: null argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_masked_reset, dim3(m->size.nworld), dim3(64), 0, st, *m, *d, key_qpos, rnd3, episode_length, max_len,
                     min_height, reset_mask, env_origins, min_up_z);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "k_masked_reset launch failed");
  return 0;
}

int mjlab_interval_push(const mjlab_model_t* m, const mjlab_data_t* d, float* time_left, const float* rnd7, float dt,
                        float interval_lo, float interval_hi, const mjlab_push_range_t* range, int root_is_free, void* stream) {
  int rc = check_model(m);
  if (rc) return rc;
  if (!time_left || !rnd7 || !range) return fail(-16, "interval_push: null argument");
  if (!root_is_free || m->size.nq < 7 || m->size.nv < 6) return fail(-17, "interval_push: the first joint must be a free joint");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_interval_push, dim3((m->size.nworld + 63) / 64), dim3(64), 0, st, *m, *d, time_left, rnd7, dt, interval_lo, interval_hi, *range);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "k_interval_push launch failed");
  return 0;
}

// ---- environment terms (env_terms.h)
static int launched(const char* what) {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, what);
}
int mjlab_event_reset_root_state_uniform(float* qpos, int nq, int q_adr, float* qvel, int nv, int v_adr, int nworld, const unsigned char* mask,
                                         const float* default_root_state, int ld_root, const float* env_origins, const float* U, int ldu,
                                         const float* pose_range, const float* velocity_range, void* stream) {
  if (!qpos || !qvel || !mask || !default_root_state || !env_origins || !U || !pose_range || !velocity_range) return fail(-22, "reset_root_state_uniform: null argument");
  if (nworld < 1 || q_adr < 0 || q_adr + 7 > nq || v_adr < 0 || v_adr + 6 > nv || ldu < 12) return fail(-22, "reset_root_state_uniform: bad sizes");
  hipLaunchKernelGGL(k_event_reset_root_state_uniform, dim3((nworld + 255) / 256), dim3(256), 0, (hipStream_t)stream, qpos, nq, q_adr, qvel, nv, v_adr,
                     nworld, mask, default_root_state, ld_root, env_origins, U, ldu, pose_range, velocity_range);
  return launched("k_event_reset_root_state_uniform launch failed");
}
int mjlab_event_reset_joints_by_scale(float* qpos, int nq, float* qvel, int nv, int nworld, const unsigned char* mask, int nj, const int* joint_ids,
                                      const int* q_adr, const int* v_adr, const float* default_joint_pos, int ld_jpos, const float* default_joint_vel,
                                      int ld_jvel, const float* soft_joint_pos_limits, int ld_lim, const float* U, int ldu, const float* ranges,
                                      void* stream) {
  if (!qpos || !qvel || !mask || !q_adr || !v_adr || !default_joint_pos || !default_joint_vel || !soft_joint_pos_limits || !U || !ranges)
    return fail(-22, "reset_joints_by_scale: null argument");
  if (nworld < 1 || nj < 1 || ldu < 2 * nj) return fail(-22, "reset_joints_by_scale: bad sizes");
  const long long total = (long long)nworld * nj;
  hipLaunchKernelGGL(k_event_reset_joints_by_scale, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, qpos, nq, qvel, nv, nworld, mask,
                     nj, joint_ids, q_adr, v_adr, default_joint_pos, ld_jpos, default_joint_vel, ld_jvel, soft_joint_pos_limits, ld_lim, U, ldu, ranges);
  return launched("k_event_reset_joints_by_scale launch failed");
}
int mjlab_event_push_by_setting_velocity(float* qvel, int nv, int v_adr, int nworld, float* time_left, float dt, const float* interval_range,
                                         const float* root_link_vel_w, int ld_vel, const float* root_link_quat_w, int ld_quat, const float* U, int ldu,
                                         const float* velocity_range, void* stream) {
  if (!qvel || !time_left || !interval_range || !root_link_vel_w || !root_link_quat_w || !U || !velocity_range) return fail(-22, "push_by_setting_velocity: null argument");
  if (nworld < 1 || v_adr < 0 || v_adr + 6 > nv || ldu < 7) return fail(-22, "push_by_setting_velocity: bad sizes");
  hipLaunchKernelGGL(k_event_push_by_setting_velocity, dim3((nworld + 255) / 256), dim3(256), 0, (hipStream_t)stream, qvel, nv, v_adr, nworld, time_left, dt,
                     interval_range, root_link_vel_w, ld_vel, root_link_quat_w, ld_quat, U, ldu, velocity_range);
  return launched("k_event_push_by_setting_velocity launch failed");
}
int mjlab_command_uniform_velocity(const mjlab_velocity_command_t* c, void* stream) {
  if (!c || !c->U || !c->ranges || !c->time_left || !c->vel_command_b || !c->is_standing_env || !c->command_counter) return fail(-22, "command_uniform_velocity: null argument");
  if (c->heading_command && (!c->heading_target || !c->is_heading_env || (!c->mask && !c->heading_w))) return fail(-22, "command_uniform_velocity: heading arguments missing");
  if (c->nworld < 1 || c->ldu < 7) return fail(-22, "command_uniform_velocity: bad sizes");
  if (c->error_vel_xy && (c->mask || !c->error_vel_yaw || !c->root_link_lin_vel_b || !c->root_link_ang_vel_b))
    return fail(-22, "command_uniform_velocity: the metrics belong to compute() and need both error arrays and both velocities");
  hipLaunchKernelGGL(k_command_uniform_velocity, dim3((c->nworld + 255) / 256), dim3(256), 0, (hipStream_t)stream, *c);
  return launched("k_command_uniform_velocity launch failed");
}

static int check_tables(const mjlab_motion_tables_t* t, const char* who) {
  if (!t || !t->joint_pos || !t->joint_vel || !t->body_pos_w || !t->body_quat_w || !t->body_lin_vel_w || !t->body_ang_vel_w || !t->body_indexes) return fail(-22, who);
  if (t->nframe < 1 || t->nj < 0 || t->nbody_m < 1 || t->nb < 1) return fail(-22, who);
  return 0;
}
int mjlab_command_motion_write(const mjlab_motion_tables_t* tab, float* qpos, int nq, int q_adr, float* qvel, int nv, int v_adr, const int* joint_q_adr,
                               const int* joint_v_adr, int nworld, const unsigned char* mask, const long long* time_steps, const float* env_origins,
                               const float* soft_joint_pos_limits, int ld_lim, const float* U, int ldu, const float* pose_range, const float* velocity_range,
                               float joint_lo, float joint_hi, void* stream) {
  int rc = check_tables(tab, "command_motion_write: bad motion tables");
  if (rc) return rc;
  if (!qpos || !qvel || !joint_q_adr || !joint_v_adr || !mask || !time_steps || !env_origins || !soft_joint_pos_limits || !U || !pose_range || !velocity_range)
    return fail(-22, "command_motion_write: null argument");
  if (nworld < 1 || q_adr < 0 || q_adr + 7 > nq || v_adr < 0 || v_adr + 6 > nv || ldu < 12 + tab->nj) return fail(-22, "command_motion_write: bad sizes");
  hipLaunchKernelGGL(k_command_motion_write, dim3((nworld + 255) / 256), dim3(256), 0, (hipStream_t)stream, *tab, qpos, nq, q_adr, qvel, nv, v_adr, joint_q_adr,
                     joint_v_adr, nworld, mask, time_steps, env_origins, soft_joint_pos_limits, ld_lim, U, ldu, pose_range, velocity_range, joint_lo, joint_hi);
  return launched("k_command_motion_write launch failed");
}
int mjlab_copy_batch(const mjlab_copy_entry_t* entries, int n, void* stream) {
  if (n < 0 || (n > 0 && !entries)) return fail(-24, "copy_batch: bad arguments");
  for (int k0 = 0; k0 < n; k0 += MJLAB_COPY_BATCH_MAX) {
    mjlab_copy_batch_t b;
    memset(&b, 0, sizeof(b));
    const int m = n - k0 < MJLAB_COPY_BATCH_MAX ? n - k0 : MJLAB_COPY_BATCH_MAX;
    unsigned long long most = 0;
    for (int k = 0; k < m; ++k) {
      if (!entries[k0 + k].dst || !entries[k0 + k].src) return fail(-24, "copy_batch: null pointer in an entry");
      b.e[k] = entries[k0 + k];
      most = b.e[k].nbytes > most ? b.e[k].nbytes : most;
    }
    unsigned long long gx = (most / 4 + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
    hipLaunchKernelGGL(k_copy_batch, dim3((unsigned)gx, (unsigned)m), dim3(256), 0, (hipStream_t)stream, b);
    int rc = launched("k_copy_batch launch failed");
    if (rc) return rc;
  }
  return 0;
}

int mjlab_sizeof_motion_sample(void) { return (int)sizeof(mjlab_motion_sample_t); }
int mjlab_command_motion_sample(const mjlab_motion_sample_t* a, void* stream) {
  if (!a) return fail(-25, "command_motion_sample: null argument");
  const void* ptrs[] = {a->mask, a->terminated, a->time_steps, a->U, a->cdf, a->entropy, a->top1_prob, a->top1_bin, a->hist_out, a->m_entropy, a->m_top1_prob, a->m_top1_bin};
  for (const void* p : ptrs) if (!p) return fail(-25, "command_motion_sample: null argument");
  if ((a->time_left != nullptr) != (a->command_counter != nullptr)) return fail(-25, "command_motion_sample: time_left and command_counter come together");
  if (a->nworld < 1 || a->ldu < 3 || a->bin_count < 1 || a->bin_count > MJLAB_MOTION_SAMPLE_MAX_BINS || a->time_step_total < 1)
    return fail(-25, "command_motion_sample: bad sizes (bin_count <= MJLAB_MOTION_SAMPLE_MAX_BINS, U needs three columns)");
  hipLaunchKernelGGL(k_command_motion_sample, dim3(1), dim3(1024), 0, (hipStream_t)stream, *a);
  return launched("k_command_motion_sample launch failed");
}

int mjlab_sizeof_motion_sampler(void) { return (int)sizeof(mjlab_motion_sampler_t); }
int mjlab_command_motion_sampler(const mjlab_motion_sampler_t* a, void* stream) {
  if (!a || !a->bin_failed_count) return fail(-26, "command_motion_sampler: null argument");
  if (a->do_update && !a->current_bin_failed) return fail(-26, "command_motion_sampler: the update needs current_bin_failed");
  if (a->do_dist && (!a->kernel || !a->cdf || !a->entropy || !a->top1_prob || !a->top1_bin || a->kernel_size < 1)) return fail(-26, "command_motion_sampler: the distribution needs kernel, cdf and the three scalars");
  if (a->bin_count < 1 || a->bin_count > MJLAB_MOTION_SAMPLE_MAX_BINS) return fail(-26, "command_motion_sampler: bin_count must be in [1, MJLAB_MOTION_SAMPLE_MAX_BINS]");
  hipLaunchKernelGGL(k_command_motion_sampler, dim3(1), dim3(256), 0, (hipStream_t)stream, *a);
  return launched("k_command_motion_sampler launch failed");
}

int mjlab_log_finish(const float* const* src, const unsigned char* div, const float* scale, int k, const float* count, int first, float* vec, void* stream) {
  if (!src || !div || !scale || !count || !vec || k < 1) return fail(-29, "log_finish: null argument or no entries");
  hipLaunchKernelGGL(k_log_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, src, div, scale, k, count, first, vec);
  return launched("k_log_finish launch failed");
}
int mjlab_flag_to_mask(const float* flag, int nworld, int* world_mask, void* stream) {
  if (!flag || !world_mask || nworld < 1) return fail(-28, "flag_to_mask: null argument or no worlds");
  hipLaunchKernelGGL(k_flag_to_mask, dim3((nworld + 255) / 256), dim3(256), 0, (hipStream_t)stream, flag, nworld, world_mask);
  return launched("k_flag_to_mask launch failed");
}

int mjlab_sizeof_motion_metrics(void) { return (int)sizeof(mjlab_motion_metrics_t); }
int mjlab_command_motion_metrics(const mjlab_motion_metrics_t* a, void* stream) {
  if (!a) return fail(-24, "command_motion_metrics: null argument");
  const void* ptrs[] = {a->body_pos_w, a->body_quat_w, a->body_lin_vel_w, a->body_ang_vel_w, a->robot_body_pos_w, a->robot_body_quat_w, a->robot_body_lin_vel_w,
                        a->robot_body_ang_vel_w, a->body_pos_relative_w, a->body_quat_relative_w, a->joint_pos, a->joint_vel, a->robot_joint_pos, a->robot_joint_vel, a->out};
  for (const void* p : ptrs) if (!p) return fail(-24, "command_motion_metrics: null argument");
  if (a->nworld < 1 || a->nb < 1 || a->nj < 1 || a->anchor_index < 0 || a->anchor_index >= a->nb || (a->nworld > 1 && (a->ld_robot_joint_pos < a->nj || a->ld_robot_joint_vel < a->nj)))
    return fail(-24, "command_motion_metrics: bad sizes");
  hipLaunchKernelGGL(k_command_motion_metrics, dim3((unsigned)a->nworld), dim3(64), 0, (hipStream_t)stream, *a);
  return launched("k_command_motion_metrics launch failed");
}

int mjlab_command_motion_frame(const mjlab_motion_tables_t* tab, int nworld, const long long* time_steps, const float* env_origins, const float* body_link_pose_w,
                               const float* body_link_vel_w, int nbody_e, const int* track_ids, float* joint_pos, float* joint_vel, float* body_pos_w,
                               float* body_quat_w, float* body_lin_vel_w, float* body_ang_vel_w, float* robot_body_pos_w, float* robot_body_quat_w,
                               float* robot_body_lin_vel_w, float* robot_body_ang_vel_w, void* stream) {
  int rc = check_tables(tab, "command_motion_frame: bad motion tables");
  if (rc) return rc;
  if (!time_steps || !env_origins || !joint_pos || !joint_vel || !body_pos_w || !body_quat_w || !body_lin_vel_w || !body_ang_vel_w)
    return fail(-23, "command_motion_frame: null argument");
  if (body_link_pose_w && (!body_link_vel_w || !track_ids || nbody_e < 1 || !robot_body_pos_w || !robot_body_quat_w || !robot_body_lin_vel_w || !robot_body_ang_vel_w))
    return fail(-23, "command_motion_frame: the robot's part needs pose, velocity, the tracked ids and four outputs");
  if (nworld < 1) return fail(-23, "command_motion_frame: bad sizes");
  const long long total = (long long)nworld * tab->nb;
  hipLaunchKernelGGL(k_command_motion_frame, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *tab, nworld, time_steps, env_origins,
                     body_link_pose_w, body_link_vel_w, nbody_e, track_ids, joint_pos, joint_vel, body_pos_w, body_quat_w, body_lin_vel_w, body_ang_vel_w,
                     robot_body_pos_w, robot_body_quat_w, robot_body_lin_vel_w, robot_body_ang_vel_w);
  return launched("k_command_motion_frame launch failed");
}

int mjlab_command_motion_relative(const mjlab_motion_tables_t* tab, int nworld, const long long* time_steps, const float* env_origins, const float* xpos,
                                  const float* xquat, int nbody, int anchor_body_id, int anchor_index, float* body_pos_relative_w, float* body_quat_relative_w,
                                  int exact, void* stream) {
  int rc = check_tables(tab, "command_motion_relative: bad motion tables");
  if (rc) return rc;
  if (!time_steps || !env_origins || !xpos || !xquat || !body_pos_relative_w || !body_quat_relative_w) return fail(-22, "command_motion_relative: null argument");
  if (nworld < 1 || anchor_body_id < 0 || anchor_body_id >= nbody || anchor_index < 0 || anchor_index >= tab->nb) return fail(-22, "command_motion_relative: bad sizes");
  const long long total = (long long)nworld * tab->nb;
  hipLaunchKernelGGL(k_command_motion_relative, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *tab, nworld, time_steps, env_origins, xpos,
                     xquat, nbody, anchor_body_id, anchor_index, body_pos_relative_w, body_quat_relative_w, exact);
  return launched("k_command_motion_relative launch failed");
}
int mjlab_sizeof_motion_tables(void) { return (int)sizeof(mjlab_motion_tables_t); }
int mjlab_masked_fill_rows(const mjlab_fill_entry_t* entries, int nentries, const unsigned char* mask, int nworld, void* stream) {
  if (!entries || !mask) return fail(-22, "masked_fill_rows: null argument");
  if (nentries < 1 || nworld < 1) return fail(-22, "masked_fill_rows: bad sizes");
  hipLaunchKernelGGL(k_masked_fill_rows, dim3(nworld), dim3(64), 0, (hipStream_t)stream, entries, nentries, mask, nworld);
  return launched("k_masked_fill_rows launch failed");
}
int mjlab_masked_sums(const mjlab_sum_entry_t* entries, int k, const unsigned char* mask, int nworld, float* out, void* stream) {
  if (!mask || !out || (k > 0 && !entries)) return fail(-22, "masked_sums: null argument");
  if (k < 0 || nworld < 1) return fail(-22, "masked_sums: bad sizes");
  hipLaunchKernelGGL(k_masked_sums, dim3(k + 1), dim3(256), 0, (hipStream_t)stream, entries, k, mask, nworld, out);
  return launched("k_masked_sums launch failed");
}
int mjlab_reward_accumulate(const float* values, const float* weights, const int* columns, int k, int nworld, float dt, float* reward,
                            float* const* episode_sums, float* step_reward, int nterm, void* stream) {
  if (!values || !weights || !columns || !reward || !episode_sums || !step_reward) return fail(-22, "reward_accumulate: null argument");
  if (k < 1 || nworld < 1 || nterm < k) return fail(-22, "reward_accumulate: bad sizes");
  hipLaunchKernelGGL(k_reward_accumulate, dim3((nworld + 255) / 256), dim3(256), 0, (hipStream_t)stream, values, weights, columns, k, nworld, dt, reward,
                     episode_sums, step_reward, nterm);
  return launched("k_reward_accumulate launch failed");
}

int mjlab_control_step(const mjlab_model_t* m, const mjlab_data_t* d, const mjlab_control_t* c, void* stream) {
  int rc = check_model(m);
  if (rc) return rc;
  if (!c || c->nsubstep < 0) return fail(-20, "control_step: bad argument");
  if (m->opt.solver == MJLAB_SOL_PGS) return fail(-20, "control_step: the control kernel carries the primal solvers only (MJLAB_SOL_PGS: separate calls)");
  if (c->action && (!c->action_offset || !c->action_scale)) return fail(-20, "control_step: action without offset / scale");
  if (c->key_qpos && (!c->rnd3 || !c->episode_length || !c->reset_mask)) return fail(-15, "control_step: reset arguments missing");
  if ((c->reset_qpos != nullptr) != (c->reset_qvel != nullptr)) return fail(-15, "control_step: reset_qpos and reset_qvel come together");
  if (c->motion && (m->size.nq < 7 || m->size.nv < 6)) return fail(-15, "control_step: motion resets need a floating base");
  if ((c->reset_qpos || c->term_ref || c->motion) && !c->key_qpos) return fail(-15, "control_step: reset_qpos / term_ref need the reset phase (key_qpos)");
  if (c->push_time_left && (!c->rnd7 || m->size.nq < 7 || m->size.nv < 6)) return fail(-17, "control_step: push needs rnd7 and a free root joint");
  if (c->forward_mode < 0 || c->forward_mode > 2) return fail(-20, "control_step: forward_mode must be 0, 1 or 2");
  hipStream_t st = (hipStream_t)stream;
  const int a = presolve_lds_floats(m->size), b = solve_stage_lds_floats(m), lds = a > b ? a : b;
  const int fold = (m->opt.flags & MJLAB_OPT_FOLD_FORWARD) ? 1 : 0;
  const NvpLaunch* L = nvp_launch(solve_nvp(m->size.nv));
  if (!L) return fail(-3, "nv must be in [1, 64]");
  hipError_t e = (m->opt.cone == MJLAB_CONE_ELLIPTIC ? L->control_cone : L->control)(m, d, c, fold, 4 * lds, st);
  if (e != hipSuccess) return fail((int)e, "k_control_step launch failed");
  return 0;
}

int mjlab_chol_selftest(int n, int nbatch, const float* A, const float* b, float* x, void* stream) {
  if (!A || !b || !x || nbatch < 1) return fail(-27, "chol_selftest: null argument or empty batch");
  const NvpLaunch* L = nvp_launch(solve_nvp(n));
  if (n < 1 || !L) return fail(-3, "nv must be in [1, 64]");
  hipError_t e = L->chol_test(A, b, x, n, nbatch, (hipStream_t)stream);
  if (e != hipSuccess) return fail((int)e, "k_chol_selftest launch failed");
  return 0;
}

int mjlab_selftest(void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int nblk = 16, n = nblk * 64;
  float h[16 * 64];
  unsigned x = 12345u;
  for (int i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (float)(int)(x >> 8) / 8388608.f - 1.f; }
  float* din = nullptr;
  int* derr = nullptr;
  int herr = -1;
  if (hipMalloc(&din, sizeof(h)) != hipSuccess || hipMalloc(&derr, sizeof(int)) != hipSuccess) return fail(-11, "selftest: hipMalloc");
  (void)hipMemcpyAsync(din, h, sizeof(h), hipMemcpyHostToDevice, st);
  (void)hipMemsetAsync(derr, 0, sizeof(int), st);
  hipLaunchKernelGGL(k_selftest, dim3(nblk), dim3(64), 0, st, din, derr);
  (void)hipMemcpyAsync(&herr, derr, sizeof(int), hipMemcpyDeviceToHost, st);
  hipError_t e = hipStreamSynchronize(st);
  (void)hipFree(din);
  (void)hipFree(derr);
  if (e != hipSuccess) return fail((int)e, "selftest failed to run");
  if (herr != 0) return fail(-12, "selftest: DPP wave reductions disagree with the shuffle reference");
  return 0;
}

int mjlab_poison_scratch(int nblocks, void* stream) {
  static unsigned* sink = nullptr;
  if (!sink && hipMalloc(&sink, sizeof(unsigned)) != hipSuccess) return fail(-13, "poison_scratch: hipMalloc");
  if (nblocks <= 0) return fail(-14, "poison_scratch: nblocks must be positive");
  // ~20 us of residency per wave (s_memtime ticks at 100 MHz): the whole launch is on the device at once up to 8192 waves
  hipLaunchKernelGGL(k_poison_scratch, dim3(nblocks), dim3(64), 0, (hipStream_t)stream, 0x7fc1a000u, 3, 2000LL, sink);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "k_poison_scratch launch failed");
  return 0;
}

int mjlab_tile_field(void* dst, const void* src, long long nelem, int nworld, int elem_size, void* stream) {
  if (nelem <= 0 || nworld <= 0) return fail(-9, "tile: empty field");
  const long long total = nelem * nworld;
  const int block = 256;
  long long grid = (total + block - 1) / block;
  if (grid > 2048) grid = 2048;
  hipStream_t st = (hipStream_t)stream;
  if (elem_size == 4) hipLaunchKernelGGL(k_tile<unsigned>, dim3((unsigned)grid), dim3(block), 0, st, (unsigned*)dst, (const unsigned*)src, nelem, total);
  else if (elem_size == 8) hipLaunchKernelGGL(k_tile<unsigned long long>, dim3((unsigned)grid), dim3(block), 0, st, (unsigned long long*)dst, (const unsigned long long*)src, nelem, total);
  else return fail(-10, "tile: elem_size must be 4 or 8");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "k_tile launch failed");
  return 0;
}

}  // extern "C"
