// env_terms.h -- MASK-BASED forms of the reference's event and command terms that touch mjData (include/mjlab_amd.h,
// "environment terms").  The reference runs each of them on a variable-length id list (`reset_buf.nonzero()`,
// `(time_left <= 0).nonzero()`), 20-90 small torch kernels per term and call; here one launch per term, one thread per
// world (per world and joint for the joint reset), every world evaluated and the mask deciding which rows are written:
// no host round trip, so the launches sit inside the hipGraph of the whole control step (mjlab_amd/graphed_env.py).
// The arithmetic follows the reference's helper functions operation by operation (no contraction into fma), so that a
// world gets the values the reference's chain of torch kernels would give it for the same uniforms.
#pragma once
#ifdef MJLAB_MAIN_TU

namespace env_terms {

struct Quat { float w, x, y, z; };

// third_party/isaaclab/isaaclab/utils/math.py:269-295
__device__ __forceinline__ Quat quat_from_euler_xyz(const float roll, const float pitch, const float yaw) {
#pragma clang fp contract(off)
  const float cy = cosf(yaw * 0.5f), sy = sinf(yaw * 0.5f);
  const float cr = cosf(roll * 0.5f), sr = sinf(roll * 0.5f);
  const float cp = cosf(pitch * 0.5f), sp = sinf(pitch * 0.5f);
  Quat q;
  q.w = cy * cr * cp + sy * sr * sp;
  q.x = cy * sr * cp - sy * cr * sp;
  q.y = cy * cr * sp + sy * sr * cp;
  q.z = sy * cr * cp - cy * sr * sp;
  return q;
}

// math.py:521-555 (the 8-multiplication form, in its order)
__device__ __forceinline__ Quat quat_mul(const Quat a, const Quat b) {
#pragma clang fp contract(off)
  const float ww = (a.z + a.x) * (b.x + b.y);
  const float yy = (a.w - a.y) * (b.w + b.z);
  const float zz = (a.w + a.y) * (b.w - b.z);
  const float xx = ww + yy + zz;
  const float qq = 0.5f * (xx + (a.z - a.x) * (b.x - b.y));
  Quat q;
  q.w = qq - ww + (a.z - a.y) * (b.y - b.z);
  q.x = qq - xx + (a.x + a.w) * (b.x + b.w);
  q.y = qq - yy + (a.w - a.x) * (b.y + b.z);
  q.z = qq - zz + (a.z + a.y) * (b.w - b.x);
  return q;
}

__device__ __forceinline__ void cross3(float* o, const float* a, const float* b) {
#pragma clang fp contract(off)
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// math.py:645-664: vec - w * t + xyz x t, t = 2 (xyz x vec)
__device__ __forceinline__ void quat_apply_inverse(float* o, const Quat q, const float* v) {
#pragma clang fp contract(off)
  const float xyz[3] = {q.x, q.y, q.z};
  float t[3], c[3];
  cross3(t, xyz, v);
  for (int k = 0; k < 3; ++k) t[k] = t[k] * 2.f;
  cross3(c, xyz, t);
  for (int k = 0; k < 3; ++k) o[k] = (v[k] - q.w * t[k]) + c[k];
}

// math.py:96-117 (torch.remainder: the result takes the sign of the divisor)
__device__ __forceinline__ float wrap_to_pi(const float a) {
#pragma clang fp contract(off)
  const float pi = 3.14159265358979323846f, two_pi = 6.28318530717958647692f;
  float m = fmodf(a + pi, two_pi);
  if (m != 0.f && m < 0.f) m = m + two_pi;
  return (m == 0.f && a > 0.f) ? pi : m - pi;
}

__device__ __forceinline__ float uniform(const float u, const float lo, const float hi) {  // math.py:1354-1373
#pragma clang fp contract(off)
  return u * (hi - lo) + lo;
}

}  // namespace env_terms

// envs/mdp/events.py:42-91 reset_root_state_uniform
__global__ __launch_bounds__(256) void k_event_reset_root_state_uniform(float* qpos, const int nq, const int q_adr, float* qvel, const int nv,
                                                                        const int v_adr, const int nworld, const unsigned char* mask,
                                                                        const float* root, const int ld_root, const float* org, const float* U,
                                                                        const int ldu, const float* pose, const float* vel) {
#pragma clang fp contract(off)
  using namespace env_terms;
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= nworld || !mask[w]) return;
  const float* u = U + (size_t)w * ldu;
  const float* r = root + (size_t)w * ld_root;
  float rs[6], vs[6];
  for (int k = 0; k < 6; ++k) {
    rs[k] = uniform(u[k], pose[k], pose[6 + k]);
    vs[k] = r[7 + k] + uniform(u[6 + k], vel[k], vel[6 + k]);
  }
  float* qp = qpos + (size_t)w * nq + q_adr;
  float* qv = qvel + (size_t)w * nv + v_adr;
  for (int k = 0; k < 3; ++k) qp[k] = (r[k] + rs[k]) + org[3 * w + k];
  const Quat q = quat_mul(Quat{r[3], r[4], r[5], r[6]}, quat_from_euler_xyz(rs[3], rs[4], rs[5]));
  qp[3] = q.w, qp[4] = q.x, qp[5] = q.y, qp[6] = q.z;
  float ang[3];
  quat_apply_inverse(ang, q, vs + 3);  // (entity.write_root_link_velocity_to_sim stores the body-frame angular velocity)
  for (int k = 0; k < 3; ++k) qv[k] = vs[k], qv[3 + k] = ang[k];
}

// envs/mdp/events.py:94-124 reset_joints_by_scale; one thread per (world, selected joint)
__global__ __launch_bounds__(256) void k_event_reset_joints_by_scale(float* qpos, const int nq, float* qvel, const int nv, const int nworld,
                                                                     const unsigned char* mask, const int nj, const int* joint_ids,
                                                                     const int* q_adr, const int* v_adr, const float* jpos, const int ld_jpos,
                                                                     const float* jvel, const int ld_jvel, const float* lim, const int ld_lim,
                                                                     const float* U, const int ldu, const float* ranges) {
#pragma clang fp contract(off)
  using namespace env_terms;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int w = i / nj, j = i - w * nj;
  if (w >= nworld || !mask[w]) return;
  const int id = joint_ids ? joint_ids[j] : j;
  const float* u = U + (size_t)w * ldu;
  float p = jpos[(size_t)w * ld_jpos + id] * uniform(u[j], ranges[0], ranges[1]);
  const float v = jvel[(size_t)w * ld_jvel + id] * uniform(u[nj + j], ranges[2], ranges[3]);
  const float* l = lim + (size_t)w * ld_lim + 2 * id;
  p = fminf(fmaxf(p, l[0]), l[1]);
  qpos[(size_t)w * nq + q_adr[j]] = p;
  qvel[(size_t)w * nv + v_adr[j]] = v;
}

// EventManager.apply(mode="interval") (managers/event_manager.py:116-138) + envs/mdp/events.py:127-143 push_by_setting_velocity
__global__ __launch_bounds__(256) void k_event_push_by_setting_velocity(float* qvel, const int nv, const int v_adr, const int nworld, float* time_left,
                                                                        const float dt, const float* interval, const float* vel_w, const int ld_vel,
                                                                        const float* quat_w, const int ld_quat, const float* U, const int ldu,
                                                                        const float* range) {
#pragma clang fp contract(off)
  using namespace env_terms;
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= nworld) return;
  const float* u = U + (size_t)w * ldu;
  float t = time_left[w] - dt;
  if (t < 1e-6f) {
    t = uniform(u[6], interval[0], interval[1]);
    const float* vw = vel_w + (size_t)w * ld_vel;
    const float* qw = quat_w + (size_t)w * ld_quat;
    float v[6], ang[3];
    for (int k = 0; k < 6; ++k) v[k] = vw[k] + uniform(u[k], range[k], range[6 + k]);
    quat_apply_inverse(ang, Quat{qw[0], qw[1], qw[2], qw[3]}, v + 3);
    float* qv = qvel + (size_t)w * nv + v_adr;
    for (int k = 0; k < 3; ++k) qv[k] = v[k], qv[3 + k] = ang[k];
  }
  time_left[w] = t;
}

// CommandTerm.reset / compute / _resample (managers/command_manager.py:44-66) around UniformVelocityCommand's _resample_command and
// _update_command (tasks/velocity/mdp/velocity_command.py:64-102, without the init-velocity branch)
__global__ __launch_bounds__(256) void k_command_uniform_velocity(const mjlab_velocity_command_t c) {
#pragma clang fp contract(off)
  using namespace env_terms;
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= c.nworld) return;
  float t = c.time_left[w];
  bool m;
  if (c.mask) m = c.mask[w] != 0;
  else {
    t = t - c.dt;
    m = t <= 0.f;
  }
  float* v = c.vel_command_b + 3 * (size_t)w;
  if (!c.mask && c.error_vel_xy) {  // _update_metrics (:50-62) comes first in CommandTerm.compute: on the command as it stands
    const float* lv = c.root_link_lin_vel_b + (size_t)w * c.ld_lin_vel;
    const float dx = v[0] - lv[0], dy = v[1] - lv[1];
    c.error_vel_xy[w] = c.error_vel_xy[w] + sqrtf(dx * dx + dy * dy) * c.inv_max_command_step;
    c.error_vel_yaw[w] = c.error_vel_yaw[w] + fabsf(v[2] - c.root_link_ang_vel_b[(size_t)w * c.ld_ang_vel + 2]) * c.inv_max_command_step;
  }
  if (m) {
    const float* u = c.U + (size_t)w * c.ldu;
    const float* rg = c.ranges;  // rows lin_vel_x, lin_vel_y, ang_vel_z, heading: [lo, hi]
    t = uniform(u[0], c.resampling_lo, c.resampling_hi);
    v[0] = uniform(u[1], rg[0], rg[1]);
    v[1] = uniform(u[2], rg[2], rg[3]);
    v[2] = uniform(u[3], rg[4], rg[5]);
    if (c.heading_command) {
      c.heading_target[w] = uniform(u[4], rg[6], rg[7]);
      c.is_heading_env[w] = u[5] <= c.rel_heading_envs;
    }
    c.is_standing_env[w] = u[6] <= c.rel_standing_envs;
    c.command_counter[w] += 1;
  }
  c.time_left[w] = t;
  if (!c.mask) {  // compute(): _update_command on every world
    if (c.heading_command && c.is_heading_env[w]) {
      const float err = wrap_to_pi(c.heading_target[w] - c.heading_w[(size_t)w * c.ld_heading]);
      v[2] = fminf(fmaxf(c.heading_control_stiffness * err, c.ranges[4]), c.ranges[5]);
    }
    if (c.is_standing_env[w]) v[0] = v[1] = v[2] = 0.f;
  }
}

// math.py:623-642: vec + w * t + xyz x t, t = 2 (xyz x vec)
namespace env_terms {
__device__ __forceinline__ void quat_apply(float* o, const Quat q, const float* v) {
#pragma clang fp contract(off)
  const float xyz[3] = {q.x, q.y, q.z};
  float t[3], c[3];
  cross3(t, xyz, v);
  for (int k = 0; k < 3; ++k) t[k] = t[k] * 2.f;
  cross3(c, xyz, t);
  for (int k = 0; k < 3; ++k) o[k] = (v[k] + q.w * t[k]) + c[k];
}
}  // namespace env_terms

// MotionCommand._resample_command after the phase has been drawn (tasks/tracking/mdp/commands.py:305-363): the motion frame of the
// world's time step plus the cfg's noise, written to the floating base and the joints.  U row: [.. 3 unused by this kernel (time_left,
// bin, within-bin) .., 6 pose, 6 velocity, nj joint draws] -- the caller passes the row pointer at the pose draws.
__global__ __launch_bounds__(256) void k_command_motion_write(const mjlab_motion_tables_t tab, float* qpos, const int nq, const int q_adr, float* qvel,
                                                              const int nv, const int v_adr, const int* joint_q_adr, const int* joint_v_adr,
                                                              const int nworld, const unsigned char* mask, const long long* time_steps,
                                                              const float* org, const float* lim, const int ld_lim, const float* U, const int ldu,
                                                              const float* pose, const float* vel, const float joint_lo, const float joint_hi) {
#pragma clang fp contract(off)
  using namespace env_terms;
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= nworld || !mask[w]) return;
  const long long t = time_steps[w];
  const size_t fb = (size_t)t * tab.nbody_m + tab.body_indexes[0];  // the first tracked body = the floating base
  const float* u = U + (size_t)w * ldu;
  float rs[6], vs[6];
  for (int k = 0; k < 6; ++k) {
    rs[k] = uniform(u[k], pose[k], pose[6 + k]);
    vs[k] = uniform(u[6 + k], vel[k], vel[6 + k]);
  }
  float* qp = qpos + (size_t)w * nq;
  float* qv = qvel + (size_t)w * nv;
  for (int k = 0; k < 3; ++k) qp[q_adr + k] = (tab.body_pos_w[3 * fb + k] + org[3 * w + k]) + rs[k];
  const float* bq = tab.body_quat_w + 4 * fb;
  const Quat q = quat_mul(quat_from_euler_xyz(rs[3], rs[4], rs[5]), Quat{bq[0], bq[1], bq[2], bq[3]});
  qp[q_adr + 3] = q.w, qp[q_adr + 4] = q.x, qp[q_adr + 5] = q.y, qp[q_adr + 6] = q.z;
  float ang_w[3], ang[3];
  for (int k = 0; k < 3; ++k) {
    qv[v_adr + k] = tab.body_lin_vel_w[3 * fb + k] + vs[k];
    ang_w[k] = tab.body_ang_vel_w[3 * fb + k] + vs[3 + k];
  }
  quat_apply_inverse(ang, q, ang_w);
  for (int k = 0; k < 3; ++k) qv[v_adr + 3 + k] = ang[k];
  const float* l = lim + (size_t)w * ld_lim;
  for (int j = 0; j < tab.nj; ++j) {
    float p = tab.joint_pos[(size_t)t * tab.nj + j] + uniform(u[12 + j], joint_lo, joint_hi);
    p = fminf(fmaxf(p, l[2 * j]), l[2 * j + 1]);
    qp[joint_q_adr[j]] = p;
    qv[joint_v_adr[j]] = tab.joint_vel[(size_t)t * tab.nj + j];
  }
}

// MotionCommand._update_command's relative body poses (commands.py:371-392): the motion's bodies moved to the robot's anchor in x, y and
// yaw.  One thread per (world, tracked body).
// The reference's math helpers are torch.jit.script functions: after the profiling executor's first calls they run as NNC-fused kernels that
// hiprtc compiles with its default -ffp-contract=fast, and THOSE are the reference's values in any run longer than two steps.  Which
// products the compiler contracts into fma is fixed by the expressions; tools/experiments/rel_probe2.py found the sites by enumeration
// against the reference's own functions on the MI355X (PyTorch 2.10 / ROCm 7.2: 0 of 229 376 elements differ per helper):
//   quat_inv   sum of squares pairwise (a^2 + b^2) + (c^2 + d^2) (the reduction kernel's order), division unfused
//   quat_mul   xx = fma(z1 + x1, x2 + y2, yy) + zz;  w, x: the last product contracted into the sum;  qq, y, z: no contraction (2-D operands;
//              with the 3-D operands of _update_command also xx's last product and z's: tools/experiments/rel_probe_env.py)
//   yaw_quat   atan2(2 fma(qw, qz, qx qy), 1 - 2 fma(qz, qz, qy qy))
//   quat_apply cross products as fma(a, b, -(c d)) (at::cross, in either mode), vec + w t contracted, the second cross added plainly
namespace env_terms {
// style 1: the kernel NNC builds for 2-D (N, 4) operands; style 2: for 3-D (n, nb, 4) operands (what _update_command passes): the generated
// source differs, and with it which products the compiler contracts -- xx's last product and z's
__device__ __forceinline__ Quat quat_mul_nnc(const Quat a, const Quat b, const int style) {
#pragma clang fp contract(off)
  const float ww = (a.z + a.x) * (b.x + b.y), yy = (a.w - a.y) * (b.w + b.z), zz = (a.w + a.y) * (b.w - b.z);
  const float in = __builtin_fmaf(a.z + a.x, b.x + b.y, yy);
  const float xx = style == 2 ? __builtin_fmaf(a.w + a.y, b.w - b.z, in) : in + zz;
  const float qq = 0.5f * (xx + (a.z - a.x) * (b.x - b.y));
  Quat q;
  q.w = __builtin_fmaf(a.z - a.y, b.y - b.z, qq - ww);
  q.x = __builtin_fmaf(a.x + a.w, b.x + b.w, qq - xx);
  q.y = (qq - yy) + (a.w - a.x) * (b.y + b.z);
  q.z = style == 2 ? __builtin_fmaf(a.z + a.y, b.w - b.x, qq - zz) : (qq - zz) + (a.z + a.y) * (b.w - b.x);
  return q;
}
__device__ __forceinline__ void cross3_nnc(float* o, const float* a, const float* b) {
#pragma clang fp contract(off)
  o[0] = __builtin_fmaf(a[1], b[2], -(a[2] * b[1]));
  o[1] = __builtin_fmaf(a[2], b[0], -(a[0] * b[2]));
  o[2] = __builtin_fmaf(a[0], b[1], -(a[1] * b[0]));
}
__device__ __forceinline__ void quat_apply_nnc(float* o, const Quat q, const float* v, const bool fused) {
#pragma clang fp contract(off)
  const float xyz[3] = {q.x, q.y, q.z};
  float t[3], c[3];
  cross3_nnc(t, xyz, v);
  for (int k = 0; k < 3; ++k) t[k] = t[k] * 2.f;
  cross3_nnc(c, xyz, t);
  for (int k = 0; k < 3; ++k) o[k] = (fused ? __builtin_fmaf(q.w, t[k], v[k]) : v[k] + q.w * t[k]) + c[k];
}
}  // namespace env_terms
// _update_command's relative body poses (tasks/tracking/mdp/commands.py:370-392).  Whether a helper runs fused at a call site depends on the
// jit's profiling history in the process (a call whose input types miss the profiled specialisations falls back to the unfused graph), so the
// sites are switchable per helper and the caller calibrates against the reference's own functions (GraphedRlEnv._calibrate_relative).
// `exact`: 8 = the reference-faithful base (quat_inv's pairwise sum, at::cross's fma form), + 2 yaw_quat fused, + 4 quat_apply's sum fused,
// + 16 * s1 + 64 * s2 with s1 / s2 = the first / second quat_mul's form (0 unfused, 1 / 2 = the fused kernel for 2-D / 3-D operands); 0 = every operation rounded separately and a sequential sum of squares (round 5's kernel: 1 ulp from any of them).
__global__ __launch_bounds__(256) void k_command_motion_relative(const mjlab_motion_tables_t tab, const int nworld, const long long* time_steps,
                                                                 const float* org, const float* xpos, const float* xquat, const int nbody,
                                                                 const int anchor_body_id, const int anchor_index, float* out_pos, float* out_quat, const int exact) {
#pragma clang fp contract(off)
  using namespace env_terms;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int w = i / tab.nb, b = i - w * tab.nb;
  if (w >= nworld) return;
  const long long t = time_steps[w];
  const size_t fa = (size_t)t * tab.nbody_m + tab.body_indexes[anchor_index], fb = (size_t)t * tab.nbody_m + tab.body_indexes[b];
  float apos[3], bpos[3];
  for (int k = 0; k < 3; ++k) {
    apos[k] = tab.body_pos_w[3 * fa + k] + org[3 * w + k];
    bpos[k] = tab.body_pos_w[3 * fb + k] + org[3 * w + k];
  }
  const float* aq = tab.body_quat_w + 4 * fa;
  const float* rp = xpos + ((size_t)w * nbody + anchor_body_id) * 3;
  const float* rq = xquat + ((size_t)w * nbody + anchor_body_id) * 4;
  // quat_inv (math.py:255-266): conjugate / clamp(sum of squares, 1e-9)
  const bool f_yaw = exact & 2, f_app = exact & 4, base = exact & 8;
  const int s_mul1 = (exact >> 4) & 3, s_mul2 = (exact >> 6) & 3;
  const float ss = base ? (aq[0] * aq[0] + aq[1] * aq[1]) + (aq[2] * aq[2] + aq[3] * aq[3]) : ((aq[0] * aq[0] + aq[1] * aq[1]) + aq[2] * aq[2]) + aq[3] * aq[3];
  const float n2 = fmaxf(ss, 1e-9f);
  const Quat inv{aq[0] / n2, -aq[1] / n2, -aq[2] / n2, -aq[3] / n2};
  const Quat rqq{rq[0], rq[1], rq[2], rq[3]};
  const Quat d = s_mul1 ? quat_mul_nnc(rqq, inv, s_mul1) : quat_mul(rqq, inv);
  // yaw_quat (math.py:560-582)
  const float y1 = f_yaw ? 2.f * __builtin_fmaf(d.w, d.z, d.x * d.y) : 2.f * (d.w * d.z + d.x * d.y);
  const float y2 = f_yaw ? 1.f - 2.f * __builtin_fmaf(d.z, d.z, d.y * d.y) : 1.f - 2.f * (d.y * d.y + d.z * d.z);
  const float yaw = atan2f(y1, y2);
  float cw = cosf(yaw / 2.f), sz = sinf(yaw / 2.f);
  const float nrm = fmaxf(sqrtf(((cw * cw + 0.f) + 0.f) + sz * sz), 1e-9f);
  const Quat dq{cw / nrm, 0.f / nrm, 0.f / nrm, sz / nrm};
  const float* bq = tab.body_quat_w + 4 * fb;
  const Quat bqq{bq[0], bq[1], bq[2], bq[3]};
  const Quat oq = s_mul2 ? quat_mul_nnc(dq, bqq, s_mul2) : quat_mul(dq, bqq);
  float* oQ = out_quat + ((size_t)w * tab.nb + b) * 4;
  oQ[0] = oq.w, oQ[1] = oq.x, oQ[2] = oq.y, oQ[3] = oq.z;
  float rel[3], rot[3];
  for (int k = 0; k < 3; ++k) rel[k] = bpos[k] - apos[k];
  if (base) quat_apply_nnc(rot, dq, rel, f_app);
  else quat_apply(rot, dq, rel);
  float* oP = out_pos + ((size_t)w * tab.nb + b) * 3;
  oP[0] = rp[0] + rot[0], oP[1] = rp[1] + rot[1], oP[2] = apos[2] + rot[2];
}

// Up to MJLAB_COPY_BATCH_MAX device-to-device copies in ONE launch (mjlab_copy_batch): GraphedRlEnv copies every tensor the reference
// REBOUND during a step back into the tensor its graph reads (graphed_env.py::_restore_bindings: 14 small copies per step of the
// tracking task, each a graph node of its own before).  The entries travel by value in the kernel arguments, so a captured graph
// holds them.  blockIdx.y = entry; words of 4 bytes where size and both addresses allow, bytes otherwise.
__global__ __launch_bounds__(256) void k_copy_batch(const mjlab_copy_batch_t batch) {
  const mjlab_copy_entry_t e = batch.e[blockIdx.y];
  const size_t stride = (size_t)gridDim.x * 256, i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (((e.nbytes | (unsigned long long)e.dst | (unsigned long long)e.src) & 3ull) == 0ull) {
    const unsigned* s = (const unsigned*)e.src;
    unsigned* d = (unsigned*)e.dst;
    for (size_t i = i0; i < e.nbytes / 4; i += stride) d[i] = s[i];
  } else {
    const unsigned char* s = (const unsigned char*)e.src;
    unsigned char* d = (unsigned char*)e.dst;
    for (size_t i = i0; i < e.nbytes; i += stride) d[i] = s[i];
  }
}

// MotionCommand's per-step PROPERTIES (commands.py:128-215) for every world at once: the reference gathers each of them from the motion
// tables / EntityData at every access -- `motion.joint_pos[time_steps]`, `motion.body_pos_w[time_steps] + env_origins[:, None, :]`,
// `robot.data.body_link_pos_w[:, body_indexes]` ... -- an index launch (plus an add) per property and phase, ~44 per control step of the
// tracking task (profiles/r06_census).  Copies and ONE addition per element: the same bits.  One thread per (world, tracked body); the
// joints of a world are dealt out over its threads.
__global__ __launch_bounds__(256) void k_command_motion_frame(const mjlab_motion_tables_t tab, const int nworld, const long long* time_steps, const float* org,
                                                              const float* link_pose, const float* link_vel, const int nbody_e, const int* track_ids,
                                                              float* joint_pos, float* joint_vel, float* body_pos_w, float* body_quat_w,
                                                              float* body_lin_vel_w, float* body_ang_vel_w, float* robot_pos, float* robot_quat,
                                                              float* robot_lin_vel, float* robot_ang_vel) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int w = i / tab.nb, b = i - w * tab.nb;
  if (w >= nworld) return;
  const long long t = time_steps[w];
  const size_t fb = (size_t)t * tab.nbody_m + tab.body_indexes[b], o = (size_t)w * tab.nb + b;
  for (int k = 0; k < 3; ++k) {
    body_pos_w[3 * o + k] = tab.body_pos_w[3 * fb + k] + org[3 * w + k];
    body_lin_vel_w[3 * o + k] = tab.body_lin_vel_w[3 * fb + k];
    body_ang_vel_w[3 * o + k] = tab.body_ang_vel_w[3 * fb + k];
  }
  for (int k = 0; k < 4; ++k) body_quat_w[4 * o + k] = tab.body_quat_w[4 * fb + k];
  for (int j = b; j < tab.nj; j += tab.nb) {
    joint_pos[(size_t)w * tab.nj + j] = tab.joint_pos[(size_t)t * tab.nj + j];
    joint_vel[(size_t)w * tab.nj + j] = tab.joint_vel[(size_t)t * tab.nj + j];
  }
  if (link_pose) {  // the robot's tracked bodies out of EntityData's body_link_pose_w (.., 7) / body_link_vel_w (.., 6: linear, angular)
    const size_t rb = (size_t)w * nbody_e + track_ids[b];
    for (int k = 0; k < 3; ++k) {
      robot_pos[3 * o + k] = link_pose[7 * rb + k];
      robot_lin_vel[3 * o + k] = link_vel[6 * rb + k];
      robot_ang_vel[3 * o + k] = link_vel[6 * rb + 3 + k];
    }
    for (int k = 0; k < 4; ++k) robot_quat[4 * o + k] = link_pose[7 * rb + 3 + k];
  }
}

// MotionCommand._update_metrics (tasks/tracking/mdp/commands.py:221-254): the ten tracking errors the command term logs at a reset --
// four of the anchor body, four averaged over the tracked bodies, two over the joints -- of every world in ONE launch; the reference
// forms each from a subtraction, a norm (or quat_error_magnitude: math.py:682-693 = quat_box_minus :584-598 = axis_angle_from_quat :472-500
// of q1 * conj(q2)) and a mean, ~130 small launches per control step.  One wave per world: lane b = tracked body b, lanes = joints for
// the joint errors.  The values feed extras["log"] only (CommandTerm.reset: the mean over the environments that reset); the sums over
// bodies / joints are formed in another order than torch's reductions, so they agree with the reference's to a few ulp, not bit for bit.
__device__ __forceinline__ float wave_sum64(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float norm3_diff(const float* a, const float* b) {
#pragma clang fp contract(off)
  const float x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
  return sqrtf(x * x + y * y + z * z);
}
__device__ __forceinline__ float quat_error_magnitude(const float* q1, const float* q2) {
#pragma clang fp contract(off)
  using namespace env_terms;
  Quat d = quat_mul(Quat{q1[0], q1[1], q1[2], q1[3]}, Quat{q2[0], -q2[1], -q2[2], -q2[3]});
  const float sgn = 1.0f - 2.0f * (d.w < 0.0f ? 1.0f : 0.0f);
  d.w *= sgn; d.x *= sgn; d.y *= sgn; d.z *= sgn;
  const float mag = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
  const float half = atan2f(mag, d.w), angle = 2.0f * half;
  const float k = fabsf(angle) > 1.0e-6f ? sinf(half) / angle : 0.5f - angle * angle / 48.0f;
  const float ax = d.x / k, ay = d.y / k, az = d.z / k;
  return sqrtf(ax * ax + ay * ay + az * az);
}
__global__ __launch_bounds__(64) void k_command_motion_metrics(const mjlab_motion_metrics_t a) {
#pragma clang fp contract(off)
  const int w = blockIdx.x, lane = threadIdx.x, nb = a.nb;
  float e_pos = 0.f, e_rot = 0.f, e_lin = 0.f, e_ang = 0.f, anchor[4] = {0.f, 0.f, 0.f, 0.f};
  for (int b = lane; b < nb; b += 64) {
    const size_t o = (size_t)w * nb + b;
    e_pos += norm3_diff(a.body_pos_relative_w + 3 * o, a.robot_body_pos_w + 3 * o);
    e_rot += quat_error_magnitude(a.body_quat_relative_w + 4 * o, a.robot_body_quat_w + 4 * o);
    e_lin += norm3_diff(a.body_lin_vel_w + 3 * o, a.robot_body_lin_vel_w + 3 * o);
    e_ang += norm3_diff(a.body_ang_vel_w + 3 * o, a.robot_body_ang_vel_w + 3 * o);
    if (b == a.anchor_index) {  // anchor_* = the anchor's rows of the body arrays, robot_anchor_* the same rows of the robot's
      anchor[0] = norm3_diff(a.body_pos_w + 3 * o, a.robot_body_pos_w + 3 * o);
      anchor[1] = quat_error_magnitude(a.body_quat_w + 4 * o, a.robot_body_quat_w + 4 * o);
      anchor[2] = norm3_diff(a.body_lin_vel_w + 3 * o, a.robot_body_lin_vel_w + 3 * o);
      anchor[3] = norm3_diff(a.body_ang_vel_w + 3 * o, a.robot_body_ang_vel_w + 3 * o);
    }
  }
  float jp = 0.f, jv = 0.f;
  for (int j = lane; j < a.nj; j += 64) {
    const float dp = a.joint_pos[(size_t)w * a.nj + j] - a.robot_joint_pos[(size_t)w * a.ld_robot_joint_pos + j];
    const float dv = a.joint_vel[(size_t)w * a.nj + j] - a.robot_joint_vel[(size_t)w * a.ld_robot_joint_vel + j];
    jp += dp * dp; jv += dv * dv;
  }
  const float inv_nb = 1.0f / (float)nb;
  float out[10] = {wave_sum64(anchor[0]), wave_sum64(anchor[1]), wave_sum64(anchor[2]), wave_sum64(anchor[3]), wave_sum64(e_pos) * inv_nb, wave_sum64(e_rot) * inv_nb,
                   wave_sum64(e_lin) * inv_nb, wave_sum64(e_ang) * inv_nb, sqrtf(wave_sum64(jp)), sqrtf(wave_sum64(jv))};
  if (lane < 10) {
    float v = out[0];
    for (int k = 1; k < 10; ++k) v = lane == k ? out[k] : v;
    a.out[(size_t)lane * a.nworld + w] = v;
  }
}

// MotionCommand._adaptive_sampling (tasks/tracking/mdp/commands.py:256-297), the per-world part, for the worlds of `mask` in ONE launch of
// ONE workgroup (the histogram and the two "any" tests are over all worlds; 4096 worlds are four trips of 1024 threads): the failed worlds'
// phase bins counted (:257-265), a new phase per masked world by inverse CDF (torch.multinomial's distribution without its host round trip:
// bin = searchsorted(cdf, u1), phase = long((bin + u2) / bin_count * (total - 1)), :283-289), the three sampling metrics filled when some
// world is masked (:292-297).  The distribution itself (cdf, entropy, top bin: once per control step) stays with the caller.
// Integer and comparison work plus one float expression evaluated as torch evaluates it (division by a host scalar = multiplication by
// its float32 reciprocal): the same phases as the torch restatement in mjlab_amd/graphed_env.py, bit for bit.
__global__ __launch_bounds__(1024) void k_command_motion_sample(const mjlab_motion_sample_t a) {
#pragma clang fp contract(off)
  __shared__ float hist[MJLAB_MOTION_SAMPLE_MAX_BINS];
  __shared__ int any_mask, any_failed;
  const int tid = threadIdx.x, nbin = a.bin_count;
  for (int b = tid; b < nbin; b += 1024) hist[b] = 0.f;
  if (tid == 0) { any_mask = 0; any_failed = 0; }
  __syncthreads();
  const long long total = a.time_step_total, den = total > 1 ? total : 1;
  const float inv_bins = __fdiv_rn(1.f, (float)nbin), span = (float)(total - 1);
  for (int w = tid; w < a.nworld; w += 1024) {
    const bool m = a.mask[w] != 0, failed = m && a.terminated[w] != 0;
    const long long t = a.time_steps[w];
    if (failed) {
      long long bin = (t * nbin) / den;  // (both operands non-negative: floor division)
      bin = bin < 0 ? 0 : bin > nbin - 1 ? nbin - 1 : bin;
      atomicAdd(&hist[bin], 1.f);
      any_failed = 1;
    }
    if (m) {
      any_mask = 1;
      if (a.time_left) {  // CommandTerm._resample (managers/command_manager.py:62-66): a new timer and one more resample on the counter
        a.time_left[w] = a.U[(size_t)w * a.ldu] * a.resampling_width + a.resampling_lo;
        a.command_counter[w] += 1;
      }
      const float u1 = a.U[(size_t)w * a.ldu + 1], u2 = a.U[(size_t)w * a.ldu + 2];
      int lo = 0, hi = nbin;  // searchsorted(cdf, u1), right = False: the first index with cdf[i] >= u1
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.cdf[mid] < u1) lo = mid + 1; else hi = mid; }
      const int bin = lo > nbin - 1 ? nbin - 1 : lo;
      a.time_steps[w] = (long long)((((float)bin + u2) * inv_bins) * span);
    }
  }
  __syncthreads();
  if (a.hist_always || any_failed)
    for (int b = tid; b < nbin; b += 1024) a.hist_out[b] = hist[b];
  if (a.any_failed_out && tid == 0) *a.any_failed_out = any_failed ? 1.f : 0.f;
  if (any_mask) {
    const float H = *a.entropy, pm = *a.top1_prob, tb = *a.top1_bin;
    for (int w = tid; w < a.nworld; w += 1024) { a.m_entropy[w] = H; a.m_top1_prob[w] = pm; a.m_top1_bin[w] = tb; }
  }
}

// The adaptive sampler's GLOBAL part (tasks/tracking/mdp/commands.py): `update` = the end of _update_command (:394-398), bin_failed_count <-
// alpha current + (1 - alpha) bin_failed_count and current <- 0 (elementwise float32 operations in the reference's order: the same bits);
// `dist` = the sampling distribution of _adaptive_sampling (:267-281, :291-294) from bin_failed_count -- p = bfc + ratio / bins, the non-causal
// smoothing kernel over p padded with its last value, normalisation, entropy, top bin, running sum -- into a cdf and three scalars that
// k_command_motion_sample reads.  ONE workgroup (bins are tens to hundreds); sums are formed in a fixed order of this kernel's own, so the
// distribution agrees with the torch expressions to float rounding, not bit for bit (it decides random draws and three logged numbers).
__global__ __launch_bounds__(256) void k_command_motion_sampler(const mjlab_motion_sampler_t a) {
#pragma clang fp contract(off)
  __shared__ float q[MJLAB_MOTION_SAMPLE_MAX_BINS];
  __shared__ float red[256];
  __shared__ int redi[256];
  const int tid = threadIdx.x, nbin = a.bin_count;
  if (a.do_update) {
    for (int b = tid; b < nbin; b += 256) {
      a.bin_failed_count[b] = a.alpha * a.current_bin_failed[b] + a.one_minus_alpha * a.bin_failed_count[b];
      a.current_bin_failed[b] = 0.f;
    }
    __syncthreads();
  }
  if (!a.do_dist) return;
  for (int b = tid; b < nbin; b += 256) {
    float acc = 0.f;
    for (int k = 0; k < a.kernel_size; ++k) {
      const int j = b + k < nbin ? b + k : nbin - 1;  // replicate padding on the right (a non-causal kernel)
      acc += a.kernel[k] * (a.bin_failed_count[j] + a.uniform_term);
    }
    q[b] = acc;
  }
  __syncthreads();
  float part = 0.f;
  for (int b = tid; b < nbin; b += 256) part += q[b];
  red[tid] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
  const float total = red[0];
  __syncthreads();
  float h = 0.f, pm = -1.f;
  int im = 0;
  for (int b = tid; b < nbin; b += 256) {
    const float p = q[b] / total;
    q[b] = p;
    h += p * logf(p + 1e-12f);
    if (p > pm) { pm = p; im = b; }
  }
  red[tid] = h;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
  const float H = -red[0] / logf((float)nbin);
  __syncthreads();
  red[tid] = pm; redi[tid] = im;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {  // the largest probability, the FIRST bin that has it (torch.max(dim=0))
    if (tid < s && (red[tid + s] > red[tid] || (red[tid + s] == red[tid] && redi[tid + s] < redi[tid]))) { red[tid] = red[tid + s]; redi[tid] = redi[tid + s]; }
    __syncthreads();
  }
  if (tid == 0) {
    *a.entropy = H;
    *a.top1_prob = red[0];
    *a.top1_bin = (float)redi[0] * __fdiv_rn(1.f, (float)nbin);
    float run = 0.f;
    for (int b = 0; b < nbin; ++b) { run += q[b]; a.cdf[b] = run; }
  }
}

// RewardManager.compute's accumulation (managers/reward_manager.py:77-89) for the k active terms whose raw values are the rows of
// `values` (k, n): value = raw * weight * dt; reward += value (in term order); episode_sum[term] += value; step_reward[:, column] =
// value / dt (as torch computes it: value * (1 / dt)).  Elementwise IEEE operations in the reference's order: the same bits as its 6 launches per term.
__global__ __launch_bounds__(256) void k_reward_accumulate(const float* values, const float* weights, const int* columns, const int k, const int n,
                                                           const float dt, float* reward, float* const* episode_sums, float* step_reward,
                                                           const int nterm) {
#pragma clang fp contract(off)
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= n) return;
  float r = 0.f;
  const float inv_dt = __fdiv_rn(1.f, dt);  // torch divides by a host scalar as a multiplication by its float32 reciprocal
  for (int i = 0; i < k; ++i) {
    const float v = (values[(size_t)i * n + w] * weights[i]) * dt;
    r = r + v;
    episode_sums[i][w] = episode_sums[i][w] + v;
    step_reward[(size_t)w * nterm + columns[i]] = v * inv_dt;
  }
  reward[w] = r;
}

// The managers' reset() bookkeeping (managers/{action,reward,command,event}_manager.py reset(), envs/manager_based_rl_env.py:246,
// entity/data.py:169-178 clear_state): `row_bytes` bytes of row w of every listed buffer are filled with the entry's pattern where
// mask[w] is set -- one launch for the ~20 masked fills of a step.  One 64-lane block per world.
__global__ __launch_bounds__(64) void k_masked_fill_rows(const mjlab_fill_entry_t* e, const int ne, const unsigned char* mask, const int nworld) {
  const int w = blockIdx.x;
  if (w >= nworld || !mask[w]) return;
  for (int i = 0; i < ne; ++i) {
    char* row = (char*)e[i].ptr + (size_t)w * e[i].row_stride_bytes;
    const int eb = e[i].elem_bytes, nel = e[i].row_bytes / eb;
    // from_device: the pattern is the ADDRESS of an integer scalar at least elem_bytes wide (little endian: its low bytes) -- a value that
    // changes from step to step (the event manager's step count, managers/event_manager.py:139-148) under a captured launch
    const long long pat = e[i].from_device ? (eb == 8 ? *(const long long*)e[i].pattern : eb == 4 ? (long long)*(const int*)e[i].pattern : (long long)*(const char*)e[i].pattern) : e[i].pattern;
    for (int k = threadIdx.x; k < nel; k += 64) {
      if (eb == 4) ((int*)row)[k] = (int)pat;
      else if (eb == 8) ((long long*)row)[k] = pat;
      else row[k] = (char)pat;
    }
  }
}

// The managers' reset() logging (reward_manager.py:67-71, command_manager.py:46-49, termination_manager.py:79-83): out[i] = the sum
// over the worlds of the mask of vector i (float, or bool counted as 0 / 1), out[k] = the number of worlds in the mask.  One
// workgroup of four waves per vector; fp32 accumulation in a fixed order (deterministic).
__global__ __launch_bounds__(256) void k_masked_sums(const mjlab_sum_entry_t* e, const int k, const unsigned char* mask, const int nworld, float* out) {
  // four waves per vector, every load issued whatever the mask says (a select, no branch: the 16 trips of a lane are in flight together --
  // one wave walking 64 dependent trips took 20 us at 4096 worlds); the partial sums meet in a fixed order
  __shared__ float part[4];
  const int i = blockIdx.x;
  const bool count = i == k, is_bool = !count && e[i].is_bool;
  const void* src = count ? nullptr : e[i].ptr;
  float acc = 0.f;
  for (int w = threadIdx.x; w < nworld; w += 256) {
    const float v = count ? 1.f : (is_bool ? (float)(((const unsigned char*)src)[w] != 0) : ((const float*)src)[w]);
    acc += mask[w] ? v : 0.f;
  }
  acc = wave_sum(acc);  // (DPP reduction: no LDS)
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[i] = (part[0] + part[1]) + (part[2] + part[3]);
}

// extras["log"] of a step (mjlab_amd/env_core.py LogBook.finish; reference: the managers' reset() logging under _reset_idx, which runs
// only when some environment reset): entry i <- (div[i] ? *src[i] / max(*count, 1) : *src[i]) * scale[i] where *count > 0 (or `first`),
// else the entry keeps the last reset step's number.  Elementwise IEEE operations in the torch twin's order: the same bits.
__global__ __launch_bounds__(64) void k_log_finish(const float* const* src, const unsigned char* div, const float* scale, const int k, const float* count,
                                                   const int first, float* vec) {
#pragma clang fp contract(off)
  const float cnt = *count;
  for (int i = threadIdx.x; i < k; i += 64) {
    const float raw = *src[i];
    const float v = (div[i] ? __fdiv_rn(raw, fmaxf(cnt, 1.f)) : raw) * scale[i];
    if (first || cnt > 0.f) vec[i] = v;
  }
}

// world_mask[w] = (*flag > 0) for every world: "forward() on ALL worlds iff SOME environment reset" (envs/manager_based_rl_env.py:129-132) with
// the count of reset environments as the flag (k_masked_sums' last output; summed over the ranks when the environment is sharded)
__global__ __launch_bounds__(256) void k_flag_to_mask(const float* flag, const int nworld, int* world_mask) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w < nworld) world_mask[w] = *flag > 0.f ? 1 : 0;
}

#endif  // MJLAB_MAIN_TU
