// extras.h -- fused entity read-back, masked reset, field tiling, self-test.
// Part of kernels.h (included there, in this order, by every translation unit of the library); not a
// stand-alone header.
#pragma once

// ====================================================================================
// Fused read-back of EntityData's derived quantities (extension, see include/mjlab_amd.h)
// ====================================================================================
__device__ __forceinline__ void quat_apply_dev(float* r, const float* q, const float* v, float sign) {
  // reference third_party/isaaclab/.../math.py:623-662: t = 2 xyz x v;  v +- w t + xyz x t
  float t[3], u[3];
  cross3(t, q + 1, v);
  for (int k = 0; k < 3; ++k) t[k] *= 2.f;
  cross3(u, q + 1, t);
  for (int k = 0; k < 3; ++k) r[k] = v[k] + sign * q[0] * t[k] + u[k];
}
// world-frame velocity at point `pos` from the c-frame spatial velocity (entity/data.py:20-31)
__device__ __forceinline__ void vel_from_cvel(float* out6, const float* pos, const float* sub, const float* cv) {
  float off[3] = {sub[0] - pos[0], sub[1] - pos[1], sub[2] - pos[2]}, c[3];
  cross3(c, cv, off);
  for (int k = 0; k < 3; ++k) { out6[k] = cv[3 + k] - c[k]; out6[3 + k] = cv[k]; }
}

__device__ __forceinline__ void entity_readback_world(const Model& m, const Data& d, const mjlab_entity_view_t& v, const int w, const int lane) {
  const int nb = m.size.nbody, nq = m.size.nq, nv = m.size.nv;
  const float* sub = d.subtree_com + ((size_t)w * nb + v.root_body_id) * 3;
  const float sc[3] = {sub[0], sub[1], sub[2]};
  const float* biq = MF(body_iquat);
  for (int i = lane; i < v.nbody; i += 64) {
    const int b = v.body_ids[i];
    const size_t wb = (size_t)w * nb + b, wi = (size_t)w * v.nbody + i;
    float pos[3], ipos[3], q[4], iq[4], cv[6], qc[4], o6[6];
    for (int k = 0; k < 3; ++k) { pos[k] = d.xpos[3 * wb + k]; ipos[k] = d.xipos[3 * wb + k]; }
    for (int k = 0; k < 4; ++k) { q[k] = d.xquat[4 * wb + k]; iq[k] = biq[4 * b + k]; }
    for (int k = 0; k < 6; ++k) cv[k] = d.cvel[6 * wb + k];
    if (v.body_link_pose_w) {
      for (int k = 0; k < 3; ++k) v.body_link_pose_w[7 * wi + k] = pos[k];
      for (int k = 0; k < 4; ++k) v.body_link_pose_w[7 * wi + 3 + k] = q[k];
    }
    if (v.body_link_vel_w) {
      vel_from_cvel(o6, pos, sc, cv);
      for (int k = 0; k < 6; ++k) v.body_link_vel_w[6 * wi + k] = o6[k];
    }
    if (v.body_com_pose_w) {
      mul_quat(qc, q, iq);
      for (int k = 0; k < 3; ++k) v.body_com_pose_w[7 * wi + k] = ipos[k];
      for (int k = 0; k < 4; ++k) v.body_com_pose_w[7 * wi + 3 + k] = qc[k];
    }
    if (v.body_com_vel_w) {
      vel_from_cvel(o6, ipos, sc, cv);
      for (int k = 0; k < 6; ++k) v.body_com_vel_w[6 * wi + k] = o6[k];
    }
  }
  if (v.root_derived && lane == 0) {
    const size_t wb = (size_t)w * nb + v.root_body_id;
    float pos[3], ipos[3], q[4], cv[6], lv[6], cvl[6], r[3];
    for (int k = 0; k < 3; ++k) { pos[k] = d.xpos[3 * wb + k]; ipos[k] = d.xipos[3 * wb + k]; }
    for (int k = 0; k < 4; ++k) q[k] = d.xquat[4 * wb + k];
    for (int k = 0; k < 6; ++k) cv[k] = d.cvel[6 * wb + k];
    vel_from_cvel(lv, pos, sc, cv);
    vel_from_cvel(cvl, ipos, sc, cv);
    float* o = v.root_derived + (size_t)w * 16;
    quat_apply_dev(r, q, v.gravity_vec_w, -1.f);
    for (int k = 0; k < 3; ++k) o[k] = r[k];
    quat_apply_dev(r, q, v.forward_vec_b, 1.f);
    o[3] = atan2f(r[1], r[0]);
    quat_apply_dev(r, q, lv, -1.f);
    for (int k = 0; k < 3; ++k) o[4 + k] = r[k];
    quat_apply_dev(r, q, lv + 3, -1.f);
    for (int k = 0; k < 3; ++k) o[7 + k] = r[k];
    quat_apply_dev(r, q, cvl, -1.f);
    for (int k = 0; k < 3; ++k) o[10 + k] = r[k];
    quat_apply_dev(r, q, cvl + 3, -1.f);
    for (int k = 0; k < 3; ++k) o[13 + k] = r[k];
  }
  for (int j = lane; j < v.njoint; j += 64) {
    const size_t wj = (size_t)w * v.njoint + j;
    if (v.joint_pos) v.joint_pos[wj] = d.qpos[(size_t)w * nq + v.joint_q_adr[j]];
    if (v.joint_vel) v.joint_vel[wj] = d.qvel[(size_t)w * nv + v.joint_v_adr[j]];
    if (v.joint_acc) v.joint_acc[wj] = d.qacc[(size_t)w * nv + v.joint_v_adr[j]];
  }
}
#ifdef MJLAB_MAIN_TU
__global__ __launch_bounds__(64) void k_entity_readback(const Model m, const Data d, const mjlab_entity_view_t v) {
  entity_readback_world(m, d, v, blockIdx.x, threadIdx.x);
}
#endif  // MJLAB_MAIN_TU

// ====================================================================================
// Masked termination + reset (extension, see include/mjlab_amd.h)
// ====================================================================================
__device__ __forceinline__ float nan_to_num_dev(float x) {
  if (x != x) return 0.f;
  if (x > 3.402823466e+38f) return 3.402823466e+38f;
  if (x < -3.402823466e+38f) return -3.402823466e+38f;
  return x;
}
// Returns the (wave-uniform) reset decision of world w.
__device__ __forceinline__ bool masked_reset_world(const Model& m, const Data& d, const int w, const int lane, const float* key_qpos, const float* rnd3,
                                                   int* episode_length, const int max_len, const float min_height, int* reset_mask,
                                                   const float* env_origins, const float min_up_z, const float* reset_qpos = nullptr,
                                                   const float* reset_qvel = nullptr, const float* term_ref = nullptr, const float term_dz = 0.f,
                                                   const float term_dup = 0.f, const mjlab_motion_reset_t* mo = nullptr) {
  const int nq = m.size.nq, nv = m.size.nv;
  const bool has_free = m.size.njnt > 0 && m.jnt_type[0] == MJLAB_JNT_FREE;
  float* qpos = d.qpos + (size_t)w * nq;
  float* qvel = d.qvel + (size_t)w * nv;
  float* ws = d.qacc_warmstart + (size_t)w * nv;
  bool bad = false;
  for (int i = lane; i < nq; i += 64) {
    const float x = qpos[i];
    bad |= !(fabsf(x) <= 3.402823466e+38f);  // NaN or inf
  }
  const int elen = episode_length[w] + 1;
  float org[3] = {0.f, 0.f, 0.f};
  if (env_origins)
    for (int k = 0; k < 3; ++k) org[k] = env_origins[3 * w + k];
  // world z of the root's up axis = 1 - 2 (qx^2 + qy^2): the bad-orientation test of the
  // reference (envs/mdp/terminations.py bad_orientation: projected gravity vs a limit angle)
  bool fell = has_free && (qpos[2] - org[2] < min_height || 1.f - 2.f * (qpos[4] * qpos[4] + qpos[5] * qpos[5]) < min_up_z);
  if (has_free && term_ref) {  // the tracking task's anchor tests against the motion frame of this world's phase
    const float up_z = 1.f - 2.f * (qpos[4] * qpos[4] + qpos[5] * qpos[5]);
    fell |= fabsf(qpos[2] - term_ref[2 * w]) > term_dz || fabsf(up_z - term_ref[2 * w + 1]) > term_dup;
  }
  int mo_t = 0;
  if (has_free && mo) {  // the tracking task's anchor tests against the motion frame of this world's phase (mjlab_motion_reset_t)
    mo_t = mo->time_steps[w];
    const int f = mo_t < mo->nframe - 1 ? mo_t : mo->nframe - 1;
    const float* rq = mo->root_quat + 4 * f;
    // products and sums rounded one by one, like the chain of torch ops this replaces (same decisions bit for bit)
    const float ref_up = __fsub_rn(1.f, __fmul_rn(2.f, __fadd_rn(__fmul_rn(rq[1], rq[1]), __fmul_rn(rq[2], rq[2]))));
    const float up_z = __fsub_rn(1.f, __fmul_rn(2.f, __fadd_rn(__fmul_rn(qpos[4], qpos[4]), __fmul_rn(qpos[5], qpos[5]))));
    const float ref_z = __fadd_rn(mo->root_pos[3 * f + 2], org[2]);
    fell |= fabsf(__fsub_rn(qpos[2], ref_z)) > mo->dz || fabsf(__fsub_rn(up_z, ref_up)) > mo->dup || mo_t + 1 >= mo->nframe;
  }
  const bool reset = __ballot(bad) != 0ull || fell || elen >= max_len;
  __syncthreads();  // every lane has read qpos[2..5] before any lane overwrites them
  if (reset && has_free && mo) {
    const int nj = nq - 7;
    const float* u = mo->rnd + (size_t)w * (14 + nj);
    int bin = (int)(u[0] * (float)mo->bins);
    bin = bin < mo->bins - 1 ? bin : mo->bins - 1;
    const int tn = (int)(((float)bin + u[1]) / (float)mo->bins * (float)(mo->nframe - 1));
    float pose[6], vel[6];
    for (int k = 0; k < 6; ++k) {
      pose[k] = mo->pose_lo[k] + (mo->pose_hi[k] - mo->pose_lo[k]) * u[2 + k];
      vel[k] = mo->vel_lo[k] + (mo->vel_hi[k] - mo->vel_lo[k]) * u[8 + k];
    }
    float sr, cr, sp, cp, sy, cy;
    sincosf(0.5f * pose[3], &sr, &cr);
    sincosf(0.5f * pose[4], &sp, &cp);
    sincosf(0.5f * pose[5], &sy, &cy);
    const float dq[4] = {cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp};
    float q[4], wb[3], wv[3];
    mul_quat(q, dq, mo->root_quat + 4 * tn);
    for (int k = 0; k < 3; ++k) wv[k] = mo->root_ang_vel[3 * tn + k] + vel[3 + k];
    quat_apply_dev(wb, q, wv, -1.f);
    for (int i = lane; i < nq; i += 64) {
      float x;
      if (i < 3) x = mo->root_pos[3 * tn + i] + pose[i] + org[i];
      else if (i < 7) x = q[i - 3];
      else {
        x = mo->joint_pos[(size_t)tn * nj + (i - 7)] + mo->joint_lo + (mo->joint_hi - mo->joint_lo) * u[14 + (i - 7)];
        x = fminf(fmaxf(x, mo->soft_limits[2 * (i - 7)]), mo->soft_limits[2 * (i - 7) + 1]);
      }
      qpos[i] = x;
    }
    for (int i = lane; i < nv; i += 64) {
      qvel[i] = i < 3 ? mo->root_lin_vel[3 * tn + i] + vel[i] : (i < 6 ? wb[i - 3] : mo->joint_vel[(size_t)tn * (nv - 6) + (i - 6)]);
      ws[i] = 0.f;
    }
    if (lane == 0) mo->time_steps[w] = tn;
  } else if (reset && reset_qpos) {  // the task's own reset state for this world (write_root_state / write_joint_state rows)
    for (int i = lane; i < nq; i += 64) qpos[i] = reset_qpos[(size_t)w * nq + i];
    for (int i = lane; i < nv; i += 64) { qvel[i] = reset_qvel[(size_t)w * nv + i]; ws[i] = 0.f; }
  } else if (reset) {
    for (int i = lane; i < nq; i += 64) {
      float x = key_qpos[i];
      if (has_free) {
        const float yaw = (rnd3[3 * w + 2] * 2.f - 1.f) * 3.14f;
        if (i < 2) x += rnd3[3 * w + i] - 0.5f + org[i];
        else if (i == 2) x += org[2];
        else if (i == 3) x = cosf(yaw * 0.5f);
        else if (i == 4 || i == 5) x = 0.f;
        else if (i == 6) x = sinf(yaw * 0.5f);
      }
      qpos[i] = x;
    }
    for (int i = lane; i < nv; i += 64) { qvel[i] = 0.f; ws[i] = 0.f; }
  } else {
    for (int i = lane; i < nv; i += 64) { qvel[i] = nan_to_num_dev(qvel[i]); ws[i] = nan_to_num_dev(ws[i]); }
  }
  if (lane == 0) {
    episode_length[w] = reset ? 0 : elen;
    reset_mask[w] = reset ? 1 : 0;
    if (has_free && mo && !reset) mo->time_steps[w] = mo_t + 1;
  }
  return reset;
}
#ifdef MJLAB_MAIN_TU
__global__ __launch_bounds__(64) void k_masked_reset(const Model m, const Data d, const float* key_qpos, const float* rnd3,
                                                      int* episode_length, const int max_len, const float min_height, int* reset_mask,
                                                      const float* env_origins, const float min_up_z) {
  (void)masked_reset_world(m, d, blockIdx.x, threadIdx.x, key_qpos, rnd3, episode_length, max_len, min_height, reset_mask, env_origins, min_up_z);
}
#endif  // MJLAB_MAIN_TU

// Interval push (reference envs/mdp/events.py:127-143 push_by_setting_velocity under the event
// manager's per-env interval timer, managers/event_manager.py:116-138): see include/mjlab_amd.h.
__device__ __forceinline__ void interval_push_world(const Model& m, const Data& d, const int w, float* time_left, const float* rnd7, const float dt,
                                                    const float t_lo, const float t_hi, const mjlab_push_range_t& range) {
  float t = time_left[w] - dt;
  if (t < 1e-6f) {
    const float* r = rnd7 + (size_t)w * 7;
    t = t_lo + r[6] * (t_hi - t_lo);
    float* qvel = d.qvel + (size_t)w * m.size.nv;
    const float* q = d.qpos + (size_t)w * m.size.nq + 3;  // root quaternion (free joint first: checked by the caller)
    float dv[6];
    for (int k = 0; k < 6; ++k) dv[k] = range.lo[k] + r[k] * (range.hi[k] - range.lo[k]);
    for (int k = 0; k < 3; ++k) qvel[k] += dv[k];  // linear: world frame, as qvel stores it
    // angular: qvel holds it in the body frame; the kick is drawn in the world frame
    const float qq[4] = {q[0], q[1], q[2], q[3]};
    float wb[3];
    quat_apply_dev(wb, qq, dv + 3, -1.f);
    for (int k = 0; k < 3; ++k) qvel[3 + k] += wb[k];
  }
  time_left[w] = t;
}
#ifdef MJLAB_MAIN_TU
__global__ __launch_bounds__(64) void k_interval_push(const Model m, const Data d, float* time_left, const float* rnd7, const float dt,
                                                       const float t_lo, const float t_hi, const mjlab_push_range_t range) {
  const int w = blockIdx.x * 64 + threadIdx.x;
  if (w >= m.size.nworld) return;
  interval_push_world(m, d, w, time_left, rnd7, dt, t_lo, t_hi, range);
}
#endif  // MJLAB_MAIN_TU

// ====================================================================================
// repeat_array_kernel replacement (reference src/mjlab/sim/randomization.py:9-17)
// ====================================================================================
template <typename T>
__global__ void k_tile(T* dst, const T* src, long long nelem, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    dst[i] = src[i % nelem];
}

// self-test of the DPP reductions against the ds_bpermute versions
#ifdef MJLAB_MAIN_TU
// Scratch poisoning (mjlab_poison_scratch; round 6).  The private segment of a wave is not cleared between
// kernels: a kernel that reloads a spill slot it never wrote sees zeros in a fresh process and the previous kernel's spills
// afterwards -- exactly how a spill store that hipcc had placed under EXEC == 0 stayed invisible until two instantiations ran back to
// back.  This kernel makes "the previous kernel" the worst case on purpose: every lane fills a frame larger than any kernel's of this
// library with words that are a NaN as a float, ~2^31 as an index and a non-canonical address as the high half of a pointer, and
// stays resident long enough for the launch to occupy every scratch slot it was given.  The run-time rotation of the index keeps the
// array in scratch (a constant index would be promoted to registers).
#define MJLAB_POISON_WORDS 320  // 1280 B per lane (tests/test_code_object.py: every kernel's frame <= 1024 B)
__global__ __launch_bounds__(64) void k_poison_scratch(const unsigned pattern, const int rot, const long long spin, unsigned* sink) {
  unsigned a[MJLAB_POISON_WORDS];
  for (int i = 0; i < MJLAB_POISON_WORDS; ++i) a[(i + rot) % MJLAB_POISON_WORDS] = pattern + (unsigned)i;
  const long long t0 = clock64();
  while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  unsigned acc = 0;
  for (int i = 0; i < MJLAB_POISON_WORDS; i += 7) acc ^= a[(i + 2 * rot) % MJLAB_POISON_WORDS];
  if (acc == 0x9e3779b9u) *sink = acc;  // (never true for the patterns used; keeps the loads, hence the stores, alive)
}
__global__ void k_selftest(const float* in, int* nerr) {
  const float v = in[blockIdx.x * 64 + threadIdx.x];
  const float a = wave_sum(v), b = wave_sum_shfl(v);
  const float c = group16_sum(v), e = group16_sum_shfl(v);
  const float tol = 1e-4f * (1.f + fabsf(b));
  if (fabsf(a - b) > tol || fabsf(c - e) > 1e-4f * (1.f + fabsf(e))) atomicAdd(nerr, 1);
}
#endif  // MJLAB_MAIN_TU
