// stage_velocity.h -- stage 3: comVel, rne, actuation, qfrc_smooth.
// Part of kernels.h (included there, in this order, by every translation unit of the library); not a
// stand-alone header.
#pragma once

// ====================================================================================
// Stage 3: velocity + smooth forces (mj_comVel, mj_passive, mj_rne, mj_fwdActuation)
// ====================================================================================
__host__ __device__ inline int velocity_lds_floats(const mjlab_sizes_t& s) {
  return 2 * s.nv + 12 * s.nv + 10 * s.nbody + 24 * s.nbody;
}

// One actuator: joint transmission, fixed gain, affine bias (reference utils/spec_config.py:441-453).
struct ActuatorConst { int trn, ctrllimited, forcelimited; float gear, ctrl, crange[2], frange[2], gain, bias[3]; };
__device__ __forceinline__ void load_actuator(ActuatorConst& c, const Model& m, const float* ctrl, const float* gain, const float* biasprm,
                                              const float* crange, const float* frange, const float* gear, int a) {
  c.trn = m.actuator_trnid[2 * a];
  c.ctrllimited = m.actuator_ctrllimited[a];
  c.forcelimited = m.actuator_forcelimited[a];
  c.gear = gear[6 * a];
  c.ctrl = ctrl[a];
  c.gain = gain[10 * a];
  for (int k = 0; k < 2; ++k) { c.crange[k] = crange[2 * a + k]; c.frange[k] = frange[2 * a + k]; }
  for (int k = 0; k < 3; ++k) c.bias[k] = biasprm[10 * a + k];
}

// v += sum_k cdof_k qvel_k (and a += sum_k cdd_k qvel_k) over the set bits k of (lo, hi), ascending.
// Four terms per round: their LDS reads are issued together (a rolled loop would wait for each
// term's operands in turn); slots past the end of the mask contribute c * 0.
template <bool WITH_A>
__device__ __forceinline__ void chain_accum(unsigned long long mk, const float* s_qvel, const float* s_cdof, const float* s_cdd,
                                            float (&v)[6], float (&a)[6]) {
  while (mk) {
    int k[4];
    float qv[4], c[4][6], cd[4][6];
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // lowest set bit, cleared; -1 once the mask is empty (all by value: registers)
      k[u] = mk ? __ffsll((long long)mk) - 1 : -1;
      mk &= mk - 1ull;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = k[u] >= 0 ? k[u] : 0;
      qv[u] = k[u] >= 0 ? s_qvel[kk] : 0.f;
      for (int e = 0; e < 6; ++e) { c[u][e] = s_cdof[6 * kk + e]; if (WITH_A) cd[u][e] = s_cdd[6 * kk + e]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      for (int e = 0; e < 6; ++e) { v[e] += c[u][e] * qv[u]; if (WITH_A) a[e] += cd[u][e] * qv[u]; }
  }
}

__device__ __forceinline__ void stage_velocity(const Model& m, const Data& d, const int w, const int lane, const int flags, float* smem) {
  const int nb = m.size.nbody, nv = m.size.nv, nq = m.size.nq, nu = m.size.nu;
  float* s_qvel = smem;
  float* s_qact = s_qvel + nv;
  float* s_cdof = s_qact + nv;
  float* s_cdd = s_cdof + 6 * nv;
  float* s_cvel = s_cdd + 6 * nv;
  float* s_cfrc = s_cvel + 6 * nb;
  float* s_cfs = s_cfrc + 6 * nb;
  PROF_INIT();
  // ---- prologue: everything this lane needs from memory in any of its roles (body / dof /
  // actuator `lane`), as one batch of independent loads plus a short second one for values reached
  // through an index; the rest of the kernel only stores (see k_position)
  const float* qpos = d.qpos + (size_t)w * nq;
  const int rb = lane < nb ? lane : 0;  // body role
  const int b_snum = m.body_subtreenum[rb];
  const unsigned b_mlo = (unsigned)m.body_dofmask[2 * rb], b_mhi = (unsigned)m.body_dofmask[2 * rb + 1];
  float b_in[10], b_xf[6];
  for (int k = 0; k < 10; ++k) b_in[k] = d.cinert[((size_t)w * nb + rb) * 10 + k];
  for (int k = 0; k < 6; ++k) b_xf[k] = d.xfrc_applied[((size_t)w * nb + rb) * 6 + k];
  const int rv = lane < nv ? lane : 0;  // dof role
  const int v_body = m.dof_bodyid[rv], v_jnt = m.dof_jntid[rv];
  const float v_damp = MF(dof_damping)[rv], v_applied = d.qfrc_applied[(size_t)w * nv + rv];
  const float *gain = MF(actuator_gainprm), *biasprm = MF(actuator_biasprm), *crange = MF(actuator_ctrlrange),
              *frange = MF(actuator_forcerange), *gear = MF(actuator_gear);
  const float* ctrl = d.ctrl + (size_t)w * nu;
  ActuatorConst act;
  if (lane < nu) load_actuator(act, m, ctrl, gain, biasprm, crange, frange, gear, lane);
  glds_to_lds(s_qvel, d.qvel + (size_t)w * nv, nv, lane);
  glds_to_lds(s_cdof, d.cdof + (size_t)w * 6 * nv, 6 * nv, lane);
  // second level
  const int v_type = m.jnt_type[v_jnt], v_dofadr = m.jnt_dofadr[v_jnt], v_qadr = m.jnt_qposadr[v_jnt];
  const float v_stiff = MF(jnt_stiffness)[v_jnt];
  const unsigned v_mlo = (unsigned)m.body_dofmask[2 * v_body], v_mhi = (unsigned)m.body_dofmask[2 * v_body + 1];
  int a_qadr = 0, a_dadr = 0;
  float a_qpos = 0.f;
  if (lane < nu) {
    a_qadr = m.jnt_qposadr[act.trn];
    a_dadr = m.jnt_dofadr[act.trn];
    a_qpos = qpos[a_qadr];
  }
  for (int i = lane; i < nv; i += 64) s_qact[i] = 0.f;
  __syncthreads();
  PROF_MARK(0);

  // ---- mj_comVel / mj_rne without a level sweep.  cvel of a body is the sum of cdof_k qvel_k over
  // the dofs k of its ancestor chain (body_dofmask, ascending = root first, the order the
  // sequential sweep adds them in), and cdof_dot_j = cvel-just-before-dof-j x cdof_j: every dof
  // and every body sums its own chain (<= depth + 5 terms from LDS), nobody waits for a parent.
  if (lane < nv) {
    // dofs strictly before j; the three rotational dofs of a free joint all use the velocity after
    // its translations (mj_comVel), i.e. the prefix before the rotational block
    const int lim = (v_type == MJLAB_JNT_FREE && lane >= v_dofadr + 3) ? v_dofadr + 3 : lane;
    const unsigned long long mk = (((unsigned long long)v_mhi << 32) | v_mlo) & ((1ull << lim) - 1ull);  // lim < 64
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, c6[6], cd[6];
    chain_accum<false>(mk, s_qvel, s_cdof, s_cdd, v, cd);
    for (int c = 0; c < 6; ++c) c6[c] = s_cdof[6 * lane + c];
    cross_motion(cd, v, c6);
    const bool zero = v_type == MJLAB_JNT_FREE && lane < v_dofadr + 3;  // translations of a free joint
    for (int c = 0; c < 6; ++c) s_cdd[6 * lane + c] = zero ? 0.f : cd[c];
  }
  __syncthreads();
  PROF_MARK(1);
  if (lane < nb) {
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, a[6];
    for (int c = 0; c < 6; ++c) a[c] = c < 3 ? 0.f : -(float)m.opt.gravity[c - 3];
    if (lane == 0) {
      for (int c = 0; c < 6; ++c) { s_cvel[c] = 0.f; s_cfrc[c] = 0.f; }
    } else {
      chain_accum<true>(((unsigned long long)b_mhi << 32) | b_mlo, s_qvel, s_cdof, s_cdd, v, a);
      float t1[6], t2[6], t3[6];
      mul_inert_vec(t1, b_in, a);
      mul_inert_vec(t2, b_in, v);
      cross_force(t3, v, t2);
      for (int k = 0; k < 6; ++k) { s_cvel[6 * lane + k] = v[k]; s_cfrc[6 * lane + k] = t1[k] + t3[k]; }
    }
  }
  __syncthreads();
  PROF_MARK(2);
  lds_to_global(d.cvel + (size_t)w * 6 * nb, s_cvel, 6 * nb, lane);
  lds_to_global(d.cdof_dot + (size_t)w * 6 * nv, s_cdd, 6 * nv, lane);
  // ---- up-sweep as subtree range sums (a subtree is a contiguous body-id range)
  for (int it0 = 0; it0 < 6 * nb; it0 += 64) {
    const int it = it0 + lane, bq = it < 6 * nb ? it / 6 : 0;
    const int snum = __shfl(b_snum, bq);
    if (it >= 6 * nb) continue;
    const int c = it - 6 * bq, e = bq + snum;
    float a0 = 0.f, a1 = 0.f;
    int j = bq;
    for (; j + 1 < e; j += 2) { a0 += s_cfrc[6 * j + c]; a1 += s_cfrc[6 * j + 6 + c]; }
    if (j < e) a0 += s_cfrc[6 * j + c];
    s_cfs[it] = a0 + a1;
  }
  PROF_MARK(3);
  // ---- actuation
  for (int a0 = 0; a0 < nu; a0 += 64) {
    const int a = a0 + lane;
    if (a >= nu) break;
    if (a0 > 0) {  // models with more than 64 actuators: later rounds load in place
      load_actuator(act, m, ctrl, gain, biasprm, crange, frange, gear, a);
      a_qadr = m.jnt_qposadr[act.trn];
      a_dadr = m.jnt_dofadr[act.trn];
      a_qpos = qpos[a_qadr];
    }
    float c = act.ctrl;
    if (act.ctrllimited) c = clipf(c, act.crange[0], act.crange[1]);
    const float len = act.gear * a_qpos, vel = act.gear * s_qvel[a_dadr];
    float f = act.gain * c + act.bias[0] + act.bias[1] * len + act.bias[2] * vel;
    if (act.forcelimited) f = clipf(f, act.frange[0], act.frange[1]);
    d.actuator_force[(size_t)w * nu + a] = f;
    atomicAdd(&s_qact[a_dadr], act.gear * f);
  }
  __syncthreads();
  // ---- bias, passive, smooth force; lanes = dofs
  // bodies with a nonzero Cartesian perturbation (usually none)
  bool xnz = false;
  if (lane > 0 && lane < nb) for (int k = 0; k < 6; ++k) xnz |= b_xf[k] != 0.f;
  const unsigned long long xmask = __ballot(xnz);
  if (lane < nv) {
    const int i = lane;
    float c6[6];
    for (int k = 0; k < 6; ++k) c6[k] = s_cdof[6 * i + k];
    float bias = 0.f;
    for (int k = 0; k < 6; ++k) bias += c6[k] * s_cfs[6 * v_body + k];
    float passive = -v_damp * s_qvel[i];
    if (v_type != MJLAB_JNT_FREE && v_stiff != 0.f) passive -= v_stiff * (qpos[v_qadr] - MF(qpos_spring)[v_qadr]);  // springref, not qpos0
    float smooth = passive - bias + v_applied + s_qact[i];
    if (xmask) {
      const float* xfrc = d.xfrc_applied + (size_t)w * 6 * nb;
      const float* xipos = d.xipos_rel + (size_t)w * 3 * nb;  // both in the world's local frame
      const float* sub = d.subtree_crel + (size_t)w * 3 * nb;
      for (int b = 1; b < nb; ++b) {
        if (!((xmask >> b) & 1ull) || !dof_in_chain(m, b, i)) continue;
        float f[6];
        for (int k = 0; k < 6; ++k) f[k] = xfrc[6 * b + k];
        const int root = m.body_rootid[b];
        float off[3], jp[3];
        for (int k = 0; k < 3; ++k) off[k] = xipos[3 * b + k] - sub[3 * root + k];
        cross3(jp, c6, off);
        for (int k = 0; k < 3; ++k) jp[k] += c6[3 + k];
        smooth += dot3(jp, f) + dot3(c6, f + 3);
      }
    }
    d.qfrc_bias[(size_t)w * nv + i] = bias;
    d.qfrc_passive[(size_t)w * nv + i] = passive;
    d.qfrc_actuator[(size_t)w * nv + i] = s_qact[i];
    d.qfrc_smooth[(size_t)w * nv + i] = smooth;
  }
  PROF_MARK(4);
  PROF_FLUSH(d.profile + (size_t)w * 64 + 24);
}

#ifdef MJLAB_MAIN_TU
__global__ __launch_bounds__(64, 4) void k_velocity(const Model m, const Data d, const int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int w = blockIdx.x, lane = threadIdx.x;
  if ((flags & FLAG_MASK) && !d.world_mask[w]) return;
  stage_velocity(m, d, w, lane, flags, smem);
}
#endif  // MJLAB_MAIN_TU
