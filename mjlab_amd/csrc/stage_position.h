// stage_position.h -- stage 1: kinematics, comPos, crb, dense M.
// Part of kernels.h (included there, in this order, by every translation unit of the library); not a
// stand-alone header.
#pragma once

// ====================================================================================
// Stage 1: position  (mj_kinematics, mj_comPos, mj_crb, mj_factorM)
// ====================================================================================
__device__ __forceinline__ void local2global(float* xp, float* xm, const float* bpos, const float* bquat,
                                             const float* bmat, const float* pos, const float* quat) {
  float q[4], t[3];
  mul_mat_vec3(t, bmat, pos);
  xp[0] = bpos[0] + t[0]; xp[1] = bpos[1] + t[1]; xp[2] = bpos[2] + t[2];
  mul_quat(q, bquat, quat);
  quat2mat(xm, q);
}

__host__ __device__ inline int position_lds_floats(const mjlab_sizes_t& s) {
  int nb = s.nbody, nv = s.nv, nj = s.njnt, ld = nv | 1;
  int persistent = (3 * nb + 10 * nb + 10 * nb + 6 * nv + 6 * nv + nb + 3) & ~3;
  int kin = s.nq + 28 * nb + 6 * nj;
  int mat = nv * ld;
  return persistent + (kin > mat ? kin : mat);
}

// Stage bodies are device functions of (world, lane): the per-stage kernels below are thin wrappers,
// and k_presolve / k_substep (mjlab_amd.hip) run several of them back to back in one launch.
// Returns true when FLAG_FOLD applies to this world (nothing was recomputed: the collision and
// constraint-build stages are to be skipped too).
__device__ __forceinline__ bool stage_position(const Model& m, const Data& d, const int w, const int lane, const int flags, float* smem) {
  const int nb = m.size.nbody, nv = m.size.nv, nq = m.size.nq, nj = m.size.njnt, ng = m.size.ngeom, ns = m.size.nsite;
  if (flags & FLAG_FOLD) {
    // The reference calls forward() on all worlds after resets and then, with a new action in
    // ctrl, step() -- whose position, collision and constraint-build stages depend on qpos, qvel
    // and the model only and would reproduce the forward pass bit for bit.  Where qpos and qvel
    // still equal the snapshot taken by forward(), those three stages are skipped
    // ("forward folded into the next step"); velocity / actuation and the
    // solve always run.
    int reuse = 0;
    if (d.fold_valid[w]) {
      bool diff = false;
      for (int i = lane; i < nq; i += 64) diff |= __float_as_int(d.qpos[(size_t)w * nq + i]) != __float_as_int(d.sh_qpos[(size_t)w * nq + i]);
      for (int i = lane; i < nv; i += 64) diff |= __float_as_int(d.qvel[(size_t)w * nv + i]) != __float_as_int(d.sh_qvel[(size_t)w * nv + i]);
      reuse = __ballot(diff) == 0ull;
    }
    if (lane == 0) d.fold_reuse[w] = reuse;
    if (reuse) return true;
  }
  float* s_sub = smem;
  float* s_cinert = s_sub + 3 * nb;
  float* s_crb = s_cinert + 10 * nb;
  float* s_cdof = s_crb + 10 * nb;
  float* s_buf = s_cdof + 6 * nv;
  float* s_mass = s_buf + 6 * nv;
  float* regA = smem + ((24 * nb + 12 * nv + 3) & ~3);  // 16-byte aligned
  float* s_qpos = regA;
  float* s_xpos = s_qpos + nq;
  float* s_xquat = s_xpos + 3 * nb;
  float* s_xmat = s_xquat + 4 * nb;
  float* s_xipos = s_xmat + 9 * nb;
  float* s_ximat = s_xipos + 3 * nb;
  float* s_xanchor = s_ximat + 9 * nb;
  float* s_xaxis = s_xanchor + 3 * nj;
  const int ld = nv | 1;  // odd: conflict-free row and column walks
  float* s_M = regA;      // aliases the kinematics region once it has been consumed

  // ---- prologue: every model constant this lane needs in any of its roles (body / dof / geom /
  // site `lane`) is requested here, as ONE batch of loads with independent addresses (a second,
  // short one for values reached through an index).  The stage is bound by dependent global
  // round trips, not by arithmetic: with the constants in registers the rest of the kernel issues
  // no global load at all, and its stores never sit in front of a load it has to wait for.
  const float* qpos0 = MF(qpos0);
  const float *body_pos = MF(body_pos), *body_quat = MF(body_quat), *jnt_axis = MF(jnt_axis), *jnt_pos = MF(jnt_pos);
  const int rb = lane < nb ? lane : 0;  // body role
  const int b_pid = m.body_parentid[rb], b_ja = m.body_jntadr[rb], b_jn = m.body_jntnum[rb];
  const int b_snum = m.body_subtreenum[rb], b_root = m.body_rootid[rb];
  float b_pos[3], b_quat[4], b_ipos[3], b_iquat[4], b_inertia[3];
  {
    const float *body_ipos = MF(body_ipos), *body_iquat = MF(body_iquat), *inertia = MF(body_inertia);
    for (int k = 0; k < 3; ++k) { b_pos[k] = body_pos[3 * rb + k]; b_ipos[k] = body_ipos[3 * rb + k]; b_inertia[k] = inertia[3 * rb + k]; }
    for (int k = 0; k < 4; ++k) { b_quat[k] = body_quat[4 * rb + k]; b_iquat[k] = body_iquat[4 * rb + k]; }
  }
  const float b_mass = MF(body_mass)[rb], b_stm = MF(body_subtreemass)[rb];
  const int rv = lane < nv ? lane : 0;  // dof role
  const int v_jnt = m.dof_jntid[rv], v_body = m.dof_bodyid[rv], v_pid = m.dof_parentid[rv];
  const float v_arm = MF(dof_armature)[rv];
  // geom role: the first two rounds of the (moving) geom loop, and the site of this lane
  const int ng0 = m.size.nstaticgeom;
  int g_body[2];
  float g_pos[2][3], g_quat[2][4];
  {
    const float *gpos = MF(geom_pos), *gquat = MF(geom_quat);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int g = ng0 + 64 * r + lane < ng ? ng0 + 64 * r + lane : (ng > 0 ? ng - 1 : 0);
      g_body[r] = ng > 0 ? m.geom_bodyid[g] : 0;
      for (int k = 0; k < 3; ++k) g_pos[r][k] = ng > 0 ? gpos[3 * g + k] : 0.f;
      for (int k = 0; k < 4; ++k) g_quat[r][k] = ng > 0 ? gquat[4 * g + k] : 0.f;
    }
  }
  const int ns0 = m.size.nstaticsite;  // sites of static bodies keep the poses written at construction
  int t_body = 0;
  float t_pos[3] = {0.f, 0.f, 0.f}, t_quat[4] = {1.f, 0.f, 0.f, 0.f};
  if (ns0 + lane < ns) {
    const float *spos = MF(site_pos), *squat = MF(site_quat);
    t_body = m.site_bodyid[ns0 + lane];
    for (int k = 0; k < 3; ++k) t_pos[k] = spos[3 * (ns0 + lane) + k];
    for (int k = 0; k < 4; ++k) t_quat[k] = squat[4 * (ns0 + lane) + k];
  }
  // second level: reached through an index loaded above
  const int v_dofadr = m.jnt_dofadr[v_jnt], v_type = m.jnt_type[v_jnt], v_root = m.body_rootid[v_body];
  int b_jtype = -1, b_qadr = 0;
  float b_jax[3] = {0.f, 0.f, 1.f}, b_jpos[3] = {0.f, 0.f, 0.f}, b_q0 = 0.f;
  if (b_jn > 0) {
    b_jtype = m.jnt_type[b_ja];
    b_qadr = m.jnt_qposadr[b_ja];
    for (int k = 0; k < 3; ++k) { b_jax[k] = jnt_axis[3 * b_ja + k]; b_jpos[k] = jnt_pos[3 * b_ja + k]; }
    b_q0 = qpos0[b_qadr];
  }

  global_to_lds(s_qpos, d.qpos + (size_t)w * nq, nq, lane);
  // The world's LOCAL FRAME (include/mjlab_fields.h, xorigin): the position of the first floating base, rounded to whole
  // metres.  qpos - org is exact in fp32 (both are multiples of the same ulp and the difference is small), every pose below is
  // composed from it and from body-relative offsets, so offsets between bodies keep full fp32 precision wherever the robot
  // stands; the public world-frame arrays get org added back when they are written.
  float org[3] = {0.f, 0.f, 0.f};
  {
    const bool floating = lane < nb && b_jn == 1 && b_jtype == MJLAB_JNT_FREE && b_pid == 0;
    const unsigned long long fm = __ballot(floating);
    if (fm && !(m.opt.flags & MJLAB_OPT_WORLD_FRAME)) {
      const int src = (int)__builtin_ctzll(fm);
      for (int k = 0; k < 3; ++k) org[k] = rintf(lane_bcast_dyn(floating ? d.qpos[(size_t)w * nq + b_qadr + k] : 0.f, src));  // wave-uniform (SGPRs)
    }
  }
  if (lane == 0) {
    for (int k = 0; k < 3; ++k) { s_xpos[k] = -org[k]; d.xorigin[(size_t)w * 3 + k] = org[k]; }
    s_xquat[0] = 1.f; s_xquat[1] = s_xquat[2] = s_xquat[3] = 0.f;
  }
  if (lane < 9) s_xmat[lane] = (lane % 4 == 0) ? 1.f : 0.f;
  __syncthreads();

  PROF_INIT();
  // ---- kinematics: lane = body (nbody <= 64, enforced by check_model), no level-by-level sweep.
  // (1) every body computes its pose RELATIVE TO ITS PARENT (body offset + joint motion; the
  //     sincos and all model loads happen once, in parallel) and its joints' anchors / axes in the
  //     parent frame;
  // (2) every body composes the relative poses of its ancestors, walking up to the world body
  //     (<= nlevel - 1 steps of one quaternion product + one rotation, inputs from LDS);
  // (3) joints' anchors / axes are taken to the world frame with the parent's pose.
  // The tree depth (11 for G1) costs 11 short dependent steps instead of 11 full passes.
  {
    float* s_lpos = s_xmat;            // relative poses live in the (not yet needed) xmat area
    float* s_lquat = s_xmat + 3 * nb;
    int* s_pid = (int*)s_mass;         // parent ids on chip (nb slots, refilled with masses later): the upward walk is a chain of dependent reads
    const int b = rb, pid = b_pid, ja = b_ja, jn = b_jn;
    float pos[3], quat[4];
    if (lane < nb) {
      s_pid[b] = pid;
      for (int k = 0; k < 3; ++k) pos[k] = b_pos[k];
      for (int k = 0; k < 4; ++k) quat[k] = b_quat[k];
      if (jn == 1 && b_jtype == MJLAB_JNT_FREE) {
        for (int k = 0; k < 3; ++k) pos[k] = s_qpos[b_qadr + k];
        for (int k = 0; k < 4; ++k) quat[k] = s_qpos[b_qadr + 3 + k];
        normalize4(quat);
        for (int k = 0; k < 3; ++k) { s_xanchor[3 * ja + k] = pos[k]; s_xaxis[3 * ja + k] = b_jax[k]; }
      } else {
        for (int j = ja; j < ja + jn; ++j) {
          float ax[3], jpos[3], xax[3], anc[3], t[3];
          int type;
          float dq;
          if (j == ja) {  // the first joint's constants are in registers already
            type = b_jtype;
            dq = s_qpos[b_qadr] - b_q0;
            for (int k = 0; k < 3; ++k) { ax[k] = b_jax[k]; jpos[k] = b_jpos[k]; }
          } else {
            const int qadr = m.jnt_qposadr[j];
            type = m.jnt_type[j];
            dq = s_qpos[qadr] - qpos0[qadr];
            for (int k = 0; k < 3; ++k) { ax[k] = jnt_axis[3 * j + k]; jpos[k] = jnt_pos[3 * j + k]; }
          }
          rot_vec_quat(xax, ax, quat);
          rot_vec_quat(t, jpos, quat);
          for (int k = 0; k < 3; ++k) anc[k] = t[k] + pos[k];
          if (type == MJLAB_JNT_SLIDE) {
            for (int k = 0; k < 3; ++k) pos[k] += xax[k] * dq;
          } else {
            float ql[4], qn[4];
            axis_angle2quat(ql, ax, dq);
            mul_quat(qn, quat, ql);
            for (int k = 0; k < 4; ++k) quat[k] = qn[k];
            rot_vec_quat(t, jpos, quat);
            for (int k = 0; k < 3; ++k) pos[k] = anc[k] - t[k];
          }
          for (int k = 0; k < 3; ++k) { s_xanchor[3 * j + k] = anc[k]; s_xaxis[3 * j + k] = xax[k]; }  // parent frame
        }
      }
      if (pid == 0 && b > 0) {  // children of the world: into the local frame (exact for the floating base)
        for (int k = 0; k < 3; ++k) pos[k] -= org[k];
        if (jn == 1 && b_jtype == MJLAB_JNT_FREE) for (int k = 0; k < 3; ++k) s_xanchor[3 * ja + k] = pos[k];
        else for (int j = ja; j < ja + jn; ++j) for (int k = 0; k < 3; ++k) s_xanchor[3 * j + k] -= org[k];
      }
      for (int k = 0; k < 3; ++k) s_lpos[3 * b + k] = pos[k];
      for (int k = 0; k < 4; ++k) s_lquat[4 * b + k] = quat[k];
    }
    __syncthreads();
    // (2) compose upwards; the world body (0) is the identity
    float ppos[3] = {0.f, 0.f, 0.f}, pquat[4] = {1.f, 0.f, 0.f, 0.f};  // pose of this body's PARENT
    if (lane < nb && lane > 0) {
      bool first = true;
      for (int a = pid; a > 0; a = s_pid[a]) {
        float ap[3], aq[4], t[3], q2[4];
        for (int k = 0; k < 3; ++k) ap[k] = s_lpos[3 * a + k];
        for (int k = 0; k < 4; ++k) aq[k] = s_lquat[4 * a + k];
        if (first) {
          for (int k = 0; k < 3; ++k) ppos[k] = ap[k];
          for (int k = 0; k < 4; ++k) pquat[k] = aq[k];
          first = false;
        } else {
          rot_vec_quat(t, ppos, aq);
          for (int k = 0; k < 3; ++k) ppos[k] = ap[k] + t[k];
          mul_quat(q2, aq, pquat);
          for (int k = 0; k < 4; ++k) pquat[k] = q2[k];
        }
      }
      normalize4(pquat);
      float t[3], q2[4];
      rot_vec_quat(t, pos, pquat);
      for (int k = 0; k < 3; ++k) pos[k] = ppos[k] + t[k];
      mul_quat(q2, pquat, quat);
      for (int k = 0; k < 4; ++k) quat[k] = q2[k];
      normalize4(quat);
    }
    __syncthreads();  // every lane has read the relative poses: the xmat area may be overwritten
    if (lane < 9) s_xmat[lane] = (lane % 4 == 0) ? 1.f : 0.f;  // world body (its slot held relative poses)
    if (lane < nb && lane > 0) {
      float R[9];
      quat2mat(R, quat);
      for (int k = 0; k < 3; ++k) s_xpos[3 * b + k] = pos[k];
      for (int k = 0; k < 4; ++k) s_xquat[4 * b + k] = quat[k];
      for (int k = 0; k < 9; ++k) s_xmat[9 * b + k] = R[k];
      // (3) this body's joints: parent frame -> world (the free joint's are already world)
      if (!(jn == 1 && b_jtype == MJLAB_JNT_FREE)) {
        for (int j = ja; j < ja + jn; ++j) {
          float al[3], xl[3], t[3], u[3];
          for (int k = 0; k < 3; ++k) { al[k] = s_xanchor[3 * j + k]; xl[k] = s_xaxis[3 * j + k]; }
          rot_vec_quat(t, al, pquat);
          rot_vec_quat(u, xl, pquat);
          for (int k = 0; k < 3; ++k) { s_xanchor[3 * j + k] = ppos[k] + t[k]; s_xaxis[3 * j + k] = u[k]; }
        }
      }
    }
    __syncthreads();
  }
  PROF_MARK(0);
  // ---- inertial frames, geoms, sites (constants from the prologue)
  if (lane < nb) {
    float bp[3], bq[4], bm[9], xp[3], xm[9];
    for (int k = 0; k < 3; ++k) bp[k] = s_xpos[3 * lane + k];
    for (int k = 0; k < 4; ++k) bq[k] = s_xquat[4 * lane + k];
    for (int k = 0; k < 9; ++k) bm[k] = s_xmat[9 * lane + k];
    local2global(xp, xm, bp, bq, bm, b_ipos, b_iquat);
    for (int k = 0; k < 3; ++k) s_xipos[3 * lane + k] = xp[k];
    for (int k = 0; k < 9; ++k) s_ximat[9 * lane + k] = xm[k];
  }
  {
    float* gx = d.geom_xpos + (size_t)w * 3 * ng;
    float* gxr = d.geom_xrel + (size_t)w * 3 * ng;  // local frame: what the collision stage reads
    float* gm = d.geom_xmat + (size_t)w * 9 * ng;
    // geoms of static bodies keep the poses written at construction (sizes.nstaticgeom)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int g = ng0 + 64 * r + lane;
      if (g < ng) {
        const int b = g_body[r];
        float bp[3], bq[4], bm[9], xp[3], xm[9];
        for (int k = 0; k < 3; ++k) bp[k] = s_xpos[3 * b + k];
        for (int k = 0; k < 4; ++k) bq[k] = s_xquat[4 * b + k];
        for (int k = 0; k < 9; ++k) bm[k] = s_xmat[9 * b + k];
        local2global(xp, xm, bp, bq, bm, g_pos[r], g_quat[r]);
        for (int k = 0; k < 3; ++k) { gx[3 * g + k] = xp[k] + org[k]; gxr[3 * g + k] = xp[k]; }
        for (int k = 0; k < 9; ++k) gm[9 * g + k] = xm[k];
      }
    }
    const float *gpos = MF(geom_pos), *gquat = MF(geom_quat);
    for (int g = ng0 + 128 + lane; g < ng; g += 64) {  // models with more than 128 moving geoms
      const int b = m.geom_bodyid[g];
      float bp[3], bq[4], bm[9], ip[3], iq[4], xp[3], xm[9];
      for (int k = 0; k < 3; ++k) { bp[k] = s_xpos[3 * b + k]; ip[k] = gpos[3 * g + k]; }
      for (int k = 0; k < 4; ++k) { bq[k] = s_xquat[4 * b + k]; iq[k] = gquat[4 * g + k]; }
      for (int k = 0; k < 9; ++k) bm[k] = s_xmat[9 * b + k];
      local2global(xp, xm, bp, bq, bm, ip, iq);
      for (int k = 0; k < 3; ++k) { gx[3 * g + k] = xp[k] + org[k]; gxr[3 * g + k] = xp[k]; }
      for (int k = 0; k < 9; ++k) gm[9 * g + k] = xm[k];
    }
    float* sx = d.site_xpos + (size_t)w * 3 * ns;
    float* sm = d.site_xmat + (size_t)w * 9 * ns;
    if (ns0 + lane < ns) {
      const int b = t_body, g = ns0 + lane;
      float bp[3], bq[4], bm[9], xp[3], xm[9];
      for (int k = 0; k < 3; ++k) bp[k] = s_xpos[3 * b + k];
      for (int k = 0; k < 4; ++k) bq[k] = s_xquat[4 * b + k];
      for (int k = 0; k < 9; ++k) bm[k] = s_xmat[9 * b + k];
      local2global(xp, xm, bp, bq, bm, t_pos, t_quat);
      for (int k = 0; k < 3; ++k) sx[3 * g + k] = xp[k] + org[k];
      for (int k = 0; k < 9; ++k) sm[9 * g + k] = xm[k];
    }
    const float *spos = MF(site_pos), *squat = MF(site_quat);
    for (int g = ns0 + 64 + lane; g < ns; g += 64) {  // models with more than 64 moving sites
      const int b = m.site_bodyid[g];
      float bp[3], bq[4], bm[9], ip[3], iq[4], xp[3], xm[9];
      for (int k = 0; k < 3; ++k) { bp[k] = s_xpos[3 * b + k]; ip[k] = spos[3 * g + k]; }
      for (int k = 0; k < 4; ++k) { bq[k] = s_xquat[4 * b + k]; iq[k] = squat[4 * g + k]; }
      for (int k = 0; k < 9; ++k) bm[k] = s_xmat[9 * b + k];
      local2global(xp, xm, bp, bq, bm, ip, iq);
      for (int k = 0; k < 3; ++k) sx[3 * g + k] = xp[k] + org[k];
      for (int k = 0; k < 9; ++k) sm[9 * g + k] = xm[k];
    }
  }
  __syncthreads();
  PROF_MARK(1);
  lds3_to_global_org(d.xpos + (size_t)w * 3 * nb, s_xpos, 3 * nb, lane, org);
  lds_to_global(d.xquat + (size_t)w * 4 * nb, s_xquat, 4 * nb, lane);
  lds_to_global(d.xmat + (size_t)w * 9 * nb, s_xmat, 9 * nb, lane);
  lds3_to_global_org(d.xipos + (size_t)w * 3 * nb, s_xipos, 3 * nb, lane, org);
  lds_to_global(d.xipos_rel + (size_t)w * 3 * nb, s_xipos, 3 * nb, lane);
  lds_to_global(d.ximat + (size_t)w * 9 * nb, s_ximat, 9 * nb, lane);
  lds3_to_global_org(d.xanchor + (size_t)w * 3 * nj, s_xanchor, 3 * nj, lane, org);
  lds_to_global(d.xaxis + (size_t)w * 3 * nj, s_xaxis, 3 * nj, lane);

  PROF_MARK(2);
  // ---- comPos: subtree_com (a subtree is a contiguous body-id range), cinert, cdof
  // body masses staged in LDS for the range sums; per-body constants of OTHER bodies come from
  // the owning lane's register (ds_bpermute), not from memory
  if (lane < nb) s_mass[lane] = b_mass;
  __syncthreads();
  for (int it0 = 0; it0 < 3 * nb; it0 += 64) {
    const int it = it0 + lane, bq_ = it < 3 * nb ? it / 3 : 0;
    const int snum = __shfl(b_snum, bq_);
    const float sm_ = __shfl(b_stm, bq_);
    if (it >= 3 * nb) continue;
    const int b = bq_, c = it - 3 * b, e = b + snum;
    // four independent partial sums: the LDS reads of a group are in flight together
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    int j = b;
    for (; j + 3 < e; j += 4) {
      a0 += s_mass[j] * s_xipos[3 * j + c];
      a1 += s_mass[j + 1] * s_xipos[3 * j + 3 + c];
      a2 += s_mass[j + 2] * s_xipos[3 * j + 6 + c];
      a3 += s_mass[j + 3] * s_xipos[3 * j + 9 + c];
      m0 += s_mass[j]; m1 += s_mass[j + 1]; m2 += s_mass[j + 2]; m3 += s_mass[j + 3];
    }
    for (; j < e; ++j) { a0 += s_mass[j] * s_xipos[3 * j + c]; m0 += s_mass[j]; }
    float acc = (a0 + a1) + (a2 + a3);
    // mj_comPos divides the mass-weighted sum of WORLD positions by body_subtreemass.  With positions in the local frame that
    // is the same number only if body_subtreemass is the sum of the masses; a model whose body_mass was randomised per world
    // without it (randomize_field writes one field) keeps MuJoCo's value through the missing term org (sum m - subtreemass)
    const float msum = (m0 + m1) + (m2 + m3), oc = c == 0 ? org[0] : (c == 1 ? org[1] : org[2]);
    if (fabsf(msum - sm_) > 2e-5f * sm_) acc += oc * (msum - sm_);
    s_sub[it] = sm_ < MINVAL ? s_xipos[it] : acc / sm_;
  }
  __syncthreads();
  if (lane < nb) {
    const int i = lane;
    float res[10];
    if (i == 0) {
      for (int k = 0; k < 10; ++k) res[k] = 0.f;
    } else {
      float mat[9], in[3], dif[3], tmp[9];
      const float ms = b_mass;
      const int root = b_root;
      for (int k = 0; k < 9; ++k) mat[k] = s_ximat[9 * i + k];
      for (int k = 0; k < 3; ++k) { in[k] = b_inertia[k]; dif[k] = s_xipos[3 * i + k] - s_sub[3 * root + k]; }
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
    for (int k = 0; k < 10; ++k) s_cinert[10 * i + k] = res[k];
  }
  if (lane < nv) {
    const int i = lane, j = v_jnt, b = v_body, k = i - v_dofadr, type = v_type, root = v_root;
    float off[3], c6[6];
    for (int a = 0; a < 3; ++a) off[a] = s_sub[3 * root + a] - s_xanchor[3 * j + a];
    if (type == MJLAB_JNT_FREE && k < 3) {
      for (int a = 0; a < 6; ++a) c6[a] = (a == 3 + k) ? 1.f : 0.f;
    } else if (type == MJLAB_JNT_FREE) {
      float ax[3] = {s_xmat[9 * b + (k - 3)], s_xmat[9 * b + 3 + (k - 3)], s_xmat[9 * b + 6 + (k - 3)]};
      for (int a = 0; a < 3; ++a) c6[a] = ax[a];
      cross3(c6 + 3, ax, off);
    } else if (type == MJLAB_JNT_SLIDE) {
      for (int a = 0; a < 3; ++a) { c6[a] = 0.f; c6[3 + a] = s_xaxis[3 * j + a]; }
    } else {
      float ax[3] = {s_xaxis[3 * j], s_xaxis[3 * j + 1], s_xaxis[3 * j + 2]};
      for (int a = 0; a < 3; ++a) c6[a] = ax[a];
      cross3(c6 + 3, ax, off);
    }
    for (int a = 0; a < 6; ++a) s_cdof[6 * i + a] = c6[a];
  }
  __syncthreads();
  PROF_MARK(3);
  lds3_to_global_org(d.subtree_com + (size_t)w * 3 * nb, s_sub, 3 * nb, lane, org);
  lds_to_global(d.subtree_crel + (size_t)w * 3 * nb, s_sub, 3 * nb, lane);
  lds_to_global(d.cinert + (size_t)w * 10 * nb, s_cinert, 10 * nb, lane);
  lds_to_global(d.cdof + (size_t)w * 6 * nv, s_cdof, 6 * nv, lane);

  PROF_MARK(4);
  // ---- crb: composite inertia = sum of cinert over the subtree range
  for (int it0 = 0; it0 < 10 * nb; it0 += 64) {
    const int it = it0 + lane, bq_ = it < 10 * nb ? it / 10 : 0;
    const int snum = __shfl(b_snum, bq_);
    if (it >= 10 * nb) continue;
    const int b = bq_, c = it - 10 * b, e = b + snum;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int j = b;
    for (; j + 3 < e; j += 4) {
      a0 += s_cinert[10 * j + c];
      a1 += s_cinert[10 * j + 10 + c];
      a2 += s_cinert[10 * j + 20 + c];
      a3 += s_cinert[10 * j + 30 + c];
    }
    for (; j < e; ++j) a0 += s_cinert[10 * j + c];
    s_crb[it] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  if (lane < nv) {
    const int i = lane;
    float in[10], v[6], r[6];
    const int b = v_body;
    for (int k = 0; k < 10; ++k) in[k] = s_crb[10 * b + k];
    for (int k = 0; k < 6; ++k) v[k] = s_cdof[6 * i + k];
    mul_inert_vec(r, in, v);
    for (int k = 0; k < 6; ++k) s_buf[6 * i + k] = r[k];
  }
  __syncthreads();
  PROF_MARK(5);
  // M[i][j] = cdof_j . (crb_i cdof_i) for j an ancestor dof of i (or i itself), else 0:
  // lane i clears row i, then walks its ancestor chain (dof_parentid) and fills both triangles
  for (int k = lane; k < nv * ld; k += 64) s_M[k] = 0.f;
  __syncthreads();
  {
    float f6[6];
    for (int k = 0; k < 6; ++k) f6[k] = lane < nv ? s_buf[6 * lane + k] : 0.f;
    // the wave walks together (the parent of dof j sits in lane j's register): trip count = the
    // longest chain, lanes that reached the root idle
    int j = lane < nv ? lane : -1;
    while (__ballot(j >= 0) != 0ull) {
      const int jj = j >= 0 ? j : 0;
      const int next = __shfl(v_pid, jj);
      if (j >= 0) {
        float v = 0.f;
        for (int k = 0; k < 6; ++k) v += s_cdof[6 * j + k] * f6[k];
        if (j == lane) v += v_arm;
        s_M[lane * ld + j] = v;
        s_M[j * ld + lane] = v;
        j = next;
      }
    }
  }
  __syncthreads();
  PROF_MARK(6);
  // the Cholesky factor of M (mj_factorM) is produced by the solve stage, where it is used
  dense_lds_to_global(d.qM + (size_t)w * nv * nv, s_M, nv, ld, lane, false);
  PROF_MARK(7);
  PROF_FLUSH(d.profile + (size_t)w * 64 + 16);
  return false;
}

#ifdef MJLAB_MAIN_TU
__global__ __launch_bounds__(64, 4) void k_position(const Model m, const Data d, const int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int w = blockIdx.x, lane = threadIdx.x;
  if ((flags & FLAG_MASK) && !d.world_mask[w]) return;
  (void)stage_position(m, d, w, lane, flags, smem);
}
#endif  // MJLAB_MAIN_TU
