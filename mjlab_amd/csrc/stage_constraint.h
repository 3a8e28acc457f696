// stage_constraint.h -- stage 4: friction loss + limits + contacts -> efc rows, contact sensors.
// Part of kernels.h (included there, in this order, by every translation unit of the library); not a
// stand-alone header.
#pragma once

// ====================================================================================
// Stage 4: constraints (mj_makeConstraint: dof friction loss, joint limits, contacts; contact sensors)
// ====================================================================================
__device__ __forceinline__ float impedance(const float* solimp, float pos, float margin) {
  const float dmin = clipf(solimp[0], MINIMP, MAXIMP), dmax = clipf(solimp[1], MINIMP, MAXIMP);
  const float width = fmaxf(solimp[2], MINVAL);
  const float mid = clipf(solimp[3], MINIMP, MAXIMP), power = fmaxf(solimp[4], 1.f);
  float x = fabsf((pos - margin) / width);
  float y;
  if (x >= 1.f) y = 1.f;
  else if (x == 0.f) y = 0.f;
  else if (x <= mid) y = (power == 2.f) ? x * x / mid : powf(x, power) / powf(mid, power - 1.f);
  else {
    const float omx = 1.f - x, omm = 1.f - mid;
    y = 1.f - ((power == 2.f) ? omx * omx / omm : powf(omx, power) / powf(omm, power - 1.f));
  }
  if (solimp[2] <= MINVAL) y = 0.5f;  // mj getimpedance's flat case (width <= mjMINVAL): the mean of dmin and dmax (the one divergence read in the restatement)
  return dmin + y * (dmax - dmin);
}
// reference acceleration and regulariser of one row
__device__ __forceinline__ void row_params(float timestep, const float* solref, const float* solimp, float pos, float margin,
                                           float vel, float diag_approx, float* aref, float* R) {
  const float imp = impedance(solimp, pos, margin);
  const float dmax = clipf(solimp[1], MINIMP, MAXIMP);
  float k, b;
  if (solref[0] > 0.f) {
    const float tc = fmaxf(solref[0], 2.f * timestep), dr = solref[1];
    k = 1.f / fmaxf(dmax * dmax * tc * tc * dr * dr, MINVAL);
    b = 2.f / fmaxf(dmax * tc, MINVAL);
  } else {
    k = -solref[0] / fmaxf(dmax * dmax, MINVAL);
    b = -solref[1] / fmaxf(dmax, MINVAL);
  }
  *R = fmaxf((1.f - imp) / imp * diag_approx, MINVAL);
  *aref = -b * vel - k * imp * (pos - margin);
}

// LDS: contact -> efc address (all contacts, for the sensors), limit rows, and one chunk of 64
// staged contacts as structure-of-arrays (CC_* rows of 64).
enum {
  CC_OFF1 = 0,    // 3: contact point relative to subtree_com[root of body 1]
  CC_OFF2 = 3,    // 3: same for body 2
  CC_FRAME = 6,   // 9: contact frame (rows: normal, tangent 1, tangent 2)
  CC_MASK = 15,   // 4: ancestor-dof bitmasks (lo1, hi1, lo2, hi2), int bits
  CC_MU = 19,     // 2: friction[0], friction[1]
  CC_B = 21,      // damping coefficient of the reference acceleration
  CC_KIP = 22,    // stiffness * impedance * (dist - margin)
  CC_D = 23,      // efc_D of every row of the contact
  CC_DIST = 24,
  CC_INC = 25,
  CC_ADR = 26,    // first efc row (int bits) or -1
  CC_DIM = 27,    // condim (int bits)
  CC_NROWS = 28
};
// capacity of the unit-row list: friction-loss rows (<= nv) + limit rows (<= 2 per joint)
__host__ __device__ inline int constraint_nlim(const mjlab_sizes_t& s) { return s.nv + 2 * s.njnt < s.njmax ? s.nv + 2 * s.njnt : s.njmax; }
__host__ __device__ inline int constraint_lds_floats(const mjlab_sizes_t& s) {
  return s.nconmax + 2 * constraint_nlim(s) + CC_NROWS * 64;
}

// ELL: mjlab_option_t.cone == MJLAB_CONE_ELLIPTIC -- a condim-3 contact gives 3 rows [normal, tangent 1, tangent 2] (the contact-frame
// components of the relative acceleration) instead of the pyramid's 4 edges; only the normal row has a position and a margin, the friction
// rows are velocity rows (aref = -b v); R_0 from the impedance and the translational weight, R_k = R_0 / impratio * friction[0]^2 /
// friction[k-1]^2 (mj_instantiateContact / mj_makeImpedance).  A separate instantiation reached through k_constraint_cone and the cone
// variants of the fused kernels (kernels.h); the pyramid's kernels instantiate the default.
template <bool ELL = false>
__device__ __forceinline__ void stage_constraint(const Model& m, const Data& d, const int w, const int lane, const int flags, float* smem) {
  const int nb = m.size.nbody, nv = m.size.nv, nq = m.size.nq, nj = m.size.njnt, ncm = m.size.nconmax, njm = m.size.njmax;
  const int nlim = constraint_nlim(m.size);
  PROF_INIT();
  int* s_cadr = (int*)smem;                  // contact -> first efc row (or -1)
  int* s_ldof = s_cadr + ncm;                // friction-loss / limit row -> dof
  float* s_lsign = (float*)(s_ldof + nlim);  // friction-loss / limit row -> Jacobian entry (+-1)
  float* s_cc = s_lsign + nlim;              // staged contact chunk, [CC_NROWS][64]
  const float timestep = (float)m.opt.timestep;
  float* J = d.efc_J + (size_t)w * njm * nv;
  const size_t wr = (size_t)w * njm;
  int nefc = 0;
  bool rows_dropped = false;  // wave-uniform: something did not fit njmax
  {
    const float *range = MF(jnt_range), *jmargin = MF(jnt_margin), *jsolref = MF(jnt_solref), *jsolimp = MF(jnt_solimp),
                *dinv = MF(dof_invweight0);
    const float* qpos = d.qpos + (size_t)w * nq;
    const float* qvel = d.qvel + (size_t)w * nv;
    // ---- friction loss (mj_instantiateFriction): lanes = dofs (nv <= 64); one row per dof with dof_frictionloss > 0,
    // in dof order, before every other row: J = unit vector of the dof, pos = margin = 0, solref / solimp of the dof
    if (m.opt.flags & MJLAB_OPT_FRICTIONLOSS) {  // otherwise the field is not read and data.nf stays 0
      const float fl = lane < nv ? MF(dof_frictionloss)[lane] : 0.f;
      const int act = fl > 0.f;
      int total;
      const int r = wave_excl_scan(act, lane, &total);
      if (total) {  // wave-uniform
        if (act && r < njm) {
          float aref, R;
          row_params(timestep, MF(dof_solref) + 2 * lane, MF(dof_solimp) + 5 * lane, 0.f, 0.f, qvel[lane], dinv[lane], &aref, &R);
          s_ldof[r] = lane;
          s_lsign[r] = 1.f;
          d.efc_pos[wr + r] = 0.f;
          d.efc_margin[wr + r] = 0.f;
          d.efc_D[wr + r] = 1.f / R;
          d.efc_aref[wr + r] = aref;
          d.efc_frictionloss[wr + r] = fl;
          d.efc_type[wr + r] = MJLAB_EFC_FRICTION_DOF;
          d.efc_id[wr + r] = lane;
        }
        rows_dropped |= total > njm;
        nefc = min(total, njm);
      }
      if (lane == 0) d.nf[w] = nefc;
    }
    // ---- joint limits: lanes = joints, rows assigned in (joint, side) order
    for (int j0 = 0; j0 < nj; j0 += 64) {
      const int j = j0 + lane;
      float dist[2] = {0.f, 0.f};
      int act[2] = {0, 0};
      float mg = 0.f;
      int da = 0;
      if (j < nj && m.jnt_limited[j] && m.jnt_type[j] != MJLAB_JNT_FREE) {
        const float value = qpos[m.jnt_qposadr[j]];
        mg = jmargin[j];
        da = m.jnt_dofadr[j];
        dist[0] = value - range[2 * j];
        dist[1] = range[2 * j + 1] - value;
        act[0] = dist[0] < mg;
        act[1] = dist[1] < mg;
      }
      int total;
      int off = nefc + wave_excl_scan(act[0] + act[1], lane, &total);
      for (int side = 0; side < 2; ++side) {
        if (!act[side]) continue;
        const int r = off++;
        if (r >= njm) continue;
        const float sgn = side == 0 ? 1.f : -1.f;
        float aref, R;
        row_params(timestep, jsolref + 2 * j, jsolimp + 5 * j, dist[side], mg, sgn * qvel[da], dinv[da], &aref, &R);
        s_ldof[r] = da;
        s_lsign[r] = sgn;
        d.efc_pos[wr + r] = dist[side];
        d.efc_margin[wr + r] = mg;
        d.efc_D[wr + r] = 1.f / R;
        d.efc_aref[wr + r] = aref;
        d.efc_type[wr + r] = MJLAB_EFC_LIMIT;
        d.efc_id[wr + r] = j;
      }
      rows_dropped |= nefc + total > njm;
      nefc = min(nefc + total, njm);
    }
    __syncthreads();
    for (int r = 0; r < nefc; ++r) {
      const int dof = s_ldof[r];
      const float sg = s_lsign[r];
      for (int i = lane; i < nv; i += 64) J[(size_t)r * nv + i] = (i == dof) ? sg : 0.f;
    }
  }
  PROF_MARK(0);
  // ---- contacts.  Phase A (lanes = contacts of a chunk): fetch the contact, its bodies'
  // chain masks and offsets, and evaluate everything that is per contact (impedance,
  // regulariser, reference stiffness/damping) once, into LDS.  Phase B (lanes = dofs): one
  // contact at a time, Jacobian rows from LDS operands only.
  const int ncon = d.ncon[w];
  const float* binv = MF(body_invweight0);
  const float* sub = d.subtree_crel + (size_t)w * 3 * nb;  // local frame, like contact_prel
  const float impratio_rs = sqrtf(1.f / (float)m.opt.impratio);
  float c6[6], qv = 0.f;  // this lane's dof (nv <= 64)
  for (int k = 0; k < 6; ++k) c6[k] = lane < nv ? d.cdof[((size_t)w * nv + lane) * 6 + k] : 0.f;
  if (lane < nv) qv = d.qvel[(size_t)w * nv + lane];
  for (int c0 = 0; c0 < ncon; c0 += 64) {
    const int c = c0 + lane;
    int nrow = 0, dim = 0;
    if (c < ncon) {
      const size_t wc = (size_t)w * ncm + c;
      dim = d.contact_dim[wc];
      const float dist = d.contact_dist[wc], inc = d.contact_includemargin[wc];
      const int g1 = d.contact_geom[2 * wc], g2 = d.contact_geom[2 * wc + 1];
      float pos[3], solref[2], solimp[5];
      for (int k = 0; k < 3; ++k) pos[k] = d.contact_prel[3 * wc + k];
      for (int k = 0; k < 9; ++k) s_cc[(CC_FRAME + k) * 64 + lane] = d.contact_frame[9 * wc + k];
      const float mu0 = d.contact_friction[5 * wc], mu1 = d.contact_friction[5 * wc + 1];
      for (int k = 0; k < 2; ++k) solref[k] = d.contact_solref[2 * wc + k];
      for (int k = 0; k < 5; ++k) solimp[k] = d.contact_solimp[5 * wc + k];
      const int b1 = m.geom_bodyid[g1], b2 = m.geom_bodyid[g2];
      const int r1 = m.body_rootid[b1], r2 = m.body_rootid[b2];
      for (int k = 0; k < 3; ++k) {
        s_cc[(CC_OFF1 + k) * 64 + lane] = pos[k] - sub[3 * r1 + k];
        s_cc[(CC_OFF2 + k) * 64 + lane] = pos[k] - sub[3 * r2 + k];
      }
      int* mk = (int*)s_cc + CC_MASK * 64 + lane;
      mk[0] = m.body_dofmask[2 * b1]; mk[64] = m.body_dofmask[2 * b1 + 1];
      mk[128] = m.body_dofmask[2 * b2]; mk[192] = m.body_dofmask[2 * b2 + 1];
      const float tran = binv[2 * b1] + binv[2 * b2];
      // reference acceleration and regulariser (same expressions as row_params)
      const float imp = impedance(solimp, dist, inc);
      const float dmax = clipf(solimp[1], MINIMP, MAXIMP);
      float kk, bb;
      if (solref[0] > 0.f) {
        const float tc = fmaxf(solref[0], 2.f * timestep), dr = solref[1];
        kk = 1.f / fmaxf(dmax * dmax * tc * tc * dr * dr, MINVAL);
        bb = 2.f / fmaxf(dmax * tc, MINVAL);
      } else {
        kk = -solref[0] / fmaxf(dmax * dmax, MINVAL);
        bb = -solref[1] / fmaxf(dmax, MINVAL);
      }
      float Dc;
      if (dim == 1 || ELL) {
        Dc = 1.f / fmaxf((1.f - imp) / imp * tran, MINVAL);
      } else {
        const float Rfirst = fmaxf((1.f - imp) / imp * (tran + mu0 * mu0 * tran), MINVAL);
        const float mu0i = mu0 * impratio_rs;
        Dc = 1.f / fmaxf(2.f * mu0i * mu0i * Rfirst, MINVAL);
      }
      s_cc[CC_MU * 64 + lane] = mu0; s_cc[(CC_MU + 1) * 64 + lane] = mu1;
      s_cc[CC_B * 64 + lane] = bb;
      s_cc[CC_KIP * 64 + lane] = kk * imp * (dist - inc);
      s_cc[CC_D * 64 + lane] = Dc;
      s_cc[CC_DIST * 64 + lane] = dist;
      s_cc[CC_INC * 64 + lane] = inc;
      ((int*)s_cc)[CC_DIM * 64 + lane] = dim;
      if (dist < inc) nrow = dim == 1 ? 1 : (ELL ? dim : 2 * (dim - 1));
    }
    // efc addresses: contacts take rows in order; one that does not fit is dropped
    int total;
    int adr = nefc + wave_excl_scan(nrow, lane, &total);
    if (nefc + total > njm) {  // rare: replay the sequential rule
      rows_dropped = true;
      int run = nefc;
      for (int l = 0; l < 64; ++l) {
        const int nr = __shfl(nrow, l);
        const bool fits = nr > 0 && run + nr <= njm;
        if (lane == l) adr = fits ? run : -1;
        if (fits) run += nr;
      }
      total = run - nefc;
    } else if (nrow == 0) {
      adr = -1;
    }
    if (c < ncon) {
      s_cadr[c] = adr;
      d.contact_efc_address[(size_t)w * ncm + c] = adr;
      ((int*)s_cc)[CC_ADR * 64 + lane] = adr;
    }
    unsigned long long todo = __ballot(adr >= 0 && c < ncon);
    __syncthreads();
    PROF_MARK(1);
    while (todo) {
      const int i = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int* icc = (const int*)s_cc;
      const int adr_i = icc[CC_ADR * 64 + i], dim_i = icc[CC_DIM * 64 + i];
      const unsigned lo1 = (unsigned)icc[CC_MASK * 64 + i], hi1 = (unsigned)icc[(CC_MASK + 1) * 64 + i];
      const unsigned lo2 = (unsigned)icc[(CC_MASK + 2) * 64 + i], hi2 = (unsigned)icc[(CC_MASK + 3) * 64 + i];
      const bool in1 = lane < 32 ? ((lo1 >> lane) & 1u) : ((hi1 >> (lane - 32)) & 1u);
      const bool in2 = lane < 32 ? ((lo2 >> lane) & 1u) : ((hi2 >> (lane - 32)) & 1u);
      float frame[9], off1[3], off2[3];
      for (int k = 0; k < 9; ++k) frame[k] = s_cc[(CC_FRAME + k) * 64 + i];
      for (int k = 0; k < 3; ++k) { off1[k] = s_cc[(CC_OFF1 + k) * 64 + i]; off2[k] = s_cc[(CC_OFF2 + k) * 64 + i]; }
      float jf[3] = {0.f, 0.f, 0.f};
      if (lane < nv) {
        float jp[3];
        if (in1) {
          cross3(jp, c6, off1);
          for (int k = 0; k < 3; ++k) jp[k] += c6[3 + k];
          for (int a = 0; a < 3; ++a) jf[a] -= dot3(frame + 3 * a, jp);
        }
        if (in2) {
          cross3(jp, c6, off2);
          for (int k = 0; k < 3; ++k) jp[k] += c6[3 + k];
          for (int a = 0; a < 3; ++a) jf[a] += dot3(frame + 3 * a, jp);
        }
      }
      const float bb = s_cc[CC_B * 64 + i], kip = s_cc[CC_KIP * 64 + i], Dc = s_cc[CC_D * 64 + i];
      const float dist = s_cc[CC_DIST * 64 + i], inc = s_cc[CC_INC * 64 + i];
      const float v0 = wave_sum(jf[0] * qv);
      const int cid = c0 + i;
      if (dim_i == 1) {
        if (lane < nv) J[(size_t)adr_i * nv + lane] = jf[0];
        if (lane == 0) {
          d.efc_pos[wr + adr_i] = dist; d.efc_margin[wr + adr_i] = inc; d.efc_D[wr + adr_i] = Dc;
          d.efc_aref[wr + adr_i] = -bb * v0 - kip;
          d.efc_type[wr + adr_i] = MJLAB_EFC_CONTACT_FRICTIONLESS; d.efc_id[wr + adr_i] = cid;
        }
      } else if constexpr (ELL) {
        const float v1 = wave_sum(jf[1] * qv), v2 = wave_sum(jf[2] * qv);
        const float mu0 = s_cc[CC_MU * 64 + i], mu1 = s_cc[(CC_MU + 1) * 64 + i];
        for (int r = 0; r < 3; ++r)
          if (lane < nv) J[(size_t)(adr_i + r) * nv + lane] = jf[r];
        if (lane < 3) {  // lanes = rows of this contact for the scalar row fields
          const float ir = (float)m.opt.impratio;
          const float R1 = (1.f / Dc) / (ir > MINVAL ? ir : MINVAL);
          const float fk = lane == 2 ? mu1 : mu0;
          const size_t rr = wr + adr_i + lane;
          d.efc_pos[rr] = lane == 0 ? dist : 0.f; d.efc_margin[rr] = lane == 0 ? inc : 0.f;
          d.efc_D[rr] = lane == 0 ? Dc : 1.f / fmaxf(R1 * mu0 * mu0 / (fk * fk), MINVAL);
          d.efc_aref[rr] = lane == 0 ? -bb * v0 - kip : -bb * (lane == 1 ? v1 : v2);
          d.efc_type[rr] = MJLAB_EFC_CONTACT_ELLIPTIC; d.efc_id[rr] = cid;
        }
      } else {
        const float v1 = wave_sum(jf[1] * qv), v2 = wave_sum(jf[2] * qv);
        const float mu0 = s_cc[CC_MU * 64 + i], mu1 = s_cc[(CC_MU + 1) * 64 + i];
        const int nrow_i = 2 * (dim_i - 1);
        for (int r = 0; r < nrow_i; ++r) {
          const float mu = (r >> 1) ? mu1 : mu0, sg = (r & 1) ? -mu : mu;
          if (lane < nv) J[(size_t)(adr_i + r) * nv + lane] = jf[0] + sg * ((r >> 1) ? jf[2] : jf[1]);
        }
        if (lane < nrow_i) {  // lanes = rows of this contact for the scalar row fields
          const float mu = (lane >> 1) ? mu1 : mu0, sg = (lane & 1) ? -mu : mu;
          const float vel = v0 + sg * ((lane >> 1) ? v2 : v1);
          const size_t rr = wr + adr_i + lane;
          d.efc_pos[rr] = dist; d.efc_margin[rr] = inc; d.efc_D[rr] = Dc;
          d.efc_aref[rr] = -bb * vel - kip;
          d.efc_type[rr] = MJLAB_EFC_CONTACT_PYRAMIDAL; d.efc_id[rr] = cid;
        }
      }
    }
    nefc += total;
    __syncthreads();
    PROF_MARK(2);
  }
  if (lane == 0) {
    d.nefc[w] = nefc;
    if (rows_dropped) d.overflow[w] |= MJLAB_OVF_NJMAX;  // k_collision wrote the word earlier in this pass
  }
  __syncthreads();
  // ---- contact sensors ("found" data spec): count of matching contacts that are in efc
  const int nsens = m.size.nsensor;
  for (int k = 0; k < nsens; ++k) {
    const int ot = m.sensor_objtype[k], oi = m.sensor_objid[k], rt = m.sensor_reftype[k], ri = m.sensor_refid[k];
    int cnt = 0;
    for (int c0 = 0; c0 < ncon; c0 += 64) {
      const int c = c0 + lane;
      bool hit = false;
      if (c < ncon && s_cadr[c] >= 0) {
        const size_t wc = (size_t)w * ncm + c;
        const int g[2] = {d.contact_geom[2 * wc], d.contact_geom[2 * wc + 1]};
        bool mo[2], mr[2];
        for (int s = 0; s < 2; ++s) {
          const int b = m.geom_bodyid[g[s]];
          mo[s] = ot == MJLAB_OBJ_GEOM ? g[s] == oi : ot == MJLAB_OBJ_BODY ? b == oi : (b >= oi && b < oi + m.body_subtreenum[oi]);
          mr[s] = rt < 0 ? true : rt == MJLAB_OBJ_GEOM ? g[s] == ri : rt == MJLAB_OBJ_BODY ? b == ri : (b >= ri && b < ri + m.body_subtreenum[ri]);
        }
        hit = (mo[0] && mr[1]) || (mo[1] && mr[0]);
      }
      cnt += __popcll(__ballot(hit));
    }
    const int adr = m.sensor_adr[k], dim = m.sensor_dim[k];
    float* sd = d.sensordata + (size_t)w * m.size.nsensordata;
    for (int i = lane; i < dim; i += 64) sd[adr + i] = i == 0 ? (float)cnt : 0.f;
  }
  PROF_MARK(3);
  PROF_FLUSH(d.profile + (size_t)w * 64 + 40);
}

#ifdef MJLAB_MAIN_TU
__global__ __launch_bounds__(64, 4) void k_constraint(const Model m, const Data d, const int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int w = blockIdx.x, lane = threadIdx.x;
  if ((flags & FLAG_MASK) && !d.world_mask[w]) return;
  if ((flags & FLAG_FOLD) && d.fold_reuse[w]) return;
  stage_constraint(m, d, w, lane, flags, smem);
}
__global__ __launch_bounds__(64, 4) void k_constraint_cone(const Model m, const Data d, const int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int w = blockIdx.x, lane = threadIdx.x;
  if ((flags & FLAG_MASK) && !d.world_mask[w]) return;
  if ((flags & FLAG_FOLD) && d.fold_reuse[w]) return;
  stage_constraint<true>(m, d, w, lane, flags, smem);
}
#endif  // MJLAB_MAIN_TU
