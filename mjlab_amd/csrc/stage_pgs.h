// stage_pgs.h -- the dual PGS solver (mjSOL_PGS), one kernel per stage only.
// Part of kernels.h (included there, after stage_solve.h, by every translation unit of the library); not a stand-alone header.
#pragma once

// ====================================================================================
// mj_solPGS with scalar rows (pyramidal / frictionless contacts, limits, friction loss: every row is its own block).
//   AR = J M^-1 J^T + diag(R), R = 1 / D, b = J qacc_smooth - aref;  per sweep and row:  f_r -= res_r / AR_rr with
//   res_r = b_r + (AR f)_r, projected on f >= 0 (inequality rows) or |f| <= frictionloss; a row update that would raise the dual
//   cost by more than 1e-10 is undone; the sweeps end when the cost improvement of one, scaled by 1 / (meaninertia max(1, nv)),
//   drops below tolerance.  Warm start: the constraint update's forces at qacc_warmstart, kept if their dual cost is negative.
//   Then qfrc_constraint = J^T f, qacc = qacc_smooth + M^-1 J^T f.
// AR is never formed (njmax^2 floats per world): B_r = M^-1 J_r^T is kept per row (data.efc_B, one substitution with the factor of
// M per row) and v = M^-1 J^T f is carried along, lane i owning v_i -- a row costs two coalesced row loads, one wave reduction and
// one axpy.  Gauss-Seidel is sequential over the rows by definition, so this solver is a wave walking ~nefc dependent steps per
// sweep: it is here because MujocoCfg.solver names it (reference sim/sim.py:56), not because it suits one world per wave; the
// fused launch structures and the control kernel carry the primal solvers only (check_model).
// LDS: factor of M | 1 / D_i of the factor | per row: f, b, 1 / AR_rr, R.
// ====================================================================================
__host__ __device__ inline int pgs_lds_floats(const mjlab_sizes_t& s) {
  const int nvp = solve_nvp(s.nv), ld = (nvp % 8 == 4) ? nvp : nvp + 4;
  return nvp * ld + nvp + 4 * s.njmax;
}

template <int NVP>
__device__ __forceinline__ void stage_solve_pgs(const Model& m, const Data& d, const int w, const int lane, float* smem) {
  constexpr int ld = CholCfg<NVP>::LD;
  const int nv = m.size.nv, njm = m.size.njmax;
  float* s_H = smem;
  float* s_invd = s_H + NVP * ld;
  float* s_f = s_invd + NVP;
  float* s_b = s_f + njm;
  float* s_ari = s_b + njm;
  float* s_R = s_ari + njm;
  const bool own = lane < nv;
  const size_t wv = (size_t)w * nv + lane, wr = (size_t)w * njm;
  const float* J = d.efc_J + wr * nv;
  float* B = d.efc_B + wr * nv;
  const int nefc = d.nefc[w], nf = (m.opt.flags & MJLAB_OPT_FRICTIONLOSS) ? d.nf[w] : 0;
  const float qs = own ? d.qfrc_smooth[wv] : 0.f;
  // mj_factorM + qacc_smooth = M^-1 qfrc_smooth
  dense_global_to_lds(s_H, d.qM + (size_t)w * nv * nv, nv, ld, lane, true);
  chol_pad_rows<NVP>(s_H, nv, lane);
  chol_pad_diag<NVP>(s_H, nv, lane);
  __syncthreads();
  chol_factor<NVP>(s_H, s_invd, nv, lane);
  __syncthreads();
  const float qas = chol_solve<NVP>(s_H, s_invd, lane, qs);
  if (own) d.qacc_smooth[wv] = qas;
  const bool ws_at_advance = (m.opt.flags & MJLAB_OPT_WARMSTART_AT_ADVANCE) != 0;
  if (nefc == 0) {
    if (own) {
      d.qacc[wv] = qas;
      d.qfrc_constraint[wv] = 0.f;
      if (!ws_at_advance) d.qacc_warmstart[wv] = qas;
    }
    if (lane == 0) d.solver_niter[w] = 0;
    return;
  }
  const float ws = own ? d.qacc_warmstart[wv] : 0.f;
  float v = 0.f;
  // ---- per row: B_r = M^-1 J_r^T, 1 / AR_rr, b_r, and the warm-start force
  for (int r = 0; r < nefc; ++r) {
    const float jr = own ? J[(size_t)r * nv + lane] : 0.f;
    const float br = chol_solve<NVP>(s_H, s_invd, lane, jr);
    if (own) B[(size_t)r * nv + lane] = br;
    const float Dr = d.efc_D[wr + r], R = 1.f / Dr, ar = d.efc_aref[wr + r];
    const float arr = R + wave_sum(jr * br), bb = wave_sum(jr * qas) - ar, x = wave_sum(jr * ws) - ar;
    float f;
    if (r < nf) {
      const float fl = d.efc_frictionloss[wr + r], rf = fl / Dr;
      f = x <= -rf ? fl : (x >= rf ? -fl : -Dr * x);
    } else {
      f = x < 0.f ? -Dr * x : 0.f;
    }
    s_f[r] = f; s_b[r] = bb; s_ari[r] = 1.f / arr; s_R[r] = R;  // wave-uniform values: every lane stores the same word
    v += f * (own ? br : 0.f);
  }
  __syncthreads();  // the rows of B are read back below
  // ---- dual cost of the warm start: 0.5 f' AR f + f' b; the cost of zero force is 0
  float cost = 0.f;
  for (int r = 0; r < nefc; ++r) {
    const float jr = own ? J[(size_t)r * nv + lane] : 0.f;
    const float jv = wave_sum(jr * v), f = s_f[r];
    cost += f * (0.5f * (jv + f * s_R[r]) + s_b[r]);
  }
  if (cost > 0.f) {
    for (int r = lane; r < nefc; r += 64) s_f[r] = 0.f;
    v = 0.f;
    __syncthreads();
  }
  // ---- sweeps
  const float nvf = (float)(nv > 1 ? nv : 1), scale = 1.f / ((float)m.opt.meaninertia * nvf), tol = (float)m.opt.tolerance;
  const int maxiter = m.opt.iterations;
  int iter = 0;
  while (iter < maxiter) {
    float improvement = 0.f;
    for (int r = 0; r < nefc; ++r) {
      const float jr = own ? J[(size_t)r * nv + lane] : 0.f, br = own ? B[(size_t)r * nv + lane] : 0.f;
      const float old = s_f[r], ari = s_ari[r];
      const float res = s_b[r] + old * s_R[r] + wave_sum(jr * v);
      float f = old - res * ari;
      if (r < nf) {
        const float fl = d.efc_frictionloss[wr + r];
        f = f < -fl ? -fl : (f > fl ? fl : f);
      } else if (f < 0.f) {
        f = 0.f;
      }
      float delta = f - old, change = 0.5f * delta * delta / ari + delta * res;
      if (change > 1e-10f) { f = old; delta = 0.f; change = 0.f; }
      v += delta * br;
      s_f[r] = f;
      improvement -= change;
    }
    ++iter;
    if (scale * improvement < tol) break;
  }
  // ---- dual -> primal
  float fc = 0.f;
  for (int r = 0; r < nefc; ++r) {
    const float f = s_f[r];
    if (f != 0.f) fc += (own ? J[(size_t)r * nv + lane] : 0.f) * f;
  }
  for (int r = lane; r < nefc; r += 64) d.efc_force[wr + r] = s_f[r];
  if (own) {
    const float qacc = qas + v;
    d.qacc[wv] = qacc;
    d.qfrc_constraint[wv] = fc;
    if (!ws_at_advance) d.qacc_warmstart[wv] = qacc;
  }
  if (lane == 0) d.solver_niter[w] = iter;
}
