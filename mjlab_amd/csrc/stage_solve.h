// stage_solve.h -- stages 5+6: Newton solver and integration.
// Part of kernels.h (included there, in this order, by every translation unit of the library); not a
// stand-alone header.
#pragma once

// ====================================================================================
// Stage 5+6: Newton solver (mj_fwdConstraint) and integration (mj_Euler / mj_implicit)
// ====================================================================================
struct LsPnt { float alpha, cost, d0, d1; };

// compile-time padded sizes the solve kernel is instantiated for
__host__ __device__ inline int solve_nvp(int nv) {
  const int sizes[] = {8, 16, 20, 24, 32, 36, 40, 48, 64};
  for (int i = 0; i < 9; ++i) if (nv <= sizes[i]) return sizes[i];
  return -1;
}
// Rows the three per-row arrays (jar, jv, D) have room for in LDS NEXT TO a packed copy of M.  A world with more rows than
// that (a robot lying on the ground in a heap of contacts) runs the same solver code in a second layout of the same
// LDS block (stage_solve's BIG instantiation): all njmax rows, no copy of M -- M is read from global memory wherever it
// is needed, as before round 2.  128 = what the line search caches in registers, and what leaves room for M next to H
// at 16 waves per CU (10 KB per wave) with the G1's 36 x 36 factor.
#ifndef MJLAB_RCAP
#define MJLAB_RCAP 128
#endif
// everything but the per-row arrays: H / factor, 1 / D, scratch (friction-loss arrays during the solve, 64 floats for the
// integrator after it), packed M
__host__ __device__ inline int solve_lds_fixed_floats(const mjlab_sizes_t& s) {
  const int nvp = solve_nvp(s.nv), ld = (nvp % 8 == 4) ? nvp : nvp + 4;
  const int scratch = 4 * nvp > 64 ? 4 * nvp : 64;
  return nvp * ld + nvp + scratch + nvp * (nvp + 1) / 2;
}
// the layout without a copy of M: all njmax rows (the BIG instantiation)
__host__ __device__ inline int solve_lds_all_rows_floats(const mjlab_sizes_t& s) {
  const int nvp = solve_nvp(s.nv);
  return solve_lds_fixed_floats(s) - nvp * (nvp + 1) / 2 + 3 * s.njmax;
}
// Rows of the layout with M: as many as fit next to it in 10 KB (models smaller than the G1: up to all of njmax), never
// fewer than MJLAB_RCAP.  -1: the model is so large that the copy of M would cost a wave of occupancy per CU (NVP 64:
// 28 KB against 21 KB) -- then every world runs the layout without it.
__host__ __device__ inline int solve_lds_rows(const mjlab_sizes_t& s) {
  int rows = (2528 - solve_lds_fixed_floats(s)) / 3;
  if (rows < MJLAB_RCAP) rows = MJLAB_RCAP;
  if (s.njmax < rows) rows = s.njmax;
  const int with_m = 4 * (solve_lds_fixed_floats(s) + 3 * rows), all_rows = 4 * solve_lds_all_rows_floats(s);
  const int lds = 160 * 1024, occ_m = lds / with_m < 16 ? lds / with_m : 16, occ_all = lds / all_rows < 16 ? lds / all_rows : 16;
  if (occ_m < occ_all) return -1;  // (16 waves per CU is all the 128-register kernels can have)
  return rows;
}
__host__ __device__ inline int solve_lds_floats(const mjlab_sizes_t& s) {
  const int rows = solve_lds_rows(s);
  const int all_rows = solve_lds_all_rows_floats(s);
  if (rows < 0) return all_rows;
  const int with_m = solve_lds_fixed_floats(s) + 3 * rows;
  return with_m > all_rows ? with_m : all_rows;
}

template <int NVP>
struct SolveCtx {
  static constexpr int NB = CholCfg<NVP>::NB;
  static constexpr int ld = CholCfg<NVP>::LD;
  const float* J;  // global, row-major nefc x nv
  const float* M;  // global, dense nv x nv (read once per pass; the Newton loop works on the packed copy s_M)
  float* s_M;      // LDS: M as a packed lower triangle, element (i, j), j <= i, at i (i + 1) / 2 + j
  float *s_H, *s_invd;
  float *s_jar, *s_jv, *s_D;  // per-row arrays in LDS: solve_lds_rows() rows, or all njmax in the BIG instantiation
  // friction-loss rows (the first nf rows, nf <= nv): s_fl[r] = efc_frictionloss, s_fdof[r] = the row's dof (its Jacobian is
  // that unit vector, so the row never goes through the J passes of the Hessian: its force and curvature are added to the
  // dof's entries directly), s_ff[dof] / s_fD[dof] = current force / curvature of the dof's row (0 for dofs without one)
  float *s_fl, *s_ff, *s_fD;
  int* s_fdof;
  int nv, nefc, lane;
  int nf;  // friction-loss rows: the first nf rows (nf <= nv <= 64).  Their cost is quadratic inside |jar| < f / D and
           // linear outside (mj_constraintUpdate); all other rows are quadratic where jar < 0 and free otherwise
  float quad_gauss[3];
  float dn1, dn2;  // rounding noise of the line-search derivative: d0_noise(alpha) = dn1 + |alpha| dn2
  float noise_ulps;  // 0 under MJLAB_OPT_LITERAL_TERMINATION (MuJoCo's rules only), else 1
  int ls_iter;
  // line search: quadratic coefficients of this lane's row (rows 0..63) for the current search
  // direction, so that an evaluation touches LDS only for rows >= 64
  float lj0, ljv, lq0, lq1, lq2;
  float mj0, mjv, mq0, mq1, mq2;  // same for row 64 + lane (worlds with more than 64 rows set the kernel's tail)
};

// x16[cb] = x[16 cb + (lane & 15)], gathered from the lane-owned layout
template <int NB>
__device__ __forceinline__ void gather16(float x, float (&x16)[NB], int lane) {
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) x16[cb] = __shfl(x, 16 * cb + (lane & 15));
}
template <int NB>
__device__ __forceinline__ float pick16(const float (&v)[NB], int lane) {
  // a chain of v_cndmask; the index is re-laundered per step because the optimiser otherwise turns
  // the chain into a per-lane indexed load from a scratch copy of v[] (a memory round trip in
  // the middle of every Newton iteration)
  float r = v[0];
#pragma unroll
  for (int cb = 1; cb < NB; ++cb) r = (launder(lane >> 4) == cb) ? v[cb] : r;
  return r;
}

// Rows are walked 16 at a time (4 MFMA-shaped groups of 4 rows x 16 columns): the loads of a
// 16-row block are issued together, so a pass over J exposes one memory round trip per 16 rows.
#ifndef MJLAB_JU
#define MJLAB_JU 4
#endif
constexpr int JU = MJLAB_JU;  // 4-row groups per unrolled block

// out[r] = sum_i J[r][i] x_i (+ out2 for a second vector); lanes form 4 row groups x 16 columns
template <int NVP, bool TWO>
__device__ __forceinline__ void jac_mul(const SolveCtx<NVP>& c, const float (&x16)[CholCfg<NVP>::NB], const float (&y16)[CholCfg<NVP>::NB], float* out, float* out2) {
  constexpr int NB = CholCfg<NVP>::NB;
  const int sub = c.lane >> 4, col = launder(c.lane & 15);
  for (int r0 = 0; r0 < c.nefc; r0 += 4 * JU) {
    float jv[JU][NB];
#pragma unroll
    for (int u = 0; u < JU; ++u) {
      const int r = r0 + 4 * u + sub;
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) {
        const int cc = 16 * cb + col;
        jv[u][cb] = (r < c.nefc && cc < c.nv) ? c.J[(size_t)r * c.nv + cc] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < JU; ++u) {
      const int r = r0 + 4 * u + sub;
      float acc = 0.f, acc2 = 0.f;
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) {
        acc += jv[u][cb] * x16[cb];
        if (TWO) acc2 += jv[u][cb] * y16[cb];
      }
      acc = group16_sum(acc);
      if (TWO) acc2 = group16_sum(acc2);
      if (col == 0 && r < c.nefc) { out[r] = acc; if (TWO) out2[r] = acc2; }
    }
  }
}

// One pass over J: the lane-owned constraint force qfrc_constraint_i = sum_r J[r][i] f_r and,
// if WITH_H, the tiles of J^T diag(D*active) J (lower-triangular 16x16 blocks) in `acc` via
// fp32 MFMA.  The tiles stay in registers: hessian_store() adds M and lays them out in LDS
// only when the Newton iteration actually needs a new factorization.
// Only ACTIVE rows (jar < 0) contribute to J^T f and to J^T D J, and at a typical state they are
// about a third of the rows, so the pass runs over a compacted list of active row indices
// (built with wave ballots into the LDS area of s_jv, which is dead between two line searches).
// Returns the number of active rows; the list is in increasing row order.
template <int NVP>
__device__ __forceinline__ int build_active_list(const SolveCtx<NVP>& c, int* s_act) {
  int nact = 0;
  for (int r0 = 0; r0 < c.nefc; r0 += 64) {
    const int r = r0 + c.lane;
    const bool act = r < c.nefc && r >= c.nf && c.s_jar[r] < 0.f;  // friction-loss rows: friction_rows() below
    const unsigned long long mask = __ballot(act);
    if (act) s_act[nact + __popcll(mask & ((1ull << c.lane) - 1ull))] = r;
    nact += __popcll(mask);
  }
  return nact;
}

template <int NVP, bool WITH_H>
__device__ __forceinline__ float hessian_accum(const SolveCtx<NVP>& c, f32x4 (&acc)[CholCfg<NVP>::NB * (CholCfg<NVP>::NB + 1) / 2], const int* s_act, int nact) {
  constexpr int NB = CholCfg<NVP>::NB;
  constexpr int NT = NB * (NB + 1) / 2;
  float jtf[NB];
  if (WITH_H) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) jtf[cb] = 0.f;
  const int sub = c.lane >> 4, col = launder(c.lane & 15);
  for (int k0 = 0; k0 < nact; k0 += 4 * JU) {
    float x[JU][NB], dact[JU], f[JU];
#pragma unroll
    for (int u = 0; u < JU; ++u) {
      const int k = k0 + 4 * u + sub;
      const bool valid = k < nact;
      const int r = valid ? s_act[k] : 0;
      dact[u] = 0.f; f[u] = 0.f;
      if (valid) { dact[u] = c.s_D[r]; f[u] = -dact[u] * c.s_jar[r]; }
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) {
        const int cc = 16 * cb + col;
        x[u][cb] = (valid && cc < c.nv) ? c.J[(size_t)r * c.nv + cc] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < JU; ++u) {
      float a[NB];
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) {
        jtf[cb] += x[u][cb] * f[u];
        a[cb] = dact[u] * x[u][cb];
      }
      if (WITH_H) {
        // Block sparsity of J, in its cheapest form: a row touches the dofs of its bodies' chains only
        // (a left-foot contact of the G1: 12 of 35, all in the first 16-column block), so the tiles of a 16-column block that is all
        // zero in this 4-row group are skipped -- decided by a ballot on the values just loaded, no per-row mask to store or fetch.
        // Adds exact zeros otherwise: results are bit-identical.  1.280 -> 1.260 ms per control step (profiles/r03_v9/ab_hskip.txt)
        bool nz[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) nz[cb] = __ballot(x[u][cb] != 0.f) != 0ull;
        int t = 0;
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
          for (int Jb = 0; Jb <= I; ++Jb) {
            if (nz[I] && nz[Jb])
            {
              if constexpr (chol_use_tiles(NVP))  // UPPER tiles U(Jb, I) = H[16 Jb + ..][16 I + ..] (the layout chol_factor_tiles eliminates in, common.h)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[Jb], x[u][I], acc[t], 0, 0, 0);
              else
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[I], x[u][Jb], acc[t], 0, 0, 0);
            }
            ++t;
          }
      }
    }
  }
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) { jtf[cb] += __shfl_xor(jtf[cb], 16); jtf[cb] += __shfl_xor(jtf[cb], 32); }
  return pick16<NB>(jtf, c.lane);
}

// Friction-loss rows at the current residuals: force and curvature of every row, scattered to its dof.  Returns this
// lane's (dof's) share of J^T f; the curvature is added to the diagonal of H by hessian_friction_diag().
template <int NVP>
__device__ __forceinline__ float friction_rows(const SolveCtx<NVP>& c) {
  if (c.lane < c.nf) {
    const int r = c.lane;
    const float x = c.s_jar[r], Dr = c.s_D[r], fl = c.s_fl[r];
    const bool lin = fabsf(x) >= fl / Dr;  // outside the quadratic zone: constant force, no curvature
    const int dof = c.s_fdof[r];
    c.s_ff[dof] = lin ? (x < 0.f ? fl : -fl) : -Dr * x;
    c.s_fD[dof] = lin ? 0.f : Dr;
  }
  __syncthreads();
  return c.lane < c.nv ? c.s_ff[c.lane] : 0.f;
}
template <int NVP>
__device__ __forceinline__ void hessian_friction_diag(const SolveCtx<NVP>& c) {
  __syncthreads();
  if (c.lane < c.nv) c.s_H[c.lane * c.ld + c.lane] += c.s_fD[c.lane];
}

// H = M + tiles -> LDS (lower triangle only)
template <int NVP, bool BIG>
__device__ __forceinline__ void hessian_store(const SolveCtx<NVP>& c, const f32x4 (&acc)[CholCfg<NVP>::NB * (CholCfg<NVP>::NB + 1) / 2]) {
  constexpr int NB = CholCfg<NVP>::NB;
  const int sub = c.lane >> 4, col = c.lane & 15;
  // BIG: per-lane part of the global M offset, opaque so that the 4 NT addresses are not hoisted out of the Newton loop
  int moff = sub * 4 * c.nv + col;
  if (BIG) asm volatile("" : "+v"(moff));
  int t = 0;
#pragma unroll
  for (int I = 0; I < NB; ++I)
#pragma unroll
    for (int Jb = 0; Jb <= I; ++Jb) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = 16 * I + sub * 4 + k, cc = 16 * Jb + col;
        if (row < c.nv && cc <= row)
          c.s_H[row * c.ld + cc] = acc[t][k] + (BIG ? c.M[(16 * I + k) * c.nv + 16 * Jb + moff] : c.s_M[((row * (row + 1)) >> 1) + cc]);
      }
      ++t;
    }
}

// cost of row r along the search direction: D/2 (j0 + alpha jv)^2 where that is negative
#ifndef MJLAB_LSNOISE
#define MJLAB_LSNOISE 4.f
#endif
// FL: the world has friction-loss rows (the line search is instantiated twice so that worlds without them -- every
// world of the reference's own robots -- run exactly the code they ran before those rows existed)
template <int NVP, bool FL>
__device__ __forceinline__ void ls_prepare(SolveCtx<NVP>& c) {
  const int r = c.lane;
  float j0 = 1.f, jv = 0.f, Dr = 0.f;  // lanes beyond nefc: never active
  if (r < c.nefc && (!FL || r >= c.nf)) { j0 = c.s_jar[r]; jv = c.s_jv[r]; Dr = c.s_D[r]; }  // friction-loss rows: see ls_eval
  c.lj0 = j0; c.ljv = jv;
  c.lq0 = 0.5f * Dr * j0 * j0; c.lq1 = Dr * j0 * jv; c.lq2 = 0.5f * Dr * jv * jv;
  j0 = 1.f; jv = 0.f; Dr = 0.f;
  if (r + 64 < c.nefc) { j0 = c.s_jar[r + 64]; jv = c.s_jv[r + 64]; Dr = c.s_D[r + 64]; }
  c.mj0 = j0; c.mjv = jv;
  c.mq0 = 0.5f * Dr * j0 * j0; c.mq1 = Dr * j0 * jv; c.mq2 = 0.5f * Dr * jv * jv;
  // The derivative d0(alpha) = sum_r (q1_r + 2 alpha q2_r) + Gauss terms is a sum of large terms
  // that cancel at the minimiser: in fp32 it cannot get below MJLAB_LSNOISE ulps of them, while
  // MuJoCo's gtol (tolerance * ls_tolerance * |search| * scale ~ 1e-7) asks for less.  Once
  // |d0| is inside that noise band the search has found the minimiser as well as fp32 can tell.
  float a1 = fabsf(c.lq1) + fabsf(c.mq1), a2 = fabsf(c.lq2) + fabsf(c.mq2);
  if (FL && r < c.nf) {
    const float dj = c.s_D[r] * c.s_jv[r];
    a1 += fabsf(dj * c.s_jar[r]); a2 += fabsf(0.5f * dj * c.s_jv[r]);
  }
  for (int k = r + 128; k < c.nefc; k += 64) {
    const float dj = c.s_D[k] * c.s_jv[k];
    a1 += fabsf(dj * c.s_jar[k]); a2 += fabsf(0.5f * dj * c.s_jv[k]);
  }
  c.dn1 = c.noise_ulps * MJLAB_LSNOISE * 5.9604645e-8f * (wave_sum(a1) + fabsf(c.quad_gauss[1]));
  c.dn2 = c.noise_ulps * MJLAB_LSNOISE * 5.9604645e-8f * 2.f * (wave_sum(a2) + fabsf(c.quad_gauss[2]));
}
__device__ __forceinline__ float ls_newton_step(float alpha, float d0, float d1) {
  return alpha - d0 * __builtin_amdgcn_rcpf(d1);  // 1 ulp reciprocal: alpha only has to meet ls_tolerance
}
template <int NVP, bool FL>
__device__ __forceinline__ void ls_eval(SolveCtx<NVP>& c, LsPnt* p, float alpha) {
  float cost = 0.f, d0 = 0.f, d1 = 0.f;
  if (c.lj0 + alpha * c.ljv < 0.f) {
    cost = alpha * alpha * c.lq2 + alpha * c.lq1 + c.lq0;
    d0 = 2.f * alpha * c.lq2 + c.lq1;
    d1 = 2.f * c.lq2;
  }
  if (c.nefc > 64 && c.mj0 + alpha * c.mjv < 0.f) {
    cost += alpha * alpha * c.mq2 + alpha * c.mq1 + c.mq0;
    d0 += 2.f * alpha * c.mq2 + c.mq1;
    d1 += 2.f * c.mq2;
  }
  if (FL && c.lane < c.nf) {  // friction loss (mj PrimalEval): linear outside |x| < f / D
    const int r = c.lane;
    const float j0 = c.s_jar[r], jv = c.s_jv[r], Dr = c.s_D[r], fl = c.s_fl[r], rf = fl / Dr;
    const float x = j0 + alpha * jv;
    if (x <= -rf) { cost += fl * (-0.5f * rf - j0) - alpha * fl * jv; d0 -= fl * jv; }
    else if (x >= rf) { cost += fl * (-0.5f * rf + j0) + alpha * fl * jv; d0 += fl * jv; }
    else {
      const float q0 = 0.5f * Dr * j0 * j0, q1 = Dr * j0 * jv, q2 = 0.5f * Dr * jv * jv;
      cost += alpha * alpha * q2 + alpha * q1 + q0;
      d0 += 2.f * alpha * q2 + q1;
      d1 += 2.f * q2;
    }
  }
  for (int r = c.lane + 128; r < c.nefc; r += 64) {
    const float j0 = c.s_jar[r], jv = c.s_jv[r], Dr = c.s_D[r];
    const float x = j0 + alpha * jv;
    if (x < 0.f) {
      const float q0 = 0.5f * Dr * j0 * j0, q1 = Dr * j0 * jv, q2 = 0.5f * Dr * jv * jv;
      cost += alpha * alpha * q2 + alpha * q1 + q0;
      d0 += 2.f * alpha * q2 + q1;
      d1 += 2.f * q2;
    }
  }
  cost = wave_sum(cost); d0 = wave_sum(d0); d1 = wave_sum(d1);
  cost += alpha * alpha * c.quad_gauss[2] + alpha * c.quad_gauss[1] + c.quad_gauss[0];
  d0 += 2.f * alpha * c.quad_gauss[2] + c.quad_gauss[1];
  d1 += 2.f * c.quad_gauss[2];
  if (d1 <= 0.f) d1 = MINVAL;
  p->alpha = alpha; p->cost = cost; p->d0 = d0; p->d1 = d1;
  c.ls_iter++;
}
template <int NVP, bool FL>
__device__ __forceinline__ int update_bracket(SolveCtx<NVP>& c, LsPnt* p, const LsPnt* cand, LsPnt* pnext) {
  int flag = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (p->d0 < 0.f && cand[i].d0 < 0.f && p->d0 < cand[i].d0) { *p = cand[i]; flag = 1; }
    else if (p->d0 > 0.f && cand[i].d0 > 0.f && p->d0 > cand[i].d0) { *p = cand[i]; flag = 2; }
  }
  if (flag) ls_eval<NVP, FL>(c, pnext, ls_newton_step(p->alpha, p->d0, p->d1));
  return flag;
}
// exact 1-D line search on the piecewise-quadratic cost (safeguarded Newton + bracketing)
// tolerance on the derivative at step alpha: MuJoCo's gtol, but never below what fp32 resolves
template <int NVP>
__device__ __forceinline__ float ls_tol(const SolveCtx<NVP>& c, float gtol, float alpha) {
  return fmaxf(gtol, c.dn1 + fabsf(alpha) * c.dn2);
}
template <int NVP, bool FL>
__device__ float line_search(SolveCtx<NVP>& c, float gtol, int lsmax) {
  LsPnt p0, p1, p2, pmid, p1next, p2next;
  c.ls_iter = 0;
  ls_prepare<NVP, FL>(c);
  ls_eval<NVP, FL>(c, &p0, 0.f);
  ls_eval<NVP, FL>(c, &p1, ls_newton_step(p0.alpha, p0.d0, p0.d1));
  if (p0.cost < p1.cost) p1 = p0;
  if (fabsf(p1.d0) < ls_tol(c, gtol, p1.alpha)) return p1.alpha;
  const float dir = p1.d0 < 0.f ? 1.f : -1.f;
  bool p2update = false;
  p2 = p1;
  while (p1.d0 * dir <= -ls_tol(c, gtol, p1.alpha) && c.ls_iter < lsmax) {
    p2 = p1;
    p2update = true;
    ls_eval<NVP, FL>(c, &p1, ls_newton_step(p1.alpha, p1.d0, p1.d1));
    if (fabsf(p1.d0) < ls_tol(c, gtol, p1.alpha)) return p1.alpha;
  }
  if (c.ls_iter >= lsmax) return p1.alpha;
  if (!p2update) return p1.alpha;
  p2next = p1;
  ls_eval<NVP, FL>(c, &p1next, ls_newton_step(p1.alpha, p1.d0, p1.d1));
  while (c.ls_iter < lsmax) {
    ls_eval<NVP, FL>(c, &pmid, 0.5f * (p1.alpha + p2.alpha));
    LsPnt cand[3] = {p1next, p2next, pmid};
    float bestcost = 0.f, bestalpha = 0.f;
    bool found = false;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (fabsf(cand[i].d0) < ls_tol(c, gtol, cand[i].alpha) && (!found || cand[i].cost < bestcost)) { bestcost = cand[i].cost; bestalpha = cand[i].alpha; found = true; }
    if (found) return bestalpha;
    const int b1 = update_bracket<NVP, FL>(c, &p1, cand, &p1next);
    const int b2 = update_bracket<NVP, FL>(c, &p2, cand, &p2next);
    if (!b1 && !b2) return pmid.cost < p0.cost ? pmid.alpha : 0.f;
  }
  if (p1.cost <= p2.cost && p1.cost < p0.cost) return p1.alpha;
  if (p2.cost <= p1.cost && p2.cost < p0.cost) return p2.alpha;
  return 0.f;
}

#ifndef MJLAB_LSP_U
#define MJLAB_LSP_U 2
#endif
// mujoco_warp's parallel line search (MJLAB_OPT_LS_PARALLEL, include/mjlab_fields.h): the cost at `lsmax` log-spaced step sizes in
// [min_step, 1], the lowest cost wins (the first one on ties).  LANES ARE CANDIDATES: lane c + lsmax g evaluates step size c over
// the rows g, g + G, g + 2 G, ... (G = up to 4 row groups, as many as fit in the wave), reading the per-row arrays from LDS -- all
// lanes of a group read the same address (a broadcast), so one trip over the rows prices every candidate at once: ~6 VALU + 3 LDS
// instructions per row trip instead of one wave reduction per candidate (20 of them for the reference's ls_iterations).
// Candidates are compared by cost(alpha) - cost(0): the same argmin, without the common constant (see the row loop).
// `literal` (MJLAB_OPT_LS_LITERAL_COST): the candidates' LITERAL total costs are compared instead -- every row contributes its full cost
// and the Gauss term keeps its constant (what an engine that sums the costs as MuJoCo's PrimalEval writes them would compare).
template <int NVP, bool FL>
__device__ float line_search_parallel(SolveCtx<NVP>& c, float min_step, int lsmax, float* best_diff, const bool literal) {
  const int nc = lsmax < 64 ? (lsmax > 1 ? lsmax : 1) : 64;  // candidates per pass over the rows
  const int G = 1 + (2 * nc <= 64) + (3 * nc <= 64) + (4 * nc <= 64);  // min(4, 64 / nc) without an integer division
  const float lo = logf(min_step), step = (0.f - lo) / (float)(lsmax > 1 ? lsmax - 1 : 1);
  const int g = (c.lane >= nc) + (c.lane >= 2 * nc) + (c.lane >= 3 * nc) + (c.lane >= 4 * nc), cnd = c.lane - g * nc;
  const bool valid = g < G;
  const lds_f32 *s_jar = (const lds_f32*)c.s_jar, *s_jv = (const lds_f32*)c.s_jv, *s_D = (const lds_f32*)c.s_D, *s_fl = (const lds_f32*)c.s_fl;
  float best_alpha = 0.f, best_cost = 0.f;
  bool have = false;
  for (int c0 = 0; c0 < lsmax; c0 += nc) {  // one trip for lsmax <= 64
    const int ci = c0 + cnd;
    const float alpha = expf(lo + (float)ci * step);
    float acc = 0.f;
    constexpr int U = MJLAB_LSP_U;  // row trips in flight (their LDS reads are issued together)
    for (int r0 = 0; r0 < c.nefc; r0 += U * G) {
      float j0[U], jv[U], Dr[U], fl[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = r0 + u * G + g;
        const bool on = valid && r < c.nefc;
        const int rr = on ? r : 0;
        j0[u] = s_jar[rr]; jv[u] = s_jv[rr];
        Dr[u] = s_D[rr] * (on ? 1.f : 0.f);
        if (FL) fl[u] = (on && r < c.nf) ? s_fl[r] : -1.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // cost(alpha) - cost(0) of the row, formed as a product of differences: the rows' constant terms (0.5 D jar^2, ~1e2..1e5
        // summed) are common to all candidates and would bury the candidates' differences under fp32 rounding -- near the
        // minimiser the LITERAL sum of costs cannot tell the candidates apart and the solve stalls at 1e-4
        const float x = fmaf(alpha, jv[u], j0[u]);
        const float xm = fminf(x, 0.f), xm0 = fminf(j0[u], 0.f);
        float t = literal ? Dr[u] * xm * xm : Dr[u] * (xm - xm0) * (xm + xm0);
        if (FL && fl[u] >= 0.f) {  // friction loss (mj PrimalEval): Huber cost, linear beyond |x| = f / D
          const float rf = fl[u] / Dr[u], ax = fabsf(x), a0 = fabsf(j0[u]);
          const float ha = ax >= rf ? 2.f * fl[u] * (ax - 0.5f * rf) : Dr[u] * x * x;
          const float h0 = a0 >= rf ? 2.f * fl[u] * (a0 - 0.5f * rf) : Dr[u] * j0[u] * j0[u];
          t = literal ? ha : ha - h0;
        }
        acc += t;
      }
    }
    {  // lanes of group 0 collect their candidate's row groups (from the partial sums as they stand: the shuffles wrap around)
      const float part = acc;
      for (int k = 1; k < G; ++k) acc += __shfl(part, c.lane + k * nc);
    }
    float cost = 0.5f * acc + alpha * (alpha * c.quad_gauss[2] + c.quad_gauss[1]);  // relative to the cost at alpha = 0
    if (literal) cost += c.quad_gauss[0];
    if (!(g == 0 && ci < lsmax)) cost = 3.0e38f;
    const float cmin = wave_min(cost);
    const unsigned long long hit = __ballot(cost == cmin && g == 0 && ci < lsmax);
    const int first = hit ? (int)__builtin_ctzll(hit) : 0;  // lane = candidate index within this trip: the first one on ties
    if (hit && (!have || cmin < best_cost)) { best_cost = cmin; best_alpha = lane_bcast_dyn(alpha, first); have = true; }
  }
  c.ls_iter = lsmax;
  *best_diff = best_cost;  // cost(best alpha) - cost(0): the caller's cost bookkeeping for this iteration
  return best_alpha;
}

// constraint cost sum_r s(jar_r) over rows held in LDS
template <int NVP>
__device__ __forceinline__ float constraint_cost(const SolveCtx<NVP>& c, const float* s_jar) {
  float cost = 0.f;
  for (int r = c.lane; r < c.nefc; r += 64) {
    const float x = s_jar[r], Dr = c.s_D[r];
    if (r < c.nf) {  // friction loss: Huber cost
      const float fl = c.s_fl[r], rf = fl / Dr, ax = fabsf(x);
      cost += ax >= rf ? fl * (ax - 0.5f * rf) : 0.5f * Dr * x * x;
    } else if (x < 0.f) {
      cost += 0.5f * Dr * x * x;
    }
  }
  return wave_sum(cost);
}

// H = M + J^T D J (+ friction-loss curvature) in the tiles hessian_accum filled, factored where they lie (common.h, chol_factor_tiles):
// the Hessian never exists in LDS.  The factor replaces the previous one in s_H; the caller has synchronised since the last solve.
template <int NVP, bool BIG>
__device__ __forceinline__ void newton_factor(const SolveCtx<NVP>& c, f32x4 (&htile)[CholCfg<NVP>::NB * (CholCfg<NVP>::NB + 1) / 2], int nv, int lane) {
  tiles_add_M<NVP, BIG, true>(htile, c.s_M, c.M, nv, lane);
  if (c.nf > 0) tiles_add_diag<NVP>(htile, c.s_fD, nv, lane);
  chol_factor_tiles<NVP>(htile, c.s_H, c.s_invd, lane);
  __syncthreads();
}

// The kernel is written as a small state machine around ONE factor + substitution site:
//   ST_SMOOTH     s_H = M,            rhs = qfrc_smooth          -> qacc_smooth
//   ST_NEWTON     s_H = H (if new),   rhs = gradient             -> search direction, line
//                 search, update, convergence test (repeats)
//   ST_INTEGRATE  s_H = M + h*diag,   rhs = qfrc_smooth + J^T f  -> implicit acceleration
// so the fully unrolled factorization is inlined exactly once: no call ABI, no callee-saved
// registers through scratch memory, and the register allocator sees the whole kernel.
enum { ST_SMOOTH = 0, ST_NEWTON = 1, ST_PREP_INTEGRATE = 2, ST_INTEGRATE = 3 };

// Rounding noise of the gradient M a - qfrc_smooth - J^T f as evaluated in fp32: MJLAB_GNOISE ulps
// of the terms it is the difference of.  A gradient below that cannot be reduced by another Newton
// step -- the step itself would be noise -- so the iteration stops (the fp64 restatement applies
// the same rule with its own epsilon, where it never binds before the tolerance does).
#ifndef MJLAB_GNOISE
#define MJLAB_GNOISE 4.f
#endif
__device__ __forceinline__ float grad_noise(float ulps, float scale, bool own, float Ma, float qs, float fc) {
  const float t = own ? fabsf(Ma) + fabsf(qs) + fabsf(fc) : 0.f;
  return ulps * MJLAB_GNOISE * 5.9604645e-8f * scale * sqrtf(wave_sum(t * t));
}

// Experiment switch (round 4): a floor under the improvement test.  An fp32 engine that forms the improvement as the difference of
// two cost totals (the reference's rule, taken literally) cannot see an improvement below a few ulps of the cost and ends the
// iteration there by accident; since the grid search hands over the improvement as a sum of differences (below) this kernel sees
// it exactly and keeps iterating down to `tolerance`.  MJLAB_INOISE > 0 stops at MJLAB_INOISE ulps of the cost instead.
#ifndef MJLAB_INOISE
#define MJLAB_INOISE 0.f
#endif
#define IMPROVEMENT_FLOOR fmaxf(tol, c.noise_ulps * (MJLAB_INOISE) * 5.9604645e-8f * scale * fabsf(cost))
#define LS_BY_DIFFERENCES ((m.opt.flags & (MJLAB_OPT_LS_PARALLEL | MJLAB_OPT_LS_LITERAL_COST)) == MJLAB_OPT_LS_PARALLEL)
#define IMPROVEMENT (LS_BY_DIFFERENCES ? -scale * ls_diff : scale * (oldcost - cost))
// BIG: this world has more rows than fit in LDS next to M: all njmax rows in LDS instead, M from global memory
// CG: mjSOL_CG -- no Hessian; the direction is M^-1 grad (the factor of M from ST_SMOOTH stays in LDS for the whole solve)
// combined with the previous direction by Polak-Ribiere (mj_solPrimal with flg_Newton = 0)
template <int NVP, bool BIG, bool CG>
__device__ __forceinline__ void stage_solve_impl(const Model& m, const Data& d, const int w, const int lane, const int do_solve, const int do_integrate, const int flags,
                                                 float* smem) {
  constexpr int NB = CholCfg<NVP>::NB, ld = CholCfg<NVP>::LD;
  constexpr bool TILES = chol_use_tiles(NVP);  // the matrices are factored as MFMA accumulator tiles (common.h); else: the LDS-broadcast column sweep
  const int nv = m.size.nv, nq = m.size.nq, nu = m.size.nu, nj = m.size.njnt, njm = m.size.njmax;
  SolveCtx<NVP> c;
  c.s_H = smem;
  c.s_invd = c.s_H + NVP * ld;
  const int nrl = BIG ? njm : solve_lds_rows(m.size);
  float* s_rows = c.s_invd + NVP;
  float* s_vec = s_rows + 3 * nrl;  // 64 floats of scratch for the integrator (new qvel for the position update) ...
  c.s_fl = s_vec;                   // ... which the friction-loss arrays of the solve share (dead by then)
  c.s_ff = c.s_fl + NVP;
  c.s_fD = c.s_ff + NVP;
  c.s_fdof = (int*)(c.s_fD + NVP);
  c.s_M = s_vec + (4 * NVP > 64 ? 4 * NVP : 64);
  c.s_jar = s_rows;
  c.s_jv = s_rows + nrl;
  c.s_D = s_rows + 2 * nrl;
  c.J = d.efc_J + (size_t)w * njm * nv;
  c.M = d.qM + (size_t)w * nv * nv;
  c.nv = nv; c.lane = lane;
  c.noise_ulps = (m.opt.flags & MJLAB_OPT_LITERAL_TERMINATION) ? 0.f : 1.f;
  const bool ws_at_advance = (m.opt.flags & MJLAB_OPT_WARMSTART_AT_ADVANCE) != 0;
  const size_t wv = (size_t)w * nv + lane;
  const size_t wr = (size_t)w * njm;
  const bool own = lane < nv;
  const float qs = own ? d.qfrc_smooth[wv] : 0.f;
  const float h = (float)m.opt.timestep;
  const float nvf = (float)(nv > 1 ? nv : 1), mi = (float)m.opt.meaninertia;
  const float scale = 1.f / (mi * nvf), tol = (float)m.opt.tolerance, lstol = (float)m.opt.ls_tolerance;
  const int maxiter = m.opt.iterations, lsmax = m.opt.ls_iterations;
  const int nefc = do_solve ? d.nefc[w] : 0;
  c.nefc = nefc;
  c.nf = (do_solve && (m.opt.flags & MJLAB_OPT_FRICTIONLOSS)) ? d.nf[w] : 0;
  float qacc = 0.f, fc = 0.f, qas = 0.f, Ma = 0.f, cost = 0.f, gauss = 0.f, rhs = 0.f;
  float cg_search = 0.f, cg_grad = 0.f, cg_Mgrad = 0.f;  // CG: the previous direction, gradient and M^-1 gradient
  int iter = 0, state;
  bool need_factor = true;
  // low-rank correction of the Newton factor (above): tile factorization, worlds whose rows fit the masks, no friction-loss rows
  PROF_INIT();

  if (do_solve) {
    // mj_factorM + qacc_smooth = M^-1 qfrc_smooth
    if constexpr (TILES) {
    if (!BIG) {
      glds_dense_to_packed(c.s_M, c.M, nv, lane);
      for (int k = ((nv * (nv + 1)) >> 1) + lane; k < NVP * (NVP + 1) / 2; k += 64) c.s_M[k] = 0.f;
      __syncthreads();
    }
    } else {
    if (BIG) {
      dense_global_to_lds(c.s_H, c.M, nv, ld, lane, true);
    } else {
      // every trip of the load in flight at once, straight into the packed copy (common.h: glds_dense_to_packed); the dense
      // lower triangle the factorization reads is laid out from that copy on chip
      glds_dense_to_packed(c.s_M, c.M, nv, lane);
      for (int k = ((nv * (nv + 1)) >> 1) + lane; k < NVP * (NVP + 1) / 2; k += 64) c.s_M[k] = 0.f;
      __syncthreads();
      packed_to_lds(c.s_H, c.s_M, nv, ld, lane);
    }
    chol_pad_rows<NVP>(c.s_H, nv, lane);
    chol_pad_diag<NVP>(c.s_H, nv, lane);
    }  // !TILES
    rhs = qs;
    state = ST_SMOOTH;
    PROF_MARK(0);
  } else {
    if (own) {
      const size_t wve = (size_t)w * nv + launder(lane);
      qacc = d.qacc[wve];
      fc = d.qfrc_constraint[wve];
    }
    state = ST_PREP_INTEGRATE;
  }

  for (;;) {
    bool skip_solve = false;
    if (state == ST_PREP_INTEGRATE) {
      // s_H = M + h * diag(-d qfrc_smooth / d qvel), rhs = qfrc_smooth + J^T f
      if (!do_integrate) break;
      // diagonal of -d(qfrc_smooth)/d(qvel): dof damping (+ actuator velocity gains for implicitfast)
      float diag = own ? MF(dof_damping)[launder(lane)] : 0.f;
      bool need = diag > 0.f;
      if (m.opt.integrator == MJLAB_INT_IMPLICITFAST) {
        // d(qfrc_actuator)/d(qvel) of the affine-bias actuators: lanes = actuators, scattered
        // to the owning dof through LDS (clamped actuators have zero derivative)
        need = true;
        const float *biasprm = MF(actuator_biasprm), *gear = MF(actuator_gear), *frange = MF(actuator_forcerange);
        __syncthreads();
        s_vec[lane] = 0.f;
        __syncthreads();
        for (int k = lane; k < nu; k += 64) {
          const int da = m.jnt_dofadr[m.actuator_trnid[2 * k]];
          const float f = d.actuator_force[(size_t)w * nu + k];
          if (m.actuator_forcelimited[k] && (f <= frange[2 * k] || f >= frange[2 * k + 1])) continue;
          atomicAdd(&s_vec[da], -gear[6 * k] * gear[6 * k] * biasprm[10 * k + 2]);
        }
        __syncthreads();
        diag += s_vec[lane];
      }
      state = ST_INTEGRATE;
      if (__ballot(need)) {
        __syncthreads();
        if constexpr (TILES) {
          s_vec[lane] = own ? h * diag : 0.f;  // added to the diagonal where the tiles of M are built (the factor site below)
          __syncthreads();
        } else {
          if (do_solve && !BIG) packed_to_lds(c.s_H, c.s_M, nv, ld, lane);  // M is still on chip
          else dense_global_to_lds(c.s_H, c.M, nv, ld, lane, true);
          chol_pad_rows<NVP>(c.s_H, nv, lane);
          chol_pad_diag<NVP>(c.s_H, nv, lane);
          __syncthreads();
          if (own) c.s_H[lane * ld + lane] += h * diag;
        }
        rhs = own ? qs + fc : 0.f;
        need_factor = true;
      } else {
        skip_solve = true;  // explicit Euler without damping: a = qacc
      }
    }

    float x = qacc;
    if (!skip_solve) {
      if constexpr (TILES) {
      if (need_factor) {
        // The two factorizations of M per pass (ST_SMOOTH: M, ST_INTEGRATE: M + h diag): the matrix goes from the packed copy on
        // chip (or, BIG / integrate-only passes, from global memory) straight into MFMA accumulator tiles and is taken apart
        // there (common.h, chol_factor_tiles).  The Newton Hessians are factored where hessian_accum leaves their tiles (below):
        // tiles never cross a loop boundary or a merge of paths.
        f32x4 mt[NB * (NB + 1) / 2];
        __syncthreads();
        if (do_solve && !BIG) tiles_add_M<NVP, false, false>(mt, c.s_M, c.M, nv, lane);
        else tiles_add_M<NVP, true, false>(mt, c.s_M, c.M, nv, lane);
        if (state == ST_INTEGRATE) tiles_add_diag<NVP>(mt, s_vec, nv, lane);
        PROF_MARK(0);
        chol_factor_tiles<NVP>(mt, c.s_H, c.s_invd, lane);
        __syncthreads();
        PROF_MARK(12);
        PROF_COUNT(14);
      }
      x = chol_solve_tiles<NVP>(c.s_H, c.s_invd, lane, rhs);
      PROF_MARK(13);
      PROF_COUNT(15);
      } else {
      if (need_factor) {
        __syncthreads();
        chol_factor<NVP>(c.s_H, c.s_invd, nv, lane);
        __syncthreads();
        PROF_MARK(12);
        PROF_COUNT(14);
      }
      x = chol_solve<NVP>(c.s_H, c.s_invd, lane, rhs);
      PROF_MARK(13);
      PROF_COUNT(15);
      }
    }

    if (state == ST_INTEGRATE) {
      // velocity / position update with acceleration x (mj_Euler / mj_implicit tail)
      if (own) {
        const size_t wvi = (size_t)w * nv + launder(lane);
        const float qv = d.qvel[wvi] + h * x;
        d.qvel[wvi] = qv;
        if (ws_at_advance) d.qacc_warmstart[wvi] = qacc;
        s_vec[lane] = qv;
      }
      __syncthreads();
      float* qpos = d.qpos + (size_t)w * nq;
      for (int j = lane; j < nj; j += 64) {
        const int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
        if (m.jnt_type[j] == MJLAB_JNT_FREE) {
          for (int k = 0; k < 3; ++k) qpos[qa + k] += h * s_vec[da + k];
          float ax[3] = {s_vec[da + 3], s_vec[da + 4], s_vec[da + 5]}, q[4], qr[4], qn[4];
          for (int k = 0; k < 4; ++k) q[k] = qpos[qa + 3 + k];
          const float ang = h * normalize3(ax);
          axis_angle2quat(qr, ax, ang);
          normalize4(q);
          mul_quat(qn, q, qr);
          normalize4(qn);
          for (int k = 0; k < 4; ++k) qpos[qa + 3 + k] = qn[k];
        } else {
          qpos[qa] += h * s_vec[da];
        }
      }
      if (lane == 0) d.time[w] += h;
      PROF_MARK(9);
      break;
    }

    bool finished = false;  // constraint solve finished in this pass
    if (state == ST_SMOOTH) {
      qas = x;
      __syncthreads();
      const size_t wvs = (size_t)w * nv + launder(lane);
      if (own) d.qacc_smooth[wvs] = qas;
      if (nefc == 0) {
        qacc = qas;
        finished = true;
      } else {
        for (int r = launder(lane); r < nefc; r += 64) c.s_D[r] = d.efc_D[wr + r];
        if (c.nf > 0) {
          if (lane < c.nf) { c.s_fl[lane] = d.efc_frictionloss[wr + launder(lane)]; c.s_fdof[lane] = d.efc_id[wr + launder(lane)]; }
          if (lane < NVP) { c.s_ff[lane] = 0.f; c.s_fD[lane] = 0.f; }
        }
        // ---- warmstart: better of qacc_warmstart and qacc_smooth
        const float ws = own ? d.qacc_warmstart[wvs] : 0.f;
        {
          float x16[NB], y16[NB];
          gather16<NB>(ws, x16, lane);
          gather16<NB>(qas, y16, lane);
          jac_mul<NVP, true>(c, x16, y16, c.s_jar, c.s_jv);
        }
        __syncthreads();
        for (int r = launder(lane); r < nefc; r += 64) { const float ar = d.efc_aref[wr + r]; c.s_jar[r] -= ar; c.s_jv[r] -= ar; }
        __syncthreads();
        const float Ma_ws = BIG ? symm_mul_global<NVP>(c.M, nv, ws, lane) : symm_mul_packed<NVP>(c.s_M, nv, ws, lane);
        const float cost_ws = constraint_cost<NVP>(c, c.s_jar) + wave_sum(own ? 0.5f * (Ma_ws - qs) * (ws - qas) : 0.f);
        const float cost_s = constraint_cost<NVP>(c, c.s_jv);
        if (cost_ws > cost_s) {
          qacc = qas;
          Ma = qs;  // M qacc_smooth = qfrc_smooth
          for (int r = lane; r < nefc; r += 64) c.s_jar[r] = c.s_jv[r];
          __syncthreads();
        } else {
          qacc = ws;
          Ma = Ma_ws;
        }
        PROF_MARK(2);
        // ---- initial constraint state, gradient, Hessian
        cost = constraint_cost<NVP>(c, c.s_jar);
        gauss = wave_sum(own ? 0.5f * (Ma - qs) * (qacc - qas) : 0.f);
        cost += gauss;
        {
          __syncthreads();
          int* s_act = (int*)c.s_jv;
          const int nact = build_active_list<NVP>(c, s_act);
          __syncthreads();
          f32x4 htile[NB * (NB + 1) / 2];
          fc = hessian_accum<NVP, !CG>(c, htile, s_act, nact);
          if (c.nf > 0) fc += friction_rows<NVP>(c);
          rhs = own ? Ma - qs - fc : 0.f;
          if (!CG) {
            if constexpr (TILES) {
              newton_factor<NVP, BIG>(c, htile, nv, lane);
            } else {
              hessian_store<NVP, BIG>(c, htile);
              if (c.nf > 0) hessian_friction_diag<NVP>(c);
              chol_pad_diag<NVP>(c.s_H, nv, lane);
            }
          }
        }
        PROF_MARK(3);
        need_factor = !CG && !TILES;  // CG: the factor of M is what the gradient goes through; tiles: factored above
        state = ST_NEWTON;
      }
    } else {
      // ---- ST_NEWTON: x = H^-1 grad (CG: M^-1 grad) -> line search along the direction, update, convergence test
      float search = own ? -x : 0.f;
      if (CG) {  // Polak-Ribiere; rhs is the gradient x was solved from
        const float Mgrad = own ? x : 0.f;
        float beta = 0.f;
        if (iter > 0) {
          const float num = wave_sum(rhs * (Mgrad - cg_Mgrad)), den = wave_sum(cg_grad * cg_Mgrad);
          beta = fmaxf(0.f, num / fmaxf(den, MINVAL));
        }
        search = beta * cg_search - Mgrad;
        cg_search = search; cg_grad = rhs; cg_Mgrad = Mgrad;
      }
      const float snorm = sqrtf(wave_sum(search * search));
      float alpha = 0.f, Mv = 0.f, ls_diff = 0.f;
      if (snorm >= MINVAL) {
        const float gtol = tol * lstol * snorm * mi * nvf;
        Mv = BIG ? symm_mul_global<NVP>(c.M, nv, search, lane) : symm_mul_packed<NVP>(c.s_M, nv, search, lane);
        {
          float x16[NB];
          gather16<NB>(search, x16, lane);
          __syncthreads();
          jac_mul<NVP, false>(c, x16, x16, c.s_jv, c.s_jv);
        }
        __syncthreads();
        c.quad_gauss[0] = gauss;
        c.quad_gauss[1] = wave_sum(own ? search * (Ma - qs) : 0.f);
        c.quad_gauss[2] = wave_sum(own ? 0.5f * search * Mv : 0.f);
        PROF_MARK(5);
        if (m.opt.flags & MJLAB_OPT_LS_PARALLEL)
          alpha = c.nf > 0 ? line_search_parallel<NVP, true>(c, (float)m.opt.ls_parallel_min_step, lsmax, &ls_diff, (m.opt.flags & MJLAB_OPT_LS_LITERAL_COST) != 0)
                           : line_search_parallel<NVP, false>(c, (float)m.opt.ls_parallel_min_step, lsmax, &ls_diff, (m.opt.flags & MJLAB_OPT_LS_LITERAL_COST) != 0);
        else
          alpha = c.nf > 0 ? line_search<NVP, true>(c, gtol, lsmax) : line_search<NVP, false>(c, gtol, lsmax);
        PROF_MARK(6);
#ifdef MJLAB_PROFILE
        prof_acc_[10] += (float)c.ls_iter;
        prof_acc_[11] += 1.f;
#endif
      }
      if (alpha == 0.f) {
        finished = true;  // no direction or no progress: keep the current iterate
      } else {
        qacc += alpha * search;
        Ma += alpha * Mv;
        bool changed = false;  // did any row switch between active and satisfied?
        for (int r = lane; r < nefc; r += 64) {
          const float o = c.s_jar[r], nw = o + alpha * c.s_jv[r];
          bool qo = o < 0.f, qn = nw < 0.f;  // in the quadratic zone before / after?
          if (r < c.nf) { const float rf = c.s_fl[r] / c.s_D[r]; qo = fabsf(o) < rf; qn = fabsf(nw) < rf; }
          changed |= qo != qn;
          c.s_jar[r] = nw;
        }
        const bool any_changed = __ballot(changed) != 0ull;

        __syncthreads();
        const float oldcost = cost;
        if (LS_BY_DIFFERENCES) {
          // the grid search has just priced this very step: cost(alpha) - cost(0), formed from differences.  The improvement
          // test below reads it directly (no second trip over the rows, no Gauss reduction, and no difference of two totals
          // of 1e3..1e5 that differ in the 6th digit); the Gauss term itself is not needed by this search
          cost = oldcost + ls_diff;
        } else {
          cost = constraint_cost<NVP>(c, c.s_jar);
          gauss = wave_sum(own ? 0.5f * (Ma - qs) * (qacc - qas) : 0.f);
          cost += gauss;
        }
        // One pass over J gives J^T f for the convergence test and, if the active set
        // changed, the new Hessian tiles (kept in registers).  Laying H out in LDS and its
        // factorization happen only when another iteration follows; with an unchanged active
        // set H is unchanged and the factor in LDS is reused.  (The two branches are spelled
        // out so that the 24 tile registers are live only inside the branch that needs them.)
        iter++;
        int* s_act = (int*)c.s_jv;  // J search is dead until the next line search
        const int nact = build_active_list<NVP>(c, s_act);
        __syncthreads();
        const bool refactor = !CG && any_changed;
        if (refactor) {
          f32x4 htile[NB * (NB + 1) / 2];
          fc = hessian_accum<NVP, true>(c, htile, s_act, nact);
          if (c.nf > 0) fc += friction_rows<NVP>(c);
          rhs = own ? Ma - qs - fc : 0.f;
          const float improvement = IMPROVEMENT;
          const float gradient = scale * sqrtf(wave_sum(rhs * rhs));
          finished = improvement < IMPROVEMENT_FLOOR || gradient < tol || gradient < grad_noise(c.noise_ulps, scale, own, Ma, qs, fc) || iter >= maxiter;
          if (!finished) {
            __syncthreads();
            if constexpr (TILES) {
              PROF_MARK(7);
              newton_factor<NVP, BIG>(c, htile, nv, lane);
              PROF_MARK(12);
              PROF_COUNT(14);
            } else {
              hessian_store<NVP, BIG>(c, htile);
              if (c.nf > 0) hessian_friction_diag<NVP>(c);
              chol_pad_diag<NVP>(c.s_H, nv, lane);
            }
          }
          need_factor = !TILES;
        } else {  // same active set -> same H -> the factor in LDS is still valid
          f32x4 unused[NB * (NB + 1) / 2];
          fc = hessian_accum<NVP, false>(c, unused, s_act, nact);
          if (c.nf > 0) fc += friction_rows<NVP>(c);
          rhs = own ? Ma - qs - fc : 0.f;
          const float improvement = IMPROVEMENT;
          const float gradient = scale * sqrtf(wave_sum(rhs * rhs));
          finished = improvement < IMPROVEMENT_FLOOR || gradient < tol || gradient < grad_noise(c.noise_ulps, scale, own, Ma, qs, fc) || iter >= maxiter;
          need_factor = false;
        }
        PROF_MARK(7);
      }
    }
    if (finished) {  // publish the solve, then hand over to the integrator
      if (lane == 0) d.solver_niter[w] = iter;
      for (int r = launder(lane); r < nefc; r += 64) {
        const float xr = c.s_jar[r];
        float fr = xr < 0.f ? -c.s_D[r] * xr : 0.f;
        if (r < c.nf) { const float fl = c.s_fl[r]; fr = fabsf(xr) >= fl / c.s_D[r] ? (xr < 0.f ? fl : -fl) : -c.s_D[r] * xr; }
        d.efc_force[wr + r] = fr;
      }
      if (own) {
        const size_t wvp = (size_t)w * nv + launder(lane);
        d.qacc[wvp] = qacc;
        if (!ws_at_advance) d.qacc_warmstart[wvp] = qacc;
        d.qfrc_constraint[wvp] = fc;
      }
      PROF_MARK(8);
      state = ST_PREP_INTEGRATE;
    }
  }
  if (do_integrate && lane == 0) d.fold_valid[w] = 0;  // the state moved on
  PROF_FLUSH(d.profile + (size_t)w * 64);
}

template <int NVP>
__device__ __forceinline__ void stage_solve(const Model& m, const Data& d, const int w, const int lane, const int do_solve, const int do_integrate, const int flags,
                                            float* smem) {
  // wave-uniform; the common instantiation is the one with the row arrays in LDS
  const int rows_with_m = solve_lds_rows(m.size);
  const bool big = rows_with_m < 0 || (do_solve && d.nefc[w] > rows_with_m);
  if (do_solve) wave_priority(__builtin_amdgcn_readfirstlane(d.nefc[w]), __builtin_amdgcn_readfirstlane(d.solver_niter[w]), d.sched_thr);
  if (m.opt.solver == MJLAB_SOL_CG) {
    if (big) stage_solve_impl<NVP, true, true>(m, d, w, lane, do_solve, do_integrate, flags, smem);
    else stage_solve_impl<NVP, false, true>(m, d, w, lane, do_solve, do_integrate, flags, smem);
  } else if (big) stage_solve_impl<NVP, true, false>(m, d, w, lane, do_solve, do_integrate, flags, smem);
  else stage_solve_impl<NVP, false, false>(m, d, w, lane, do_solve, do_integrate, flags, smem);
}

template <int NVP>
__device__ __forceinline__ void stage_solve_pgs(const Model& m, const Data& d, const int w, const int lane, float* smem);  // stage_pgs.h
template <int NVP>
__global__ __launch_bounds__(64, 4) void k_solve_integrate(const Model m, const Data d, const int do_solve, const int do_integrate, const int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int w = blockIdx.x, lane = threadIdx.x;
  if ((flags & FLAG_MASK) && !d.world_mask[w]) return;
  if (m.opt.solver == MJLAB_SOL_PGS) {  // the dual solver (stage_pgs.h), then the integrator of the primal path with the solve switched off
    if (do_solve) {
      stage_solve_pgs<NVP>(m, d, w, lane, smem);
      __syncthreads();
    }
    if (do_integrate) stage_solve<NVP>(m, d, w, lane, 0, 1, flags, smem);
    return;
  }
  stage_solve<NVP>(m, d, w, lane, do_solve, do_integrate, flags, smem);
}

// forward(): remember the qpos / qvel the pass was computed from (see FLAG_FOLD in k_position)
__device__ __forceinline__ void fold_snapshot(const Model& m, const Data& d, const int w, const int lane) {
  const int nq = m.size.nq, nv = m.size.nv;
  for (int i = lane; i < nq; i += 64) d.sh_qpos[(size_t)w * nq + i] = d.qpos[(size_t)w * nq + i];
  for (int i = lane; i < nv; i += 64) d.sh_qvel[(size_t)w * nv + i] = d.qvel[(size_t)w * nv + i];
  if (lane == 0) d.fold_valid[w] = 1;
}
#ifdef MJLAB_MAIN_TU
__global__ __launch_bounds__(64) void k_fold_snapshot(const Model m, const Data d, const int flags) {
  const int w = blockIdx.x, lane = threadIdx.x;
  if ((flags & FLAG_MASK) && !d.world_mask[w]) return;
  fold_snapshot(m, d, w, lane);
}
#endif  // MJLAB_MAIN_TU

