// stage_cone.h -- the Newton solver with ELLIPTIC friction cones (mjlab_option_t.cone = MJLAB_CONE_ELLIPTIC): kernels of their own.
// Part of kernels.h (included there, after stage_pgs.h, by every translation unit of the library); not a stand-alone header.
#pragma once

// ====================================================================================
// mj_solPrimal (Newton) where a condim-3 contact is ONE three-row block [normal, tangent 1, tangent 2] whose cost is the distance to the
// dual friction cone instead of three scalar rows (MujocoCfg.cone = "elliptic", reference sim/sim.py:49,52; no registered task
// configures it).  The cone model is restated from MuJoCo's documentation (mj_constraintUpdate's elliptic branch) and UNPINNED; the CPU
// restatement the parity tests compare against is itself held to Coulomb's law, the optimality conditions and cone membership:
//   x = (J qacc - aref) of the block, mu = friction[0] / sqrt(impratio), U = (mu x0, f1 x1, f2 x2), N = U0, T = |(U1, U2)|
//   top zone     N >= mu T: cost 0;      bottom zone  mu N + T <= 0: 0.5 sum_k D_k x_k^2;
//   middle zone  0.5 Dm (N - mu T)^2, Dm = D_0 / (mu^2 (1 + mu^2)), whose Hessian is the sum of two rank-one terms
//                Dm g g^T + (Dm (mu T - N) mu / T) q q^T,  g = d(N - mu T) / dx,  q = (0, -f1 U2 / T, f2 U1 / T):
//                the block enters H = M + J^T (...) J as two virtual rows g^T Jc and q^T Jc.
// Like the dual solver (stage_pgs.h) this is here because the configuration names it, as a wave-per-world kernel next to the
// optimised pyramid path, which it leaves untouched.  Rows are evaluated one per lane from LDS; what they contribute to J^T f and to
// H is written down as a list of VIRTUAL rows (base row, three coefficients, weight D, force f: a quadratic scalar row is (r; 1 0 0;
// D_r; -D_r x_r), a friction-loss row in its linear zone has D = 0, a cone gives its three rows in the bottom zone and g^T Jc, q^T Jc in
// the middle zone, nothing in the top zone), and ONE pass over that list forms J^T f and the 16 x 16 tiles of J^T D J on the matrix
// cores (v_mfma_f32_16x16x4_f32, four virtual rows per instruction, the loads of 16 virtual rows in flight together).  The factor is
// the LDS column sweep (common.h chol_factor); both line searches of the primal path (the exact one and mujoco_warp's grid).  Launched
// as k_solve_cone (one kernel per stage) or inside the cone variants of the fused kernels (kernels.h: k_substep_cone, k_control_step_cone).
//
// LDS, 10 KB per wave like the pyramid's solve (16 waves per CU: one round for 4096 worlds):
//   H / its factor | 1 / D_i of the factor | friction loss of the first nf rows | role bits (2 per row) | M packed | jar, J search, D
// for as many rows as fit (cone_lds_rows: 168 for the G1); a world with more rows runs the layout without M (all njmax rows, M read from
// global memory: the BIG instantiation, as in stage_solve.h).  A cone's three D slots hold (-D_0, friction_1, friction_2): D_1 = D_0
// impratio and D_2 = D_1 friction_2^2 / friction_1^2 follow from them (mj_makeImpedance's rule, without its MINVAL clamp), mu =
// friction_1 / sqrt(impratio).  The virtual rows are written 64 rows at a time (<= 192 per trip) into the factor's block, which is dead
// between the solve for the search direction and the next H; so are the grid search's cone costs at step 0.
// ====================================================================================
// the factor's block also holds, while the factor is dead, the virtual rows of one 64-row trip (6 arrays of <= 192) or one value per row
__host__ __device__ constexpr int cone_hblk(int nvp_ld, int njmax) { return nvp_ld > 6 * 192 ? (nvp_ld > njmax ? nvp_ld : njmax) : (6 * 192 > njmax ? 6 * 192 : njmax); }
__host__ __device__ inline int cone_lds_fixed_floats(const mjlab_sizes_t& s) {  // everything but M and the per-row arrays
  const int nvp = solve_nvp(s.nv), ld = (nvp % 8 == 4) ? nvp : nvp + 4;
  return cone_hblk(nvp * ld, s.njmax) + 2 * nvp + (s.njmax + 15) / 16;
}
__host__ __device__ inline int cone_lds_rows(const mjlab_sizes_t& s) {  // rows next to M in 10 KB; -1: no such layout (every world without M)
  const int nvp = solve_nvp(s.nv);
  int rows = (2560 - cone_lds_fixed_floats(s) - nvp * (nvp + 1) / 2) / 3;
  rows = rows > 0 ? rows & ~3 : 0;
  if (rows >= s.njmax) return s.njmax;
  return rows >= 32 ? rows : -1;
}
__host__ __device__ inline int cone_lds_floats(const mjlab_sizes_t& s) {
  const int nvp = solve_nvp(s.nv), rows = cone_lds_rows(s);
  const int all = cone_lds_fixed_floats(s) + 3 * s.njmax, with_m = rows < 0 ? 0 : cone_lds_fixed_floats(s) + nvp * (nvp + 1) / 2 + 3 * rows;
  return all > with_m ? all : with_m;
}

enum { CONE_ROLE_ROW = 0, CONE_ROLE_START = 1, CONE_ROLE_MEMBER = 2 };

struct ConeCtx {
  const float* J;
  float *s_jar, *s_jv, *s_D, *s_fl;
  const unsigned* s_role;
  int* s_vr;                                   // virtual rows: base row ...
  float *s_vc0, *s_vc1, *s_vc2, *s_vD, *s_vf;  // ... coefficients of rows r, r + 1, r + 2, weight, force
  float impratio, mu_scale;                    // (clamped) impratio, 1 / sqrt(impratio)
  int nv, nefc, nf, lane;
  float quad_gauss[3];
  int ls_iter;
};
__device__ __forceinline__ int cone_role(const ConeCtx& c, int r) { return (c.s_role[r >> 4] >> (2 * (r & 15))) & 3; }
// D_k and (mu, friction_1, friction_2) of the cone whose first row is r
__device__ __forceinline__ void cone_params(const ConeCtx& c, int r, float (&D)[3], float (&fr)[3]) {
  D[0] = -c.s_D[r]; fr[1] = c.s_D[r + 1]; fr[2] = c.s_D[r + 2];
  fr[0] = fr[1] * c.mu_scale;
  D[1] = D[0] * c.impratio;
  D[2] = D[1] * (fr[2] * fr[2]) / (fr[1] * fr[1]);
}

// one cone at residuals x: zone (0 top, 1 bottom, 2 middle), cost, force = -d cost / dx, and for the middle zone the two rank-one
// terms of the Hessian (Da g g^T + Db q q^T).  D = the three rows' efc_D, fr = (mu, friction[0], friction[1]).
__device__ __forceinline__ int cone_block(const float (&x)[3], const float (&D)[3], const float (&fr)[3], float& cost, float (&force)[3], float (&g)[3],
                                          float (&q)[3], float& Da, float& Db) {
  const float mu = fr[0];
  const float U0 = x[0] * mu, U1 = x[1] * fr[1], U2 = x[2] * fr[2];
  const float N = U0, T = sqrtf(U1 * U1 + U2 * U2);
  cost = 0.f; force[0] = force[1] = force[2] = 0.f;
  if (N >= mu * T || (T <= 0.f && N >= 0.f)) return 0;
  if (mu * N + T <= 0.f || (T <= 0.f && N < 0.f)) {
    for (int k = 0; k < 3; ++k) { cost += 0.5f * D[k] * x[k] * x[k]; force[k] = -D[k] * x[k]; }
    return 1;
  }
  const float Dm = D[0] / (mu * mu * (1.f + mu * mu)), phi = N - mu * T, it = 1.f / T;
  cost = 0.5f * Dm * phi * phi;
  g[0] = mu; g[1] = -mu * fr[1] * U1 * it; g[2] = -mu * fr[2] * U2 * it;
  for (int k = 0; k < 3; ++k) force[k] = -Dm * phi * g[k];
  q[0] = 0.f; q[1] = -fr[1] * U2 * it; q[2] = fr[2] * U1 * it;
  Da = Dm; Db = Dm * (-phi) * mu * it;
  return 2;
}
// the cone whose first row is r, at residuals xs[r .. r + 2] + alpha dx[r .. r + 2] (dx may be null)
__device__ __forceinline__ int cone_at(const ConeCtx& c, int r, const float* xs, const float* dx, float alpha, float (&D)[3], float& cost, float (&force)[3],
                                       float (&g)[3], float (&q)[3], float& Da, float& Db) {
  float fr[3], x[3];
  cone_params(c, r, D, fr);
  for (int k = 0; k < 3; ++k) x[k] = dx ? fmaf(alpha, dx[r + k], xs[r + k]) : xs[r + k];
  return cone_block(x, D, fr, cost, force, g, q, Da, Db);
}

// cost / force of the scalar row r at residual x (mj_constraintUpdate); returns true in the quadratic zone
__device__ __forceinline__ bool cone_scalar_row(const ConeCtx& c, int r, float x, float& cost, float& force) {
  const float Dr = c.s_D[r];
  if (r < c.nf) {
    const float f = c.s_fl[r], rf = f / Dr;
    if (x <= -rf) { force = f; cost = f * (-0.5f * rf - x); return false; }
    if (x >= rf) { force = -f; cost = f * (-0.5f * rf + x); return false; }
    force = -Dr * x; cost = 0.5f * Dr * x * x; return true;
  }
  if (x < 0.f) { force = -Dr * x; cost = 0.5f * Dr * x * x; return true; }
  force = 0.f; cost = 0.f; return false;
}

// sum of the rows' costs at residuals xs[] (LDS), lanes = rows: the warm start's comparison
__device__ __forceinline__ float cone_rows_cost(const ConeCtx& c, const float* xs) {
  float cost = 0.f;
  for (int r = c.lane; r < c.nefc; r += 64) {
    const int role = cone_role(c, r);
    if (role == CONE_ROLE_MEMBER) continue;
    float rc;
    if (role == CONE_ROLE_START) {
      float D[3], fo[3], g[3], q[3], Da, Db;
      cone_at(c, r, xs, nullptr, 0.f, D, rc, fo, g, q, Da, Db);
    } else {
      float fo;
      cone_scalar_row(c, r, xs[r], rc, fo);
    }
    cost += rc;
  }
  return wave_sum(cost);
}

// cost, slope and curvature along the search direction at step alpha (mj PrimalEval)
__device__ __forceinline__ void cone_ls_eval(ConeCtx& c, LsPnt* p, float alpha) {
  float cost = 0.f, d0 = 0.f, d1 = 0.f;
  for (int r = c.lane; r < c.nefc; r += 64) {
    const int role = cone_role(c, r);
    if (role == CONE_ROLE_MEMBER) continue;
    if (role == CONE_ROLE_START) {
      const float jv[3] = {c.s_jv[r], c.s_jv[r + 1], c.s_jv[r + 2]};
      float D[3], rc, fo[3], g[3], q[3], Da, Db;
      const int zone = cone_at(c, r, c.s_jar, c.s_jv, alpha, D, rc, fo, g, q, Da, Db);
      cost += rc;
      d0 -= fo[0] * jv[0] + fo[1] * jv[1] + fo[2] * jv[2];
      if (zone == 1) d1 += D[0] * jv[0] * jv[0] + D[1] * jv[1] * jv[1] + D[2] * jv[2] * jv[2];
      else if (zone == 2) {
        const float gj = g[0] * jv[0] + g[1] * jv[1] + g[2] * jv[2], qj = q[1] * jv[1] + q[2] * jv[2];
        d1 += Da * gj * gj + Db * qj * qj;
      }
      continue;
    }
    const float j0 = c.s_jar[r], jv = c.s_jv[r], Dr = c.s_D[r], x = fmaf(alpha, jv, j0);
    if (r < c.nf) {
      const float f = c.s_fl[r], rf = f / Dr;
      if (x <= -rf) { cost += f * (-0.5f * rf - j0) - alpha * f * jv; d0 -= f * jv; continue; }
      if (x >= rf) { cost += f * (-0.5f * rf + j0) + alpha * f * jv; d0 += f * jv; continue; }
    }
    if (x < 0.f || r < c.nf) {
      const float q0 = 0.5f * Dr * j0 * j0, q1 = Dr * j0 * jv, q2 = 0.5f * Dr * jv * jv;
      cost += alpha * alpha * q2 + alpha * q1 + q0;
      d0 += 2.f * alpha * q2 + q1;
      d1 += 2.f * q2;
    }
  }
  cost = wave_sum(cost) + alpha * alpha * c.quad_gauss[2] + alpha * c.quad_gauss[1] + c.quad_gauss[0];
  d0 = wave_sum(d0) + 2.f * alpha * c.quad_gauss[2] + c.quad_gauss[1];
  d1 = wave_sum(d1) + 2.f * c.quad_gauss[2];
  if (d1 <= 0.f) d1 = MINVAL;
  p->alpha = alpha; p->cost = cost; p->d0 = d0; p->d1 = d1;
  c.ls_iter++;
}

// the grid search's price of step alpha over the rows g, g + G, g + 2 G, ... (lanes = candidates x G row groups; every LDS read is a
// broadcast within a group): cost(alpha) - cost(0) formed row by row as differences (stage_solve.h line_search_parallel), or -- literal --
// the cost itself (times two, without the Gauss term).  s_c0 = the cones' costs at alpha = 0.
__device__ __forceinline__ float cone_cost_at(const ConeCtx& c, float alpha, bool literal, int g, int G, const float* s_c0) {
  float acc = 0.f;
  for (int r = g; r < c.nefc; r += G) {
    const int role = cone_role(c, r);
    if (role == CONE_ROLE_MEMBER) continue;
    if (role == CONE_ROLE_START) {
      float D[3], ca, fo[3], gg[3], q[3], Da, Db;
      cone_at(c, r, c.s_jar, c.s_jv, alpha, D, ca, fo, gg, q, Da, Db);
      acc += literal ? 2.f * ca : 2.f * (ca - s_c0[r]);
      continue;
    }
    const float j0 = c.s_jar[r], Dr = c.s_D[r], x = fmaf(alpha, c.s_jv[r], j0);
    if (r < c.nf) {
      const float fl = c.s_fl[r], rf = fl / Dr, ax = fabsf(x), a0 = fabsf(j0);
      const float ha = ax >= rf ? 2.f * fl * (ax - 0.5f * rf) : Dr * x * x;
      const float h0 = a0 >= rf ? 2.f * fl * (a0 - 0.5f * rf) : Dr * j0 * j0;
      acc += literal ? ha : ha - h0;
    } else {
      const float xm = fminf(x, 0.f), xm0 = fminf(j0, 0.f);
      acc += literal ? Dr * xm * xm : Dr * (xm - xm0) * (xm + xm0);
    }
  }
  return acc;
}

__device__ __forceinline__ int cone_update_bracket(ConeCtx& c, LsPnt* p, const LsPnt* cand, LsPnt* pnext) {
  int flag = 0;
  for (int i = 0; i < 3; ++i) {
    if (p->d0 < 0.f && cand[i].d0 < 0.f && p->d0 < cand[i].d0) { *p = cand[i]; flag = 1; }
    else if (p->d0 > 0.f && cand[i].d0 > 0.f && p->d0 > cand[i].d0) { *p = cand[i]; flag = 2; }
  }
  if (flag) cone_ls_eval(c, pnext, p->alpha - p->d0 / p->d1);
  return flag;
}

// MuJoCo's exact search (mj_solPrimal's PrimalSearch), all values wave-uniform
__device__ __forceinline__ float cone_line_search(ConeCtx& c, float gtol, float dn1, float dn2, int lsmax) {
#define CONE_LS_TOL(a_) fmaxf(gtol, dn1 + fabsf(a_) * dn2)
  LsPnt p0, p1, p2, pmid, p1next, p2next;
  cone_ls_eval(c, &p0, 0.f);
  cone_ls_eval(c, &p1, p0.alpha - p0.d0 / p0.d1);
  if (p0.cost < p1.cost) p1 = p0;
  if (fabsf(p1.d0) < CONE_LS_TOL(p1.alpha)) return p1.alpha;
  const float dir = p1.d0 < 0.f ? 1.f : -1.f;
  bool p2update = false;
  p2 = p1;
  while (p1.d0 * dir <= -CONE_LS_TOL(p1.alpha) && c.ls_iter < lsmax) {
    p2 = p1;
    p2update = true;
    cone_ls_eval(c, &p1, p1.alpha - p1.d0 / p1.d1);
    if (fabsf(p1.d0) < CONE_LS_TOL(p1.alpha)) return p1.alpha;
  }
  if (c.ls_iter >= lsmax || !p2update) return p1.alpha;
  p2next = p1;
  cone_ls_eval(c, &p1next, p1.alpha - p1.d0 / p1.d1);
  while (c.ls_iter < lsmax) {
    cone_ls_eval(c, &pmid, 0.5f * (p1.alpha + p2.alpha));
    const LsPnt cand[3] = {p1next, p2next, pmid};
    float bestcost = 0.f;
    int best = -1;
    for (int i = 0; i < 3; ++i)
      if (fabsf(cand[i].d0) < CONE_LS_TOL(cand[i].alpha) && (best == -1 || cand[i].cost < bestcost)) { bestcost = cand[i].cost; best = i; }
    if (best >= 0) return cand[best].alpha;
    const int b1 = cone_update_bracket(c, &p1, cand, &p1next);
    const int b2 = cone_update_bracket(c, &p2, cand, &p2next);
    if (!b1 && !b2) return pmid.cost < p0.cost ? pmid.alpha : 0.f;
  }
  if (p1.cost <= p2.cost && p1.cost < p0.cost) return p1.alpha;
  if (p2.cost <= p1.cost && p2.cost < p0.cost) return p2.alpha;
  return 0.f;
#undef CONE_LS_TOL
}

// Constraint update of rows r0 .. r0 + 63 at the residuals s_jar (lanes = rows): their virtual rows (header) into the list, which
// starts empty; returns this lane's row's cost (the caller sums) and the length of the list.
__device__ __forceinline__ float cone_update_trip(const ConeCtx& c, int r0, int* nvirt) {
  float cost = 0.f;
  const int r = r0 + c.lane;
  int n = 0;
  float e0[3] = {0.f, 0.f, 0.f}, e1[3] = {0.f, 0.f, 0.f}, e2[3] = {0.f, 0.f, 0.f}, eD[3] = {0.f, 0.f, 0.f}, ef[3] = {0.f, 0.f, 0.f};
  if (r < c.nefc) {
    const int role = cone_role(c, r);
    if (role == CONE_ROLE_START) {
      float D[3], fo[3], g[3], q[3], Da, Db;
      const int zone = cone_at(c, r, c.s_jar, nullptr, 0.f, D, cost, fo, g, q, Da, Db);
      if (zone == 1) {
        n = 3;
        e0[0] = 1.f; e1[1] = 1.f; e2[2] = 1.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) { eD[k] = D[k]; ef[k] = fo[k]; }
      } else if (zone == 2) {
        n = 2;
        e0[0] = g[0]; e1[0] = g[1]; e2[0] = g[2]; eD[0] = Da; ef[0] = fo[0] / g[0];  // (-Dm phi: force_0 = -Dm phi mu, g_0 = mu)
        e1[1] = q[1]; e2[1] = q[2]; eD[1] = Db;
      }
    } else if (role == CONE_ROLE_ROW) {
      float fo;
      const bool quad = cone_scalar_row(c, r, c.s_jar[r], cost, fo);
      if (fo != 0.f || quad) { n = 1; e0[0] = 1.f; eD[0] = quad ? c.s_D[r] : 0.f; ef[0] = fo; }
    }
  }
  int total;
  const int off = wave_excl_scan(n, c.lane, &total);
#pragma unroll
  for (int k = 0; k < 3; ++k)  // (static indices: the entries stay in registers)
    if (k < n) { c.s_vr[off + k] = r; c.s_vc0[off + k] = e0[k]; c.s_vc1[off + k] = e1[k]; c.s_vc2[off + k] = e2[k]; c.s_vD[off + k] = eD[k]; c.s_vf[off + k] = ef[k]; }
  *nvirt = total;
  return cost;
}

// The list's share of J^T f (per 16-column block, this lane's column of it) and of the tiles of J^T D J.  Lanes form 4 virtual rows x 16 columns.
#ifndef CONE_JU
#define CONE_JU 2  // 4-row groups per trip of the accumulation (three loads per element: 18 in flight per lane)
#endif
template <int NVP>
__device__ __forceinline__ void cone_accum(const ConeCtx& c, int nvirt, f32x4 (&acc)[CholCfg<NVP>::NB * (CholCfg<NVP>::NB + 1) / 2], float (&jtf)[CholCfg<NVP>::NB]) {
  constexpr int NB = CholCfg<NVP>::NB;
  const int sub = c.lane >> 4, col = launder(c.lane & 15);
  for (int k0 = 0; k0 < nvirt; k0 += 4 * CONE_JU) {
    float x[CONE_JU][NB], dv[CONE_JU], fv[CONE_JU];
#pragma unroll
    for (int u = 0; u < CONE_JU; ++u) {
      const int k = k0 + 4 * u + sub;
      const bool valid = k < nvirt;
      const int kk = valid ? k : 0;
      const int r = c.s_vr[kk];
      const float c0 = valid ? c.s_vc0[kk] : 0.f, c1 = valid ? c.s_vc1[kk] : 0.f, c2 = valid ? c.s_vc2[kk] : 0.f;
      dv[u] = valid ? c.s_vD[kk] : 0.f;
      fv[u] = valid ? c.s_vf[kk] : 0.f;
      // every load unconditional, from a clamped (always valid) address: per-element conditions would put each load in a branch of its
      // own and serialise the round trips.  Rows r + 1, r + 2 are fetched for scalar rows too (coefficient 0; the same cache lines mostly)
      const int ra = valid ? r : 0, rb = min(ra + 1, c.nefc - 1), rc = min(ra + 2, c.nefc - 1);
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) {
        const int cc = 16 * cb + col, ci = min(cc, c.nv - 1);
        const float j0 = c.J[(size_t)ra * c.nv + ci], j1 = c.J[(size_t)rb * c.nv + ci], j2 = c.J[(size_t)rc * c.nv + ci];
        x[u][cb] = (valid && cc < c.nv) ? c0 * j0 + c1 * j1 + c2 * j2 : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < CONE_JU; ++u) {
      float a[NB];
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) {
        jtf[cb] += x[u][cb] * fv[u];
        a[cb] = dv[u] * x[u][cb];
      }
      // (a 16-column block that is all zero in these four rows adds exact zeros: skipped, like the pyramid path's pass)
      bool nz[NB];
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) nz[cb] = __ballot(x[u][cb] != 0.f) != 0ull;
      int t = 0;
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int Jb = 0; Jb <= I; ++Jb) {
          if (nz[I] && nz[Jb]) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[I], x[u][Jb], acc[t], 0, 0, 0);
          ++t;
        }
    }
  }
}

// out[r] = sum_i J[r][i] x_i (and out2 for y) through the pyramid path's pass (stage_solve.h jac_mul: 4 rows x 16 columns per trip of the lanes)
template <int NVP, bool TWO>
__device__ __forceinline__ void cone_jac_mul(const ConeCtx& c, float x, float y, float* out, float* out2) {
  constexpr int NB = CholCfg<NVP>::NB;
  SolveCtx<NVP> sc;
  sc.J = c.J; sc.nv = c.nv; sc.nefc = c.nefc; sc.lane = c.lane;
  float x16[NB], y16[NB];
  gather16<NB>(x, x16, c.lane);
  gather16<NB>(TWO ? y : 0.f, y16, c.lane);
  jac_mul<NVP, TWO>(sc, x16, y16, out, out2);
}

// BIG: this world has more rows than fit next to M: all njmax rows in LDS instead, M from global memory
// CG: mjSOL_CG with elliptic cones (reference sim/sim.py:49-56 accepts the pair) -- mj_solPrimal's Polak-Ribiere directions preconditioned by
// M: the same constraint update, cost and line search; the matrix factored at the loop's top is M itself (the Hessian's tiles are dropped),
// so x = M^-1 grad.  An off-path combination: one instantiation (the layout without M in LDS, any row count).
template <int NVP, bool BIG, bool CG = false>
__device__ __forceinline__ void stage_solve_cone_impl(const Model& m, const Data& d, const int w, const int lane, float* smem) {
  constexpr int ld = CholCfg<NVP>::LD, NB = CholCfg<NVP>::NB, NT = NB * (NB + 1) / 2;
  const int nv = m.size.nv, njm = m.size.njmax, ncm = m.size.nconmax;
  float* s_H = smem;
  float* s_invd = s_H + cone_hblk(NVP * ld, njm);
  ConeCtx c;
  c.s_fl = s_invd + NVP;
  unsigned* s_role = (unsigned*)(c.s_fl + NVP);
  c.s_role = s_role;
  float* s_M = (float*)(s_role + (njm + 15) / 16);  // (!BIG only)
  const int nrl = BIG ? njm : cone_lds_rows(m.size);
  c.s_jar = s_M + (BIG ? 0 : NVP * (NVP + 1) / 2);
  c.s_jv = c.s_jar + nrl;
  c.s_D = c.s_jv + nrl;
  // the virtual rows of one 64-row trip live in the factor's block while the factor is dead
  c.s_vr = (int*)s_H;
  c.s_vc0 = s_H + 192; c.s_vc1 = s_H + 2 * 192; c.s_vc2 = s_H + 3 * 192; c.s_vD = s_H + 4 * 192; c.s_vf = s_H + 5 * 192;
  const bool own = lane < nv;
  const size_t wv = (size_t)w * nv + lane, wr = (size_t)w * njm;
  const float* J = d.efc_J + wr * nv;
  const float* M = d.qM + (size_t)w * nv * nv;
  const int nefc = d.nefc[w];
  c.J = J; c.nv = nv; c.nefc = nefc; c.lane = lane;
  c.nf = (m.opt.flags & MJLAB_OPT_FRICTIONLOSS) ? d.nf[w] : 0;
  const float ir = (float)m.opt.impratio;
  c.impratio = ir > MINVAL ? ir : MINVAL;
  c.mu_scale = 1.f / sqrtf(c.impratio);
  const float qs = own ? d.qfrc_smooth[wv] : 0.f;
  const bool ws_at_advance = (m.opt.flags & MJLAB_OPT_WARMSTART_AT_ADVANCE) != 0;
  PROF_INIT();
  // mj_factorM + qacc_smooth = M^-1 qfrc_smooth
  if (BIG) {
    dense_global_to_lds(s_H, M, nv, ld, lane, true);
  } else {
    glds_dense_to_packed(s_M, M, nv, lane);
    for (int k = ((nv * (nv + 1)) >> 1) + lane; k < NVP * (NVP + 1) / 2; k += 64) s_M[k] = 0.f;
    __syncthreads();
    packed_to_lds(s_H, s_M, nv, ld, lane);
  }
  chol_pad_rows<NVP>(s_H, nv, lane);
  chol_pad_diag<NVP>(s_H, nv, lane);
  __syncthreads();
  chol_factor<NVP>(s_H, s_invd, nv, lane);
  __syncthreads();
  const float qas = chol_solve<NVP>(s_H, s_invd, lane, qs);
  if (own) d.qacc_smooth[wv] = qas;
  PROF_MARK(0);
  if (nefc == 0) {
    if (own) {
      d.qacc[wv] = qas;
      d.qfrc_constraint[wv] = 0.f;
      if (!ws_at_advance) d.qacc_warmstart[wv] = qas;
    }
    if (lane == 0) d.solver_niter[w] = 0;
    return;
  }
  // ---- per row: D (a cone's slots: -D_0, friction_1, friction_2), friction loss, role bits (a row's role parked in the J search array,
  // then 16 rows packed per word by one lane: no atomics)
  for (int r = lane; r < nefc; r += 64) {
    float Dr = d.efc_D[wr + r];
    int role = CONE_ROLE_ROW;
    if (d.efc_type[wr + r] == MJLAB_EFC_CONTACT_ELLIPTIC) {
      const int cid = d.efc_id[wr + r], k = r - d.contact_efc_address[(size_t)w * ncm + cid];
      Dr = k == 0 ? -Dr : d.contact_friction[5 * ((size_t)w * ncm + cid) + k - 1];
      role = k == 0 ? CONE_ROLE_START : CONE_ROLE_MEMBER;
    }
    c.s_D[r] = Dr;
    ((int*)c.s_jv)[r] = role;
    if (r < c.nf) c.s_fl[r] = d.efc_frictionloss[wr + r];
  }
  __syncthreads();
  for (int k = lane; k < (nefc + 15) / 16; k += 64) {
    unsigned bits = 0u;
    for (int j = 0; j < 16; ++j) {
      const int r = 16 * k + j;
      if (r < nefc) bits |= (unsigned)((const int*)c.s_jv)[r] << (2 * j);
    }
    s_role[k] = bits;
  }
  __syncthreads();
  // ---- warm start: the better of qacc_warmstart and qacc_smooth (mj_fwdConstraint)
  const float ws = own ? d.qacc_warmstart[wv] : 0.f;
  cone_jac_mul<NVP, true>(c, ws, qas, c.s_jar, c.s_jv);
  __syncthreads();
  for (int r = lane; r < nefc; r += 64) { const float ar = d.efc_aref[wr + r]; c.s_jar[r] -= ar; c.s_jv[r] -= ar; }
  __syncthreads();
  auto mul_M = [&](float x) __attribute__((always_inline)) { return BIG ? symm_mul_global<NVP>(M, nv, x, lane) : symm_mul_packed<NVP>(s_M, nv, x, lane); };
  float Ma = mul_M(ws);
  const float cw = cone_rows_cost(c, c.s_jar) + wave_sum(own ? 0.5f * (Ma - qs) * (ws - qas) : 0.f);
  const float cs = cone_rows_cost(c, c.s_jv);
  float qacc = ws;
  if (cw > cs) {
    qacc = qas;
    Ma = mul_M(qas);
    __syncthreads();
    for (int r = lane; r < nefc; r += 64) c.s_jar[r] = c.s_jv[r];
  }
  __syncthreads();
  const float nvf = (float)(nv > 1 ? nv : 1), mi = (float)m.opt.meaninertia;
  const float scale = 1.f / (mi * nvf), tol = (float)m.opt.tolerance, lstol = (float)m.opt.ls_tolerance;
  const int maxiter = m.opt.iterations, lsmax = m.opt.ls_iterations;
  const float ulp4 = (m.opt.flags & MJLAB_OPT_LITERAL_TERMINATION) ? 0.f : 4.f * 5.9604645e-8f;
  PROF_MARK(1);
  // constraint update: cost, J^T f, and H = M + J^T (.) J laid out for the factor -- 64 rows at a time: their virtual rows into the
  // list (the dead factor's block), the list through the matrix cores; the tiles stay in registers until the last trip
  float cost, gauss, fc;
  auto update = [&]() {
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float jtf[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) jtf[cb] = 0.f;
    float rows = 0.f;
    if (nefc <= 64) {  // the common case spelled out: no loop around the tiles (carried across a loop they cost register copies)
      int nvirt;
      rows = cone_update_trip(c, 0, &nvirt);
      __syncthreads();
      cone_accum<NVP>(c, nvirt, acc, jtf);
      __syncthreads();
    } else {
      for (int r0 = 0; r0 < nefc; r0 += 64) {  // wave-uniform
        int nvirt;
        rows += cone_update_trip(c, r0, &nvirt);
        __syncthreads();
        cone_accum<NVP>(c, nvirt, acc, jtf);
        __syncthreads();
      }
    }
    {  // H = M + tiles -> LDS (lower triangle), identity beyond nv
      const int sub = lane >> 4, col = lane & 15;
      int t = 0;
#pragma unroll
      for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int Jb = 0; Jb <= I; ++Jb) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int row = 16 * I + sub * 4 + k, cc = 16 * Jb + col;
            if (row < nv && cc <= row) s_H[row * ld + cc] = (CG ? 0.f : acc[t][k]) + (BIG ? M[row * nv + cc] : s_M[((row * (row + 1)) >> 1) + cc]);
          }
          ++t;
        }
    }
    chol_pad_rows<NVP>(s_H, nv, lane);
    chol_pad_diag<NVP>(s_H, nv, lane);
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) { jtf[cb] += __shfl_xor(jtf[cb], 16); jtf[cb] += __shfl_xor(jtf[cb], 32); }
    fc = pick16<NB>(jtf, lane);
    gauss = wave_sum(own ? 0.5f * (Ma - qs) * (qacc - qas) : 0.f);
    cost = wave_sum(rows) + gauss;
    __syncthreads();
  };
  update();
  PROF_MARK(2);
  int iter = 0;
  float cg_search = 0.f, cg_grad = 0.f, cg_Mgrad = 0.f;  // CG: the previous direction, gradient and M^-1 gradient
  while (iter < maxiter) {
    // ---- search = -H^-1 grad (CG: -M^-1 grad + beta x the previous direction)
    const float grad = own ? Ma - qs - fc : 0.f;
    chol_factor<NVP>(s_H, s_invd, nv, lane);
    __syncthreads();
    float search = -chol_solve<NVP>(s_H, s_invd, lane, grad);
    if (CG) {  // Polak-Ribiere (mj_solPrimal): beta = grad . (Mgrad - Mgrad_old) / max(grad_old . Mgrad_old, mjMINVAL), clamped at 0
      const float Mgrad = own ? -search : 0.f;
      float beta = 0.f;
      if (iter > 0) {
        const float num = wave_sum(grad * (Mgrad - cg_Mgrad)), den = wave_sum(cg_grad * cg_Mgrad);
        beta = fmaxf(0.f, num / fmaxf(den, MINVAL));
      }
      search = beta * cg_search - Mgrad;
      cg_search = search; cg_grad = grad; cg_Mgrad = Mgrad;
    }
    PROF_MARK(4);
    // ---- line search
    const float snorm = sqrtf(wave_sum(search * search));
    if (snorm < MINVAL) break;
    const float Mv = mul_M(search);
    __syncthreads();
    cone_jac_mul<NVP, false>(c, search, 0.f, c.s_jv, nullptr);
    __syncthreads();
    c.quad_gauss[0] = gauss;
    c.quad_gauss[1] = wave_sum(search * (Ma - qs));
    c.quad_gauss[2] = wave_sum(0.5f * search * Mv);
    c.ls_iter = 0;
    PROF_MARK(5);
    float alpha;
    if (m.opt.flags & MJLAB_OPT_LS_PARALLEL) {
      // mujoco_warp's grid: ls_iterations log-spaced steps in [ls_parallel_min_step, 1], lanes = candidates x row groups, lowest cost wins,
      // the first one on ties; the cones' costs at alpha = 0 once per search (in the dead factor's block)
      const float lo = logf((float)m.opt.ls_parallel_min_step), step = (0.f - lo) / (float)(lsmax > 1 ? lsmax - 1 : 1);
      const bool literal = (m.opt.flags & MJLAB_OPT_LS_LITERAL_COST) != 0;
      float* s_c0 = s_H;
      for (int r = lane; r < nefc; r += 64)
        if (cone_role(c, r) == CONE_ROLE_START) {
          float D[3], cb, fo[3], gg[3], q[3], Da, Db;
          cone_at(c, r, c.s_jar, nullptr, 0.f, D, cb, fo, gg, q, Da, Db);
          s_c0[r] = cb;
        }
      __syncthreads();
      const int nc = lsmax < 64 ? (lsmax > 1 ? lsmax : 1) : 64;  // candidates per trip
      const int G = 1 + (2 * nc <= 64) + (3 * nc <= 64) + (4 * nc <= 64);  // row groups per candidate
      const int g = (lane >= nc) + (lane >= 2 * nc) + (lane >= 3 * nc) + (lane >= 4 * nc), cnd = lane - g * nc;
      float best_cost = 0.f;
      bool have = false;
      alpha = 0.f;
      for (int c0 = 0; c0 < lsmax; c0 += nc) {
        const int ci = c0 + cnd;
        const float a = expf(lo + (float)ci * step);
        float part = g < G ? cone_cost_at(c, a, literal, g, G, s_c0) : 0.f, acc = part;
        for (int k = 1; k < G; ++k) acc += __shfl(part, lane + k * nc);  // group 0 collects its candidate's row groups
        float cc = 0.5f * acc + a * (a * c.quad_gauss[2] + c.quad_gauss[1]);
        if (literal) cc += c.quad_gauss[0];
        if (!(g == 0 && ci < lsmax)) cc = 3.0e38f;
        const float cmin = wave_min(cc);
        const unsigned long long hit = __ballot(cc == cmin && g == 0 && ci < lsmax);
        if (hit && (!have || cmin < best_cost)) { best_cost = cmin; alpha = lane_bcast_dyn(a, (int)__builtin_ctzll(hit)); have = true; }
      }
      __syncthreads();  // (the update below writes its list over the cone costs)
    } else {
      float a1 = 0.f, a2 = 0.f;
      for (int r = lane; r < nefc; r += 64) {
        const int role = cone_role(c, r);
        float Dr = c.s_D[r];
        if (role != CONE_ROLE_ROW) {  // (a cone's D slots hold -D_0, friction_1, friction_2)
          const int rs = role == CONE_ROLE_START ? r : (cone_role(c, r - 1) == CONE_ROLE_START ? r - 1 : r - 2);
          float D[3], fr[3];
          cone_params(c, rs, D, fr);
          Dr = r == rs ? D[0] : (r == rs + 1 ? D[1] : D[2]);
        }
        const float dj = Dr * c.s_jv[r];
        a1 += fabsf(dj * c.s_jar[r]); a2 += fabsf(0.5f * dj * c.s_jv[r]);
      }
      const float dn1 = ulp4 * (wave_sum(a1) + fabsf(c.quad_gauss[1])), dn2 = 2.f * ulp4 * (wave_sum(a2) + fabsf(c.quad_gauss[2]));
      alpha = cone_line_search(c, tol * lstol * snorm * (mi * nvf), dn1, dn2, lsmax);
    }
    PROF_MARK(6);
    if (alpha == 0.f) break;
    qacc += alpha * search;
    Ma += alpha * Mv;
    __syncthreads();
    for (int r = lane; r < nefc; r += 64) c.s_jar[r] = fmaf(alpha, c.s_jv[r], c.s_jar[r]);
    __syncthreads();
    const float oldcost = cost;
    update();
    PROF_MARK(3);
    const float gnew = own ? Ma - qs - fc : 0.f;
    const float tn = own ? fabsf(Ma) + fabsf(qs) + fabsf(fc) : 0.f;
    const float improvement = scale * (oldcost - cost), gradient = scale * sqrtf(wave_sum(gnew * gnew));
    const float noise = ulp4 * scale * sqrtf(wave_sum(tn * tn));
    ++iter;
    PROF_MARK(7);
    PROF_COUNT(8);
    if (improvement < tol || gradient < tol || gradient < noise) break;
  }
  // ---- publish (the forces from the residuals once more: no array of them is kept)
  __syncthreads();
  for (int r = lane; r < nefc; r += 64) {
    const int role = cone_role(c, r);
    if (role == CONE_ROLE_MEMBER) continue;
    if (role == CONE_ROLE_START) {
      float D[3], rc, fo[3], g[3], q[3], Da, Db;
      cone_at(c, r, c.s_jar, nullptr, 0.f, D, rc, fo, g, q, Da, Db);
      for (int k = 0; k < 3; ++k) d.efc_force[wr + r + k] = fo[k];
    } else {
      float rc, fo;
      cone_scalar_row(c, r, c.s_jar[r], rc, fo);
      d.efc_force[wr + r] = fo;
    }
  }
  if (own) {
    d.qacc[wv] = qacc;
    d.qfrc_constraint[wv] = fc;
    if (!ws_at_advance) d.qacc_warmstart[wv] = qacc;
  }
  if (lane == 0) d.solver_niter[w] = iter;
  PROF_FLUSH(d.profile + (size_t)w * 64 + 48);  // (slots 48..63: the constraint stage's block uses its first four only)
}

template <int NVP>
__device__ __forceinline__ void stage_solve_cone(const Model& m, const Data& d, const int w, const int lane, float* smem) {
  const int rows_with_m = cone_lds_rows(m.size);  // wave-uniform; the common instantiation is the one with M in LDS
  if (m.opt.solver == MJLAB_SOL_CG) stage_solve_cone_impl<NVP, true, true>(m, d, w, lane, smem);
  else if (rows_with_m < 0 || d.nefc[w] > rows_with_m) stage_solve_cone_impl<NVP, true>(m, d, w, lane, smem);
  else stage_solve_cone_impl<NVP, false>(m, d, w, lane, smem);
}

// the solve of a world with elliptic cones; the integrator follows as k_solve_integrate<NVP> with the solve switched off (like the dual
// solver).  A kernel of its own so that the pyramid path's kernels carry none of its registers or scratch.  Two waves per SIMD: nothing
// spilled.  (At four -- 128 VGPRs, ~130 spilled -- this kernel ran 7 % faster and passed every test; the FUSED cone kernels with spills
// faulted when instantiations of different sizes ran back to back, cause not found: none of the cone kernels is built with spills.)
template <int NVP>
__global__ __launch_bounds__(64, 2) void k_solve_cone(const Model m, const Data d, const int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int w = blockIdx.x, lane = threadIdx.x;
  if ((flags & FLAG_MASK) && !d.world_mask[w]) return;
  stage_solve_cone<NVP>(m, d, w, lane, smem);
}
