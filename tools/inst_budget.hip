// Static instruction budget of the solve stage's building blocks (no GPU needed): every block is instantiated ALONE in a
// small kernel for the G1's padded size (NVP = 36), compiled to gfx950 assembly, and tools/inst_budget.py counts the
// instructions per class.  Loops (row blocks, Newton iterations) appear once, so the figures are "per trip": per
// factorization, per 16 rows of a J pass, per line-search evaluation, ...  Multiplied by the dynamic trip counts of the phase
// profile (profiles/<tag>/phases.txt) they give the instruction budget of a solve pass.
//   hipcc -O3 -std=c++17 -ffp-contract=on --offload-arch=gfx950 -S --cuda-device-only -DMJLAB_NVP=36 tools/inst_budget.hip -o /tmp/inst_budget.s
#include "../mjlab_amd/csrc/kernels.h"

constexpr int N = MJLAB_NVP;
constexpr int NBk = CholCfg<N>::NB;

struct Args {
  float *J, *M, *out;
  int nv, nefc, nf, nact;
};
__device__ __forceinline__ SolveCtx<N> make_ctx(const Args& a, float* smem) {
  SolveCtx<N> c;
  c.s_H = smem;
  c.s_invd = c.s_H + N * CholCfg<N>::LD;
  float* rows = c.s_invd + N;
  c.s_jar = rows; c.s_jv = rows + 128; c.s_D = rows + 256;
  c.s_fl = rows + 384; c.s_ff = c.s_fl + N; c.s_fD = c.s_ff + N; c.s_fdof = (int*)(c.s_fD + N);
  c.s_M = c.s_fl + 4 * N;
  c.J = a.J; c.M = a.M; c.nv = a.nv; c.nefc = a.nefc; c.nf = a.nf; c.lane = threadIdx.x;
  c.quad_gauss[0] = a.J[0]; c.quad_gauss[1] = a.J[1]; c.quad_gauss[2] = a.J[2];
  c.noise_ulps = 1.f; c.ls_iter = 0; c.dn1 = c.dn2 = 0.f;
  c.lj0 = c.ljv = c.lq0 = c.lq1 = c.lq2 = c.mj0 = c.mjv = c.mq0 = c.mq1 = c.mq2 = 0.f;
  return c;
}
#define KB(name) extern "C" __global__ __launch_bounds__(64, 4) void name(const Args a)
#define SMEM extern __shared__ __attribute__((aligned(16))) float smem[]

KB(kb_empty) { SMEM; SolveCtx<N> c = make_ctx(a, smem); a.out[threadIdx.x] = c.quad_gauss[0] + c.s_jar[threadIdx.x]; }
KB(kb_factor) { SMEM; SolveCtx<N> c = make_ctx(a, smem); chol_factor<N>(c.s_H, c.s_invd, a.nv, threadIdx.x); a.out[threadIdx.x] = c.quad_gauss[0] + c.s_jar[threadIdx.x]; }
KB(kb_solve) { SMEM; SolveCtx<N> c = make_ctx(a, smem); a.out[threadIdx.x] = chol_solve<N>(c.s_H, c.s_invd, threadIdx.x, c.s_jar[threadIdx.x]) + c.quad_gauss[0]; }
KB(kb_symm) { SMEM; SolveCtx<N> c = make_ctx(a, smem); a.out[threadIdx.x] = symm_mul_packed<N>(c.s_M, a.nv, c.s_jar[threadIdx.x], threadIdx.x) + c.quad_gauss[0]; }
KB(kb_jacmul1) {
  SMEM; SolveCtx<N> c = make_ctx(a, smem);
  float x16[NBk]; gather16<NBk>(c.s_jar[threadIdx.x], x16, threadIdx.x);
  jac_mul<N, false>(c, x16, x16, c.s_jv, c.s_jv);
  a.out[threadIdx.x] = c.quad_gauss[0] + c.s_jar[threadIdx.x];
}
KB(kb_jacmul2) {
  SMEM; SolveCtx<N> c = make_ctx(a, smem);
  float x16[NBk], y16[NBk]; gather16<NBk>(c.s_jar[threadIdx.x], x16, threadIdx.x); gather16<NBk>(c.s_D[threadIdx.x], y16, threadIdx.x);
  jac_mul<N, true>(c, x16, y16, c.s_jv, c.s_jar);
  a.out[threadIdx.x] = c.quad_gauss[0] + c.s_jar[threadIdx.x];
}
KB(kb_gather16) { SMEM; SolveCtx<N> c = make_ctx(a, smem); float x16[NBk]; gather16<NBk>(c.s_jar[threadIdx.x], x16, threadIdx.x); a.out[threadIdx.x] = c.quad_gauss[0] + c.s_jar[threadIdx.x] + x16[0] + x16[NBk - 1]; }
KB(kb_hess_h) {
  SMEM; SolveCtx<N> c = make_ctx(a, smem);
  f32x4 t[NBk * (NBk + 1) / 2];
  float fc = hessian_accum<N, true>(c, t, (const int*)c.s_jv, a.nact);
  hessian_store<N, false>(c, t);
  a.out[threadIdx.x] = fc + c.quad_gauss[0] + c.s_jar[threadIdx.x];
}
KB(kb_hess_noh) {
  SMEM; SolveCtx<N> c = make_ctx(a, smem);
  f32x4 t[NBk * (NBk + 1) / 2];
  float fc = hessian_accum<N, false>(c, t, (const int*)c.s_jv, a.nact);
  a.out[threadIdx.x] = fc + c.quad_gauss[0] + c.s_jar[threadIdx.x];
}
KB(kb_active) { SMEM; SolveCtx<N> c = make_ctx(a, smem); int n = build_active_list<N>(c, (int*)c.s_jv); a.out[threadIdx.x] = (float)n + c.quad_gauss[0] + c.s_jar[threadIdx.x]; }
KB(kb_cost) { SMEM; SolveCtx<N> c = make_ctx(a, smem); a.out[threadIdx.x] = constraint_cost<N>(c, c.s_jar) + c.quad_gauss[0] + c.s_jar[threadIdx.x]; }
KB(kb_lsprep) { SMEM; SolveCtx<N> c = make_ctx(a, smem); ls_prepare<N, false>(c); a.out[threadIdx.x] = c.lq0 + c.lq1 + c.lq2 + c.mq0 + c.dn1 + c.dn2 + c.lj0 + c.ljv + c.mj0 + c.mjv + c.mq1 + c.mq2 + c.quad_gauss[0] + c.s_jar[threadIdx.x]; }
KB(kb_lseval) {
  SMEM; SolveCtx<N> c = make_ctx(a, smem);
  c.lj0 = c.s_jar[threadIdx.x]; c.ljv = c.s_jv[threadIdx.x]; c.lq0 = c.s_D[threadIdx.x]; c.lq1 = c.s_fl[threadIdx.x]; c.lq2 = c.s_ff[threadIdx.x];
  c.mj0 = c.s_jar[threadIdx.x + 64]; c.mjv = c.s_jv[threadIdx.x + 64]; c.mq0 = c.s_D[threadIdx.x + 64]; c.mq1 = c.s_fl[threadIdx.x + 1]; c.mq2 = c.s_ff[threadIdx.x + 1];
  LsPnt p; ls_eval<N, false>(c, &p, c.quad_gauss[1]);
  a.out[threadIdx.x] = p.cost + p.d0 + p.d1 + c.quad_gauss[0] + c.s_jar[threadIdx.x];
}
KB(kb_lspar) {  // the whole parallel (grid) line search: 20 candidates priced in one trip over the rows
  SMEM; SolveCtx<N> c = make_ctx(a, smem);
  float bd = 0.f;
  a.out[threadIdx.x] = line_search_parallel<N, false>(c, c.quad_gauss[1], a.nact, &bd) + bd + c.quad_gauss[0] + c.s_jar[threadIdx.x];
}
KB(kb_loadM) {
  SMEM; SolveCtx<N> c = make_ctx(a, smem);
  dense_global_to_lds_packed(c.s_H, c.s_M, c.M, a.nv, CholCfg<N>::LD, threadIdx.x);
  chol_pad_rows<N>(c.s_H, a.nv, threadIdx.x); chol_pad_diag<N>(c.s_H, a.nv, threadIdx.x);
  a.out[threadIdx.x] = c.quad_gauss[0] + c.s_jar[threadIdx.x];
}
KB(kb_packed2lds) { SMEM; SolveCtx<N> c = make_ctx(a, smem); packed_to_lds(c.s_H, c.s_M, a.nv, CholCfg<N>::LD, threadIdx.x); a.out[threadIdx.x] = c.quad_gauss[0] + c.s_jar[threadIdx.x]; }
KB(kb_wavesum) { SMEM; SolveCtx<N> c = make_ctx(a, smem); a.out[threadIdx.x] = wave_sum(c.s_jar[threadIdx.x]) + c.quad_gauss[0]; }
KB(kb_hstore) {
  SMEM; SolveCtx<N> c = make_ctx(a, smem);
  f32x4 t[NBk * (NBk + 1) / 2];
  for (int i = 0; i < NBk * (NBk + 1) / 2; ++i) t[i] = *(const f32x4*)(c.s_jar + 4 * i);
  hessian_store<N, false>(c, t);
  a.out[threadIdx.x] = c.quad_gauss[0] + c.s_jar[threadIdx.x];
}
