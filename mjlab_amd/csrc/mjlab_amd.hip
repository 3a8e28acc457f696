// mjlab_amd.hip -- batched MuJoCo-style physics step for MI355X (gfx950 / CDNA4).
//
// Replaces the reference's foreign calls mjwarp.step / mjwarp.forward
// (reference: src/mjlab/sim/sim.py:136,139,187,195).  Stage names follow MuJoCo's
// pipeline as catalogued by the reference's stubs (typings/mujoco/_functions.pyi:
// mj_kinematics :803, mj_comPos :358, mj_crb :399, mj_factorM :449, mj_collision :353,
// mj_makeConstraint :835, mj_comVel :363, mj_rne :1070, mj_fwdActuation :493,
// mj_fwdAcceleration :488, mj_fwdConstraint :498, mj_implicit :555).
//
// Execution model: ONE WORLD (environment) PER WAVEFRONT.  A workgroup is a single
// 64-lane wave, so every stage kernel launches `nworld` workgroups; with 4096 worlds that
// is 16 waves per CU on the 256 CUs of an MI355X, all resident at once: every stage keeps
// its LDS footprint at or below 10 KB and its registers at or below 128 (4 waves per SIMD).
// Inside a wave, lanes own bodies, dofs, candidate geom pairs, contacts, constraint rows or
// matrix rows, depending on the stage.  The tree recursions are not swept level by level: a
// body composes the relative poses of its ancestors (kinematics) or sums over the dofs of its
// ancestor chain (velocities) on its own, and the stage kernels request every model constant a
// lane needs in one batch at kernel start -- at 4096 worlds these kernels are bound by
// dependent latency (global round trips above all), not by throughput.  Public mjData
// arrays are [nworld][n] row-major, so "lanes = elements of one world's row" gives coalesced
// HBM traffic; intermediates that never leave a stage live in LDS or registers.  The only
// GEMM-shaped work -- the Newton Hessian H = M + J^T D J over the ACTIVE constraint rows --
// is streamed row-major from L2 straight into fp32 MFMA (v_mfma_f32_16x16x4_f32) operands.
//
// Five stage kernels per physics step:
//   k_position    kinematics, comPos, crb, dense M           (skipped after an unchanged forward())
//   k_collision   static pair list + box terrain through an xy grid, analytic primitives (   "   )
//   k_velocity    comVel, rne, actuation, qfrc_smooth
//   k_constraint  limits + contacts -> efc rows, sensors      (   "   )
//   k_solve_integrate<NVP>  Newton solver (LDL^T in registers/LDS, exact line search),
//                 implicitfast / Euler integration; a state machine around one factor site
// plus helpers: k_tile (expand_model_fields), k_fold_snapshot, k_masked_reset,
// k_entity_readback.
//
// All arithmetic is fp32 (like the reference's Warp kernels); ids are int32.

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "../../include/mjlab_amd.h"

typedef mjlab_model_t Model;
typedef mjlab_data_t Data;
typedef float __attribute__((ext_vector_type(4))) f32x4;
typedef float __attribute__((ext_vector_type(2))) f32x2;

// Opt-in phase profiling (tools/profile_phases.py builds a second library with
// -DMJLAB_PROFILE): accumulates shader-clock deltas per phase into data.profile[world][16].
#ifdef MJLAB_PROFILE
#define PROF_INIT() long long prof_last_ = clock64(); float prof_acc_[16] = {0}
#define PROF_MARK(id) do { long long n_ = clock64(); prof_acc_[id] += (float)(n_ - prof_last_); prof_last_ = n_; } while (0)
#define PROF_COUNT(id) prof_acc_[id] += 1.f
#define PROF_FLUSH(ptr) do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 16; ++i_) (ptr)[i_] += prof_acc_[i_]; } while (0)
#else
#define PROF_INIT() do {} while (0)
#define PROF_MARK(id) do {} while (0)
#define PROF_COUNT(id) do {} while (0)
#define PROF_FLUSH(ptr) do {} while (0)
#endif

// The dense factor / substitution routines are large fully-unrolled bodies; the solve kernel is
// organised so that each is instantiated exactly once (see k_solve_integrate).
#define CHOL_INLINE __forceinline__

#ifndef MJLAB_CB
#define MJLAB_CB 12
#endif
#define MINVAL 1e-15f
#define MINIMP 0.0001f
#define MAXIMP 0.9999f
#define MF(name) (m.name + (size_t)w * (size_t)m.name##_ws)

// ------------------------------------------------------------------------------------
// wave-level helpers (wave = 64 lanes)
// ------------------------------------------------------------------------------------
// Hides a per-lane index from the optimiser at the point of use: global addresses derived from it
// are then formed where they are needed instead of being hoisted to the top of a long kernel and
// kept alive (spilled) across all of it.
__device__ __forceinline__ int launder(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
__device__ __forceinline__ float lane_bcast(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
// DPP cross-lane moves (no LDS traffic): dpp_ctrl encodings of the GFX9 family --
// quad_perm 0x00-0xFF, row_mirror 0x140, row_half_mirror 0x141, row_bcast:15 0x142,
// row_bcast:31 0x143.  Lanes disabled by row_mask receive `old` (= 0 here).
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, BOUND));
}
// Sum over each row of 16 lanes (xor-butterfly 1,2 | half-mirror | mirror); every lane of the
// row ends with the row total.
__device__ __forceinline__ float group16_sum(float v) {
  v += dpp_mov<0xB1, 0xF, true>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E, 0xF, true>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141, 0xF, true>(v);  // row_half_mirror
  v += dpp_mov<0x140, 0xF, true>(v);  // row_mirror
  return v;
}
// Sum over the wave; the result is made explicitly wave-uniform (SGPR) so that the solver's
// control flow compiles to scalar branches.
__device__ __forceinline__ float wave_sum(float v) {
  v = group16_sum(v);
  v += dpp_mov<0x142, 0xA, false>(v);  // rows 1,3 += lane 15 of rows 0,2
  v += dpp_mov<0x143, 0xC, false>(v);  // rows 2,3 += lane 31
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// reference implementations through the LDS crossbar (used by the self-test only)
__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float group16_sum_shfl(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int* total) {
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  *total = __shfl(incl, 63);
  return incl - v;
}
__device__ __forceinline__ void lds_to_global(float* dst, const float* src, int n, int lane) {
  for (int k = lane; k < n; k += 64) dst[k] = src[k];
}
__device__ __forceinline__ void global_to_lds(float* dst, const float* src, int n, int lane) {
  for (int k = lane; k < n; k += 64) dst[k] = src[k];
}

// Dense n x n matrix copies between row-major global memory (leading dimension n) and LDS
// (leading dimension ld); lanes walk consecutive global elements, (i, j) tracked without
// integer division.
__device__ __forceinline__ void dense_global_to_lds(float* dst, const float* src, int n, int ld, int lane, bool lower_only) {
  lane = launder(lane);
  int i = 0, j = lane;
  while (j >= n) { j -= n; ++i; }
  for (int k = lane; k < n * n; k += 64) {
    if (!lower_only || j <= i) dst[i * ld + j] = src[k];
    j += 64;
    while (j >= n) { j -= n; ++i; }
  }
}
__device__ __forceinline__ void dense_lds_to_global(float* dst, const float* src, int n, int ld, int lane, bool zero_upper) {
  int i = 0, j = lane;
  while (j >= n) { j -= n; ++i; }
  for (int k = lane; k < n * n; k += 64) {
    dst[k] = (zero_upper && j > i) ? 0.f : src[i * ld + j];
    j += 64;
    while (j >= n) { j -= n; ++i; }
  }
}

// ------------------------------------------------------------------------------------
// small math (quaternions w-x-y-z, row-major 3x3)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void cross3(float* r, const float* a, const float* b) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ float normalize3(float* v) {
  float n = sqrtf(dot3(v, v));
  if (n < MINVAL) { v[0] = 1; v[1] = 0; v[2] = 0; return 0; }
  float inv = 1.0f / n;
  v[0] *= inv; v[1] *= inv; v[2] *= inv;
  return n;
}
__device__ __forceinline__ void normalize4(float* q) {
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  float inv = 1.0f / n;
  q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}
__device__ __forceinline__ void mul_quat(float* r, const float* a, const float* b) {
  float w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  float x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  float y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  float z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
__device__ __forceinline__ void quat2mat(float* R, const float* q) {
  float q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  float q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
  float q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  R[0] = q00 + q11 - q22 - q33; R[4] = q00 - q11 + q22 - q33; R[8] = q00 - q11 - q22 + q33;
  R[1] = 2 * (q12 - q03); R[2] = 2 * (q13 + q02);
  R[3] = 2 * (q12 + q03); R[5] = 2 * (q23 - q01);
  R[6] = 2 * (q13 - q02); R[7] = 2 * (q23 + q01);
}
__device__ __forceinline__ void mul_mat_vec3(float* r, const float* R, const float* v) {
  float x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  float y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  float z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ void rot_vec_quat(float* r, const float* v, const float* q) {
  float R[9];
  quat2mat(R, q);
  mul_mat_vec3(r, R, v);
}
__device__ __forceinline__ void axis_angle2quat(float* q, const float* axis, float angle) {
  float s, c;
  sincosf(angle * 0.5f, &s, &c);
  q[0] = c; q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
__device__ __forceinline__ float clipf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// spatial algebra; motion vectors are [angular(3), linear(3)] about subtree_com[root]
__device__ __forceinline__ void mul_inert_vec(float* r, const float* i, const float* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
__device__ __forceinline__ void cross_motion(float* r, const float* vel, const float* v) {
  r[0] = -vel[2] * v[1] + vel[1] * v[2];
  r[1] = vel[2] * v[0] - vel[0] * v[2];
  r[2] = -vel[1] * v[0] + vel[0] * v[1];
  r[3] = -vel[2] * v[4] + vel[1] * v[5] - vel[5] * v[1] + vel[4] * v[2];
  r[4] = vel[2] * v[3] - vel[0] * v[5] + vel[5] * v[0] - vel[3] * v[2];
  r[5] = -vel[1] * v[3] + vel[0] * v[4] - vel[4] * v[0] + vel[3] * v[1];
}
__device__ __forceinline__ void cross_force(float* r, const float* vel, const float* f) {
  r[0] = -vel[2] * f[1] + vel[1] * f[2] - vel[5] * f[4] + vel[4] * f[5];
  r[1] = vel[2] * f[0] - vel[0] * f[2] + vel[5] * f[3] - vel[3] * f[5];
  r[2] = -vel[1] * f[0] + vel[0] * f[1] - vel[4] * f[3] + vel[3] * f[4];
  r[3] = -vel[2] * f[4] + vel[1] * f[5];
  r[4] = vel[2] * f[3] - vel[0] * f[5];
  r[5] = -vel[1] * f[3] + vel[0] * f[4];
}

__device__ __forceinline__ bool dof_in_chain(const Model& m, int body, int dof) {
  unsigned lo = (unsigned)m.body_dofmask[2 * body], hi = (unsigned)m.body_dofmask[2 * body + 1];
  return dof < 32 ? ((lo >> dof) & 1u) : ((hi >> (dof - 32)) & 1u);
}

// ------------------------------------------------------------------------------------
// Dense Cholesky A = L L^T for one world, n <= 64, REGISTER-RESIDENT: lane i owns row i of
// the lower triangle in NVP VGPRs (NVP = nv padded to a compile-time size; rows >= nv are
// identity).  The left-looking column sweep is fully unrolled, so L[j][k] is a
// v_readlane of lane j's k-th register feeding an FMA with an SGPR operand: no LDS traffic
// and no barriers inside the factorization (about 2 instructions per multiply-add instead
// of the ~12 an LDS-resident sweep needs).  The matrix travels through LDS only to move
// between layouts: MFMA tiles -> rows (before), rows -> columns of L for the backward
// substitution (after).  LDS leading dimension LD is a multiple of 4 with LD/4 odd, so the
// per-lane 128-bit row accesses are bank-conflict free.
// ------------------------------------------------------------------------------------
template <int NVP>
struct CholCfg {
  static constexpr int LD = (NVP % 8 == 4) ? NVP : NVP + 4;
  static constexpr int NB = (NVP + 15) / 16;  // 16-column blocks for the MFMA Hessian
};

// LDS-qualified views: the factor routines are out-of-line functions, and a generic `float*`
// argument would make every access pay for an address-space check.
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

template <int NVP>
__device__ __forceinline__ void chol_pad_rows(float* A, int n, int lane) {
  constexpr int LD = CholCfg<NVP>::LD;
  for (int k = n * LD + lane; k < NVP * LD; k += 64) A[k] = 0.f;
}
template <int NVP>
__device__ __forceinline__ void chol_pad_diag(float* A, int n, int lane) {
  constexpr int LD = CholCfg<NVP>::LD;
  if (lane >= n && lane < NVP) A[lane * LD + lane] = 1.f;
}
// Column sweep of the factorization below, written as compile-time recursion over the column J
// and the batch BI so that every register-array index and the choice of ping-pong buffer is a
// constant (the arrays must live in VGPRs, never in scratch).
template <int NVP, int CB>
struct CholSweep {
  static constexpr int LD = CholCfg<NVP>::LD;
  template <int R, int K0>
  static __device__ __forceinline__ void load_batch(lds_f32* A, float (&dst)[CB]) {
#pragma unroll
    for (int q = 0; q < CB / 4; ++q) {
      if (K0 + 4 * q < R) {
        const f32x4 v = *(lds_f32x4*)(A + R * LD + K0 + 4 * q);
        dst[4 * q] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
      }
    }
  }
  // batch BI of column J: request the next batch (same row, or first batch of row J+1) into
  // `nxt`, feed `cur` to the FMAs, recurse with the buffers swapped
  template <int J, int BI>
  static __device__ __forceinline__ void batches(const float (&a)[NVP], f32x2& acc, float (&cur)[CB], float (&nxt)[CB], lds_f32* A) {
    constexpr int NBJ = (J + CB - 1) / CB, K0 = BI * CB;
    if constexpr (BI < NBJ) {
      if constexpr (BI + 1 < NBJ) load_batch<J, K0 + CB>(A, nxt);
      else if constexpr (J + 1 < NVP) load_batch<J + 1, 0>(A, nxt);
#pragma unroll
      for (int k = 0; k < CB; k += 2) {
        if (K0 + k + 1 < J) {
          const f32x2 av = {a[K0 + k], a[K0 + k + 1]};
          const f32x2 sv = {cur[k], cur[k + 1]};
          acc -= av * sv;
        } else if (K0 + k < J) {
          acc.x -= a[K0 + k] * cur[k];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      batches<J, BI + 1>(a, acc, nxt, cur, A);
    }
  }
  // column J; `cur` holds (or is about to receive) the first batch of row J
  template <int J>
  static __device__ __forceinline__ void col(float (&a)[NVP], float (&cur)[CB], float (&oth)[CB], lds_f32* A, lds_f32* row, int rowid, float& myinvd) {
    constexpr int NBJ = (J + CB - 1) / CB;
    f32x2 acc = {a[J], 0.f};  // two accumulators, products in pairs (v_pk_fma_f32)
    batches<J, 0>(a, acc, cur, oth, A);
    const float t = acc.x + acc.y;
    const float djj = fmaxf(lane_bcast(t, J), MINVAL);
    float invd = __builtin_amdgcn_rcpf(djj);  // v_rcp_f32 (1 ulp) + one Newton step
    invd = invd * (2.f - djj * invd);
    a[J] = t;
    const float lu = rowid > J ? t * invd : 0.f;
    row[J] = lu;
    myinvd = rowid == J ? invd : myinvd;
    if constexpr (J + 1 < NVP) {
      // after NBJ swaps the first batch of row J+1 sits in `cur` (NBJ even) or `oth` (NBJ odd)
      if constexpr (NBJ == 0) {
        load_batch<J + 1, 0>(A, cur);  // J == 0: nothing was in flight
      } else if constexpr (J < CB) {   // entry written after the request: patch from lane J+1
        const float e = lane_bcast(lu, J + 1);
        if constexpr (NBJ % 2 == 0) cur[J] = e; else oth[J] = e;
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NBJ % 2 == 0) col<J + 1>(a, cur, oth, A, row, rowid, myinvd);
      else col<J + 1>(a, oth, cur, A, row, rowid, myinvd);
    }
  }
};

// A (LDS, lower triangle valid for rows < n) -> unit-lower factor of A = Lu D Lu^T in place:
// Lu[i][j] (i > j), ZERO on and above the diagonal, s_invd[i] = 1 / D_i.  With the zero
// diagonal the substitutions below are a bare v_readlane + v_fma per step.
//
// Lane i owns row i in NVP registers.  Left-looking column sweep, fully unrolled:
//   t_i = A[i][j] - sum_{k<j} W[i][k] * Lu[j][k],   W[i][k] = t_i of step k (kept in a[k]),
//   D_j = t_j,  Lu[i][j] = t_i / D_j  -> written to LDS column j by every lane.
// Row j of Lu, which every lane needs in step j, is read back from LDS as 128-bit
// *broadcast* reads (all lanes, same address: conflict-free), ceil(j/4) instructions instead
// of j cross-lane v_readlane's; the wave's DS queue is in order, so the column written in
// step j-1 is visible without a barrier.  No square roots (LDL^T).
// Lanes >= NVP mirror lane NVP-1 (same row, same arithmetic, same values stored), which keeps
// the sweep free of exec-mask branches.  Rows n <= i < NVP must hold identity rows on entry:
// producers call chol_pad_rows() once per kernel (zero fill; the factor keeps those rows'
// off-diagonals at zero) and chol_pad_diag() after every (re)write of the matrix.
// Caller synchronises before and after.
template <int NVP>
__device__ CHOL_INLINE void chol_factor(float* A_, float* s_invd_, int n, int lane) {
  constexpr int LD = CholCfg<NVP>::LD;
  lds_f32* A = (lds_f32*)A_;
  lds_f32* s_invd = (lds_f32*)s_invd_;
  int rowid = lane < NVP ? lane : NVP - 1;
  // opaque: otherwise the 2 NVP lane-mask compares below are loop invariant for the caller's
  // solver loop, get hoisted out of it and live (spilled) in ~150 SGPRs
  asm volatile("" : "+v"(rowid));
  lds_f32* row = A + rowid * LD;
  float a[NVP];
#pragma unroll
  for (int c = 0; c < NVP / 4; ++c) {
    const f32x4 v = *(lds_f32x4*)(row + 4 * c);
    a[4 * c] = v.x; a[4 * c + 1] = v.y; a[4 * c + 2] = v.z; a[4 * c + 3] = v.w;
  }
  (void)n;  // rows >= n are identity rows already (chol_pad_rows / chol_pad_diag by the producer)
  float myinvd = 1.f;
  // Row j of Lu is consumed in batches of CB columns.  The batches are software pipelined
  // through two register buffers: while batch i feeds the FMAs, the reads of batch i+1 --
  // the next batch of the same row, or the first batch of the next row -- are already in
  // flight, so the sweep does not stall on an LDS round trip per column.  The first batch of
  // row j+1 is requested before column j is written; its one missing entry Lu[j+1][j] is
  // patched in from lane j+1's register.
  float bufA[MJLAB_CB], bufB[MJLAB_CB];
  CholSweep<NVP, MJLAB_CB>::template col<0>(a, bufA, bufB, A, row, rowid, myinvd);
  s_invd[rowid] = myinvd;
}
// Solves Lu D Lu^T x = b with the factor in LDS (as left by chol_factor); lane i owns
// b_i / x_i (lanes >= n must pass 0).  Forward substitution uses row i of Lu, backward
// substitution row i of Lu^T (= column i of Lu, read with unit stride across lanes).
template <int NVP>
__device__ CHOL_INLINE float chol_solve(const float* L_, const float* s_invd_, int lane, float b) {
  constexpr int LD = CholCfg<NVP>::LD;
  const lds_f32* L = (const lds_f32*)L_;
  const lds_f32* s_invd = (const lds_f32*)s_invd_;
  const int li = lane < NVP ? lane : NVP - 1;
  const float invd = s_invd[li];
  {
    float a[NVP];
#pragma unroll
    for (int c = 0; c < NVP / 4; ++c) {
      const f32x4 v = *(const lds_f32x4*)(L + li * LD + 4 * c);
      a[4 * c] = v.x; a[4 * c + 1] = v.y; a[4 * c + 2] = v.z; a[4 * c + 3] = v.w;
    }
#pragma unroll
    for (int k = 0; k < NVP; ++k) b = fmaf(-a[k], lane_bcast(b, k), b);  // a[k] = 0 for lanes <= k
  }
  b *= invd;
  __builtin_amdgcn_sched_barrier(0);
  {
    float at[NVP];
#pragma unroll
    for (int k = 0; k < NVP; ++k) at[k] = L[k * LD + li];  // Lu[k][i]: zero for k <= i
#pragma unroll
    for (int k = NVP - 1; k >= 0; --k) b = fmaf(-at[k], lane_bcast(b, k), b);
  }
  return b;
}
// y_i = sum_j M[i][j] v_j with M symmetric, dense row-major in GLOBAL memory (ld = n);
// lane i owns v_i and y_i.  Row j is read coalesced (M[j][i] = M[i][j]), v_j comes from
// lane j by v_readlane; fully unrolled so all loads are in flight together.
template <int NVP>
__device__ __forceinline__ float symm_mul_global(const float* M, int n, float v, int lane) {
  // The element offset is made opaque to the optimiser: otherwise the NVP row addresses are
  // loop-invariant 64-bit VGPR pairs that get hoisted out of the Newton loop and spilled.
  int off = lane < n ? lane : 0;
  asm volatile("" : "+v"(off));
  constexpr int CH = 12;  // loads in flight per chunk
  float y0 = 0.f, y1 = 0.f;
#pragma unroll
  for (int j0 = 0; j0 < NVP; j0 += CH) {
    float mv[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int j = j0 + u;
      mv[u] = (j < NVP && j < n) ? M[j * n + off] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int j = j0 + u;
      if (j < NVP) {
        if (u & 1) y1 = fmaf(mv[u], lane_bcast(v, j), y1);
        else y0 = fmaf(mv[u], lane_bcast(v, j), y0);
      }
    }
  }
  return lane < n ? y0 + y1 : 0.f;
}

// launch flags shared by the stage kernels
enum {
  FLAG_MASK = 1,      // skip worlds whose world_mask entry is 0 (mjlab_forward_masked)
  FLAG_FOLD = 2,      // step(): reuse the position / collision / constraint stages of the last forward()
                      // in worlds whose qpos and qvel are still bit-identical (fold_reuse, set by k_position)
  FLAG_SNAPSHOT = 4   // forward(): record qpos / qvel next to the derived arrays (fold_valid = 1)
};

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

__global__ __launch_bounds__(64, 4) void k_position(const Model m, const Data d, const int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int w = blockIdx.x, lane = threadIdx.x;
  if ((flags & FLAG_MASK) && !d.world_mask[w]) return;
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
    if (reuse) return;
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
  int t_body = 0;
  float t_pos[3] = {0.f, 0.f, 0.f}, t_quat[4] = {1.f, 0.f, 0.f, 0.f};
  if (lane < ns) {
    const float *spos = MF(site_pos), *squat = MF(site_quat);
    t_body = m.site_bodyid[lane];
    for (int k = 0; k < 3; ++k) t_pos[k] = spos[3 * lane + k];
    for (int k = 0; k < 4; ++k) t_quat[k] = squat[4 * lane + k];
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
  if (lane == 0) {
    s_xpos[0] = s_xpos[1] = s_xpos[2] = 0.f;
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
        for (int k = 0; k < 3; ++k) gx[3 * g + k] = xp[k];
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
      for (int k = 0; k < 3; ++k) gx[3 * g + k] = xp[k];
      for (int k = 0; k < 9; ++k) gm[9 * g + k] = xm[k];
    }
    float* sx = d.site_xpos + (size_t)w * 3 * ns;
    float* sm = d.site_xmat + (size_t)w * 9 * ns;
    if (lane < ns) {
      const int b = t_body;
      float bp[3], bq[4], bm[9], xp[3], xm[9];
      for (int k = 0; k < 3; ++k) bp[k] = s_xpos[3 * b + k];
      for (int k = 0; k < 4; ++k) bq[k] = s_xquat[4 * b + k];
      for (int k = 0; k < 9; ++k) bm[k] = s_xmat[9 * b + k];
      local2global(xp, xm, bp, bq, bm, t_pos, t_quat);
      for (int k = 0; k < 3; ++k) sx[3 * lane + k] = xp[k];
      for (int k = 0; k < 9; ++k) sm[9 * lane + k] = xm[k];
    }
    const float *spos = MF(site_pos), *squat = MF(site_quat);
    for (int g = 64 + lane; g < ns; g += 64) {  // models with more than 64 sites
      const int b = m.site_bodyid[g];
      float bp[3], bq[4], bm[9], ip[3], iq[4], xp[3], xm[9];
      for (int k = 0; k < 3; ++k) { bp[k] = s_xpos[3 * b + k]; ip[k] = spos[3 * g + k]; }
      for (int k = 0; k < 4; ++k) { bq[k] = s_xquat[4 * b + k]; iq[k] = squat[4 * g + k]; }
      for (int k = 0; k < 9; ++k) bm[k] = s_xmat[9 * b + k];
      local2global(xp, xm, bp, bq, bm, ip, iq);
      for (int k = 0; k < 3; ++k) sx[3 * g + k] = xp[k];
      for (int k = 0; k < 9; ++k) sm[9 * g + k] = xm[k];
    }
  }
  __syncthreads();
  PROF_MARK(1);
  lds_to_global(d.xpos + (size_t)w * 3 * nb, s_xpos, 3 * nb, lane);
  lds_to_global(d.xquat + (size_t)w * 4 * nb, s_xquat, 4 * nb, lane);
  lds_to_global(d.xmat + (size_t)w * 9 * nb, s_xmat, 9 * nb, lane);
  lds_to_global(d.xipos + (size_t)w * 3 * nb, s_xipos, 3 * nb, lane);
  lds_to_global(d.ximat + (size_t)w * 9 * nb, s_ximat, 9 * nb, lane);
  lds_to_global(d.xanchor + (size_t)w * 3 * nj, s_xanchor, 3 * nj, lane);
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
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int j = b;
    for (; j + 3 < e; j += 4) {
      a0 += s_mass[j] * s_xipos[3 * j + c];
      a1 += s_mass[j + 1] * s_xipos[3 * j + 3 + c];
      a2 += s_mass[j + 2] * s_xipos[3 * j + 6 + c];
      a3 += s_mass[j + 3] * s_xipos[3 * j + 9 + c];
    }
    for (; j < e; ++j) a0 += s_mass[j] * s_xipos[3 * j + c];
    const float acc = (a0 + a1) + (a2 + a3);
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
  lds_to_global(d.subtree_com + (size_t)w * 3 * nb, s_sub, 3 * nb, lane);
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
}

// ====================================================================================
// Stage 2: collision (static candidate pair list; plane/sphere/capsule/box primitives)
// ====================================================================================
struct RawCon { float dist, pos[3], frame[6]; };
// dst = take ? src : dst, field by field (v_cndmask).  Contact slots are filled through VALUE
// selects with compile-time slot indices: a conditional store to `slot[n]` makes the compiler keep
// the whole slot array in scratch memory.
__device__ __forceinline__ void rc_take(RawCon& dst, const RawCon& src, bool take) {
  dst.dist = take ? src.dist : dst.dist;
#pragma unroll
  for (int k = 0; k < 3; ++k) dst.pos[k] = take ? src.pos[k] : dst.pos[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) dst.frame[k] = take ? src.frame[k] : dst.frame[k];
}

__device__ __forceinline__ int plane_sphere(RawCon* c, float margin, const float* ppos, const float* pn, const float* spos, float r) {
  float dif[3] = {spos[0] - ppos[0], spos[1] - ppos[1], spos[2] - ppos[2]};
  float cdist = dot3(dif, pn);
  if (cdist > margin + r) return 0;
  c->dist = cdist - r;
  for (int k = 0; k < 3; ++k) { c->pos[k] = spos[k] + pn[k] * (-c->dist * 0.5f - r); c->frame[k] = pn[k]; c->frame[3 + k] = 0.f; }
  return 1;
}
__device__ __forceinline__ int sphere_sphere(RawCon* c, float margin, const float* p1, float r1, const float* p2, float r2) {
  float dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  float cd2 = dot3(dif, dif), mn = margin + r1 + r2;
  if (cd2 > mn * mn) return 0;
  float len = sqrtf(cd2);
  if (len < MINVAL) { dif[0] = 1.f; dif[1] = dif[2] = 0.f; }
  else { float inv = 1.0f / len; dif[0] *= inv; dif[1] *= inv; dif[2] *= inv; }
  c->dist = len - r1 - r2;
  for (int k = 0; k < 3; ++k) { c->pos[k] = p1[k] + dif[k] * (r1 + c->dist * 0.5f); c->frame[k] = dif[k]; c->frame[3 + k] = 0.f; }
  return 1;
}
__device__ __forceinline__ int capsule_capsule(RawCon* c, float margin, const float* pos1, const float* axis1, const float* size1,
                               const float* pos2, const float* axis2, const float* size2) {
  float dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  float ma = dot3(axis1, axis1), mb = -dot3(axis1, axis2), mc = dot3(axis2, axis2);
  float u = -dot3(axis1, dif), v = dot3(axis2, dif), det = ma * mc - mb * mb;
  float vec1[3], vec2[3];
  if (fabsf(det) >= MINVAL) {
    float x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > size1[1]) { x1 = size1[1]; x2 = (v - mb * size1[1]) / mc; }
    else if (x1 < -size1[1]) { x1 = -size1[1]; x2 = (v + mb * size1[1]) / mc; }
    if (x2 > size2[1]) { x2 = size2[1]; x1 = clipf((u - mb * size2[1]) / ma, -size1[1], size1[1]); }
    else if (x2 < -size2[1]) { x2 = -size2[1]; x1 = clipf((u + mb * size2[1]) / ma, -size1[1], size1[1]); }
    for (int k = 0; k < 3; ++k) { vec1[k] = pos1[k] + axis1[k] * x1; vec2[k] = pos2[k] + axis2[k] * x2; }
    return sphere_sphere(c, margin, vec1, size1[0], vec2, size2[0]);
  }
  // parallel axes: up to two contacts out of four end-point candidates, taken in order.  The
  // output slots are written with compile-time indices (a run-time `c + n` would push the
  // whole contact array into scratch memory).
  int n = 0;
  RawCon t;
  auto push = [&](bool ok) {
    rc_take(c[0], t, ok && n == 0);
    rc_take(c[1], t, ok && n == 1);
    n += ok ? 1 : 0;
  };
  float x1, x2;
  for (int k = 0; k < 3; ++k) vec1[k] = pos1[k] + axis1[k] * size1[1];
  x2 = clipf((v - mb * size1[1]) / mc, -size2[1], size2[1]);
  for (int k = 0; k < 3; ++k) vec2[k] = pos2[k] + axis2[k] * x2;
  push(sphere_sphere(&t, margin, vec1, size1[0], vec2, size2[0]) != 0);
  for (int k = 0; k < 3; ++k) vec1[k] = pos1[k] - axis1[k] * size1[1];
  x2 = clipf((v + mb * size1[1]) / mc, -size2[1], size2[1]);
  for (int k = 0; k < 3; ++k) vec2[k] = pos2[k] + axis2[k] * x2;
  push(sphere_sphere(&t, margin, vec1, size1[0], vec2, size2[0]) != 0);
  if (n == 2) return n;
  for (int k = 0; k < 3; ++k) vec2[k] = pos2[k] + axis2[k] * size2[1];
  x1 = clipf((u - mb * size2[1]) / ma, -size1[1], size1[1]);
  for (int k = 0; k < 3; ++k) vec1[k] = pos1[k] + axis1[k] * x1;
  push(sphere_sphere(&t, margin, vec1, size1[0], vec2, size2[0]) != 0);
  if (n == 2) return n;
  for (int k = 0; k < 3; ++k) vec2[k] = pos2[k] - axis2[k] * size2[1];
  x1 = clipf((u + mb * size2[1]) / ma, -size1[1], size1[1]);
  for (int k = 0; k < 3; ++k) vec1[k] = pos1[k] + axis1[k] * x1;
  push(sphere_sphere(&t, margin, vec1, size1[0], vec2, size2[0]) != 0);
  return n < 2 ? n : 2;
}


// Sphere vs (static) box: centre into the box frame, clamp; outside the normal runs along
// clamped point -> centre, inside through the nearest face.  Normal points from the sphere
// (geom1) into the box (geom2), pos midway between the surfaces.  bmat is row major (world =
// bmat * local).
__device__ __forceinline__ int sphere_box(RawCon* c, float margin, const float* spos, float r, const float* bpos, const float* bmat, const float* bsize) {
  const float dif[3] = {spos[0] - bpos[0], spos[1] - bpos[1], spos[2] - bpos[2]};
  float loc[3], dv[3], nl[3], pl[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) loc[i] = bmat[i] * dif[0] + bmat[3 + i] * dif[1] + bmat[6 + i] * dif[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) dv[i] = loc[i] - clipf(loc[i], -bsize[i], bsize[i]);
  const float d2 = dot3(dv, dv), mn = margin + r;
  if (d2 > mn * mn) return 0;
  if (d2 > 0.f) {
    const float len = sqrtf(d2);
    c->dist = len - r;
#pragma unroll
    for (int i = 0; i < 3; ++i) { nl[i] = dv[i] / len; pl[i] = (loc[i] - dv[i]) + nl[i] * (c->dist * 0.5f); }
  } else {
    // centre inside the box: leave through the nearest face (first one on ties)
    int k = 0;
    float depth = bsize[0] - fabsf(loc[0]);
#pragma unroll
    for (int i = 1; i < 3; ++i) { const float di = bsize[i] - fabsf(loc[i]); if (di < depth) { depth = di; k = i; } }
#pragma unroll
    for (int i = 0; i < 3; ++i) nl[i] = (i == k) ? (loc[i] >= 0.f ? 1.f : -1.f) : 0.f;
    c->dist = -depth - r;
#pragma unroll
    for (int i = 0; i < 3; ++i) pl[i] = loc[i] + nl[i] * ((depth - r) * 0.5f);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    c->pos[i] = bpos[i] + bmat[3 * i] * pl[0] + bmat[3 * i + 1] * pl[1] + bmat[3 * i + 2] * pl[2];
    c->frame[i] = -(bmat[3 * i] * nl[0] + bmat[3 * i + 1] * nl[1] + bmat[3 * i + 2] * nl[2]);
    c->frame[3 + i] = 0.f;
  }
  return 1;
}
// d/dt of half the squared distance between pc + t h (box frame) and the box, and the squared distance
__device__ __forceinline__ float seg_box_slope(const float* pc, const float* h, const float* bsize, float t) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) { const float p = pc[i] + t * h[i]; s += (p - clipf(p, -bsize[i], bsize[i])) * h[i]; }
  return s;
}
__device__ __forceinline__ float seg_box_dist2(const float* pc, const float* h, const float* bsize, float t) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) { const float p = pc[i] + t * h[i], e = p - clipf(p, -bsize[i], bsize[i]); s += e * e; }
  return s;
}
// Capsule vs box: up to 4 sphere_box() contacts of spheres of the capsule's radius on its axis
// (point cpos + axis * halflen * t): the two ends, plus the ends ta <= tb of the interval where
// the axis is closest to the box when they are interior points outside the box (if the axis runs
// through the box with both ends outside: the inside point nearest to the capsule's centre).  The distance
// along the axis is convex, so its slope is monotone: ta / tb come from two bisections.
#define MJLAB_CAPBOX_ITERS 24
__device__ __forceinline__ int capsule_box(RawCon* c, float margin, const float* cpos, const float* axis, const float* csize, const float* bpos,
                                           const float* bmat, const float* bsize) {
  const float dif[3] = {cpos[0] - bpos[0], cpos[1] - bpos[1], cpos[2] - bpos[2]};
  float pc[3], h[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    pc[i] = bmat[i] * dif[0] + bmat[3 + i] * dif[1] + bmat[6 + i] * dif[2];
    h[i] = (bmat[i] * axis[0] + bmat[3 + i] * axis[1] + bmat[6 + i] * axis[2]) * csize[1];
  }
  const float sm = seg_box_slope(pc, h, bsize, -1.f), sp = seg_box_slope(pc, h, bsize, 1.f);
  // ta = smallest t with slope >= 0, tb = largest t with slope <= 0; both searches run in one loop
  float alo = -1.f, ahi = 1.f, blo = -1.f, bhi = 1.f;
  for (int it = 0; it < MJLAB_CAPBOX_ITERS; ++it) {
    const float am = 0.5f * (alo + ahi), bm = 0.5f * (blo + bhi);
    const bool ag = seg_box_slope(pc, h, bsize, am) >= 0.f, bl = seg_box_slope(pc, h, bsize, bm) <= 0.f;
    ahi = ag ? am : ahi; alo = ag ? alo : am;
    blo = bl ? bm : blo; bhi = bl ? bhi : bm;
  }
  const float ta = sm >= 0.f ? -1.f : (sp < 0.f ? 1.f : ahi);
  const float tb = sp <= 0.f ? 1.f : (sm > 0.f ? -1.f : blo);
  const float eps = 1e-6f;
  const bool ia = ta > -1.f + eps && ta < 1.f - eps, ib = tb > -1.f + eps && tb < 1.f - eps;
  const bool oa = seg_box_dist2(pc, h, bsize, ta) > 0.f, ob = seg_box_dist2(pc, h, bsize, tb) > 0.f;
  // the axis runs THROUGH the box with both ends outside (a thin capsule across an edge, deeper than
  // its radius): the inside point nearest to the capsule's centre carries the contact (the middle of
  // a chord through opposite faces would be equidistant from both)
  const bool pierce = ia && ib && !oa && !ob;
  const float tmid = pierce ? clipf(0.f, ta, tb) : ta;
  const bool use_a = pierce || (ia && oa);
  const bool use_b = ib && tb - ta > eps && ob;
  int n = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float t = q == 0 ? 1.f : (q == 1 ? -1.f : (q == 2 ? tmid : tb));
    const bool use = q < 2 ? true : (q == 2 ? use_a : use_b);
    float p[3];
    RawCon tc;
    for (int k = 0; k < 3; ++k) p[k] = cpos[k] + axis[k] * (csize[1] * t);
    const bool hit = use && sphere_box(&tc, margin, p, csize[0], bpos, bmat, bsize) != 0;
    for (int k = 0; k < 3; ++k) tc.frame[3 + k] = axis[k];
    rc_take(c[0], tc, hit && n == 0); rc_take(c[1], tc, hit && n == 1); rc_take(c[2], tc, hit && n == 2); rc_take(c[3], tc, hit && n == 3);
    n += hit ? 1 : 0;
  }
  return n;
}

// Terrain broadphase of one moving geom: walk the grid cells under its bounding sphere and keep
// the (at most MJLAB_TCAND_MAX, smallest ids first) boxes within reach in ascending order in
// `cand` (this lane's LDS slots).  A box listed in several cells is looked at once, in the lowest
// cell the two footprints share.
__device__ __forceinline__ int terrain_walk(const Model& m, const float* centre, float reach, int* cand) {
  const int nx = m.size.tgrid_nx, ny = m.size.tgrid_ny;
  const float x0 = (float)m.opt.tgrid_x0, y0 = (float)m.opt.tgrid_y0, inv = 1.0f / (float)m.opt.tgrid_cell;
  int ix0 = (int)floorf((centre[0] - reach - x0) * inv), ix1 = (int)floorf((centre[0] + reach - x0) * inv);
  int iy0 = (int)floorf((centre[1] - reach - y0) * inv), iy1 = (int)floorf((centre[1] + reach - y0) * inv);
  ix0 = min(max(ix0, 0), nx - 1); ix1 = min(max(ix1, 0), nx - 1);
  iy0 = min(max(iy0, 0), ny - 1); iy1 = min(max(iy1, 0), ny - 1);
  int n = 0;
  for (int ix = ix0; ix <= ix1; ++ix)
    for (int iy = iy0; iy <= iy1; ++iy) {
      const int c = ix * ny + iy;
      if (centre[2] - reach > m.tgrid_ztop[c]) continue;  // wholly above everything in this cell
      const int kend = m.tgrid_start[c + 1];
      for (int k = m.tgrid_start[c]; k < kend; ++k) {
        const int b = m.tgrid_item[k];
        const int bx = m.tbox_cell0[2 * b], by = m.tbox_cell0[2 * b + 1];
        if (ix != max(ix0, bx) || iy != max(iy0, by)) continue;
        const float *bpos = m.tbox_pos + 3 * b, *bmat = m.tbox_mat + 9 * b, *bsize = m.tbox_size + 3 * b;
        const float dif[3] = {centre[0] - bpos[0], centre[1] - bpos[1], centre[2] - bpos[2]};
        float d2 = 0.f;
        for (int i = 0; i < 3; ++i) {
          const float loc = bmat[i] * dif[0] + bmat[3 + i] * dif[1] + bmat[6 + i] * dif[2];
          const float dv = loc - clipf(loc, -bsize[i], bsize[i]);
          d2 += dv * dv;
        }
        if (d2 > reach * reach) continue;
        // sorted insert, bounded: the largest id falls off the end
        int pos = n;
        while (pos > 0 && cand[pos - 1] > b) --pos;
        if (pos >= MJLAB_TCAND_MAX) continue;
        for (int q = n < MJLAB_TCAND_MAX ? n : MJLAB_TCAND_MAX - 1; q > pos; --q) cand[q] = cand[q - 1];
        cand[pos] = b;
        n = n < MJLAB_TCAND_MAX ? n + 1 : n;
      }
    }
  return n;
}

__device__ __forceinline__ void make_frame(float* f9, const float* f6) {
  float x[3] = {f6[0], f6[1], f6[2]}, y[3] = {f6[3], f6[4], f6[5]};
  if (sqrtf(dot3(y, y)) < 0.5f) {
    y[0] = y[1] = y[2] = 0.f;
    if (x[1] < 0.5f && x[1] > -0.5f) y[1] = 1.f; else y[2] = 1.f;
  }
  float t = dot3(x, y);
  y[0] -= t * x[0]; y[1] -= t * x[1]; y[2] -= t * x[2];
  normalize3(y);
  float z[3];
  cross3(z, x, y);
  for (int k = 0; k < 3; ++k) { f9[k] = x[k]; f9[3 + k] = y[k]; f9[6 + k] = z[k]; }
}

// LDS: poses (12) and constants (8: type, size, rbound, margin, gap) of geoms [geom_lds0, ngeom) | per moving geom TCAND_MAX
// candidate boxes | flat pair lists (non-box, box)
__host__ __device__ inline int collision_lds_floats(const mjlab_sizes_t& s) {
  const int lists = 3 * s.ntgeom * MJLAB_TCAND_MAX;  // terrain lists; the close-pair list (npair) aliases them
  return 20 * (s.ngeom - s.geom_lds0) + (lists > s.npair ? lists : s.npair);
}

// Contact parameters (mj_contactParam) of the pair (g1, g2) + ordered append of this lane's n
// raw contacts at slots base + off ..
__device__ __forceinline__ void emit_contacts(const Model& m, const Data& d, int w, int g1, int g2, float margin, float gap, const RawCon (&rc)[4],
                                              int n, int first, const float* gfri, const float* gsolref, const float* gsolimp, const float* gsolmix) {
  const int ncm = m.size.nconmax;
  int condim;
  float fri[3], solref[2], solimp[5];
  const int pr1 = m.geom_priority[g1], pr2 = m.geom_priority[g2];
  if (pr1 != pr2) {
    const int gi = pr1 > pr2 ? g1 : g2;
    condim = m.geom_condim[gi];
    for (int k = 0; k < 3; ++k) fri[k] = gfri[3 * gi + k];
    for (int k = 0; k < 2; ++k) solref[k] = gsolref[2 * gi + k];
    for (int k = 0; k < 5; ++k) solimp[k] = gsolimp[5 * gi + k];
  } else {
    condim = max(m.geom_condim[g1], m.geom_condim[g2]);
    for (int k = 0; k < 3; ++k) fri[k] = fmaxf(gfri[3 * g1 + k], gfri[3 * g2 + k]);
    const float sm1 = gsolmix[g1], sm2 = gsolmix[g2];
    float mix;
    if (sm1 >= MINVAL && sm2 >= MINVAL) mix = sm1 / (sm1 + sm2);
    else if (sm1 < MINVAL && sm2 < MINVAL) mix = 0.5f;
    else if (sm1 < MINVAL) mix = 0.f;
    else mix = 1.f;
    if (gsolref[2 * g1] > 0.f && gsolref[2 * g2] > 0.f)
      for (int k = 0; k < 2; ++k) solref[k] = mix * gsolref[2 * g1 + k] + (1.f - mix) * gsolref[2 * g2 + k];
    else
      for (int k = 0; k < 2; ++k) solref[k] = fminf(gsolref[2 * g1 + k], gsolref[2 * g2 + k]);
    for (int k = 0; k < 5; ++k) solimp[k] = mix * gsolimp[5 * g1 + k] + (1.f - mix) * gsolimp[5 * g2 + k];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // no early exit: rc[i] must stay a compile-time index (registers, not scratch)
    const int c = first + i;
    if (i >= n || c >= ncm) continue;
    float f9[9];
    make_frame(f9, rc[i].frame);
    const size_t wc = (size_t)w * ncm + c;
    d.contact_dist[wc] = rc[i].dist;
    for (int k = 0; k < 3; ++k) d.contact_pos[3 * wc + k] = rc[i].pos[k];
    for (int k = 0; k < 9; ++k) d.contact_frame[9 * wc + k] = f9[k];
    d.contact_includemargin[wc] = margin - gap;
    float* f5 = d.contact_friction + 5 * wc;
    f5[0] = f5[1] = fri[0]; f5[2] = fri[1]; f5[3] = f5[4] = fri[2];
    for (int k = 0; k < 2; ++k) d.contact_solref[2 * wc + k] = solref[k];
    for (int k = 0; k < 5; ++k) d.contact_solimp[5 * wc + k] = solimp[k];
    d.contact_dim[wc] = condim;
    d.contact_geom[2 * wc] = g1;
    d.contact_geom[2 * wc + 1] = g2;
    d.contact_efc_address[wc] = -1;
  }
}

__global__ __launch_bounds__(64, 4) void k_collision(const Model m, const Data d, const int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int w = blockIdx.x, lane = threadIdx.x;
  if ((flags & FLAG_MASK) && !d.world_mask[w]) return;
  if ((flags & FLAG_FOLD) && d.fold_reuse[w]) return;
  const int ng = m.size.ngeom, npair = m.size.npair;
  const int g0 = m.size.geom_lds0, nl = ng - g0;  // geoms [g0, ng) are staged; s_gx / s_gm are indexed by g - g0
  float* s_gx = smem;
  float* s_gm = s_gx + 3 * nl;
  PROF_INIT();
  float* s_gc = s_gm + 9 * nl;  // per staged geom: type (as int bits), size[3], rbound, margin, gap, -
  const float *gsize = MF(geom_size), *rbound = MF(geom_rbound), *gmargin = MF(geom_margin), *ggap = MF(geom_gap);
  const float *gfri = MF(geom_friction), *gsolref = MF(geom_solref), *gsolimp = MF(geom_solimp), *gsolmix = MF(geom_solmix);
  // The stage is bound by dependent global round trips: every per-geom constant the narrow phase
  // needs goes to LDS in this one batch, and the pair list is fetched one sweep ahead, so a sweep
  // finds all of its operands on chip.
  int ng1 = 0, ng2 = 0;  // geoms of pair (sweep 0, this lane)
  if (lane < npair) { ng1 = m.pair_geom[2 * lane]; ng2 = m.pair_geom[2 * lane + 1]; }
  global_to_lds(s_gx, d.geom_xpos + ((size_t)w * ng + g0) * 3, 3 * nl, lane);
  global_to_lds(s_gm, d.geom_xmat + ((size_t)w * ng + g0) * 9, 9 * nl, lane);
  for (int l = lane; l < nl; l += 64) {
    const int g = g0 + l;
    ((int*)s_gc)[8 * l] = m.geom_type[g];
    for (int k = 0; k < 3; ++k) s_gc[8 * l + 1 + k] = gsize[3 * g + k];
    s_gc[8 * l + 4] = rbound[g];
    s_gc[8 * l + 5] = gmargin[g];
    s_gc[8 * l + 6] = ggap[g];
  }
  __syncthreads();
  PROF_MARK(0);
  // ---- static pairs, pass 1: the cheap bounding test for every pair, survivors compacted IN PAIR
  // ORDER into an LDS list.  Few of the 502 G1 pairs are ever close, so the divergent narrow
  // phase below runs over one or two sweeps instead of eight.
  int* s_near = (int*)(s_gc + 8 * nl);  // (g1 << 16) | g2; aliases the terrain lists (built later)
  int nnear = 0;
  for (int p0 = 0; p0 < npair; p0 += 64) {
    const int p = p0 + lane;
    const int g1 = ng1, g2 = ng2;
    if (p + 64 < npair) { ng1 = m.pair_geom[2 * (p + 64)]; ng2 = m.pair_geom[2 * (p + 64) + 1]; }  // next sweep
    bool near = false;
    if (p < npair) {
      const int l1 = g1 - g0, l2 = g2 - g0;
      const float margin = fmaxf(s_gc[8 * l1 + 5], s_gc[8 * l2 + 5]);
      float dif[3];
      for (int k = 0; k < 3; ++k) dif[k] = s_gx[3 * l2 + k] - s_gx[3 * l1 + k];
      if (((const int*)s_gc)[8 * l1] == MJLAB_GEOM_PLANE) {
        const float z1[3] = {s_gm[9 * l1 + 2], s_gm[9 * l1 + 5], s_gm[9 * l1 + 8]};
        near = dot3(dif, z1) <= margin + s_gc[8 * l2 + 4];
      } else {
        const float bound = margin + s_gc[8 * l1 + 4] + s_gc[8 * l2 + 4];
        near = dot3(dif, dif) <= bound * bound;
      }
    }
    const unsigned long long nm = __ballot(near);
    if (near) s_near[nnear + __popcll(nm & ((1ull << lane) - 1ull))] = (g1 << 16) | g2;
    nnear += __popcll(nm);
  }
  __syncthreads();
  int base = 0;  // contacts emitted so far (wave-uniform)
  // ---- pass 2: narrow phase over the close pairs
  for (int p0 = 0; p0 < nnear; p0 += 64) {
    const int p = p0 + lane;
    RawCon rc[4];
    int n = 0, g1 = 0, g2 = 0;
    float margin = 0.f, gap = 0.f;
    if (p < nnear) {
      const int code = s_near[p];
      g1 = code >> 16; g2 = code & 0xffff;
      const int l1 = g1 - g0, l2 = g2 - g0;
      const int t1 = ((const int*)s_gc)[8 * l1], t2 = ((const int*)s_gc)[8 * l2];
      margin = fmaxf(s_gc[8 * l1 + 5], s_gc[8 * l2 + 5]);
      gap = fmaxf(s_gc[8 * l1 + 6], s_gc[8 * l2 + 6]);
      float p1[3], p2[3], z1[3], z2[3], s1[3], s2[3];
      for (int k = 0; k < 3; ++k) {
        p1[k] = s_gx[3 * l1 + k]; p2[k] = s_gx[3 * l2 + k];
        z1[k] = s_gm[9 * l1 + 3 * k + 2]; z2[k] = s_gm[9 * l2 + 3 * k + 2];
        s1[k] = s_gc[8 * l1 + 1 + k]; s2[k] = s_gc[8 * l2 + 1 + k];
      }
      float dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
      {
        if (t1 == MJLAB_GEOM_PLANE && t2 == MJLAB_GEOM_SPHERE) {
          n = plane_sphere(rc, margin, p1, z1, p2, s2[0]);
        } else if (t1 == MJLAB_GEOM_PLANE && t2 == MJLAB_GEOM_CAPSULE) {
          float q[3];
          RawCon t;
          for (int k = 0; k < 3; ++k) q[k] = p2[k] + z2[k] * s2[1];
          bool hit = plane_sphere(&t, margin, p1, z1, q, s2[0]) != 0;
          for (int k = 0; k < 3; ++k) t.frame[3 + k] = z2[k];
          rc_take(rc[0], t, hit);
          n = hit ? 1 : 0;
          for (int k = 0; k < 3; ++k) q[k] = p2[k] - z2[k] * s2[1];
          hit = plane_sphere(&t, margin, p1, z1, q, s2[0]) != 0;
          for (int k = 0; k < 3; ++k) t.frame[3 + k] = z2[k];
          rc_take(rc[0], t, hit && n == 0);
          rc_take(rc[1], t, hit && n == 1);
          n += hit ? 1 : 0;
        } else if (t1 == MJLAB_GEOM_PLANE && t2 == MJLAB_GEOM_BOX) {
          const float dist = dot3(dif, z1);
          float bm[9];
          for (int k = 0; k < 9; ++k) bm[k] = s_gm[9 * l2 + k];
          for (int i = 0; i < 8 && n < 4; ++i) {
            float vec[3] = {(i & 1) ? s2[0] : -s2[0], (i & 2) ? s2[1] : -s2[1], (i & 4) ? s2[2] : -s2[2]}, corner[3];
            mul_mat_vec3(corner, bm, vec);
            const float ldist = dot3(z1, corner);
            if (dist + ldist > margin || ldist > 0.f) continue;
            RawCon t;
            t.dist = dist + ldist;
            for (int k = 0; k < 3; ++k) {
              t.pos[k] = corner[k] + p2[k] + z1[k] * (-t.dist * 0.5f);
              t.frame[k] = z1[k]; t.frame[3 + k] = 0.f;
            }
            rc_take(rc[0], t, n == 0); rc_take(rc[1], t, n == 1); rc_take(rc[2], t, n == 2); rc_take(rc[3], t, n == 3);
            n++;
          }
        } else if (t1 == MJLAB_GEOM_SPHERE && t2 == MJLAB_GEOM_SPHERE) {
          n = sphere_sphere(rc, margin, p1, s1[0], p2, s2[0]);
        } else if (t1 == MJLAB_GEOM_SPHERE && t2 == MJLAB_GEOM_CAPSULE) {
          float vec[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
          const float x = clipf(dot3(z2, vec), -s2[1], s2[1]);
          for (int k = 0; k < 3; ++k) vec[k] = p2[k] + z2[k] * x;
          n = sphere_sphere(rc, margin, p1, s1[0], vec, s2[0]);
        } else if (t1 == MJLAB_GEOM_CAPSULE && t2 == MJLAB_GEOM_CAPSULE) {
          n = capsule_capsule(rc, margin, p1, z1, s1, p2, z2, s2);
        }
      }
    }
    int total;
    const int off = wave_excl_scan(n, lane, &total);
    if (n > 0) emit_contacts(m, d, w, g1, g2, margin, gap, rc, n, base + off, gfri, gsolref, gsolimp, gsolmix);
    base += total;
  }
  PROF_MARK(1);
  // ---- box terrain: moving spheres / capsules vs static boxes found through the xy grid ----
  const int ntg = m.size.ntgeom;
  if (ntg > 0) {
    __syncthreads();  // the close-pair list shares its LDS with the lists built below
    int* s_cand = (int*)(s_gc + 8 * nl);            // [ntg][TCAND_MAX] box ids, ascending per geom
    int* s_pair = s_cand + ntg * MJLAB_TCAND_MAX;   // flat, ordered candidate list: (ti << 24) | slot
    int* s_pairb = s_pair + ntg * MJLAB_TCAND_MAX;  // the same for moving BOX geoms (own sweep below)
    int pbase = 0, bbase = 0;
    for (int t0 = 0; t0 < ntg; t0 += 64) {          // lanes = moving geoms
      const int ti = t0 + lane;
      int nc = 0;
      bool isbox = false;
      if (ti < ntg) {
        const int g = m.tgeom[ti];
        isbox = ((const int*)s_gc)[8 * (g - g0)] == MJLAB_GEOM_BOX;
        nc = terrain_walk(m, s_gx + 3 * (g - g0), s_gc[8 * (g - g0) + 4] + s_gc[8 * (g - g0) + 5], s_cand + ti * MJLAB_TCAND_MAX);
      }
      int total, totalb;
      const int off = wave_excl_scan(isbox ? 0 : nc, lane, &total);
      const int offb = wave_excl_scan(isbox ? nc : 0, lane, &totalb);
      int* dst = isbox ? s_pairb + bbase + offb : s_pair + pbase + off;
      for (int q = 0; q < nc; ++q) dst[q] = (ti << 24) | q;
      pbase += total;
      bbase += totalb;
    }
    __syncthreads();
    PROF_MARK(3);
    for (int p0 = 0; p0 < pbase; p0 += 64) {        // lanes = candidate (geom, box) pairs
      const int p = p0 + lane;
      RawCon rc[4];
      int n = 0, g = 0, gb = 0;
      float margin = 0.f, gap = 0.f;
      if (p < pbase) {
        const int code = s_pair[p], ti = code >> 24;
        const int b = s_cand[ti * MJLAB_TCAND_MAX + (code & 0xffffff)];
        g = m.tgeom[ti];
        gb = m.tbox_geom[b];
        const int l = g - g0;
        margin = s_gc[8 * l + 5];  // terrain boxes carry no margin / gap (checked when the model is compiled)
        gap = s_gc[8 * l + 6];
        float cp[3], cz[3], cs[3], bpos[3], bmat[9], bsize[3];
        for (int k = 0; k < 3; ++k) {
          cp[k] = s_gx[3 * l + k]; cz[k] = s_gm[9 * l + 3 * k + 2]; cs[k] = s_gc[8 * l + 1 + k];
          bpos[k] = m.tbox_pos[3 * b + k]; bsize[k] = m.tbox_size[3 * b + k];
        }
        for (int k = 0; k < 9; ++k) bmat[k] = m.tbox_mat[9 * b + k];
        if (((const int*)s_gc)[8 * l] == MJLAB_GEOM_SPHERE) n = sphere_box(rc, margin, cp, cs[0], bpos, bmat, bsize);
        else n = capsule_box(rc, margin, cp, cz, cs, bpos, bmat, bsize);
      }
      int total;
      const int off = wave_excl_scan(n, lane, &total);
      if (n > 0) emit_contacts(m, d, w, g, gb, margin, gap, rc, n, base + off, gfri, gsolref, gsolimp, gsolmix);
      base += total;
    }
    // Moving boxes: the 8 corners of the box as points (sphere_box with radius 0), lanes = (pair,
    // corner), 8 pairs per sweep; the first 4 hits of a pair in corner order are kept -- on a face
    // exactly the plane-box contacts.  Not a full box-box test: see DESIGN.md section 7 (row 4).
    for (int p0 = 0; p0 < 8 * bbase; p0 += 64) {
      const int p = p0 + lane, corner_id = lane & 7;
      RawCon rc[4];
      int g = 0, gb = 0;
      float margin = 0.f, gap = 0.f;
      bool hit = false;
      if (p < 8 * bbase) {
        const int code = s_pairb[p >> 3], ti = code >> 24;
        const int b = s_cand[ti * MJLAB_TCAND_MAX + (code & 0xffffff)];
        g = m.tgeom[ti];
        gb = m.tbox_geom[b];
        const int l = g - g0;
        margin = s_gc[8 * l + 5];
        gap = s_gc[8 * l + 6];
        float cp[3], vec[3], corner[3], bpos[3], bmat[9], bsize[3];
        for (int k = 0; k < 3; ++k) {
          const float sz = s_gc[8 * l + 1 + k];
          cp[k] = s_gx[3 * l + k]; vec[k] = ((corner_id >> k) & 1) ? sz : -sz;
          bpos[k] = m.tbox_pos[3 * b + k]; bsize[k] = m.tbox_size[3 * b + k];
        }
        mul_mat_vec3(corner, s_gm + 9 * l, vec);
        for (int k = 0; k < 3; ++k) corner[k] += cp[k];
        for (int k = 0; k < 9; ++k) bmat[k] = m.tbox_mat[9 * b + k];
        hit = sphere_box(rc, margin, corner, 0.f, bpos, bmat, bsize) != 0;
      }
      // rank of this hit among the hits of the same pair (8 consecutive lanes)
      const unsigned long long hits = __ballot(hit);
      const int rank = __popcll(hits & (0xffull << (lane & 56)) & ((1ull << lane) - 1ull));
      const int n = hit && rank < 4 ? 1 : 0;
      int total;
      const int off = wave_excl_scan(n, lane, &total);
      if (n > 0) emit_contacts(m, d, w, g, gb, margin, gap, rc, n, base + off, gfri, gsolref, gsolimp, gsolmix);
      base += total;
    }
  }
  const int ncm = m.size.nconmax;
  if (lane == 0) d.ncon[w] = base < ncm ? base : ncm;
  PROF_MARK(2);
  PROF_FLUSH(d.profile + (size_t)w * 64 + 32);
}

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

__global__ __launch_bounds__(64, 4) void k_velocity(const Model m, const Data d, const int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int w = blockIdx.x, lane = threadIdx.x;
  if ((flags & FLAG_MASK) && !d.world_mask[w]) return;
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
  global_to_lds(s_qvel, d.qvel + (size_t)w * nv, nv, lane);
  global_to_lds(s_cdof, d.cdof + (size_t)w * 6 * nv, 6 * nv, lane);
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
    if (v_type != MJLAB_JNT_FREE && v_stiff != 0.f) passive -= v_stiff * (qpos[v_qadr] - MF(qpos0)[v_qadr]);
    float smooth = passive - bias + v_applied + s_qact[i];
    if (xmask) {
      const float* xfrc = d.xfrc_applied + (size_t)w * 6 * nb;
      const float* xipos = d.xipos + (size_t)w * 3 * nb;
      const float* sub = d.subtree_com + (size_t)w * 3 * nb;
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

// ====================================================================================
// Stage 4: constraints (mj_makeConstraint: joint limits + contacts; contact sensors)
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
__host__ __device__ inline int constraint_nlim(const mjlab_sizes_t& s) { return 2 * s.njnt < s.njmax ? 2 * s.njnt : s.njmax; }
__host__ __device__ inline int constraint_lds_floats(const mjlab_sizes_t& s) {
  return s.nconmax + 2 * constraint_nlim(s) + CC_NROWS * 64;
}

__global__ __launch_bounds__(64, 4) void k_constraint(const Model m, const Data d, const int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int w = blockIdx.x, lane = threadIdx.x;
  if ((flags & FLAG_MASK) && !d.world_mask[w]) return;
  if ((flags & FLAG_FOLD) && d.fold_reuse[w]) return;
  const int nb = m.size.nbody, nv = m.size.nv, nq = m.size.nq, nj = m.size.njnt, ncm = m.size.nconmax, njm = m.size.njmax;
  const int nlim = constraint_nlim(m.size);
  PROF_INIT();
  int* s_cadr = (int*)smem;                  // contact -> first efc row (or -1)
  int* s_ldof = s_cadr + ncm;                // limit row -> dof
  float* s_lsign = (float*)(s_ldof + nlim);  // limit row -> Jacobian entry (+-1)
  float* s_cc = s_lsign + nlim;              // staged contact chunk, [CC_NROWS][64]
  const float timestep = (float)m.opt.timestep;
  float* J = d.efc_J + (size_t)w * njm * nv;
  const size_t wr = (size_t)w * njm;
  int nefc = 0;
  // ---- joint limits: lanes = joints, rows assigned in (joint, side) order
  {
    const float *range = MF(jnt_range), *jmargin = MF(jnt_margin), *jsolref = MF(jnt_solref), *jsolimp = MF(jnt_solimp),
                *dinv = MF(dof_invweight0);
    const float* qpos = d.qpos + (size_t)w * nq;
    const float* qvel = d.qvel + (size_t)w * nv;
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
  const float* sub = d.subtree_com + (size_t)w * 3 * nb;
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
      for (int k = 0; k < 3; ++k) pos[k] = d.contact_pos[3 * wc + k];
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
      if (dim == 1) {
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
      if (dist < inc) nrow = dim == 1 ? 1 : 2 * (dim - 1);
    }
    // efc addresses: contacts take rows in order; one that does not fit is dropped
    int total;
    int adr = nefc + wave_excl_scan(nrow, lane, &total);
    if (nefc + total > njm) {  // rare: replay the sequential rule
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
  if (lane == 0) d.nefc[w] = nefc;
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
__host__ __device__ inline int solve_lds_floats(const mjlab_sizes_t& s) {
  const int nvp = solve_nvp(s.nv), ld = (nvp % 8 == 4) ? nvp : nvp + 4;
  return nvp * ld + nvp + 3 * s.njmax + 64;
}

template <int NVP>
struct SolveCtx {
  static constexpr int NB = CholCfg<NVP>::NB;
  static constexpr int ld = CholCfg<NVP>::LD;
  const float* J;  // global, row-major nefc x nv
  const float* M;  // global, dense nv x nv
  float *s_H, *s_invd, *s_jar, *s_jv, *s_D;
  int nv, nefc, lane;
  float quad_gauss[3];
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
    const bool act = r < c.nefc && c.s_jar[r] < 0.f;
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
        int t = 0;
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
          for (int Jb = 0; Jb <= I; ++Jb) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[I], x[u][Jb], acc[t], 0, 0, 0);
            ++t;
          }
      }
    }
  }
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) { jtf[cb] += __shfl_xor(jtf[cb], 16); jtf[cb] += __shfl_xor(jtf[cb], 32); }
  return pick16<NB>(jtf, c.lane);
}

// H = M + tiles -> LDS (lower triangle only)
template <int NVP>
__device__ __forceinline__ void hessian_store(const SolveCtx<NVP>& c, const f32x4 (&acc)[CholCfg<NVP>::NB * (CholCfg<NVP>::NB + 1) / 2]) {
  constexpr int NB = CholCfg<NVP>::NB;
  const int sub = c.lane >> 4, col = c.lane & 15;
  // per-lane part of the M offset, opaque so that the 4 NT addresses are not hoisted out of
  // the Newton loop as 64-bit VGPR pairs (and then spilled)
  int moff = sub * 4 * c.nv + col;
  asm volatile("" : "+v"(moff));
  int t = 0;
#pragma unroll
  for (int I = 0; I < NB; ++I)
#pragma unroll
    for (int Jb = 0; Jb <= I; ++Jb) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = 16 * I + sub * 4 + k, cc = 16 * Jb + col;
        if (row < c.nv && cc <= row) c.s_H[row * c.ld + cc] = acc[t][k] + c.M[(16 * I + k) * c.nv + 16 * Jb + moff];
      }
      ++t;
    }
}

// cost of row r along the search direction: D/2 (j0 + alpha jv)^2 where that is negative
template <int NVP>
__device__ __forceinline__ void ls_prepare(SolveCtx<NVP>& c) {
  const int r = c.lane;
  float j0 = 1.f, jv = 0.f, Dr = 0.f;  // lanes beyond nefc: never active
  if (r < c.nefc) { j0 = c.s_jar[r]; jv = c.s_jv[r]; Dr = c.s_D[r]; }
  c.lj0 = j0; c.ljv = jv;
  c.lq0 = 0.5f * Dr * j0 * j0; c.lq1 = Dr * j0 * jv; c.lq2 = 0.5f * Dr * jv * jv;
  j0 = 1.f; jv = 0.f; Dr = 0.f;
  if (r + 64 < c.nefc) { j0 = c.s_jar[r + 64]; jv = c.s_jv[r + 64]; Dr = c.s_D[r + 64]; }
  c.mj0 = j0; c.mjv = jv;
  c.mq0 = 0.5f * Dr * j0 * j0; c.mq1 = Dr * j0 * jv; c.mq2 = 0.5f * Dr * jv * jv;
}
__device__ __forceinline__ float ls_newton_step(float alpha, float d0, float d1) {
  return alpha - d0 * __builtin_amdgcn_rcpf(d1);  // 1 ulp reciprocal: alpha only has to meet ls_tolerance
}
template <int NVP>
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
template <int NVP>
__device__ __forceinline__ int update_bracket(SolveCtx<NVP>& c, LsPnt* p, const LsPnt* cand, LsPnt* pnext) {
  int flag = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (p->d0 < 0.f && cand[i].d0 < 0.f && p->d0 < cand[i].d0) { *p = cand[i]; flag = 1; }
    else if (p->d0 > 0.f && cand[i].d0 > 0.f && p->d0 > cand[i].d0) { *p = cand[i]; flag = 2; }
  }
  if (flag) ls_eval(c, pnext, ls_newton_step(p->alpha, p->d0, p->d1));
  return flag;
}
// exact 1-D line search on the piecewise-quadratic cost (safeguarded Newton + bracketing)
template <int NVP>
__device__ float line_search(SolveCtx<NVP>& c, float gtol, int lsmax) {
  LsPnt p0, p1, p2, pmid, p1next, p2next;
  c.ls_iter = 0;
  ls_prepare(c);
  ls_eval(c, &p0, 0.f);
  ls_eval(c, &p1, ls_newton_step(p0.alpha, p0.d0, p0.d1));
  if (p0.cost < p1.cost) p1 = p0;
  if (fabsf(p1.d0) < gtol) return p1.alpha;
  const float dir = p1.d0 < 0.f ? 1.f : -1.f;
  bool p2update = false;
  p2 = p1;
  while (p1.d0 * dir <= -gtol && c.ls_iter < lsmax) {
    p2 = p1;
    p2update = true;
    ls_eval(c, &p1, ls_newton_step(p1.alpha, p1.d0, p1.d1));
    if (fabsf(p1.d0) < gtol) return p1.alpha;
  }
  if (c.ls_iter >= lsmax) return p1.alpha;
  if (!p2update) return p1.alpha;
  p2next = p1;
  ls_eval(c, &p1next, ls_newton_step(p1.alpha, p1.d0, p1.d1));
  while (c.ls_iter < lsmax) {
    ls_eval(c, &pmid, 0.5f * (p1.alpha + p2.alpha));
    LsPnt cand[3] = {p1next, p2next, pmid};
    float bestcost = 0.f, bestalpha = 0.f;
    bool found = false;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (fabsf(cand[i].d0) < gtol && (!found || cand[i].cost < bestcost)) { bestcost = cand[i].cost; bestalpha = cand[i].alpha; found = true; }
    if (found) return bestalpha;
    const int b1 = update_bracket(c, &p1, cand, &p1next);
    const int b2 = update_bracket(c, &p2, cand, &p2next);
    if (!b1 && !b2) return pmid.cost < p0.cost ? pmid.alpha : 0.f;
  }
  if (p1.cost <= p2.cost && p1.cost < p0.cost) return p1.alpha;
  if (p2.cost <= p1.cost && p2.cost < p0.cost) return p2.alpha;
  return 0.f;
}

// constraint cost sum_r s(jar_r) over rows held in LDS
__device__ __forceinline__ float constraint_cost(const float* s_jar, const float* s_D, int nefc, int lane) {
  float cost = 0.f;
  for (int r = lane; r < nefc; r += 64) {
    const float x = s_jar[r];
    if (x < 0.f) cost += 0.5f * s_D[r] * x * x;
  }
  return wave_sum(cost);
}

// The kernel is written as a small state machine around ONE factor + substitution site:
//   ST_SMOOTH     s_H = M,            rhs = qfrc_smooth          -> qacc_smooth
//   ST_NEWTON     s_H = H (if new),   rhs = gradient             -> search direction, line
//                 search, update, convergence test (repeats)
//   ST_INTEGRATE  s_H = M + h*diag,   rhs = qfrc_smooth + J^T f  -> implicit acceleration
// so the fully unrolled factorization is inlined exactly once: no call ABI, no callee-saved
// registers through scratch memory, and the register allocator sees the whole kernel.
enum { ST_SMOOTH = 0, ST_NEWTON = 1, ST_PREP_INTEGRATE = 2, ST_INTEGRATE = 3 };

template <int NVP>
__global__ __launch_bounds__(64, 4) void k_solve_integrate(const Model m, const Data d, const int do_solve, const int do_integrate, const int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NB = CholCfg<NVP>::NB, ld = CholCfg<NVP>::LD;
  const int w = blockIdx.x, lane = threadIdx.x;
  if ((flags & FLAG_MASK) && !d.world_mask[w]) return;
  const int nv = m.size.nv, nq = m.size.nq, nu = m.size.nu, nj = m.size.njnt, njm = m.size.njmax;
  SolveCtx<NVP> c;
  c.s_H = smem;
  c.s_invd = c.s_H + NVP * ld;
  c.s_jar = c.s_invd + NVP;
  c.s_jv = c.s_jar + njm;
  c.s_D = c.s_jv + njm;
  float* s_vec = c.s_D + njm;  // 64 floats of scratch (new qvel for the position update)
  c.J = d.efc_J + (size_t)w * njm * nv;
  c.M = d.qM + (size_t)w * nv * nv;
  c.nv = nv; c.lane = lane;
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
  float qacc = 0.f, fc = 0.f, qas = 0.f, Ma = 0.f, cost = 0.f, gauss = 0.f, rhs = 0.f;
  int iter = 0, state;
  bool need_factor = true;
  PROF_INIT();

  if (do_solve) {
    // mj_factorM + qacc_smooth = M^-1 qfrc_smooth
    dense_global_to_lds(c.s_H, c.M, nv, ld, lane, true);
    chol_pad_rows<NVP>(c.s_H, nv, lane);
    chol_pad_diag<NVP>(c.s_H, nv, lane);
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
        dense_global_to_lds(c.s_H, c.M, nv, ld, lane, true);
        chol_pad_rows<NVP>(c.s_H, nv, lane);
        chol_pad_diag<NVP>(c.s_H, nv, lane);
        __syncthreads();
        if (own) c.s_H[lane * ld + lane] += h * diag;
        rhs = own ? qs + fc : 0.f;
        need_factor = true;
      } else {
        skip_solve = true;  // explicit Euler without damping: a = qacc
      }
    }

    float x = qacc;
    if (!skip_solve) {
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

    if (state == ST_INTEGRATE) {
      // velocity / position update with acceleration x (mj_Euler / mj_implicit tail)
      if (own) {
        const size_t wvi = (size_t)w * nv + launder(lane);
        const float qv = d.qvel[wvi] + h * x;
        d.qvel[wvi] = qv;
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
        const float Ma_ws = symm_mul_global<NVP>(c.M, nv, ws, lane);
        const float cost_ws = constraint_cost(c.s_jar, c.s_D, nefc, lane) + wave_sum(own ? 0.5f * (Ma_ws - qs) * (ws - qas) : 0.f);
        const float cost_s = constraint_cost(c.s_jv, c.s_D, nefc, lane);
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
        cost = constraint_cost(c.s_jar, c.s_D, nefc, lane);
        gauss = wave_sum(own ? 0.5f * (Ma - qs) * (qacc - qas) : 0.f);
        cost += gauss;
        {
          __syncthreads();
          int* s_act = (int*)c.s_jv;
          const int nact = build_active_list<NVP>(c, s_act);
          __syncthreads();
          f32x4 htile[NB * (NB + 1) / 2];
          fc = hessian_accum<NVP, true>(c, htile, s_act, nact);
          rhs = own ? Ma - qs - fc : 0.f;
          hessian_store<NVP>(c, htile);
          chol_pad_diag<NVP>(c.s_H, nv, lane);
        }
        PROF_MARK(3);
        need_factor = true;
        state = ST_NEWTON;
      }
    } else {
      // ---- ST_NEWTON: x = H^-1 grad -> line search along -x, update, convergence test
      const float search = own ? -x : 0.f;
      const float snorm = sqrtf(wave_sum(search * search));
      float alpha = 0.f, Mv = 0.f;
      if (snorm >= MINVAL) {
        const float gtol = tol * lstol * snorm * mi * nvf;
        Mv = symm_mul_global<NVP>(c.M, nv, search, lane);
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
        alpha = line_search<NVP>(c, gtol, lsmax);
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
          changed |= (o < 0.f) != (nw < 0.f);
          c.s_jar[r] = nw;
        }
        const bool any_changed = __ballot(changed) != 0ull;
        __syncthreads();
        const float oldcost = cost;
        cost = constraint_cost(c.s_jar, c.s_D, nefc, lane);
        gauss = wave_sum(own ? 0.5f * (Ma - qs) * (qacc - qas) : 0.f);
        cost += gauss;
        // One pass over J gives J^T f for the convergence test and, if the active set
        // changed, the new Hessian tiles (kept in registers).  Laying H out in LDS and its
        // factorization happen only when another iteration follows; with an unchanged active
        // set H is unchanged and the factor in LDS is reused.  (The two branches are spelled
        // out so that the 24 tile registers are live only inside the branch that needs them.)
        iter++;
        int* s_act = (int*)c.s_jv;  // J search is dead until the next line search
        const int nact = build_active_list<NVP>(c, s_act);
        __syncthreads();
        if (any_changed) {
          f32x4 htile[NB * (NB + 1) / 2];
          fc = hessian_accum<NVP, true>(c, htile, s_act, nact);
          rhs = own ? Ma - qs - fc : 0.f;
          const float improvement = scale * (oldcost - cost);
          const float gradient = scale * sqrtf(wave_sum(rhs * rhs));
          finished = improvement < tol || gradient < tol || iter >= maxiter;
          if (!finished) {
            __syncthreads();
            hessian_store<NVP>(c, htile);
            chol_pad_diag<NVP>(c.s_H, nv, lane);
          }
          need_factor = true;
        } else {  // same active set -> same H -> the factor in LDS is still valid
          f32x4 unused[NB * (NB + 1) / 2];
          fc = hessian_accum<NVP, false>(c, unused, s_act, nact);
          rhs = own ? Ma - qs - fc : 0.f;
          const float improvement = scale * (oldcost - cost);
          const float gradient = scale * sqrtf(wave_sum(rhs * rhs));
          finished = improvement < tol || gradient < tol || iter >= maxiter;
          need_factor = false;
        }
        PROF_MARK(7);
      }
    }
    if (finished) {  // publish the solve, then hand over to the integrator
      if (lane == 0) d.solver_niter[w] = iter;
      for (int r = launder(lane); r < nefc; r += 64) {
        const float xr = c.s_jar[r];
        d.efc_force[wr + r] = xr < 0.f ? -c.s_D[r] * xr : 0.f;
      }
      if (own) {
        const size_t wvp = (size_t)w * nv + launder(lane);
        d.qacc[wvp] = qacc;
        d.qacc_warmstart[wvp] = qacc;
        d.qfrc_constraint[wvp] = fc;
      }
      PROF_MARK(8);
      state = ST_PREP_INTEGRATE;
    }
  }
  if (do_integrate && lane == 0) d.fold_valid[w] = 0;  // the state moved on
  PROF_FLUSH(d.profile + (size_t)w * 64);
}

// forward(): remember the qpos / qvel the pass was computed from (see FLAG_FOLD in k_position)
__global__ __launch_bounds__(64) void k_fold_snapshot(const Model m, const Data d, const int flags) {
  const int w = blockIdx.x, lane = threadIdx.x;
  if ((flags & FLAG_MASK) && !d.world_mask[w]) return;
  const int nq = m.size.nq, nv = m.size.nv;
  for (int i = lane; i < nq; i += 64) d.sh_qpos[(size_t)w * nq + i] = d.qpos[(size_t)w * nq + i];
  for (int i = lane; i < nv; i += 64) d.sh_qvel[(size_t)w * nv + i] = d.qvel[(size_t)w * nv + i];
  if (lane == 0) d.fold_valid[w] = 1;
}

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

__global__ __launch_bounds__(64) void k_entity_readback(const Model m, const Data d, const mjlab_entity_view_t v) {
  const int w = blockIdx.x, lane = threadIdx.x;
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

// ====================================================================================
// Masked termination + reset (extension, see include/mjlab_amd.h)
// ====================================================================================
__device__ __forceinline__ float nan_to_num_dev(float x) {
  if (x != x) return 0.f;
  if (x > 3.402823466e+38f) return 3.402823466e+38f;
  if (x < -3.402823466e+38f) return -3.402823466e+38f;
  return x;
}
__global__ __launch_bounds__(64) void k_masked_reset(const Model m, const Data d, const float* key_qpos, const float* rnd3,
                                                      int* episode_length, const int max_len, const float min_height, int* reset_mask,
                                                      const float* env_origins, const float min_up_z) {
  const int w = blockIdx.x, lane = threadIdx.x;
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
  const bool fell = has_free && (qpos[2] - org[2] < min_height || 1.f - 2.f * (qpos[4] * qpos[4] + qpos[5] * qpos[5]) < min_up_z);
  const bool reset = __ballot(bad) != 0ull || fell || elen >= max_len;
  if (reset) {
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
  }
}

// ====================================================================================
// repeat_array_kernel replacement (reference src/mjlab/sim/randomization.py:9-17)
// ====================================================================================
template <typename T>
__global__ void k_tile(T* dst, const T* src, long long nelem, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    dst[i] = src[i % nelem];
}

// self-test of the DPP reductions against the ds_bpermute versions
__global__ void k_selftest(const float* in, int* nerr) {
  const float v = in[blockIdx.x * 64 + threadIdx.x];
  const float a = wave_sum(v), b = wave_sum_shfl(v);
  const float c = group16_sum(v), e = group16_sum_shfl(v);
  const float tol = 1e-4f * (1.f + fabsf(b));
  if (fabsf(a - b) > tol || fabsf(c - e) > 1e-4f * (1.f + fabsf(e))) atomicAdd(nerr, 1);
}

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

int mjlab_lds_bytes(const mjlab_model_t* m, int stage) {
  switch (stage) {
    case MJLAB_STAGE_POSITION: return 4 * position_lds_floats(m->size);
    case MJLAB_STAGE_COLLISION: return 4 * collision_lds_floats(m->size);
    case MJLAB_STAGE_VELOCITY: return 4 * velocity_lds_floats(m->size);
    case MJLAB_STAGE_CONSTRAINT: return 4 * constraint_lds_floats(m->size);
    case MJLAB_STAGE_SOLVE:
    case MJLAB_STAGE_INTEGRATE: return 4 * solve_lds_floats(m->size);
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
  if (m->opt.cone != 0) return fail(-5, "only the pyramidal friction cone is implemented");
  if (m->opt.integrator != MJLAB_INT_EULER && m->opt.integrator != MJLAB_INT_IMPLICITFAST)
    return fail(-6, "integrator must be Euler or implicitfast");
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
  const int lds = solve_lds_floats(m->size);
  switch (solve_nvp(m->size.nv)) {
    case 8: LAUNCH(k_solve_integrate<8>, lds, *m, *d, do_solve, do_integrate, flags); break;
    case 16: LAUNCH(k_solve_integrate<16>, lds, *m, *d, do_solve, do_integrate, flags); break;
    case 20: LAUNCH(k_solve_integrate<20>, lds, *m, *d, do_solve, do_integrate, flags); break;
    case 24: LAUNCH(k_solve_integrate<24>, lds, *m, *d, do_solve, do_integrate, flags); break;
    case 32: LAUNCH(k_solve_integrate<32>, lds, *m, *d, do_solve, do_integrate, flags); break;
    case 36: LAUNCH(k_solve_integrate<36>, lds, *m, *d, do_solve, do_integrate, flags); break;
    case 40: LAUNCH(k_solve_integrate<40>, lds, *m, *d, do_solve, do_integrate, flags); break;
    case 48: LAUNCH(k_solve_integrate<48>, lds, *m, *d, do_solve, do_integrate, flags); break;
    case 64: LAUNCH(k_solve_integrate<64>, lds, *m, *d, do_solve, do_integrate, flags); break;
    default: return fail(-3, "nv must be in [1, 64]");
  }
  return 0;
}

static int forward_stages_impl(const mjlab_model_t* m, const mjlab_data_t* d, int stages, int flags, void* stream) {
  int rc = check_model(m);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (stages & MJLAB_STAGE_POSITION) LAUNCH(k_position, position_lds_floats(m->size), *m, *d, flags);
  if (stages & MJLAB_STAGE_COLLISION) LAUNCH(k_collision, collision_lds_floats(m->size), *m, *d, flags);
  if (stages & MJLAB_STAGE_VELOCITY) LAUNCH(k_velocity, velocity_lds_floats(m->size), *m, *d, flags);
  if (stages & MJLAB_STAGE_CONSTRAINT) LAUNCH(k_constraint, constraint_lds_floats(m->size), *m, *d, flags);
  if (stages & (MJLAB_STAGE_SOLVE | MJLAB_STAGE_INTEGRATE)) {
    rc = launch_solve(m, d, (stages & MJLAB_STAGE_SOLVE) != 0, (stages & MJLAB_STAGE_INTEGRATE) != 0, flags, st);
    if (rc) return rc;
  }
  if (flags & FLAG_SNAPSHOT) LAUNCH(k_fold_snapshot, 0, *m, *d, flags);
  return 0;
}

// process-wide switch for "forward folded into the next step" (default on)
static int g_fold = 1;
int mjlab_set_fold(int enable) {
  const int old = g_fold;
  g_fold = enable != 0;
  return old;
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
  for (int k = 0; k < nsubstep; ++k) {
    int rc = forward_stages_impl(m, d, MJLAB_STAGE_STEP, (k == 0 && g_fold) ? FLAG_FOLD : 0, stream);
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
  hipMemcpyAsync(din, h, sizeof(h), hipMemcpyHostToDevice, st);
  hipMemsetAsync(derr, 0, sizeof(int), st);
  hipLaunchKernelGGL(k_selftest, dim3(nblk), dim3(64), 0, st, din, derr);
  hipMemcpyAsync(&herr, derr, sizeof(int), hipMemcpyDeviceToHost, st);
  hipError_t e = hipStreamSynchronize(st);
  hipFree(din);
  hipFree(derr);
  if (e != hipSuccess) return fail((int)e, "selftest failed to run");
  if (herr != 0) return fail(-12, "selftest: DPP wave reductions disagree with the shuffle reference");
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
