// common.h -- wave-level helpers, small math, register-resident LDL^T factor / substitution.
// Part of kernels.h (included there, in this order, by every translation unit of the library); not a
// stand-alone header.
#pragma once

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
// Issue priority of this wave among the (up to four) waves of its SIMD, from the number of constraint rows of its world.
// A launch ends with its slowest waves, and those are the worlds with many rows: beyond 64 rows every per-row loop takes
// a second trip and the line search leaves its register-cached rows.  Letting them issue ahead of the cheap worlds they
// share a SIMD with (which have slack anyway) shortens the launch by 8-10 %; results do not depend
// on it.  Set when a world's row count is known (start of every solve) and kept until the next solve.
// With thresholds from the host (data.sched_thr: quantiles of the score over all worlds, refreshed every few steps) the
// classes follow the batch at hand -- a Go1 world never has 32 rows, its expensive worlds are the ones whose solver
// needs more iterations; without them the row-count thresholds above apply.
#ifndef MJLAB_NO_WAVE_PRIORITY
__device__ __forceinline__ void wave_priority(int nefc, int niter_prev, const int* thr) {
  int x = nefc, t1 = 32, t2 = 64, t3 = 80;
  if (thr[2] > 0) { x = nefc * (niter_prev + 2); t1 = thr[0]; t2 = thr[1]; t3 = thr[2]; }
  if (x > t3) __builtin_amdgcn_s_setprio(3);
  else if (x > t2) __builtin_amdgcn_s_setprio(2);
  else if (x > t1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}
#else
__device__ __forceinline__ void wave_priority(int, int, const int*) {}
#endif
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
// Minimum over the wave, wave-uniform.  Same DPP pattern; lanes a step does not write keep their own value (`old` operand).
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_mov_keep(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, BOUND));
}
__device__ __forceinline__ float wave_min(float v) {
  v = fminf(v, dpp_mov_keep<0xB1, 0xF, true>(v));
  v = fminf(v, dpp_mov_keep<0x4E, 0xF, true>(v));
  v = fminf(v, dpp_mov_keep<0x141, 0xF, true>(v));
  v = fminf(v, dpp_mov_keep<0x140, 0xF, true>(v));
  v = fminf(v, dpp_mov_keep<0x142, 0xA, false>(v));  // rows 1,3: min with lane 15 of rows 0,2
  v = fminf(v, dpp_mov_keep<0x143, 0xC, false>(v));  // rows 2,3: min with lane 31
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// value of lane `src` (wave-uniform, not a compile-time constant)
__device__ __forceinline__ float lane_bcast_dyn(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_amdgcn_readfirstlane(src)));
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
// ---- copies between global memory and LDS.  A rolled `dst[k] = src[k]` loop over generic pointers cannot be pipelined by the
// compiler (the store of one trip may alias the load of the next as far as it can tell), so every trip waits for its own memory
// round trip -- ~600 cycles from L2 per 64 floats on the way in, an LDS round trip per 64 floats on the way out.  With
// MJLAB_COPY_BATCH > 1 that many independent accesses are issued before the first dependent instruction (copies only: results
// are bit-identical).  MEASURED (round 4, profiles/r04_v8, r04_v9): with batches of 4 the solve stage's load of M drops from 17.7 k
// to 8.4 k cycles on one wave per SIMD and the four pre-solve STAGE KERNELS get 4-8 % shorter (418.5 k -> 412.7 k cycles per step
// summed over the stage kernels), but the fused control kernel -- the default path, at the 128-register cap -- is 0.4 % SLOWER
// (3.341 vs 3.356 M env-steps/s, three interleaved repetitions): the four extra live registers per copy move its spills into the
// Newton loop (LS prep +4 k, post-LS +3 k cycles).  The default therefore stays 1 (the rolled loop); the switch is the record.
#ifndef MJLAB_COPY_BATCH
#define MJLAB_COPY_BATCH 1
#endif
constexpr int CPB = MJLAB_COPY_BATCH;
#ifndef MJLAB_COPY_BATCH_M
#define MJLAB_COPY_BATCH_M MJLAB_COPY_BATCH
#endif
constexpr int CPBM = MJLAB_COPY_BATCH_M;  // the solve stage's load of M (its register allocation is the tightest of the kernel)
__device__ __forceinline__ void lds_to_global(float* dst, const float* src, int n, int lane) {
  for (int k0 = lane; k0 < n; k0 += 64 * CPB) {
    float v[CPB];
#pragma unroll
    for (int u = 0; u < CPB; ++u) { const int k = k0 + 64 * u; v[u] = k < n ? src[k] : 0.f; }
#pragma unroll
    for (int u = 0; u < CPB; ++u) { const int k = k0 + 64 * u; if (k < n) dst[k] = v[u]; }
  }
}
// rows of 3 (positions in the world's local frame) -> public world-frame array: element k gets org[k mod 3] added
__device__ __forceinline__ void lds3_to_global_org(float* dst, const float* src, int n, int lane, const float (&org)[3]) {
  for (int k0 = lane; k0 < n; k0 += 64 * CPB) {
    float v[CPB];
#pragma unroll
    for (int u = 0; u < CPB; ++u) { const int k = k0 + 64 * u; v[u] = k < n ? src[k] : 0.f; }
#pragma unroll
    for (int u = 0; u < CPB; ++u) {
      const int k = k0 + 64 * u;
      const int c = k - 3 * (k / 3);
      if (k < n) dst[k] = v[u] + (c == 0 ? org[0] : c == 1 ? org[1] : org[2]);
    }
  }
}
__device__ __forceinline__ void global_to_lds(float* dst, const float* src, int n, int lane) {
  for (int k0 = lane; k0 < n; k0 += 64 * CPB) {
    float v[CPB];
#pragma unroll
    for (int u = 0; u < CPB; ++u) { const int k = k0 + 64 * u; v[u] = k < n ? src[k] : 0.f; }
#pragma unroll
    for (int u = 0; u < CPB; ++u) { const int k = k0 + 64 * u; if (k < n) dst[k] = v[u]; }
  }
}

// ---- global -> LDS without registers (round 5): gfx950's LDS-DMA load (`global_load_lds_dword`: each lane names its own global
// address, the data lands at a wave-uniform LDS base + 4 * lane).  No destination VGPR means every trip of a copy can be in flight
// at once whatever the kernel's register allocation looks like -- the one way to batch the copies of the fused control kernel, which
// sits on the 128-register cap (MJLAB_COPY_BATCH above buys the same overlap with registers and loses it again to spills).  The
// compiler counts these loads (vmcnt) and waits before the first LDS access that follows; copies only: results are bit-identical.
// MJLAB_GLDS bit 0: the solve stage's load of M; bit 1: the other stages' prologue copies.
typedef __attribute__((address_space(1))) const void* glds_src_t;
typedef __attribute__((address_space(3))) void* glds_dst_t;
__device__ __forceinline__ void glds_to_lds(float* dst, const float* src, int n, int lane) {
  lane = launder(lane);
  for (int k0 = 0; k0 < n; k0 += 64)
    if (k0 + lane < n) __builtin_amdgcn_global_load_lds((glds_src_t)(src + k0 + lane), (glds_dst_t)(dst + k0), 4, 0, 0);
}
// lower triangle of a dense row-major n x n matrix in global memory -> packed lower triangle in LDS (element (i, j), j <= i, at
// i (i + 1) / 2 + j): lane k of a trip fetches the element whose packed index is k0 + k.  (i, j) from the packed index by a square
// root (exact for these sizes after one correction either way).
__device__ __forceinline__ void glds_dense_to_packed(float* packed, const float* src, int n, int lane) {
  lane = launder(lane);
  const int total = (n * (n + 1)) >> 1;
  for (int k0 = 0; k0 < total; k0 += 64) {
    const int k = k0 + lane;
    int i = (int)((sqrtf((float)(8 * k + 1)) - 1.f) * 0.5f);
    if (((i + 1) * (i + 2)) >> 1 <= k) ++i;
    if ((i * (i + 1)) >> 1 > k) --i;
    const int j = k - ((i * (i + 1)) >> 1);
    if (k < total) __builtin_amdgcn_global_load_lds((glds_src_t)(src + i * n + j), (glds_dst_t)(packed + k0), 4, 0, 0);
  }
}

// Dense n x n matrix copies between row-major global memory (leading dimension n) and LDS
// (leading dimension ld); lanes walk consecutive global elements.  (i, j) of element k advance by 64 per step:
// j += 64 mod n, i += 64 div n, one conditional carry -- no division and no data-dependent loop inside the batches.
struct RowCol {
  int i, j, q, r, n;
  __device__ __forceinline__ RowCol(int lane, int n_) : n(n_) {
    q = 64 / n_; r = 64 - q * n_;  // wave-uniform
    i = lane / n_; j = lane - i * n_;
  }
  __device__ __forceinline__ void next() {
    j += r; i += q;
    if (j >= n) { j -= n; ++i; }
  }
};
__device__ __forceinline__ void dense_global_to_lds(float* dst, const float* src, int n, int ld, int lane, bool lower_only) {
  lane = launder(lane);
  RowCol rc(lane, n);
  for (int k0 = lane; k0 < n * n; k0 += 64 * CPBM) {
    float v[CPBM];
#pragma unroll
    for (int u = 0; u < CPBM; ++u) { const int k = k0 + 64 * u; v[u] = k < n * n ? src[k] : 0.f; }
#pragma unroll
    for (int u = 0; u < CPBM; ++u) {
      if (k0 + 64 * u < n * n && (!lower_only || rc.j <= rc.i)) dst[rc.i * ld + rc.j] = v[u];
      rc.next();
    }
  }
}
// same copy (lower triangle), plus a packed copy of the lower triangle: element (i, j) at i (i + 1) / 2 + j
__device__ __forceinline__ void dense_global_to_lds_packed(float* dst, float* packed, const float* src, int n, int ld, int lane) {
  lane = launder(lane);
  RowCol rc(lane, n);
  for (int k0 = lane; k0 < n * n; k0 += 64 * CPBM) {
    float v[CPBM];
#pragma unroll
    for (int u = 0; u < CPBM; ++u) { const int k = k0 + 64 * u; v[u] = k < n * n ? src[k] : 0.f; }
#pragma unroll
    for (int u = 0; u < CPBM; ++u) {
      if (k0 + 64 * u < n * n && rc.j <= rc.i) { dst[rc.i * ld + rc.j] = v[u]; packed[((rc.i * (rc.i + 1)) >> 1) + rc.j] = v[u]; }
      rc.next();
    }
  }
}
// packed lower triangle (LDS) -> lower triangle of a dense LDS matrix
__device__ __forceinline__ void packed_to_lds(float* dst, const float* packed, int n, int ld, int lane) {
  int i = 0, tri = 0;  // tri = i (i + 1) / 2
  const int total = (n * (n + 1)) >> 1;
  for (int k = launder(lane); k < total; k += 64) {
    while (k >= tri + i + 1) { tri += i + 1; ++i; }
    dst[i * ld + (k - tri)] = packed[k];
  }
}
__device__ __forceinline__ void dense_lds_to_global(float* dst, const float* src, int n, int ld, int lane, bool zero_upper) {
  RowCol rc(lane, n);
  for (int k0 = lane; k0 < n * n; k0 += 64 * CPB) {
    float v[CPB];
#pragma unroll
    for (int u = 0; u < CPB; ++u) {
      v[u] = (k0 + 64 * u < n * n && !(zero_upper && rc.j > rc.i)) ? src[rc.i * ld + rc.j] : 0.f;
      rc.next();
    }
#pragma unroll
    for (int u = 0; u < CPB; ++u) { const int k = k0 + 64 * u; if (k < n * n) dst[k] = v[u]; }
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
#ifndef MJLAB_CHOL_ACC
#define MJLAB_CHOL_ACC 1
#endif
// the sweep's scheduling fences (between batches and between columns)
#define CHOL_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
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
  // MJLAB_CHOL_ACC pairs of accumulators (default 1 = two partial sums; 2 measured equal): the products of a column's dot product go round-robin into
  // independent v_pk_fma_f32 chains, so the dependent chain of a column is J / (2 ACC) multiply-adds long instead of J / 2
  template <int J, int BI>
  static __device__ __forceinline__ void batches(const float (&a)[NVP], f32x2 (&acc)[MJLAB_CHOL_ACC], float (&cur)[CB], float (&nxt)[CB], lds_f32* A) {
    constexpr int NBJ = (J + CB - 1) / CB, K0 = BI * CB;
    if constexpr (BI < NBJ) {
      if constexpr (BI + 1 < NBJ) load_batch<J, K0 + CB>(A, nxt);
      else if constexpr (J + 1 < NVP) load_batch<J + 1, 0>(A, nxt);
#pragma unroll
      for (int k = 0; k < CB; k += 2) {
        constexpr int dummy = 0; (void)dummy;
        const int slot = ((K0 + k) >> 1) % MJLAB_CHOL_ACC;
        if (K0 + k + 1 < J) {
          const f32x2 av = {a[K0 + k], a[K0 + k + 1]};
          const f32x2 sv = {cur[k], cur[k + 1]};
          acc[slot] -= av * sv;
        } else if (K0 + k < J) {
          acc[slot].x -= a[K0 + k] * cur[k];
        }
      }
      CHOL_SCHED_BARRIER();
      batches<J, BI + 1>(a, acc, nxt, cur, A);
    }
  }
  // column J; `cur` holds (or is about to receive) the first batch of row J
  template <int J>
  static __device__ __forceinline__ void col(float (&a)[NVP], float (&cur)[CB], float (&oth)[CB], lds_f32* A, lds_f32* row, lds_f32* s_invd, int rowid) {
    constexpr int NBJ = (J + CB - 1) / CB;
    f32x2 acc[MJLAB_CHOL_ACC];  // products in pairs (v_pk_fma_f32), round-robin over MJLAB_CHOL_ACC independent chains
    acc[0] = (f32x2){a[J], 0.f};
#pragma unroll
    for (int q = 1; q < MJLAB_CHOL_ACC; ++q) acc[q] = (f32x2){0.f, 0.f};
    batches<J, 0>(a, acc, cur, oth, A);
    f32x2 accs = acc[0];
#pragma unroll
    for (int q = 1; q < MJLAB_CHOL_ACC; ++q) if (2 * q < J) accs += acc[q];
    const float t = accs.x + accs.y;
    // pivot clamped from below by one v_med3 (no canonicalising v_max pair), reciprocal = v_rcp_f32 (1 ulp):
    // 3 of the ~14 scalar-like instructions every column costs on top of its multiply-adds
    const float djj = __builtin_amdgcn_fmed3f(lane_bcast(t, J), MINVAL, 3.0e38f);
    const float invd = __builtin_amdgcn_rcpf(djj);
    a[J] = t;
    const float lu = rowid > J ? t * invd : 0.f;
    row[J] = lu;
    s_invd[J] = invd;  // wave-uniform value, every lane stores it to the same address (one instruction, no select)
    if constexpr (J + 1 < NVP) {
      // after NBJ swaps the first batch of row J+1 sits in `cur` (NBJ even) or `oth` (NBJ odd)
      if constexpr (NBJ == 0) {
        load_batch<J + 1, 0>(A, cur);  // J == 0: nothing was in flight
      } else if constexpr (J < CB) {   // entry written after the request: patch from lane J+1
        const float e = lane_bcast(lu, J + 1);
        if constexpr (NBJ % 2 == 0) cur[J] = e; else oth[J] = e;
      }
      CHOL_SCHED_BARRIER();
      if constexpr (NBJ % 2 == 0) col<J + 1>(a, cur, oth, A, row, s_invd, rowid);
      else col<J + 1>(a, oth, cur, A, row, s_invd, rowid);
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
  // Row j of Lu is consumed in batches of CB columns.  The batches are software pipelined
  // through two register buffers: while batch i feeds the FMAs, the reads of batch i+1 --
  // the next batch of the same row, or the first batch of the next row -- are already in
  // flight, so the sweep does not stall on an LDS round trip per column.  The first batch of
  // row j+1 is requested before column j is written; its one missing entry Lu[j+1][j] is
  // patched in from lane j+1's register.
  float bufA[MJLAB_CB], bufB[MJLAB_CB];
  CholSweep<NVP, MJLAB_CB>::template col<0>(a, bufA, bufB, A, row, s_invd, rowid);
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
// ------------------------------------------------------------------------------------
// REGISTER-RESIDENT BLOCKED LDL^T ON MFMA TILES (round 5; lane-level model: tools/chol_mfma_model.py, which this code follows
// operation by operation).  The left-looking sweep above spends its time waiting: every batch of a column's dot product needs an
// LDS broadcast of the factor's row (2.7 exposed round trips per column, 9.4 k cycles per 36 x 36 factorization on one wave per SIMD,
// profiles/r04_v3/latency_table.md), and the Hessian it factors was laid out in LDS from MFMA accumulators just before.  Here the
// matrix never leaves those accumulators.  The symmetric matrix is held as UPPER block tiles U(Jb, I), Jb <= I, in the C / D layout
// of v_mfma_f32_16x16x4_f32:
//     U(Jb, I)[reg r][lane l] = A[16 Jb + 4 kk + r][16 I + j],   kk = l >> 4, j = l & 15
// so ONE REGISTER r of tile row Jb holds four matrix rows {16 Jb + 4 kk + r}, one per 16-lane group -- which is exactly the shape of
// an MFMA A / B operand (K index = lane group).  The columns of a 16-column block are eliminated in that order: panel r = columns
// c(kk) = 16 Jb + 4 kk + r, kk = 0..3, for r = 0..3 (a fixed permutation inside every block; the substitutions walk the same order), and
// a panel's rows are operands AS THEY LIE -- no transposition, no LDS round trip:
//   1. the panel's 4 x 4 diagonal block: 10 v_readlane, a wave-uniform LDL^T of it (4 reciprocals: the only sequential chain) and
//      the inverse of its unit factor;
//   2. W = panel x Linv^T: one MFMA per tile of the block row (A operand: Linv on rows 0, 4, 8, 12; the result's register 0 is W in
//      operand layout), masked to the rows eliminated later, L = W / D;
//   3. L goes to LDS as columns of the factor (for the substitutions; nothing waits for it);
//   4. the trailing update U(J', I) -= W(J') L(I)^T: one MFMA per tile (6 for the G1's first block row, then 3, then none).
// A last block of 4 real columns (NVP 20, 36) is one uniform 4 x 4 factorization.  Same LDL^T, no square roots, the same pivot clamp;
// sums are formed in a different order than in the sweep above, so results differ in the last bits (parity gate: DESIGN.md section 4).
// Factor storage (what chol_solve_tiles reads): C[c][i] = Lu[i][c], column-major with leading dimension LD, natural indices, exactly
// zero unless column c is eliminated before row i.
// ------------------------------------------------------------------------------------
#ifndef MJLAB_CHOL_TILES
#define MJLAB_CHOL_TILES 1
#endif
// Per padded size (measured, profiles/r05_v14/nvp_sweep.txt: chain models with 75-230 rows per world, 4096 worlds, us per physics step,
// tiles | sweep): NVP 8: 109 | 106, 20: 706 | 703, 32: 595 | 664, 36: 754 | 793, 40: 1247 | 962, 48: 1870 | 1885, 64: 3108 | 3230;
// Go1 (NVP 20, few rows): 7.94 | 8.34 M env-steps/s.  The tile factorization pays where the blocks are full (32, 48, 64) or the
// remainder is one 4 x 4 block (36); with 8 real columns in the last block (40) its four half-empty panels cost more than they save, and
// below two blocks the column sweep's few short columns are cheaper than the panels' uniform 4 x 4 chains.  MJLAB_CHOL_TILES_MIN overrides
// (8: tiles everywhere, 99: nowhere) for A/B builds.
#ifndef MJLAB_CHOL_TILES_MIN
#define MJLAB_CHOL_TILES_MIN 0
#endif
__host__ __device__ constexpr bool chol_use_tiles(int nvp) {
  return MJLAB_CHOL_TILES && (MJLAB_CHOL_TILES_MIN ? nvp >= MJLAB_CHOL_TILES_MIN : (nvp == 32 || nvp == 36 || nvp == 48 || nvp == 64));
}
template <int NVP>
struct CholT {
  static constexpr int NB = CholCfg<NVP>::NB, NT = NB * (NB + 1) / 2, LD = CholCfg<NVP>::LD;
  static constexpr bool SMALL_LAST = NVP - 16 * (NB - 1) == 4;  // the last block is one 4 x 4 block (lane group 0, registers 0..3)
  __host__ __device__ static constexpr int tix(int jb, int i) { return i * (i + 1) / 2 + jb; }  // tile (jb, i), jb <= i
  // the s-th column (< NVP) in elimination order
  __host__ __device__ static constexpr int order(int s) {
    int n = 0;
    for (int jb = 0; jb < NB; ++jb) {
      if (SMALL_LAST && jb == NB - 1) {
        for (int r = 0; r < 4; ++r) { if (n == s) return 16 * jb + r; ++n; }
        continue;
      }
      for (int r = 0; r < 4; ++r)
        for (int kk = 0; kk < 4; ++kk) {
          const int c = 16 * jb + 4 * kk + r;
          if (c < NVP) { if (n == s) return c; ++n; }
        }
    }
    return -1;
  }
};
__device__ __forceinline__ float chol_pivot(float x) { return __builtin_amdgcn_fmed3f(x, MINVAL, 3.0e38f); }

// tiles (+)= M: the packed lower triangle in LDS (zero-padded to NVP; stage_solve) or, GLOBAL, the dense row-major matrix in global
// memory; identity beyond nv.  ADD = false overwrites the tiles.  Branch-free: every load is issued by every lane (index clamped into
// the array), the bounds are selects, and off-diagonal tiles address through one per-lane base per block column plus an immediate.
template <int NVP, bool GLOBAL, bool ADD>
__device__ __forceinline__ void tiles_add_M(f32x4 (&T)[CholT<NVP>::NT], const float* sM_, const float* Mg, int nv, int lane) {
  using K = CholT<NVP>;
  const lds_f32* sM = (const lds_f32*)sM_;
  asm volatile("" : "+v"(lane));
  const int kk = lane >> 4, j = lane & 15;
#pragma unroll
  for (int i = 0; i < K::NB; ++i) {
    const bool colin = 16 * i + 16 <= NVP || j < NVP - 16 * i;  // column 16 i + j exists in the padded matrix
    const int col = colin ? 16 * i + j : 0;
#pragma unroll
    for (int jb = 0; jb <= i; ++jb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool rowin = 16 * jb + 16 <= NVP || 4 * kk + r < NVP - 16 * jb;
        const int row = rowin ? 16 * jb + 4 * kk + r : 0;
        int hi = col, lo = row;  // off-diagonal tiles lie above the diagonal: column > row
        if (i == jb) { hi = row > col ? row : col; lo = row > col ? col : row; }
        float v;
        if (GLOBAL) { const bool in = hi < nv; v = Mg[in ? hi * nv + lo : 0]; v = in ? v : 0.f; }
        else v = sM[((hi * (hi + 1)) >> 1) + lo];
        if (!(16 * i + 16 <= NVP)) v = colin ? v : 0.f;
        if (!(16 * jb + 16 <= NVP)) v = rowin ? v : 0.f;
        if (i == jb) v = (j == 4 * kk + r && 16 * jb + j >= nv) ? 1.f : v;  // identity beyond nv
        if (ADD) T[K::tix(jb, i)][r] += v; else T[K::tix(jb, i)][r] = v;
      }
    }
  }
}
// Ends the tiles' live range: every register is "written" by an empty asm statement (no instruction), so that the allocator does not
// keep 6 tiles alive around the solver loop on the paths where they are not refilled (the factor site is one place in a state
// machine; path-insensitive liveness would carry them from one visit to the next).
template <int NT>
__device__ __forceinline__ void tiles_kill(f32x4 (&T)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    asm volatile("" : "=v"(T[t]));
  }
}
// tiles += diag(s[dof]) for dofs < nv (s in LDS: friction-loss curvature, the integrator's h * damping)
template <int NVP>
__device__ __forceinline__ void tiles_add_diag(f32x4 (&T)[CholT<NVP>::NT], const float* s_, int nv, int lane) {
  using K = CholT<NVP>;
  const lds_f32* s = (const lds_f32*)s_;
  asm volatile("" : "+v"(lane));
  const int kk = lane >> 4, j = lane & 15;
#pragma unroll
  for (int jb = 0; jb < K::NB; ++jb) {
    const int dof = 16 * jb + j;
    float v = s[dof < NVP ? dof : 0];
    v = dof < nv ? v : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) T[K::tix(jb, jb)][r] += (j == 4 * kk + r) ? v : 0.f;
  }
}

template <int NVP>
__device__ CHOL_INLINE void chol_factor_tiles(f32x4 (&T)[CholT<NVP>::NT], float* C_, float* s_invd_, int lane) {
  using K = CholT<NVP>;
  constexpr int NB = K::NB, LD = K::LD;
  lds_f32* C = (lds_f32*)C_;
  lds_f32* s_invd = (lds_f32*)s_invd_;
  asm volatile("" : "+v"(lane));  // (per-lane predicates below: formed here, not hoisted out of the caller's solver loop)
  const int kk = lane >> 4, j = lane & 15, jr = j & 3, jq = j >> 2;
  lds_f32* cl = C + 4 * kk * LD + j;  // column 4 kk (+ 16 jb + r), row j (+ 16 i) of the factor
#pragma unroll
  for (int jb = 0; jb < NB; ++jb) {
    if (K::SMALL_LAST && jb == NB - 1) {
      // ---- the last 4 columns 16 jb + r: a[r][r'] = U(jb, jb)[reg r][lane r']
      const f32x4 t = T[K::tix(jb, jb)];
      const float a00 = lane_bcast(t[0], 0), a10 = lane_bcast(t[1], 0), a11 = lane_bcast(t[1], 1), a20 = lane_bcast(t[2], 0), a21 = lane_bcast(t[2], 1),
                  a22 = lane_bcast(t[2], 2), a30 = lane_bcast(t[3], 0), a31 = lane_bcast(t[3], 1), a32 = lane_bcast(t[3], 2), a33 = lane_bcast(t[3], 3);
      const float d0 = chol_pivot(a00), i0 = __builtin_amdgcn_rcpf(d0);
      const float l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
      const float d1 = chol_pivot(a11 - l10 * a10), i1 = __builtin_amdgcn_rcpf(d1);
      const float t21 = a21 - l20 * a10, l21 = t21 * i1, t31 = a31 - l30 * a10, l31 = t31 * i1;
      const float d2 = chol_pivot(a22 - l20 * a20 - l21 * t21), i2 = __builtin_amdgcn_rcpf(d2);
      const float t32 = a32 - l30 * a20 - l31 * t21, l32 = t32 * i2;
      const float d3 = chol_pivot(a33 - l30 * a30 - l31 * t31 - l32 * t32), i3 = __builtin_amdgcn_rcpf(d3);
      const int b0 = 16 * jb;
      if (lane < NVP) {
        C[(b0 + 0) * LD + lane] = lane == b0 + 1 ? l10 : lane == b0 + 2 ? l20 : lane == b0 + 3 ? l30 : 0.f;
        C[(b0 + 1) * LD + lane] = lane == b0 + 2 ? l21 : lane == b0 + 3 ? l31 : 0.f;
        C[(b0 + 2) * LD + lane] = lane == b0 + 3 ? l32 : 0.f;
        C[(b0 + 3) * LD + lane] = 0.f;
      }
      if (lane < 4) s_invd[b0 + lane] = lane == 0 ? i0 : lane == 1 ? i1 : lane == 2 ? i2 : i3;
      continue;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // ---- 1. diagonal block of the panel: a[kk][kk'] = U(jb, jb)[reg r][lane 16 kk + 4 kk' + r]; uniform LDL^T and inverse factor
      const float s = T[K::tix(jb, jb)][r];
      const float a00 = lane_bcast(s, r), a10 = lane_bcast(s, 16 + r), a11 = lane_bcast(s, 20 + r), a20 = lane_bcast(s, 32 + r), a21 = lane_bcast(s, 36 + r),
                  a22 = lane_bcast(s, 40 + r), a30 = lane_bcast(s, 48 + r), a31 = lane_bcast(s, 52 + r), a32 = lane_bcast(s, 56 + r), a33 = lane_bcast(s, 60 + r);
      const float d0 = chol_pivot(a00), i0 = __builtin_amdgcn_rcpf(d0);
      const float l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
      const float d1 = chol_pivot(a11 - l10 * a10), i1 = __builtin_amdgcn_rcpf(d1);
      const float t21 = a21 - l20 * a10, l21 = t21 * i1, t31 = a31 - l30 * a10, l31 = t31 * i1;
      const float d2 = chol_pivot(a22 - l20 * a20 - l21 * t21), i2 = __builtin_amdgcn_rcpf(d2);
      const float t32 = a32 - l30 * a20 - l31 * t21, l32 = t32 * i2;
      const float d3 = chol_pivot(a33 - l30 * a30 - l31 * t31 - l32 * t32), i3 = __builtin_amdgcn_rcpf(d3);
      const float invl = kk == 0 ? i0 : kk == 1 ? i1 : kk == 2 ? i2 : i3;
      const bool below = jr > r || (jr == r && jq > kk);  // rows of the panel's own block that are eliminated later
      float Wm[NB], Lm[NB];
      // ---- 2. W = panel x Linv^T, Linv = inverse of the block's unit factor (uniform, 6 multiply-adds): ONE MFMA per tile of the block
      // row.  A operand: row 4 q of a 16 x 4 matrix = Linv[q][:] (lanes 0 4 8 12 | 20 24 28 | 40 44 | 60), B operand: the panel's
      // register as it lies; register 0 of the result is W in operand layout (lane (g, j): row 16 i + j, column c(g)).  Masked to the
      // rows eliminated later; L = W / D
      const float m10 = -l10, m21 = -l21, m32 = -l32;
      const float m20 = -(l20 + l21 * m10), m31 = -(l31 + l32 * m21), m30 = -(l30 + l31 * m10 + l32 * m20);
      float G = j == 4 * kk ? 1.f : 0.f;  // unit diagonal: lanes 0, 20, 40, 60
      G = lane == 4 ? m10 : G; G = lane == 24 ? m21 : G; G = lane == 8 ? m20 : G;  // (in the order the entries become available)
      G = lane == 44 ? m32 : G; G = lane == 28 ? m31 : G; G = lane == 12 ? m30 : G;
#pragma unroll
      for (int i = jb; i < NB; ++i) {
        const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x4f32(G, T[K::tix(jb, i)][r], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        Wm[i] = (i > jb || below) ? o[0] : 0.f;
        Lm[i] = Wm[i] * invl;
      }
      // ---- 3. columns c(kk) = 16 jb + 4 kk + r of the factor, every row (zero in the blocks above)
      const bool colok = 16 * jb + 4 * kk + r < NVP;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if (16 * i < NVP) {
          const float v = i >= jb ? Lm[i < jb ? jb : i] : 0.f;
          if (colok && (16 * i + 16 <= NVP || j < NVP - 16 * i)) cl[(16 * jb + r) * LD + 16 * i] = v;
        }
      }
      if (colok && j == 0) s_invd[16 * jb + 4 * kk + r] = invl;
      // ---- 4. trailing update of every tile that still holds entries to be eliminated (this block's own row first: the next panel)
#pragma unroll
      for (int jp = jb; jp < NB; ++jp) {
        if (jp == jb && r == 3) continue;
        const float nw = -Wm[jp];
#pragma unroll
        for (int i = jp; i < NB; ++i) T[K::tix(jp, i)] = __builtin_amdgcn_mfma_f32_16x16x4f32(nw, Lm[i], T[K::tix(jp, i)], 0, 0, 0);
      }
    }
  }
}
// Solves A x = b with the factor as chol_factor_tiles left it; lane i owns b_i / x_i (lanes >= n must pass 0).
template <int NVP>
struct CholSolveT {
  using K = CholT<NVP>;
  template <int S>
  static __device__ __forceinline__ void fwd(const float (&a)[NVP], float& b) {
    if constexpr (S < NVP) {
      constexpr int c = K::order(S);
      b = fmaf(-a[c], lane_bcast(b, c), b);  // a[c] = Lu[i][c]: zero unless c is eliminated before i
      fwd<S + 1>(a, b);
    }
  }
  template <int S>
  static __device__ __forceinline__ void bwd(const float (&at)[NVP], float& b) {
    if constexpr (S >= 0) {
      constexpr int c = K::order(S);
      b = fmaf(-at[c], lane_bcast(b, c), b);  // at[c] = Lu[c][i]: zero unless c is eliminated after i
      bwd<S - 1>(at, b);
    }
  }
};
template <int NVP>
__device__ CHOL_INLINE float chol_solve_tiles(const float* C_, const float* s_invd_, int lane, float b) {
  constexpr int LD = CholT<NVP>::LD;
  const lds_f32* C = (const lds_f32*)C_;
  const lds_f32* s_invd = (const lds_f32*)s_invd_;
  const int li = lane < NVP ? lane : NVP - 1;
  const float invd = s_invd[li];
  {
    float a[NVP];
#pragma unroll
    for (int c = 0; c < NVP; ++c) a[c] = C[c * LD + li];  // row li of Lu, one column at a time (unit stride across lanes)
    CholSolveT<NVP>::template fwd<0>(a, b);
  }
  b *= invd;
  __builtin_amdgcn_sched_barrier(0);
  {
    float at[NVP];
#pragma unroll
    for (int q = 0; q < NVP / 4; ++q) {
      const f32x4 v = *(const lds_f32x4*)(C + li * LD + 4 * q);  // column li of Lu = row li of C
      at[4 * q] = v.x; at[4 * q + 1] = v.y; at[4 * q + 2] = v.z; at[4 * q + 3] = v.w;
    }
    CholSolveT<NVP>::template bwd<NVP - 1>(at, b);
  }
  return b;
}

// y_i = sum_j M[i][j] v_j with M symmetric, dense row-major in GLOBAL memory (ld = n) -- worlds with more rows than
// fit in LDS next to a packed copy of M (stage_solve's BIG instantiation) read M this way;
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
// y_i = sum_j M[i][j] v_j with M symmetric; lane i owns v_i and y_i, v_j comes from lane j by v_readlane.
// M is a packed lower triangle in LDS: element (i, j), j <= i, at i (i + 1) / 2 + j.  Lane i reads
// its own row for j <= i and column i of the rows below it for j > i (consecutive addresses across lanes).
template <int NVP>
__device__ __forceinline__ float symm_mul_packed(const float* sM_, int n, float v, int lane) {
  const lds_f32* sM = (const lds_f32*)sM_;
  int li = lane < NVP ? lane : NVP - 1;
  asm volatile("" : "+v"(li));
  const int tri_i = (li * (li + 1)) >> 1;
  float y0 = 0.f, y1 = 0.f;
#pragma unroll
  for (int j = 0; j < NVP; ++j) {
    const float mij = li >= j ? sM[tri_i + j] : sM[((j * (j + 1)) >> 1) + li];
    if (j & 1) y1 = fmaf(mij, lane_bcast(v, j), y1);
    else y0 = fmaf(mij, lane_bcast(v, j), y0);
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

