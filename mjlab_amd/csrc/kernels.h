#pragma once
// kernels.h -- batched MuJoCo-style physics step for MI355X (gfx950 / CDNA4): every device function and kernel.
// Included by mjlab_amd.hip (C ABI, the kernels that do not depend on the padded dof count) and, once per padded dof
// count NVP, by nvp_inst.hip (solve / substep / control-step kernels of that size): ten translation units that
// compile in parallel instead of one that takes four minutes.
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
#define MJLAB_CB 8
#endif
#define MINVAL 1e-15f
#define MINMU 1e-5f  // mjMINMU: the smallest contact friction coefficient (mj_contactParam's clamp)
#define MINIMP 0.0001f
#define MAXIMP 0.9999f
#define MF(name) (m.name + (size_t)w * (size_t)m.name##_ws)


// The kernels live in one translation unit, split by stage for readability:
#include "common.h"  // wave-level helpers, small math, register-resident LDL^T factor / substitution
#include "stage_position.h"  // stage 1: kinematics, comPos, crb, dense M
#include "stage_collision.h"  // stage 2: static pair list + box terrain, analytic primitives
#include "stage_velocity.h"  // stage 3: comVel, rne, actuation, qfrc_smooth
#include "stage_constraint.h"  // stage 4: limits + contacts -> efc rows, contact sensors
#include "stage_solve.h"  // stages 5+6: Newton solver and integration
#include "stage_pgs.h"  // the dual PGS solver (one kernel per stage only)
#include "stage_cone.h"  // the Newton solver with elliptic friction cones (one kernel per stage only)
#include "extras.h"  // fused entity read-back, masked reset, field tiling, self-test

// ====================================================================================
// Fused launches (MJLAB_OPT_FUSE_PRESOLVE / MJLAB_OPT_FUSE_STEP): the same stage bodies back to back
// in one kernel, one world per wave as before.  No kernel boundary between the stages means a fast
// world runs ahead instead of waiting for the slowest wave of every stage, and waves of one SIMD
// drift into different phases (memory-bound prologues of one overlap the arithmetic of another).
// The stages still hand their results over through the public mjData arrays (written anyway);
// __syncthreads() between stages orders those global writes for the wave's other lanes and
// separates the LDS lifetimes (every stage lays out the dynamic LDS block for itself).
// ====================================================================================
// Every stage of a fused kernel gets its arguments through FUSED_ARGS: the world / lane indices and
// the address of the two argument structs are made opaque right before the stage.  Without that the
// optimiser sees one long function (or, with several substeps, one loop body), hoists model constants
// and addresses of LATER stages to the top and keeps them alive -- spilled -- across everything in
// between (the multi-substep kernel: 405 spilled VGPRs, 2x slower than separate launches).  The two
// structs are the first two kernel arguments (kernarg offsets 0 and sizeof(Model), 8-byte aligned);
// reading them through the laundered kernarg pointer keeps the loads scalar.
#define FUSED_ARGS                                                                                                   \
  int w = wsel_, lane = threadIdx.x;                                                                                 \
  unsigned long long ka_ = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();                               \
  asm volatile("" : "+s"(w), "+v"(lane), "+s"(ka_));                                                                 \
  const Model& m = *(const Model*)(const __attribute__((address_space(4))) Model*)(kptr_t)ka_;                       \
  const Data& d = *(const Data*)(const __attribute__((address_space(4))) Data*)((kptr_t)ka_ + sizeof(Model))
typedef const __attribute__((address_space(4))) char* kptr_t;
static_assert(sizeof(Model) % 8 == 0 && alignof(Data) == 8, "kernarg layout assumed by FUSED_ARGS");

// wsel_ = the world this wave works on (blockIdx.x, or mjlab_control_t.world_order[blockIdx.x])
template <bool ELL = false>  // ELL: the constraint stage's elliptic-cone instantiation (the cone variants of the fused kernels, below)
__device__ __forceinline__ void fused_presolve(const int wsel_, const int flags, float* smem) {
  bool reuse;
  { FUSED_ARGS; reuse = stage_position(m, d, w, lane, flags, smem); }
  __syncthreads();
  if (!reuse) {
    { FUSED_ARGS; stage_collision(m, d, w, lane, flags, smem); }
    __syncthreads();
  }
  { FUSED_ARGS; stage_velocity(m, d, w, lane, flags, smem); }
  __syncthreads();
  if (!reuse) {
    { FUSED_ARGS; stage_constraint<ELL>(m, d, w, lane, flags, smem); }
    __syncthreads();
  }
}
#ifdef MJLAB_MAIN_TU
__global__ __launch_bounds__(64, 4) void k_presolve(const Model m_, const Data d_, const int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((flags & FLAG_MASK) && !d_.world_mask[blockIdx.x]) return;
  fused_presolve(blockIdx.x, flags, smem);
}
#endif  // MJLAB_MAIN_TU
// nsub physics steps of this world back to back (nsub > 1: mjlab_step's nsubstep; ctrl / qfrc_applied /
// xfrc_applied are the same for all of them, as in the reference's decimation loop,
// envs/manager_based_rl_env.py:109-114, where the action is fixed during a control step).
// INTEGRATE = false is forward() (one pass, no integration); the two get different kernel names in profiles.
template <int NVP, bool INTEGRATE>
__global__ __launch_bounds__(64, 4) void k_substep(const Model m_, const Data d_, const int flags, const int nsub) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((flags & FLAG_MASK) && !d_.world_mask[blockIdx.x]) return;
  const int wsel_ = blockIdx.x;
  for (int s = 0; s < nsub; ++s) {
    const int f = s == 0 ? flags : (flags & ~FLAG_FOLD);
    fused_presolve(wsel_, f, smem);
    { FUSED_ARGS; stage_solve<NVP>(m, d, w, lane, 1, INTEGRATE ? 1 : 0, f, smem); }
    __syncthreads();
  }
  if (!INTEGRATE && (flags & FLAG_SNAPSHOT)) { FUSED_ARGS; fold_snapshot(m, d, w, lane); }
}

// One CONTROL step of one world per wave (mjlab_control_step): action -> ctrl, nsubstep physics steps,
// termination test + reset, forward(), interval push -- the physics-facing part of the reference's
// ManagerBasedRlEnv.step (envs/manager_based_rl_env.py:106-139) without a kernel boundary in between.
#ifndef MJLAB_WPE
#define MJLAB_WPE 4  // waves per SIMD the fused kernels are compiled for (512 / MJLAB_WPE registers per lane)
#endif
template <int NVP>
__global__ __launch_bounds__(64, MJLAB_WPE) void k_control_step(const Model m_, const Data d_, const mjlab_control_t c, const int fold) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef MJLAB_PROFILE
  const long long t_begin_ = clock64();
#endif
  // Which world this workgroup takes.  The launch ends with its slowest SIMD, and a SIMD's four waves share
  // its issue slots: world_order lets the host deal the expensive worlds (many contacts last step) out evenly
  // over the SIMDs instead of wherever their index happens to land them.
  const int wsel_ = __builtin_amdgcn_readfirstlane(c.world_order ? c.world_order[blockIdx.x] : (int)blockIdx.x);
  if (c.action) {
    FUSED_ARGS;
    const int nu = m.size.nu;
    // product and sum rounded separately (no FMA contraction): bit-identical to the torch expression offset + scale * action
    for (int a = lane; a < nu; a += 64) d.ctrl[(size_t)w * nu + a] = __fadd_rn(c.action_offset[a], __fmul_rn(c.action_scale[a], c.action[(size_t)w * nu + a]));
    __syncthreads();
  }
  // ONE copy of the stage code: passes 0 .. nsubstep-1 are the physics steps, pass nsubstep is forward()
  // (preceded by the termination test + reset); a second inlined copy doubled the kernel to 294 KB of code
  bool reset = false;
  for (int s = 0; s <= c.nsubstep; ++s) {
    const bool fwd = s == c.nsubstep;
    if (fwd) {
      if (c.key_qpos) {
        FUSED_ARGS;
        reset = masked_reset_world(m, d, w, lane, c.key_qpos, c.rnd3, c.episode_length, c.max_len, c.min_height, c.reset_mask, c.env_origins, c.min_up_z,
                                   c.reset_qpos, c.reset_qvel, c.term_ref, c.term_dz, c.term_dup, c.motion);
        __syncthreads();
      }
      if (!(c.forward_mode == 1 || (c.forward_mode == 2 && reset))) break;
    }
    const int f = (s == 0 && fold && !fwd) ? FLAG_FOLD : 0;
    fused_presolve(wsel_, f, smem);
    { FUSED_ARGS; stage_solve<NVP>(m, d, w, lane, 1, fwd ? 0 : 1, f, smem); }
    __syncthreads();
    if (fwd) { FUSED_ARGS; fold_snapshot(m, d, w, lane); }
  }
  if (c.readback_on) {  // EntityData's derived quantities of the state just forwarded, before the push touches qvel
    FUSED_ARGS;
    __syncthreads();
    entity_readback_world(m, d, c.readback, w, lane);
  }
  if (c.push_time_left) {
    FUSED_ARGS;
    __syncthreads();
    if (lane == 0) interval_push_world(m, d, w, c.push_time_left, c.rnd7, c.push_dt, c.push_interval_lo, c.push_interval_hi, c.push_range);
  }
#ifdef MJLAB_PROFILE
  if (threadIdx.x == 0) {
    d_.profile[(size_t)wsel_ * 64 + 63] += (float)(clock64() - t_begin_);  // this world's share of the launch
    // where the wave ran: HW_REG_HW_ID (wave / simd / cu / sh / se fields) and HW_REG_XCC_ID, of workgroup blockIdx.x
    d_.profile[(size_t)blockIdx.x * 64 + 61] = (float)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffff);
    d_.profile[(size_t)blockIdx.x * 64 + 62] = (float)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf);
  }
#endif
}

// ====================================================================================
// The cone variants of the fused launches (mjlab_option_t.cone = MJLAB_CONE_ELLIPTIC; stage_cone.h): the same structures with the
// constraint stage's ELL instantiation and, for the solve, the cone solver followed by the pyramid path's integrator with its solve
// switched off (the hand-over of qacc / qfrc_constraint through the public arrays, as between any two stages).  Kernels of their own --
// the bodies are spelled out a second time rather than shared through a template parameter -- so that the pyramid's kernels, the
// measured path, are the code they were instruction for instruction.  Built at two waves per SIMD (one for the 64-dof instantiation), without
// spills.  The round-5 fault of the four-waves build is explained (profiles/r06_fault): hipcc placed ONE spill store of
// k_substep_cone<32, true> in a divergent loop's exit block ahead of the `s_or_b64 exec` that restores the lanes, so it wrote nothing and the
// reload returned stale scratch; tools/exec_zero_check.py finds that pattern in the code objects and tests/test_code_object.py forbids it.
// ====================================================================================
#ifdef MJLAB_CONE_WPE  // waves per SIMD the cone kernels are compiled for: mjlab_amd/native.py tries 4, then 3, and keeps the first build whose
#define CONE_WAVES(NVP) MJLAB_CONE_WPE  // code object is free of the EXEC == 0 spill-store miscompile (mjlab_amd/code_check.py)
#else  // ... else this default: two waves per SIMD (one for the 64-dof instantiation), where nothing is spilled
#define CONE_WAVES(NVP) ((NVP) <= 48 ? 2 : 1)
#endif
template <int NVP>
__device__ __forceinline__ void fused_solve_cone(const int wsel_, const bool integrate, const int flags, float* smem) {
  { FUSED_ARGS; stage_solve_cone<NVP>(m, d, w, lane, smem); }
  __syncthreads();
  if (integrate) {
    { FUSED_ARGS; stage_solve<NVP>(m, d, w, lane, 0, 1, flags, smem); }
    __syncthreads();
  }
}
template <int NVP, bool INTEGRATE>
__global__ __launch_bounds__(64, CONE_WAVES(NVP)) void k_substep_cone(const Model m_, const Data d_, const int flags, const int nsub) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((flags & FLAG_MASK) && !d_.world_mask[blockIdx.x]) return;
  const int wsel_ = blockIdx.x;
  for (int s = 0; s < nsub; ++s) {
    const int f = s == 0 ? flags : (flags & ~FLAG_FOLD);
    fused_presolve<true>(wsel_, f, smem);
    fused_solve_cone<NVP>(wsel_, INTEGRATE, f, smem);
  }
  if (!INTEGRATE && (flags & FLAG_SNAPSHOT)) { FUSED_ARGS; fold_snapshot(m, d, w, lane); }
}
template <int NVP>
__global__ __launch_bounds__(64, CONE_WAVES(NVP)) void k_control_step_cone(const Model m_, const Data d_, const mjlab_control_t c, const int fold) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wsel_ = __builtin_amdgcn_readfirstlane(c.world_order ? c.world_order[blockIdx.x] : (int)blockIdx.x);
  if (c.action) {
    FUSED_ARGS;
    const int nu = m.size.nu;
    for (int a = lane; a < nu; a += 64) d.ctrl[(size_t)w * nu + a] = __fadd_rn(c.action_offset[a], __fmul_rn(c.action_scale[a], c.action[(size_t)w * nu + a]));
    __syncthreads();
  }
  bool reset = false;
  for (int s = 0; s <= c.nsubstep; ++s) {
    const bool fwd = s == c.nsubstep;
    if (fwd) {
      if (c.key_qpos) {
        FUSED_ARGS;
        reset = masked_reset_world(m, d, w, lane, c.key_qpos, c.rnd3, c.episode_length, c.max_len, c.min_height, c.reset_mask, c.env_origins, c.min_up_z,
                                   c.reset_qpos, c.reset_qvel, c.term_ref, c.term_dz, c.term_dup, c.motion);
        __syncthreads();
      }
      if (!(c.forward_mode == 1 || (c.forward_mode == 2 && reset))) break;
    }
    const int f = (s == 0 && fold && !fwd) ? FLAG_FOLD : 0;
    fused_presolve<true>(wsel_, f, smem);
    fused_solve_cone<NVP>(wsel_, !fwd, f, smem);
    if (fwd) { FUSED_ARGS; fold_snapshot(m, d, w, lane); }
  }
  if (c.readback_on) {
    FUSED_ARGS;
    __syncthreads();
    entity_readback_world(m, d, c.readback, w, lane);
  }
  if (c.push_time_left) {
    FUSED_ARGS;
    __syncthreads();
    if (lane == 0) interval_push_world(m, d, w, c.push_time_left, c.rnd7, c.push_dt, c.push_interval_lo, c.push_interval_hi, c.push_range);
  }
}
