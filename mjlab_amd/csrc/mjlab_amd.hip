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


// The kernels live in one translation unit, split by stage for readability:
#include "common.h"  // wave-level helpers, small math, register-resident LDL^T factor / substitution
#include "stage_position.h"  // stage 1: kinematics, comPos, crb, dense M
#include "stage_collision.h"  // stage 2: static pair list + box terrain, analytic primitives
#include "stage_velocity.h"  // stage 3: comVel, rne, actuation, qfrc_smooth
#include "stage_constraint.h"  // stage 4: limits + contacts -> efc rows, contact sensors
#include "stage_solve.h"  // stages 5+6: Newton solver and integration
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
    { FUSED_ARGS; stage_constraint(m, d, w, lane, flags, smem); }
    __syncthreads();
  }
}
__global__ __launch_bounds__(64, 4) void k_presolve(const Model m_, const Data d_, const int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((flags & FLAG_MASK) && !d_.world_mask[blockIdx.x]) return;
  fused_presolve(blockIdx.x, flags, smem);
}
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
template <int NVP>
__global__ __launch_bounds__(64, 4) void k_control_step(const Model m_, const Data d_, const mjlab_control_t c, const int fold) {
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
    for (int a = lane; a < nu; a += 64) d.ctrl[(size_t)w * nu + a] = c.action_offset[a] + c.action_scale[a] * c.action[(size_t)w * nu + a];
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
        reset = masked_reset_world(m, d, w, lane, c.key_qpos, c.rnd3, c.episode_length, c.max_len, c.min_height, c.reset_mask, c.env_origins, c.min_up_z);
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
  if (c.push_time_left) {
    FUSED_ARGS;
    __syncthreads();
    if (lane == 0) interval_push_world(m, d, w, c.push_time_left, c.rnd7, c.push_dt, c.push_interval_lo, c.push_interval_hi, c.push_range);
  }
#ifdef MJLAB_PROFILE
  if (threadIdx.x == 0) d_.profile[(size_t)wsel_ * 64 + 63] += (float)(clock64() - t_begin_);  // this world's share of the launch
#endif
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

static int max4(int a, int b, int c, int e) { return (a > b ? a : b) > (c > e ? c : e) ? (a > b ? a : b) : (c > e ? c : e); }
static int presolve_lds_floats(const mjlab_sizes_t& s) {
  return max4(position_lds_floats(s), collision_lds_floats(s), velocity_lds_floats(s), constraint_lds_floats(s));
}
static int launch_substep(const mjlab_model_t* m, const mjlab_data_t* d, int do_integrate, int flags, int nsub, hipStream_t st) {
#define SUBSTEP_(N) do { if (do_integrate) LAUNCH((k_substep<N, true>), lds, *m, *d, flags, nsub); else LAUNCH((k_substep<N, false>), lds, *m, *d, flags, 1); } while (0)
  const int a = presolve_lds_floats(m->size), b = solve_lds_floats(m->size), lds = a > b ? a : b;
  switch (solve_nvp(m->size.nv)) {
    case 8: SUBSTEP_(8); break;
    case 16: SUBSTEP_(16); break;
    case 20: SUBSTEP_(20); break;
    case 24: SUBSTEP_(24); break;
    case 32: SUBSTEP_(32); break;
    case 36: SUBSTEP_(36); break;
    case 40: SUBSTEP_(40); break;
    case 48: SUBSTEP_(48); break;
    case 64: SUBSTEP_(64); break;
    default: return fail(-3, "nv must be in [1, 64]");
  }
  return 0;
}

static int forward_stages_impl(const mjlab_model_t* m, const mjlab_data_t* d, int stages, int flags, void* stream) {
  int rc = check_model(m);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int all = MJLAB_STAGE_FORWARD;
  if ((stages & all) == all && (m->opt.flags & MJLAB_OPT_FUSE_STEP)) {  // a whole forward() / step() as ONE launch
    return launch_substep(m, d, (stages & MJLAB_STAGE_INTEGRATE) != 0, flags, 1, st);  // forward(): the snapshot is taken by the same launch
  }
  const int pre = MJLAB_STAGE_POSITION | MJLAB_STAGE_COLLISION | MJLAB_STAGE_VELOCITY | MJLAB_STAGE_CONSTRAINT;
  if ((stages & pre) == pre && (m->opt.flags & MJLAB_OPT_FUSE_PRESOLVE)) {  // the four pre-solve stages as one launch
    LAUNCH(k_presolve, presolve_lds_floats(m->size), *m, *d, flags);
    stages &= ~pre;
  }
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
  if (nsubstep > 1 && (m->opt.flags & MJLAB_OPT_FUSE_STEP)) {  // all substeps in one launch
    int rc = check_model(m);
    if (rc) return rc;
    return launch_substep(m, d, 1, (m->opt.flags & MJLAB_OPT_FOLD_FORWARD) ? FLAG_FOLD : 0, nsubstep, (hipStream_t)stream);
  }
  for (int k = 0; k < nsubstep; ++k) {
    int rc = forward_stages_impl(m, d, MJLAB_STAGE_STEP, (k == 0 && (m->opt.flags & MJLAB_OPT_FOLD_FORWARD)) ? FLAG_FOLD : 0, stream);
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

int mjlab_interval_push(const mjlab_model_t* m, const mjlab_data_t* d, float* time_left, const float* rnd7, float dt,
                        float interval_lo, float interval_hi, const mjlab_push_range_t* range, int root_is_free, void* stream) {
  int rc = check_model(m);
  if (rc) return rc;
  if (!time_left || !rnd7 || !range) return fail(-16, "interval_push: null argument");
  if (!root_is_free || m->size.nq < 7 || m->size.nv < 6) return fail(-17, "interval_push: the first joint must be a free joint");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_interval_push, dim3((m->size.nworld + 63) / 64), dim3(64), 0, st, *m, *d, time_left, rnd7, dt, interval_lo, interval_hi, *range);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "k_interval_push launch failed");
  return 0;
}

int mjlab_control_step(const mjlab_model_t* m, const mjlab_data_t* d, const mjlab_control_t* c, void* stream) {
  int rc = check_model(m);
  if (rc) return rc;
  if (!c || c->nsubstep < 0) return fail(-20, "control_step: bad argument");
  if (c->action && (!c->action_offset || !c->action_scale)) return fail(-20, "control_step: action without offset / scale");
  if (c->key_qpos && (!c->rnd3 || !c->episode_length || !c->reset_mask)) return fail(-15, "control_step: reset arguments missing");
  if (c->push_time_left && (!c->rnd7 || m->size.nq < 7 || m->size.nv < 6)) return fail(-17, "control_step: push needs rnd7 and a free root joint");
  if (c->forward_mode < 0 || c->forward_mode > 2) return fail(-20, "control_step: forward_mode must be 0, 1 or 2");
  hipStream_t st = (hipStream_t)stream;
  const int a = presolve_lds_floats(m->size), b = solve_lds_floats(m->size), lds = a > b ? a : b;
  const int fold = (m->opt.flags & MJLAB_OPT_FOLD_FORWARD) ? 1 : 0;
  switch (solve_nvp(m->size.nv)) {
    case 8: LAUNCH(k_control_step<8>, lds, *m, *d, *c, fold); break;
    case 16: LAUNCH(k_control_step<16>, lds, *m, *d, *c, fold); break;
    case 20: LAUNCH(k_control_step<20>, lds, *m, *d, *c, fold); break;
    case 24: LAUNCH(k_control_step<24>, lds, *m, *d, *c, fold); break;
    case 32: LAUNCH(k_control_step<32>, lds, *m, *d, *c, fold); break;
    case 36: LAUNCH(k_control_step<36>, lds, *m, *d, *c, fold); break;
    case 40: LAUNCH(k_control_step<40>, lds, *m, *d, *c, fold); break;
    case 48: LAUNCH(k_control_step<48>, lds, *m, *d, *c, fold); break;
    case 64: LAUNCH(k_control_step<64>, lds, *m, *d, *c, fold); break;
    default: return fail(-3, "nv must be in [1, 64]");
  }
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
  (void)hipMemcpyAsync(din, h, sizeof(h), hipMemcpyHostToDevice, st);
  (void)hipMemsetAsync(derr, 0, sizeof(int), st);
  hipLaunchKernelGGL(k_selftest, dim3(nblk), dim3(64), 0, st, din, derr);
  (void)hipMemcpyAsync(&herr, derr, sizeof(int), hipMemcpyDeviceToHost, st);
  hipError_t e = hipStreamSynchronize(st);
  (void)hipFree(din);
  (void)hipFree(derr);
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
