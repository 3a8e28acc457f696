/* mjlab_amd.h -- C ABI of the MI355X-native batched physics step.
 *
 * Drop-in boundary for the reference's only hot path:
 *
 *   mjwarp.step(wp_model, wp_data)      reference src/mjlab/sim/sim.py:136,195
 *   mjwarp.forward(wp_model, wp_data)   reference src/mjlab/sim/sim.py:139,187
 *   repeat_array_kernel / expand_model_fields
 *                                       reference src/mjlab/sim/randomization.py:9-55
 *
 * Plain pointers and sizes only (no torch types).  All array pointers inside
 * mjlab_model_t / mjlab_data_t are DEVICE pointers owned by the caller (the Python
 * host side allocates them as torch tensors so that `sim.data.<field>` is zero-copy,
 * reference src/mjlab/sim/sim_data.py:15-64); the structs themselves live on the host
 * and are passed by pointer.  Every call only ENQUEUES work on `stream` (a
 * hipStream_t passed as void*) and returns; nothing synchronises.
 *
 * Return value: 0 on success, otherwise a hipError_t / negative library error;
 * mjlab_last_error() returns a static description of the last failure of the
 * calling thread.
 */
#ifndef MJLAB_AMD_H_
#define MJLAB_AMD_H_

#define MJLAB_REAL float
#define MJLAB_MODEL_T mjlab_model_t
#define MJLAB_DATA_T mjlab_data_t
#include "mjlab_fields.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MJLAB_ABI_VERSION 4 /* 2: the round-3 struct layouts (option: ls_parallel_min_step; sizes: nstaticsite; control: motion, read-back); 3: + the environment terms; 4: + round 6's entry points (mjlab_command_motion_frame, mjlab_command_motion_metrics, mjlab_copy_batch, mjlab_poison_scratch) */

/* stage bits for mjlab_forward_stages (testing / profiling of single stages) */
enum {
  MJLAB_STAGE_POSITION = 1,   /* kinematics, comPos, crb, factorM          */
  MJLAB_STAGE_COLLISION = 2,  /* broad + narrow phase -> contacts          */
  MJLAB_STAGE_VELOCITY = 4,   /* comVel, passive, rne, actuation, qacc_smooth */
  MJLAB_STAGE_CONSTRAINT = 8, /* makeConstraint (friction loss + limits + contacts), contact sensors */
  MJLAB_STAGE_SOLVE = 16,     /* Newton / CG solver (mjlab_option_t.solver) -> qacc, efc_force */
  MJLAB_STAGE_INTEGRATE = 32, /* Euler / implicitfast position+velocity update */
  MJLAB_STAGE_FORWARD = 31,
  MJLAB_STAGE_STEP = 63
};

int mjlab_abi_version(void);
const char* mjlab_last_error(void);

/* Struct layouts as "kind:name:ncol:count," lists in struct order (see mjlab_fields.h). */
const char* mjlab_model_layout(void);
const char* mjlab_data_layout(void);
int mjlab_sizeof_model(void);
int mjlab_sizeof_data(void);
int mjlab_sizeof_option(void); /* sizeof(mjlab_option_t), sizeof(mjlab_sizes_t): bindings that mirror the two host structs by hand check them at load */
int mjlab_sizeof_sizes(void);

/* Replaces mjwarp.step: advance every world by `nsubstep` physics steps
 * (forward dynamics + integration each).  Reference: sim/sim.py:189-195; the
 * decimation loop that calls it 4x per env step is envs/manager_based_rl_env.py:109-114. */
int mjlab_step(const mjlab_model_t* m, const mjlab_data_t* d, int nsubstep, void* stream);

/* Solver / cone options (mjlab_option_t, include/mjlab_fields.h; reference sim/sim.py:43-82): MJLAB_SOL_NEWTON and MJLAB_SOL_CG with the
 * pyramidal cone run under every launch structure and in mjlab_control_step; MJLAB_SOL_PGS with one kernel per stage only;
 * MJLAB_CONE_ELLIPTIC (round 5; MJLAB_SOL_NEWTON only) has kernels of its own for one kernel per stage, MJLAB_OPT_FUSE_STEP and
 * mjlab_control_step -- not for MJLAB_OPT_FUSE_PRESOLVE.  Anything else is refused with a message (mjlab_last_error). */

/* Contact generation.  Plane / sphere / capsule pairs restate MuJoCo's analytic primitives; sphere-box
 * restates mjc_SphereBox.  Two primitives are NOT upstream-identical (documented rules of this library,
 * mirrored by the test oracle): capsule-box (sphere-box contacts of axis points
 * chosen by a convex 1-D search instead of mjc_CapsuleBox's case analysis) and box-box for a moving box
 * against a static terrain box (corners of either box as points + the terrain box's edges clipped to the
 * moving box, first 4 hits, instead of mjc_BoxBox's separating-axis + face clipping).  Contact SETS from
 * these two can differ from upstream's on edges and corners even where the deepest penetration agrees. */

/* Replaces mjwarp.forward: everything of a step except the integration
 * (reference: sim/sim.py:182-187).
 *
 * "Forward folded into the next step" (SURVEY.md section 8f row 2): the reference calls forward() on
 * all worlds whenever an env was reset, writes the next action to ctrl and calls step(), whose
 * position, collision and constraint-build stages -- functions of qpos, qvel and the model only
 * -- recompute exactly what forward() just produced (envs/manager_based_rl_env.py:106-132).
 * mjlab_forward snapshots qpos / qvel per world (data.sh_qpos, sh_qvel, fold_valid); the first
 * sub-step of the next mjlab_step compares them bit by bit and skips those three stages where
 * nothing changed.  Velocity / actuation and the constraint solve always run, so the results are
 * bit-identical to a full recomputation.  The mechanism is per model: it is active only when
 * m->opt.flags has MJLAB_OPT_FOLD_FORWARD (mjlab_fields.h).  Callers that modify MODEL arrays between
 * forward() and step() -- interval domain randomisation -- must zero data.fold_valid
 * (mjlab_amd.sim.Simulation does whenever a writable model field is handed out). */
int mjlab_forward(const mjlab_model_t* m, const mjlab_data_t* d, void* stream);

/* Extension (not in the reference API): mjlab_forward restricted to
 * the worlds whose d->world_mask entry is non-zero; the other worlds' arrays are left untouched.
 * The reference runs forward() on ALL worlds whenever any env resets
 * (envs/manager_based_rl_env.py:128-132). */
int mjlab_forward_masked(const mjlab_model_t* m, const mjlab_data_t* d, void* stream);

/* Extension: fused post-step read-back of the quantities the
 * reference's EntityData derives with 3-6 small torch kernels each
 * (src/mjlab/entity/data.py:20-31 compute_velocity_from_cvel, :190-261 root_* / body_* pose
 * and velocity, :315-328 joint_pos/vel/acc, :487-516 projected_gravity_b, heading_w,
 * root_*_vel_b).  One launch, one wave per world; every pointer is a device pointer with a
 * leading dimension nworld; outputs may be NULL to skip them. */
typedef struct mjlab_entity_view {
  int nbody;        /* entity bodies */
  int njoint;       /* entity joints that are not the free joint */
  int root_body_id; /* global body id of the entity's root link */
  int pad_;
  const int* body_ids;    /* [nbody] global body ids */
  const int* joint_q_adr; /* [njoint] qpos addresses */
  const int* joint_v_adr; /* [njoint] dof addresses */
  float gravity_vec_w[3]; /* (0, 0, -1) in the reference, entity/entity.py:418 */
  float forward_vec_b[3]; /* (1, 0, 0), entity/entity.py:419 */
  float* body_link_pose_w; /* (nworld, nbody, 7): xpos, xquat */
  float* body_link_vel_w;  /* (nworld, nbody, 6): linear, angular (world frame, at the link origin) */
  float* body_com_pose_w;  /* (nworld, nbody, 7): xipos, xquat * body_iquat */
  float* body_com_vel_w;   /* (nworld, nbody, 6): at the body centre of mass */
  float* root_derived;     /* (nworld, 16): projected_gravity_b 0:3, heading_w 3, root_link_lin_vel_b 4:7,
                              root_link_ang_vel_b 7:10, root_com_lin_vel_b 10:13, root_com_ang_vel_b 13:16 */
  float* joint_pos;        /* (nworld, njoint) */
  float* joint_vel;        /* (nworld, njoint) */
  float* joint_acc;        /* (nworld, njoint) */
} mjlab_entity_view_t;
int mjlab_entity_readback(const mjlab_model_t* m, const mjlab_data_t* d, const mjlab_entity_view_t* v, void* stream);

/* Extension ("masked in-kernel reset"): the termination test and
 * the state reset the reference performs with a chain of small torch kernels and a nonzero()
 * host sync (envs/manager_based_rl_env.py:121-132, envs/mdp/events.py:27-124,
 * tasks/velocity/velocity_env_cfg.py:136-144), as one launch without host involvement.
 * Per world: episode_length += 1; reset = non-finite qpos | root height above the world's
 * origin < min_height | world z of the root's up axis < min_up_z (free root only; pass -2 to
 * switch the orientation test off) | episode_length >= max_len.  Reset worlds get qpos =
 * key_qpos + env_origins[w] (NULL = no offset; the terrain spawn points of
 * terrains/terrain_importer.py:196-229) with x, y += U(-0.5, 0.5) and yaw = U(-3.14, 3.14)
 * from rnd3[w] in [0,1)^3, qvel = qacc_warmstart = 0, episode_length = 0; the others keep
 * their state with NaN -> 0 and +-inf -> +-FLT_MAX in qvel and qacc_warmstart.
 * reset_mask[w] = 1 / 0 (it may alias d->world_mask). */
int mjlab_masked_reset(const mjlab_model_t* m, const mjlab_data_t* d, const float* key_qpos, const float* rnd3,
                       int* episode_length, int max_len, float min_height, int* reset_mask, const float* env_origins,
                       float min_up_z, void* stream);

/* Extension: the reference's interval event `push_robot` (envs/mdp/events.py:127-143
 * push_by_setting_velocity, scheduled per env by managers/event_manager.py:116-138 with
 * interval_range_s = (1, 3) s and velocity_range x, y = +-0.5 m/s for G1-flat:
 * tasks/velocity/velocity_env_cfg.py:155-161, config/g1/flat_env_cfg.py:20-24) as one launch without a
 * nonzero() host sync.  Per world: time_left[w] -= dt; where it drops below 1e-6 a new interval
 * t_lo + U * (t_hi - t_lo) is drawn and the root velocity is kicked by U(range.lo, range.hi) per
 * component [x, y, z, roll, pitch, yaw] (world frame; the angular part is rotated into the body frame
 * qvel stores).  rnd7 = (nworld, 7) uniforms in [0, 1): 6 components + the next interval.  The model's
 * first joint must be the free joint (otherwise an error is returned).  The reference reads the
 * velocity it adds to from cvel, which equals qvel right after the forward() it calls on resets. */
typedef struct mjlab_push_range { float lo[6], hi[6]; } mjlab_push_range_t;
int mjlab_interval_push(const mjlab_model_t* m, const mjlab_data_t* d, float* time_left, const float* rnd7, float dt,
                        float interval_lo, float interval_hi, const mjlab_push_range_t* range, int root_is_free, void* stream);

/* Extension: the physics-facing part of one CONTROL step of the
 * reference's ManagerBasedRlEnv.step (envs/manager_based_rl_env.py:106-139) as ONE launch, one world
 * per wavefront, no kernel boundary between the phases (a fast world runs ahead instead of waiting
 * for the slowest world of every kernel):
 *   action != NULL:   ctrl[w][a] = action_offset[a] + action_scale[a] * action[w][a]   (JointPositionAction,
 *                     envs/mdp/actions/joint_actions.py; ctrl is left as it is when action == NULL)
 *   nsubstep x        one physics step (mjlab_step; the reference re-applies the same action before each)
 *   key_qpos != NULL: termination test + reset, the arguments and semantics of mjlab_masked_reset
 *   forward_mode:     0 none, 1 forward() on every world (the reference's behaviour whenever any env was
 *                     reset), 2 on the reset worlds only (mjlab_forward_masked's extension)
 *   readback_on:      the fused EntityData read-back (mjlab_entity_readback) of the forwarded state
 *   push_time_left != NULL: the interval push, the arguments and semantics of mjlab_interval_push
 * Results are bit-identical to the same sequence of separate calls. */
/* Extension: the TRACKING task's reset and terminations inside the control step (reference tasks/tracking/mdp/commands.py:299-369
 * MotionCommand._resample_command / _update_command, terminations.py:27-53 bad_anchor_pos_z_only / bad_anchor_ori), so that
 * BASELINE config 4 runs under its own reset distribution without ~90 small torch launches per control step.  Motion tables as
 * the reference's MotionLoader holds them (commands.py:30-65), for the ANCHOR body = the floating base.  Per world, at the
 * reset phase of mjlab_control_step:
 *   t = time_steps[w]; frame f = min(t, nframe - 1);
 *   terminate when |qpos[2] - (root_pos[f].z + env_origin.z)| > dz  or  |up_z(root) - up_z(root_quat[f])| > dup
 *   (up_z(q) = 1 - 2 (qx^2 + qy^2)), or when the motion has ended (t + 1 >= nframe), besides the usual tests;
 *   on reset, from the uniforms rnd[w][0 .. 14 + nq - 7): bin = min(int(u0 * bins), bins - 1),
 *     t' = int((bin + u1) / bins * (nframe - 1))                      (uniform over the adaptive sampler's bins: nothing has failed)
 *     root position  = root_pos[t'] + env_origin + U(pose_lo[0..3), pose_hi[0..3))
 *     root quaternion = quat_from_euler_xyz(U(pose[3..6))) * root_quat[t']
 *     root velocity  = root_lin_vel[t'] + U(vel[0..3)),  body-frame angular velocity = R(root quaternion)^T (root_ang_vel[t'] + U(vel[3..6)))
 *     joints         = clip(joint_pos[t'] + U(joint_lo, joint_hi), soft_limits), joint velocities = joint_vel[t']
 *     qacc_warmstart = 0, time_steps[w] = t'; otherwise time_steps[w] = t + 1. */
typedef struct mjlab_motion_reset {
  const float* joint_pos;    /* (nframe, nq - 7) */
  const float* joint_vel;    /* (nframe, nv - 6) */
  const float* root_pos;     /* (nframe, 3) */
  const float* root_quat;    /* (nframe, 4) w x y z */
  const float* root_lin_vel; /* (nframe, 3) world frame */
  const float* root_ang_vel; /* (nframe, 3) world frame */
  const float* soft_limits;  /* (nq - 7, 2) lower, upper (+-inf for unlimited joints) */
  const float* rnd;          /* (nworld, 14 + nq - 7) uniforms in [0, 1), refreshed by the host before every control step */
  int* time_steps;           /* (nworld) */
  int nframe, bins;
  float pose_lo[6], pose_hi[6], vel_lo[6], vel_hi[6];
  float joint_lo, joint_hi, dz, dup;
} mjlab_motion_reset_t;

typedef struct mjlab_control {
  int nsubstep, forward_mode, max_len, pad_;
  const float* action;        /* (nworld, nu) or NULL */
  const float* action_offset; /* (nu) */
  const float* action_scale;  /* (nu) */
  const float* key_qpos;      /* (nq) or NULL: no termination / reset */
  const float* rnd3;          /* (nworld, 3) uniforms in [0, 1) */
  int* episode_length;        /* (nworld) */
  int* reset_mask;            /* (nworld) out */
  const float* env_origins;   /* (nworld, 3) or NULL */
  const int* world_order;     /* (nworld) a permutation: workgroup b works on world world_order[b]; NULL = identity.
                                 Results do not depend on it; it only decides which worlds share a SIMD (load balance) */
  float* push_time_left;      /* (nworld) or NULL: no push */
  const float* rnd7;          /* (nworld, 7) */
  float min_height, min_up_z, push_dt, push_interval_lo, push_interval_hi;
  mjlab_push_range_t push_range;
  int readback_on, pad2_;       /* non-zero: mjlab_entity_readback's outputs are refreshed by the same launch, right after the
                                   forward() pass ("emitted by the step kernel's epilogue") */
  mjlab_entity_view_t readback; /* by value: the view's pointers are device pointers */
  /* Task-provided reset states and a reference-relative termination: what the TRACKING task's reset does (reference
   * tasks/tracking/mdp/commands.py:299-363: MotionCommand._resample_command writes a motion frame plus noise through
   * write_joint_state_to_sim / write_root_state_to_sim; terminations.py:27-53 bad_anchor_pos_z_only / bad_anchor_ori).
   *   reset_qpos != NULL: a world that resets copies row w of reset_qpos (nworld, nq) and reset_qvel (nworld, nv) instead of
   *                       the key_qpos + rnd3 rule; the host refreshes EVERY row before each control step (no host sync:
   *                       which worlds reset is decided here).  key_qpos must still be non-NULL (it switches the phase on).
   *   term_ref != NULL:   row w = [reference height, reference world-z of the root's up axis]; the world also terminates
   *                       when |qpos[2] - term_ref[2 w]| > term_dz or |up_z - term_ref[2 w + 1]| > term_dup */
  const float* reset_qpos;
  const float* reset_qvel;
  const float* term_ref;
  float term_dz, term_dup;
  const mjlab_motion_reset_t* motion; /* DEVICE pointer to the struct above, or NULL (then reset_qpos / key_qpos apply) */
} mjlab_control_t;
int mjlab_control_step(const mjlab_model_t* m, const mjlab_data_t* d, const mjlab_control_t* c, void* stream);
int mjlab_sizeof_control(void); /* sizeof(mjlab_control_t), for bindings that mirror the struct */
int mjlab_sizeof_motion_reset(void); /* sizeof(mjlab_motion_reset_t) */

/* Extension: ENVIRONMENT TERMS -- the reference's event and command terms that write mjData, in
 * mask-based form, one launch per term.  The reference runs each on a variable-length id list (`reset_buf.nonzero()`,
 * `(time_left <= 0).nonzero()`: a host round trip, which a hipGraph cannot hold) as 20-90 small torch kernels; here every world
 * is evaluated by one thread and `mask` (nworld bytes, torch.bool) decides which rows are written, with the reference's
 * arithmetic operation by operation (its helpers third_party/isaaclab/isaaclab/utils/math.py: sample_uniform :1354,
 * quat_from_euler_xyz :269, quat_mul :521, quat_apply_inverse :645, wrap_to_pi :96; no fma contraction).  Uniforms in [0, 1)
 * come in from the caller: row w of U (leading dimension ldu floats) holds world w's draws, so a world's result depends on its
 * own row only.  All pointers are device pointers; ranges are device arrays so that a curriculum may change them between
 * replays of a captured graph.  Used by mjlab_amd/graphed_env.py; the torch restatements there compute the same from the same U.
 *
 * reset_root_state_uniform (envs/mdp/events.py:42-91): U row = 6 pose draws, 6 velocity draws; pose_range / velocity_range =
 * (2, 6) [lo row, hi row] over x y z roll pitch yaw; default_root_state rows of 13 with leading dimension ld_root (0 = one
 * shared row); writes qpos[w][q_adr .. +7) and qvel[w][v_adr .. +6) (angular part rotated into the body frame). */
int mjlab_event_reset_root_state_uniform(float* qpos, int nq, int q_adr, float* qvel, int nv, int v_adr, int nworld,
                                         const unsigned char* mask, const float* default_root_state, int ld_root,
                                         const float* env_origins, const float* U, int ldu, const float* pose_range,
                                         const float* velocity_range, void* stream);
/* reset_joints_by_scale (envs/mdp/events.py:94-124) for nj selected joints: joint_ids[j] (NULL = j) indexes the entity-local
 * default_joint_pos / default_joint_vel (leading dimensions ld_*; 0 = shared row) and soft_joint_pos_limits (rows of 2 per
 * joint, leading dimension ld_lim); q_adr[j] / v_adr[j] are the qpos / qvel addresses of the selected joints; U row = nj
 * position draws then nj velocity draws; ranges = device [pos_lo, pos_hi, vel_lo, vel_hi]. */
int mjlab_event_reset_joints_by_scale(float* qpos, int nq, float* qvel, int nv, int nworld, const unsigned char* mask, int nj,
                                      const int* joint_ids, const int* q_adr, const int* v_adr, const float* default_joint_pos,
                                      int ld_jpos, const float* default_joint_vel, int ld_jvel, const float* soft_joint_pos_limits,
                                      int ld_lim, const float* U, int ldu, const float* ranges, void* stream);
/* The interval event push_by_setting_velocity (envs/mdp/events.py:127-143) under the event manager's per-env timer
 * (managers/event_manager.py:116-138): time_left[w] -= dt; where it drops below 1e-6 a new interval is drawn from
 * interval_range (device [lo, hi]) and qvel[w][v_adr .. +6) = root_link_vel_w[w] + U(velocity_range), the angular part rotated
 * by the inverse of root_link_quat_w[w].  root_link_vel_w / root_link_quat_w are the tensors the reference's EntityData
 * properties return (entity/data.py:213-236; leading dimensions ld_vel, ld_quat) -- unlike mjlab_interval_push, which reads
 * qvel and qpos.  U row = 6 velocity draws, 1 interval draw. */
int mjlab_event_push_by_setting_velocity(float* qvel, int nv, int v_adr, int nworld, float* time_left, float dt,
                                         const float* interval_range, const float* root_link_vel_w, int ld_vel,
                                         const float* root_link_quat_w, int ld_quat, const float* U, int ldu,
                                         const float* velocity_range, void* stream);
/* UniformVelocityCommand (tasks/velocity/mdp/velocity_command.py:64-102) inside CommandTerm.reset / compute
 * (managers/command_manager.py:44-66).  mask != NULL -- reset(): the worlds of the mask draw a new command
 * (time_left, vel_command_b, heading_target, is_heading_env, is_standing_env; command_counter += 1).  mask == NULL --
 * compute(dt): time_left -= dt, the worlds with time_left <= 0 draw a new command, then _update_command on every world
 * (heading control from heading_w, zero command for standing envs).  U row = [time_left, lin_vel_x, lin_vel_y, ang_vel_z,
 * heading, is_heading, is_standing, (unused)].  The init-velocity branch (init_velocity_prob > 0) is not covered. */
typedef struct mjlab_velocity_command {
  int nworld, ldu, ld_heading, heading_command;
  const unsigned char* mask;      /* NULL = compute() */
  const float* U;
  const float* ranges;            /* device (4, 2): lin_vel_x, lin_vel_y, ang_vel_z, heading rows of [lo, hi] */
  const float* heading_w;         /* EntityData.heading_w (compute() with heading_command only) */
  float* time_left;               /* (nworld) */
  float* vel_command_b;           /* (nworld, 3) */
  float* heading_target;          /* (nworld) */
  unsigned char* is_heading_env;  /* (nworld) torch.bool */
  unsigned char* is_standing_env; /* (nworld) torch.bool */
  long long* command_counter;     /* (nworld) torch.long */
  float dt, resampling_lo, resampling_hi, rel_heading_envs, rel_standing_envs, heading_control_stiffness;
  /* compute() only, optional (round 6; all NULL = skipped): _update_metrics (:50-62), which CommandTerm.compute runs FIRST, on the command as
   * it stands -- error_vel_xy[w] += |vel_command_b[w][:2] - root_link_lin_vel_b[w][:2]| * inv_max_command_step, error_vel_yaw[w] +=
   * |vel_command_b[w][2] - root_link_ang_vel_b[w][2]| * inv_max_command_step (the reference divides by the host scalar max_command_step =
   * resampling_hi / step_dt: torch multiplies by its float32 reciprocal).  Logging quantities; the norm may round 1 ulp from torch's reduction. */
  float* error_vel_xy;               /* (nworld) */
  float* error_vel_yaw;              /* (nworld) */
  const float* root_link_lin_vel_b;  /* (nworld, >= 2), ld_lin_vel floats between rows */
  const float* root_link_ang_vel_b;  /* (nworld, 3), ld_ang_vel floats between rows */
  int ld_lin_vel, ld_ang_vel;
  float inv_max_command_step;
} mjlab_velocity_command_t;
int mjlab_command_uniform_velocity(const mjlab_velocity_command_t* c, void* stream);
int mjlab_sizeof_velocity_command(void);

/* MotionCommand of the tracking task (tasks/tracking/mdp/commands.py:68-392) -- the two parts of it that are per-world arithmetic;
 * the adaptive phase sampler (:255-297: statistics over all environments) stays with the caller.  The motion tables are the tensors
 * MotionLoader holds (:28-65), whole (all bodies of the file; body_indexes picks the tracked ones, the first of which is the
 * floating base). */
typedef struct mjlab_motion_tables {
  const float* joint_pos;      /* (nframe, nj) */
  const float* joint_vel;      /* (nframe, nj) */
  const float* body_pos_w;     /* (nframe, nbody_m, 3) */
  const float* body_quat_w;    /* (nframe, nbody_m, 4) w x y z */
  const float* body_lin_vel_w; /* (nframe, nbody_m, 3) */
  const float* body_ang_vel_w; /* (nframe, nbody_m, 3) */
  const int* body_indexes;     /* (nb) tracked body -> index on the tables' body axis */
  int nframe, nj, nbody_m, nb;
} mjlab_motion_tables_t;
/* _resample_command (:305-363) for the worlds of `mask`, at the phase time_steps[w] already drawn: root pose / velocity of the motion
 * frame + U(pose_range) / U(velocity_range) (device (2, 6) each), joints + U(joint_lo, joint_hi) clipped to the soft limits, written to
 * qpos / qvel.  U row (>= 12 + nj): 6 pose draws, 6 velocity draws, nj joint draws.  clear_state stays with the caller. */
int mjlab_command_motion_write(const mjlab_motion_tables_t* tab, float* qpos, int nq, int q_adr, float* qvel, int nv, int v_adr,
                               const int* joint_q_adr, const int* joint_v_adr, int nworld, const unsigned char* mask,
                               const long long* time_steps, const float* env_origins, const float* soft_joint_pos_limits, int ld_lim,
                               const float* U, int ldu, const float* pose_range, const float* velocity_range, float joint_lo,
                               float joint_hi, void* stream);
/* _update_command's body_pos_relative_w / body_quat_relative_w (:371-392) for every world and tracked body; xpos / xquat are mjData's
 * (nworld, nbody, 3 / 4), anchor_body_id the anchor's body id there, anchor_index its place among the tracked bodies.  `exact` selects where
 * the kernel rounds as the reference's jit-scripted helpers do on this stack (csrc/env_terms.h): 8 = their unfused form (pairwise sum of
 * squares, cross products as fma), + 2 / 4 = yaw_quat / quat_apply as their NNC-fused kernels, + 16 s1 + 64 s2 = the two quat_mul calls (0 unfused, 1 / 2 = the fused kernel for 2-D / 3-D operands); 0: every operation rounded
 * separately (1 ulp from any of them).  Which helpers run fused depends on the process's jit history: the caller calibrates. */
int mjlab_command_motion_relative(const mjlab_motion_tables_t* tab, int nworld, const long long* time_steps, const float* env_origins,
                                  const float* xpos, const float* xquat, int nbody, int anchor_body_id, int anchor_index,
                                  float* body_pos_relative_w, float* body_quat_relative_w, int exact, void* stream);
/* MotionCommand's gathered properties (:128-215) of every world in one launch: joint_pos / joint_vel (nworld, nj), body_pos_w (+ the
 * world's env origin) / body_quat_w / body_lin_vel_w / body_ang_vel_w (nworld, nb, 3 | 4) of the motion frame time_steps[w], and -- when
 * body_link_pose_w (nworld, nbody_e, 7) / body_link_vel_w (nworld, nbody_e, 6: linear, angular) are given -- the robot's tracked bodies
 * `robot.data.body_link_*_w[:, body_indexes]` (track_ids: nb indices on the entity's body axis).  Copies and one addition: the bits
 * of the reference's index launches (SURVEY 8f row 1 for the tracking task's command term). */
int mjlab_command_motion_frame(const mjlab_motion_tables_t* tab, int nworld, const long long* time_steps, const float* env_origins,
                               const float* body_link_pose_w, const float* body_link_vel_w, int nbody_e, const int* track_ids, float* joint_pos,
                               float* joint_vel, float* body_pos_w, float* body_quat_w, float* body_lin_vel_w, float* body_ang_vel_w,
                               float* robot_body_pos_w, float* robot_body_quat_w, float* robot_body_lin_vel_w, float* robot_body_ang_vel_w,
                               void* stream);
int mjlab_sizeof_motion_tables(void);
/* world_mask[w] = (*flag > 0) for every world (flag: one device float, e.g. mjlab_masked_sums' count of masked worlds): what
 * mjlab_forward_masked then reads for the reference's "forward() on all worlds iff some environment reset" -- one launch instead of a
 * comparison, a cast and a copy */
int mjlab_flag_to_mask(const float* flag, int nworld, int* world_mask, void* stream);
/* extras["log"] of one step from the RAW masked sums (the managers' reset() logging, envs/manager_based_rl_env.py:214-249): vec[i] <-
 * (div[i] ? *src[i] / max(*count, 1) : *src[i]) * scale[i] if *count > 0 or `first`, else vec[i] stays (a step without resets keeps the
 * last reset step's numbers).  src: DEVICE array of k device pointers to float scalars; div (k bytes), scale (k floats): device. */
int mjlab_log_finish(const float* const* src, const unsigned char* div, const float* scale, int k, const float* count, int first, float* vec, void* stream);
/* MotionCommand._adaptive_sampling (:256-297) for the worlds of `mask`, the per-world part, in one launch: hist_out (bin_count floats) <-
 * the number of worlds with mask & terminated per phase bin clamp(time_steps * bin_count // max(time_step_total, 1)) -- written when some
 * world failed, or always (hist_always; any_failed_out then receives 0 / 1: a sharded caller all-reduces both); time_steps[w] <-
 * long((searchsorted(cdf, U[w][1]) + U[w][2]) / bin_count * (time_step_total - 1)) where mask[w]; the three sampling metrics (nworld
 * floats each) <- *entropy / *top1_prob / *top1_bin when some world is masked; optionally the command term's timer and counter.  cdf (bin_count floats, the running sum of the sampling
 * probabilities) and the three scalars are DEVICE values the caller computes once per control step.  bin_count <= MJLAB_MOTION_SAMPLE_MAX_BINS. */
#define MJLAB_MOTION_SAMPLE_MAX_BINS 4096
typedef struct mjlab_motion_sample {
  const unsigned char* mask;       /* (nworld) torch.bool */
  const unsigned char* terminated; /* (nworld) torch.bool */
  long long* time_steps;           /* (nworld) torch.long, in / out */
  const float* U;                  /* (nworld, ldu): column 1 picks the bin, column 2 the place inside it */
  const float* cdf;                /* (bin_count) */
  const float *entropy, *top1_prob, *top1_bin; /* device scalars */
  float* hist_out;                 /* (bin_count) */
  float* any_failed_out;           /* 1 float or NULL */
  float *m_entropy, *m_top1_prob, *m_top1_bin; /* (nworld) each */
  long long time_step_total;
  int nworld, ldu, bin_count, hist_always;
  /* optional (NULL: skipped): CommandTerm._resample's own two lines for the masked worlds (managers/command_manager.py:62-66) --
   * time_left[w] = U[w][0] * resampling_width + resampling_lo (width = hi - lo rounded to float, as torch multiplies by the scalar),
   * command_counter[w] += 1 */
  float* time_left;
  long long* command_counter;
  float resampling_width, resampling_lo;
} mjlab_motion_sample_t;
int mjlab_command_motion_sample(const mjlab_motion_sample_t* a, void* stream);
/* The sampler's global part in one single-workgroup launch: do_update -- bin_failed_count <- alpha current_bin_failed + one_minus_alpha
 * bin_failed_count, current_bin_failed <- 0 (commands.py:394-398; elementwise, the reference's bits); do_dist -- the sampling distribution
 * from bin_failed_count (:267-281, :291-294: + uniform_term = adaptive_uniform_ratio / bin_count, smoothing kernel of kernel_size taps over
 * the right-padded probabilities, normalisation) as its running sum `cdf` (bin_count floats) and the three logged scalars: normalised
 * entropy, top-1 probability, top-1 bin / bin_count.  Float rounding from the reference's torch expressions (other summation order). */
typedef struct mjlab_motion_sampler {
  float* bin_failed_count;     /* (bin_count) */
  float* current_bin_failed;   /* (bin_count) */
  const float* kernel;         /* (kernel_size) device */
  float* cdf;                  /* (bin_count) out */
  float *entropy, *top1_prob, *top1_bin; /* device scalars, out */
  float alpha, one_minus_alpha, uniform_term;
  int bin_count, kernel_size, do_update, do_dist;
} mjlab_motion_sampler_t;
int mjlab_command_motion_sampler(const mjlab_motion_sampler_t* a, void* stream);
int mjlab_sizeof_motion_sampler(void);
int mjlab_sizeof_motion_sample(void);
/* MotionCommand._update_metrics (:221-254): the ten tracking errors of every world in one launch, rows of `out` (10, nworld) in the
 * order error_anchor_pos, error_anchor_rot, error_anchor_lin_vel, error_anchor_ang_vel, error_body_pos, error_body_rot,
 * error_body_lin_vel, error_body_ang_vel (means over the nb tracked bodies), error_joint_pos, error_joint_vel.  The body arrays are the
 * command term's properties, dense (nworld, nb, 3 | 4); the anchor's errors are row `anchor_index` of the body arrays (the term's
 * anchor_* / robot_anchor_* properties are those rows); the robot's joint arrays may be row views (ld_* floats between rows).
 * Logging quantities: a few ulp from the reference's torch reductions (summation order), not bit for bit. */
typedef struct mjlab_motion_metrics {
  const float *body_pos_w, *body_quat_w, *body_lin_vel_w, *body_ang_vel_w;
  const float *robot_body_pos_w, *robot_body_quat_w, *robot_body_lin_vel_w, *robot_body_ang_vel_w;
  const float *body_pos_relative_w, *body_quat_relative_w;
  const float *joint_pos, *joint_vel;             /* (nworld, nj) dense */
  const float *robot_joint_pos, *robot_joint_vel; /* (nworld, nj), ld_robot_joint_* floats between rows */
  float* out;                                     /* (10, nworld) */
  int ld_robot_joint_pos, ld_robot_joint_vel;
  int nworld, nb, nj, anchor_index;
} mjlab_motion_metrics_t;
int mjlab_command_motion_metrics(const mjlab_motion_metrics_t* a, void* stream);
int mjlab_sizeof_motion_metrics(void);


/* RewardManager.compute's accumulation loop (managers/reward_manager.py:77-89) as one launch: `values` (k, nworld) holds the raw
 * outputs of the k terms with non-zero weight, in term order; per world and term value = raw * weights[i] * dt,
 * reward += value, episode_sums[i][w] += value, step_reward[w][columns[i]] = value / dt (step_reward is (nworld, nterm); the
 * columns of zero-weight terms are the caller's).  weights / columns / episode_sums are device arrays of k entries. */
int mjlab_reward_accumulate(const float* values, const float* weights, const int* columns, int k, int nworld, float dt, float* reward,
                            float* const* episode_sums, float* step_reward, int nterm, void* stream);

/* The managers' reset() bookkeeping and logging for the worlds of a mask (envs/manager_based_rl_env.py:214-249 calls
 * managers/{action,reward,command,event,termination}_manager.py reset(); entity/data.py:169-178 clear_state), each as one launch:
 * mjlab_masked_fill_rows -- row w (row_bytes bytes at ptr + w * row_stride_bytes) of every entry is filled with `pattern`
 *   (elements of 1, 4 or 8 bytes) where mask[w] is set;
 * mjlab_masked_sums -- out[i] = sum over the masked worlds of vector i (nworld floats, or nworld bools counted as 0 / 1),
 *   out[k] = the number of masked worlds.  `entries` are DEVICE arrays. */
/* from_device != 0: `pattern` is the device ADDRESS of an integer scalar at least elem_bytes wide whose low bytes are the fill value, read
 * when the launch runs (a captured launch fills with the step's value, e.g. the event manager's step count) */
typedef struct mjlab_fill_entry { void* ptr; long long pattern; int row_stride_bytes, row_bytes, elem_bytes, from_device; } mjlab_fill_entry_t;
typedef struct mjlab_sum_entry { const void* ptr; int is_bool, pad_; } mjlab_sum_entry_t;
/* n device-to-device copies (dst <- src, nbytes; non-overlapping) in ceil(n / 32) launches: what GraphedRlEnv copies back at the end
 * of a step body (the tensors the reference rebound).  `entries` is a HOST array; the launches carry it by value (graph safe). */
#define MJLAB_COPY_BATCH_MAX 32
typedef struct mjlab_copy_entry { void* dst; const void* src; unsigned long long nbytes; } mjlab_copy_entry_t;
typedef struct mjlab_copy_batch { mjlab_copy_entry_t e[MJLAB_COPY_BATCH_MAX]; } mjlab_copy_batch_t;
int mjlab_copy_batch(const mjlab_copy_entry_t* entries, int n, void* stream);
int mjlab_masked_fill_rows(const mjlab_fill_entry_t* entries, int nentries, const unsigned char* mask, int nworld, void* stream);
int mjlab_masked_sums(const mjlab_sum_entry_t* entries, int k, const unsigned char* mask, int nworld, float* out, void* stream);

/* Runs only the selected stages once (bit mask of MJLAB_STAGE_*), in pipeline order. */
int mjlab_forward_stages(const mjlab_model_t* m, const mjlab_data_t* d, int stages, void* stream);

/* Replaces repeat_array_kernel (reference sim/randomization.py:9-17):
 * dst[w * nelem + i] = src[i] for w < nworld, elements of `elem_size` bytes (4 or 8). */
int mjlab_tile_field(void* dst, const void* src, long long nelem, int nworld, int elem_size, void* stream);

/* Device self-test of the wave-level primitives (DPP reductions); synchronises `stream`. */
int mjlab_selftest(void* stream);
/* Diagnostic: x[i] = A[i]^-1 b[i] for `nbatch` symmetric positive definite n x n matrices (dense row-major, lower triangle read) through the
 * solve stage's own factor + substitution code of the padded size n maps to (the MFMA-tile LDL^T for 32 / 36 / 48 / 64, the LDS-broadcast
 * column sweep otherwise): one wave per matrix.  What tests/test_gpu_chol.py holds against an fp64 factorization. */
int mjlab_chol_selftest(int n, int nbatch, const float* A, const float* b, float* x, void* stream);

/* Diagnostics: fills the private segment (scratch) of `nblocks` waves on `stream`'s queue with poison words (NaN as a float, a
 * non-canonical address as the high half of a pointer), 1280 B per lane -- more than any kernel's frame here.  Scratch is not
 * cleared between kernels, so a kernel that reloads a spill slot it never wrote reads whatever ran before it; launched in front of a
 * step this makes such a defect visible instead of dependent on process history (tests/test_gpu_scratch.py; DESIGN.md section 7:
 * the round-5 "memory aperture violation" was such a reload, of a spill store the compiler had placed under EXEC == 0). */
int mjlab_poison_scratch(int nblocks, void* stream);

/* LDS bytes per workgroup of each stage kernel for this model (occupancy reporting). */
int mjlab_lds_bytes(const mjlab_model_t* m, int stage);

#ifdef __cplusplus
}
#endif
#endif /* MJLAB_AMD_H_ */
