/* mjlab_fields.h -- field catalogue of the mjModel / mjData subset that crosses the C ABI.
 *
 * One X-macro list per (struct, element type).  The HIP library instantiates it with
 * `real = float` and DEVICE pointers, the CPU oracle with `real = double` (or float) and
 * HOST pointers; the Python host side discovers the struct layout at run time through
 * mjlab_model_layout() / mjlab_data_layout() (a comma-separated list of
 * "kind:name:ncol" items in struct order), so there is a single source of truth.
 *
 * Names, shapes and conventions follow mjModel / mjData (reference catalogue:
 * typings/mujoco/_structs.pyi:113-867 (MjData), :916ff (MjModel)); the reference reads
 * and writes them through `sim.model.<field>` / `sim.data.<field>`
 * (reference: src/mjlab/sim/sim_data.py:174-229, src/mjlab/entity/data.py:69-351).
 *
 * X(name, ncol, count)  -- one row of `ncol` elements per item, `count` = size symbol.
 */
#ifndef MJLAB_FIELDS_H_
#define MJLAB_FIELDS_H_

/* ---- model: int32, shared by all worlds -------------------------------------- */
#define MJLAB_MODEL_INT_FIELDS(X)                                               \
  X(body_parentid, 1, nbody)                                                    \
  X(body_rootid, 1, nbody)                                                      \
  X(body_weldid, 1, nbody)                                                      \
  X(body_jntnum, 1, nbody)                                                      \
  X(body_jntadr, 1, nbody)                                                      \
  X(body_dofnum, 1, nbody)                                                      \
  X(body_dofadr, 1, nbody)                                                      \
  X(body_depth, 1, nbody)                                                       \
  X(body_subtreenum, 1, nbody) /* bodies in the DFS-contiguous subtree, self included */ \
  X(level_body, 1, nbody)      /* body ids sorted by tree depth */                \
  X(level_adr, 1, nlevelp1)    /* level L = level_body[level_adr[L] .. level_adr[L+1]) */ \
  X(body_dofmask, 2, nbody) /* lo, hi 32 bits of the ancestor-dof bitmask */     \
  X(jnt_type, 1, njnt)                                                          \
  X(jnt_qposadr, 1, njnt)                                                       \
  X(jnt_dofadr, 1, njnt)                                                        \
  X(jnt_bodyid, 1, njnt)                                                        \
  X(jnt_limited, 1, njnt)                                                       \
  X(dof_bodyid, 1, nv)                                                          \
  X(dof_jntid, 1, nv)                                                           \
  X(dof_parentid, 1, nv)                                                        \
  X(geom_type, 1, ngeom)                                                        \
  X(geom_bodyid, 1, ngeom)                                                      \
  X(geom_condim, 1, ngeom)                                                      \
  X(geom_priority, 1, ngeom)                                                    \
  X(site_bodyid, 1, nsite)                                                      \
  X(actuator_trnid, 2, nu)                                                      \
  X(actuator_ctrllimited, 1, nu)                                                \
  X(actuator_forcelimited, 1, nu)                                               \
  X(sensor_objtype, 1, nsensor)                                                 \
  X(sensor_objid, 1, nsensor)                                                   \
  X(sensor_reftype, 1, nsensor)                                                 \
  X(sensor_refid, 1, nsensor)                                                   \
  X(sensor_intprm, 3, nsensor)                                                  \
  X(sensor_dim, 1, nsensor)                                                     \
  X(sensor_adr, 1, nsensor)                                                     \
  X(pair_geom, 2, npair)                                                        \
  X(tgeom, 1, ntgeom)         /* moving geoms that can touch the terrain, ascending */ \
  X(tbox_geom, 1, nterrain)   /* geom id of terrain box i */                      \
  X(tbox_cell0, 2, nterrain)  /* lowest grid cell (ix, iy) of the box footprint */ \
  X(tgrid_start, 1, ntcellp1) /* cell c = ix * ny + iy lists tgrid_item[start[c] .. start[c+1]) */ \
  X(tgrid_item, 1, ntitem)    /* terrain box indices, ascending inside a cell */

/* ---- model: real; each carries a per-world stride (0 = shared, else elements) -- */
#define MJLAB_MODEL_REAL_FIELDS(X)                                              \
  X(qpos0, 1, nq)                                                               \
  X(qpos_spring, 1, nq) /* reference pose of the joint springs (springref) */    \
  X(body_pos, 3, nbody)                                                         \
  X(body_quat, 4, nbody)                                                        \
  X(body_ipos, 3, nbody)                                                        \
  X(body_iquat, 4, nbody)                                                       \
  X(body_mass, 1, nbody)                                                        \
  X(body_subtreemass, 1, nbody)                                                 \
  X(body_inertia, 3, nbody)                                                     \
  X(body_invweight0, 2, nbody)                                                  \
  X(jnt_pos, 3, njnt)                                                           \
  X(jnt_axis, 3, njnt)                                                          \
  X(jnt_range, 2, njnt)                                                         \
  X(jnt_margin, 1, njnt)                                                        \
  X(jnt_stiffness, 1, njnt)                                                     \
  X(jnt_solref, 2, njnt)                                                        \
  X(jnt_solimp, 5, njnt)                                                        \
  X(dof_armature, 1, nv)                                                        \
  X(dof_damping, 1, nv)                                                         \
  X(dof_frictionloss, 1, nv)                                                    \
  X(dof_solref, 2, nv) /* solreffriction / solimpfriction of the dof's joint: friction-loss rows */ \
  X(dof_solimp, 5, nv)                                                          \
  X(dof_invweight0, 1, nv)                                                      \
  X(geom_size, 3, ngeom)                                                        \
  X(geom_pos, 3, ngeom)                                                         \
  X(geom_quat, 4, ngeom)                                                        \
  X(geom_friction, 3, ngeom)                                                    \
  X(geom_solref, 2, ngeom)                                                      \
  X(geom_solimp, 5, ngeom)                                                      \
  X(geom_solmix, 1, ngeom)                                                      \
  X(geom_margin, 1, ngeom)                                                      \
  X(geom_gap, 1, ngeom)                                                         \
  X(geom_rbound, 1, ngeom)                                                      \
  X(site_pos, 3, nsite)                                                         \
  X(site_quat, 4, nsite)                                                        \
  X(actuator_gainprm, 10, nu)                                                   \
  X(actuator_biasprm, 10, nu)                                                   \
  X(actuator_ctrlrange, 2, nu)                                                  \
  X(actuator_forcerange, 2, nu)                                                 \
  X(actuator_gear, 6, nu)                                                       \
  X(tbox_pos, 3, nterrain)  /* terrain boxes in the WORLD frame (static, never per world) */ \
  X(tbox_mat, 9, nterrain)                                                      \
  X(tbox_size, 3, nterrain)                                                     \
  X(tgrid_ztop, 1, ntcell) /* highest box top in the cell: geoms wholly above it skip the cell */

/* ---- data: real, leading dimension nworld ------------------------------------ */
#define MJLAB_DATA_REAL_FIELDS(X)                                               \
  X(time, 1, one)                                                               \
  X(qpos, 1, nq)                                                                \
  X(qvel, 1, nv)                                                                \
  X(ctrl, 1, nu)                                                                \
  X(qacc_warmstart, 1, nv)                                                      \
  X(qfrc_applied, 1, nv)                                                        \
  X(xfrc_applied, 6, nbody)                                                     \
  X(qacc, 1, nv)                                                                \
  X(xpos, 3, nbody)                                                             \
  X(xquat, 4, nbody)                                                            \
  X(xmat, 9, nbody)                                                             \
  X(xipos, 3, nbody)                                                            \
  X(ximat, 9, nbody)                                                            \
  X(xanchor, 3, njnt)                                                           \
  X(xaxis, 3, njnt)                                                             \
  X(geom_xpos, 3, ngeom)                                                        \
  X(geom_xmat, 9, ngeom)                                                        \
  X(site_xpos, 3, nsite)                                                        \
  X(site_xmat, 9, nsite)                                                        \
  X(subtree_com, 3, nbody)                                                      \
  X(cinert, 10, nbody)                                                          \
  X(cdof, 6, nv)                                                                \
  X(cvel, 6, nbody)                                                             \
  X(cdof_dot, 6, nv)                                                            \
  X(qM, 1, nvnv)  /* dense nv x nv joint-space inertia (row-major) */            \
  X(qLD, 1, nvnv) /* dense lower Cholesky factor of qM (oracle only; the HIP factor stays in LDS) */                      \
  X(qfrc_bias, 1, nv)                                                           \
  X(qfrc_passive, 1, nv)                                                        \
  X(qfrc_actuator, 1, nv)                                                       \
  X(actuator_force, 1, nu)                                                      \
  X(qfrc_smooth, 1, nv)                                                         \
  X(qacc_smooth, 1, nv)                                                         \
  X(qfrc_constraint, 1, nv)                                                     \
  X(sensordata, 1, nsensordata)                                                 \
  X(contact_dist, 1, nconmax)                                                   \
  X(contact_pos, 3, nconmax)                                                    \
  X(contact_frame, 9, nconmax)                                                  \
  X(contact_includemargin, 1, nconmax)                                          \
  X(contact_friction, 5, nconmax)                                               \
  X(contact_solref, 2, nconmax)                                                 \
  X(contact_solimp, 5, nconmax)                                                 \
  X(efc_J, 1, njmaxnv) /* row-major njmax x nv */                                \
  X(efc_pos, 1, njmax)                                                          \
  X(efc_margin, 1, njmax)                                                       \
  X(efc_D, 1, njmax)                                                            \
  X(efc_aref, 1, njmax)                                                         \
  X(efc_force, 1, njmax)                                                        \
  X(efc_frictionloss, 1, njmax) /* written for the friction-loss rows only: rows [0, nf) */ \
  X(efc_B, 1, njmaxnv) /* MJLAB_SOL_PGS only: row r = M^-1 J_r^T (the dual solver walks AR = J M^-1 J^T + R row by row) */ \
  /* PRIVATE hand-over arrays of the pipeline, in the world's LOCAL FRAME: positions minus xorigin, the free base's position     \
   * rounded to whole metres (0 for models without a floating base).  A robot standing 100 m from the origin has fp32 world       \
   * coordinates that resolve 7.6 um; offsets between its bodies (contact point - centre of mass, geom - terrain box) formed     \
   * from them lose 2-3 digits.  The stages therefore compute in the local frame, hand these arrays to each other, and add        \
   * xorigin only when they write the public world-frame arrays (xpos, xipos, xanchor, subtree_com, geom_xpos, site_xpos,         \
   * contact_pos), which nothing inside the step reads back */                                                                    \
  X(xorigin, 3, one)                                                            \
  X(geom_xrel, 3, ngeom)      /* geoms [nstaticgeom, ngeom): geom_xpos - xorigin */ \
  X(subtree_crel, 3, nbody)   /* subtree_com - xorigin */                         \
  X(xipos_rel, 3, nbody)      /* xipos - xorigin */                               \
  X(contact_prel, 3, nconmax) /* contact_pos - xorigin */                         \
  X(sh_qpos, 1, nq) /* qpos / qvel as they were at the last forward(): see fold_valid */ \
  X(sh_qvel, 1, nv)                                                             \
  X(profile, 64, one) /* per-world per-phase cycle counts; written only by -DMJLAB_PROFILE builds */

/* ---- data: int32, leading dimension nworld ------------------------------------ */
#define MJLAB_DATA_INT_FIELDS(X)                                                \
  X(ncon, 1, one)                                                               \
  X(nefc, 1, one)                                                               \
  X(nf, 1, one) /* friction-loss rows: the first nf of the nefc rows (dofs with dof_frictionloss > 0, dof order) */ \
  X(solver_niter, 1, one)                                                       \
  X(world_mask, 1, one) /* mjlab_forward_masked: worlds with 0 are skipped */     \
  X(fold_valid, 1, one) /* 1: position / collision / constraint arrays are those of (sh_qpos, sh_qvel) */ \
  X(fold_reuse, 1, one) /* scratch of the current step: 1 = this world skips those three stages */ \
  X(overflow, 1, one)   /* MJLAB_OVF_* bits of the last collision / constraint pass of this world */ \
  X(sched_thr, 4, one)  /* elements 0..2 (NOT per world; 4 ints per world are allocated so that the three exist even for nworld = 1): thresholds of the wave-priority classes 1..3 on the score      \
                           nefc x (solver_niter + 2), written by the host from the score's quantiles; all 0 = the built-in \
                           row-count thresholds (common.h::wave_priority).  Scheduling hint only: results do not depend on it */ \
  X(contact_dim, 1, nconmax)                                                    \
  X(contact_geom, 2, nconmax)                                                   \
  X(contact_efc_address, 1, nconmax)                                            \
  X(efc_type, 1, njmax)                                                         \
  X(efc_id, 1, njmax)

/* mjtJoint, mjtGeom, mjtObj, mjtIntegrator subsets */
enum { MJLAB_JNT_FREE = 0, MJLAB_JNT_BALL = 1, MJLAB_JNT_SLIDE = 2, MJLAB_JNT_HINGE = 3 };
enum {
  MJLAB_GEOM_PLANE = 0, MJLAB_GEOM_HFIELD = 1, MJLAB_GEOM_SPHERE = 2, MJLAB_GEOM_CAPSULE = 3,
  MJLAB_GEOM_ELLIPSOID = 4, MJLAB_GEOM_CYLINDER = 5, MJLAB_GEOM_BOX = 6, MJLAB_GEOM_MESH = 7
};
enum { MJLAB_OBJ_BODY = 1, MJLAB_OBJ_XBODY = 2, MJLAB_OBJ_GEOM = 5, MJLAB_OBJ_SITE = 6 };
enum { MJLAB_INT_EULER = 0, MJLAB_INT_IMPLICITFAST = 3 };
enum { MJLAB_SOL_PGS = 0, MJLAB_SOL_CG = 1, MJLAB_SOL_NEWTON = 2 };
/* mjtConstraint (reference typings/mujoco/_enums.pyi:1029): values of efc_type */
enum { MJLAB_EFC_FRICTION_DOF = 1, MJLAB_EFC_LIMIT = 3, MJLAB_EFC_CONTACT_FRICTIONLESS = 5, MJLAB_EFC_CONTACT_PYRAMIDAL = 6,
       MJLAB_EFC_CONTACT_ELLIPTIC = 7 /* rows [normal, tangent 1, tangent 2] of a condim-3 contact under mjlab_option_t.cone = MJLAB_CONE_ELLIPTIC */ };
/* mjtCone: values of mjlab_option_t.cone */
enum { MJLAB_CONE_PYRAMIDAL = 0, MJLAB_CONE_ELLIPTIC = 1 };

/* Sizes shared by model and data (host struct, passed by pointer). */
typedef struct mjlab_sizes {
  int nq, nv, nu, nbody, njnt, ngeom, nsite, nsensor, nsensordata, npair;
  int nlevel;  /* number of tree levels (world = level 0) */
  int nworld;  /* number of worlds (environments) */
  int nconmax; /* contact capacity PER WORLD */
  int njmax;   /* constraint-row capacity PER WORLD */
  /* static geometry: geoms [0, nstaticgeom) hang off the world (poses written once, at
   * construction); the collision stage stages geoms [geom_lds0, ngeom) on chip */
  int nstaticgeom, geom_lds0;
  /* sites [0, nstaticsite) hang off static bodies as well (the reference's scenes carry one marker site per environment
   * origin on the world body: terrains/terrain_importer.py:96-120 -- num_envs of them): posed once, like the static geoms */
  int nstaticsite;
  /* box terrain (static colliding boxes) and its uniform xy broadphase grid */
  int nterrain, ntgeom, ntcell, ntcellp1, ntitem, tgrid_nx, tgrid_ny;
} mjlab_sizes_t;

/* mjlab_data_t.overflow: capacity overflows, which DROP work silently otherwise */
enum {
  MJLAB_OVF_NCONMAX = 1, /* contacts beyond sizes.nconmax were dropped */
  MJLAB_OVF_NJMAX = 2,   /* a friction-loss row, a limit row or a contact's rows did not fit sizes.njmax and were dropped */
  MJLAB_OVF_TCAND = 4    /* a moving geom had more than MJLAB_TCAND_MAX terrain boxes within reach */
};
/* terrain boxes kept per moving geom and step: the MJLAB_TCAND_MAX with the smallest ids */
#define MJLAB_TCAND_MAX 12

/* mjOption subset (always double on the host side). */
typedef struct mjlab_option {
  double timestep;
  double gravity[3];
  double impratio;
  double tolerance;
  double ls_tolerance;
  double meaninertia; /* mjModel.stat.meaninertia */
  double tgrid_x0, tgrid_y0, tgrid_cell; /* terrain grid: origin (lowest corner) and cell edge */
  double ls_parallel_min_step; /* MJLAB_OPT_LS_PARALLEL: smallest step of the log-spaced grid (mujoco_warp Option.ls_parallel_min_step, 1e-6) */
  int iterations;
  int ls_iterations;
  int integrator;
  int cone; /* MJLAB_CONE_PYRAMIDAL everywhere; MJLAB_CONE_ELLIPTIC: MJLAB_SOL_NEWTON only; every launch structure but MJLAB_OPT_FUSE_PRESOLVE (kernels of its own) */
  int flags; /* MJLAB_OPT_* bits below */
  int solver; /* mjtSolver: MJLAB_SOL_NEWTON, MJLAB_SOL_CG, or MJLAB_SOL_PGS (the dual solver: only with one kernel per stage, i.e.
                 neither MJLAB_OPT_FUSE_PRESOLVE nor MJLAB_OPT_FUSE_STEP, and not through mjlab_control_step) */
} mjlab_option_t;
/* mjlab_option_t.flags */
enum {
  /* step() right after forward() reuses the position / collision / constraint-build stages of that
   * pass in worlds whose qpos and qvel are still bit-identical (see mjlab_forward in mjlab_amd.h) */
  MJLAB_OPT_FOLD_FORWARD = 1,
  /* MuJoCo's literal termination rules for the Newton iteration and the line search (tolerance /
   * gtol only).  Default (bit clear): both are additionally bounded from below by the fp32 rounding
   * noise of the quantity they test */
  MJLAB_OPT_LITERAL_TERMINATION = 2,
  /* qacc_warmstart <- qacc is written by the integrator's advance only, so forward() leaves it
   * untouched.  Default (bit clear): written at the end of every constraint solve, forward()
   * included (mj_fwdConstraint: "save result for next step warmstart") */
  MJLAB_OPT_WARMSTART_AT_ADVANCE = 4,
  /* launch structure (results are bit-identical either way): the four pre-solve stages in one
   * kernel, or a whole forward() / step() substep in one kernel, instead of one kernel per stage */
  MJLAB_OPT_FUSE_PRESOLVE = 8,
  MJLAB_OPT_FUSE_STEP = 16,
  /* model.dof_frictionloss may hold non-zero values: the constraint stage reads it and builds the friction-loss
   * rows (mj_instantiateFriction).  Bit clear: the field is not read and no such rows exist (the reference's robots
   * have none; its `randomize_field("dof_frictionloss")` is what sets values later -- the host side sets the bit
   * when the field is non-zero at construction, expanded per world, or handed out writable) */
  MJLAB_OPT_FRICTIONLOSS = 32,
  /* mujoco_warp's PARALLEL line search, what the reference configures (`wp_model.opt.ls_parallel = cfg.ls_parallel`, reference
   * src/mjlab/sim/sim.py:89,111): the cost along the search direction is evaluated at `ls_iterations` step sizes, log-spaced
   * from ls_parallel_min_step to 1, alpha_i = exp(log(min_step) + i (log 1 - log(min_step)) / max(1, ls_iterations - 1)), and the
   * step with the lowest cost is taken (first one on ties).  Restated from memory of mujoco_warp's solver (`_log_scale`,
   * `linesearch_parallel_best_alpha`); the pinned source is not available here, so the grid is UNVERIFIED.
   * Bit clear: MuJoCo's exact iterative search (mj_solPrimal's bracketing Newton search, <= ls_iterations evaluations) */
  MJLAB_OPT_LS_PARALLEL = 64,
  /* xorigin = 0: the stages compute in plain world coordinates (what they did before the local frame existed, and what the
   * reference's engine does).  The host sets it for the ONE pass that poses the static geoms and sites (world / terrain
   * bodies), so that their stored world poses do not depend on where the robots happen to be at construction; it is also
   * the switch for A/B runs of the local frame (SimulationCfg.local_frame = False) */
  MJLAB_OPT_WORLD_FRAME = 128,
  /* MJLAB_OPT_LS_PARALLEL: the grid search compares its candidates by their LITERAL total costs cost(alpha_i) = Gauss term +
   * sum_r s_r(jar_r + alpha_i jv_r), and the Newton iteration forms its improvement as the difference of two such totals.  Default
   * (bit clear) on the device and in the fp32 build of the restatement: candidates are compared by cost(alpha_i) - cost(0), formed
   * row by row as a product of differences -- the same argmin in exact arithmetic (what the fp64 restatement computes either way),
   * without the common constant that buries the candidates' differences under fp32 rounding (DESIGN.md section 3;
   * tests/test_oracle_flags.py asserts what the literal form costs in fp32).  Which form upstream's fp32 engine takes is for
   * upstream vectors to decide (tests/test_golden.py tries both) */
  MJLAB_OPT_LS_LITERAL_COST = 256
};

#define MJLAB_DECL_INT_(name, ncol, count) const int* name;
#define MJLAB_DECL_REAL_(name, ncol, count) \
  const MJLAB_REAL* name;                   \
  int name##_ws; /* per-world stride in elements, 0 = shared */
#define MJLAB_DECL_DREAL_(name, ncol, count) MJLAB_REAL* name;
#define MJLAB_DECL_DINT_(name, ncol, count) int* name;

/* The two structs are declared by including this header after defining MJLAB_REAL and
 * the struct names:
 *   #define MJLAB_REAL float
 *   #define MJLAB_MODEL_T mjlab_model_t
 *   #define MJLAB_DATA_T  mjlab_data_t
 */
#ifdef MJLAB_REAL
typedef struct MJLAB_MODEL_T {
  mjlab_sizes_t size;
  mjlab_option_t opt;
  MJLAB_MODEL_INT_FIELDS(MJLAB_DECL_INT_)
  MJLAB_MODEL_REAL_FIELDS(MJLAB_DECL_REAL_)
} MJLAB_MODEL_T;

typedef struct MJLAB_DATA_T {
  MJLAB_DATA_REAL_FIELDS(MJLAB_DECL_DREAL_)
  MJLAB_DATA_INT_FIELDS(MJLAB_DECL_DINT_)
} MJLAB_DATA_T;
#endif

#define MJLAB_STR_(x) #x
#define MJLAB_LAYOUT_INT_(name, ncol, count) "i:" #name ":" MJLAB_STR_(ncol) ":" #count ","
#define MJLAB_LAYOUT_REAL_(name, ncol, count) "r:" #name ":" MJLAB_STR_(ncol) ":" #count ","
#define MJLAB_MODEL_LAYOUT_STRING \
  MJLAB_MODEL_INT_FIELDS(MJLAB_LAYOUT_INT_) MJLAB_MODEL_REAL_FIELDS(MJLAB_LAYOUT_REAL_)
#define MJLAB_DATA_LAYOUT_STRING \
  MJLAB_DATA_REAL_FIELDS(MJLAB_LAYOUT_REAL_) MJLAB_DATA_INT_FIELDS(MJLAB_LAYOUT_INT_)

#endif /* MJLAB_FIELDS_H_ */
