// nvp_launch.h -- host-side launch functions of the kernels that are instantiated per padded dof count NVP
// (nvp_inst.hip: three translation units per size, compiled in parallel).  A kernel has to be launched from the
// translation unit that holds its device code, so every size exports eight plain host functions and
// mjlab_amd.hip picks them from a table.
#pragma once

typedef hipError_t (*nvp_solve_fn)(const mjlab_model_t*, const mjlab_data_t*, int do_solve, int do_integrate, int flags, int lds_bytes, hipStream_t);
typedef hipError_t (*nvp_substep_fn)(const mjlab_model_t*, const mjlab_data_t*, int flags, int nsub, int lds_bytes, hipStream_t);
typedef hipError_t (*nvp_cone_fn)(const mjlab_model_t*, const mjlab_data_t*, int flags, int lds_bytes, hipStream_t);
typedef hipError_t (*nvp_control_fn)(const mjlab_model_t*, const mjlab_data_t*, const mjlab_control_t*, int fold, int lds_bytes, hipStream_t);
typedef hipError_t (*nvp_chol_test_fn)(const float* A, const float* b, float* x, int n, int nbatch, hipStream_t);
struct NvpLaunch {
  nvp_solve_fn solve;      // k_solve_integrate<NVP>                      (part 0)
  nvp_substep_fn forward;  // k_substep<NVP, false>: forward()            (part 0)
  nvp_substep_fn step;     // k_substep<NVP, true>: nsub physics steps    (part 1)
  nvp_control_fn control;  // k_control_step<NVP>                         (part 1)
  nvp_cone_fn cone;        // k_solve_cone<NVP>: elliptic friction cones  (part 2)
  nvp_substep_fn forward_cone;  // k_substep_cone<NVP, false>              (part 2)
  nvp_substep_fn step_cone;     // k_substep_cone<NVP, true>               (part 2)
  nvp_control_fn control_cone;  // k_control_step_cone<NVP>                (part 2)
  nvp_chol_test_fn chol_test;   // k_chol_selftest<NVP>: diagnostic         (part 2)
};
#ifdef MJLAB_NVP_ONLY  // experiment builds (tools/ab_bench.sh): a library that carries one size only
#define MJLAB_NVP_SIZES(X) X(MJLAB_NVP_ONLY)
#else
#define MJLAB_NVP_SIZES(X) X(8) X(16) X(20) X(24) X(32) X(36) X(40) X(48) X(64)
#endif
#define MJLAB_NVP_DECL_(N) MJLAB_NVP_DECL2_(N)
#define MJLAB_NVP_DECL2_(N)                                                                                                       \
  hipError_t mjlab_nvp_solve_##N(const mjlab_model_t*, const mjlab_data_t*, int, int, int, int, hipStream_t);                      \
  hipError_t mjlab_nvp_forward_##N(const mjlab_model_t*, const mjlab_data_t*, int, int, int, hipStream_t);                         \
  hipError_t mjlab_nvp_step_##N(const mjlab_model_t*, const mjlab_data_t*, int, int, int, hipStream_t);                            \
  hipError_t mjlab_nvp_control_##N(const mjlab_model_t*, const mjlab_data_t*, const mjlab_control_t*, int, int, hipStream_t);            \
  hipError_t mjlab_nvp_cone_##N(const mjlab_model_t*, const mjlab_data_t*, int, int, hipStream_t);                                     \
  hipError_t mjlab_nvp_forward_cone_##N(const mjlab_model_t*, const mjlab_data_t*, int, int, int, hipStream_t);                    \
  hipError_t mjlab_nvp_step_cone_##N(const mjlab_model_t*, const mjlab_data_t*, int, int, int, hipStream_t);                       \
  hipError_t mjlab_nvp_control_cone_##N(const mjlab_model_t*, const mjlab_data_t*, const mjlab_control_t*, int, int, hipStream_t);       \
  hipError_t mjlab_nvp_chol_test_##N(const float*, const float*, float*, int, int, hipStream_t);
MJLAB_NVP_SIZES(MJLAB_NVP_DECL_)
#ifdef MJLAB_MAIN_TU
static const NvpLaunch* nvp_launch(int nvp) {
  switch (nvp) {
#define MJLAB_NVP_CASE_(N) MJLAB_NVP_CASE2_(N)
#define MJLAB_NVP_CASE2_(N) \
  case N: { static const NvpLaunch t = {mjlab_nvp_solve_##N, mjlab_nvp_forward_##N, mjlab_nvp_step_##N, mjlab_nvp_control_##N, mjlab_nvp_cone_##N, \
                                            mjlab_nvp_forward_cone_##N, mjlab_nvp_step_cone_##N, mjlab_nvp_control_cone_##N, mjlab_nvp_chol_test_##N}; return &t; }
    MJLAB_NVP_SIZES(MJLAB_NVP_CASE_)
  }
  return nullptr;
}
#endif
