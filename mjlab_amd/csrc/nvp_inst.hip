// nvp_inst.hip -- the kernels that are templates of the padded dof count, instantiated for ONE size and ONE half:
// compiled with -DMJLAB_NVP=<8|16|20|24|32|36|40|48|64> -DMJLAB_NVP_PART=<0|1|2> (mjlab_amd/native.py).
//   part 0: k_solve_integrate<NVP>, k_substep<NVP, false> (forward());   part 1: k_substep<NVP, true>, k_control_step<NVP>;
//   part 2: the elliptic-cone kernels k_solve_cone<NVP>, k_substep_cone<NVP, false / true>, k_control_step_cone<NVP>
#if !defined(MJLAB_NVP) || !defined(MJLAB_NVP_PART)
#error "compile with -DMJLAB_NVP=<padded dof count> -DMJLAB_NVP_PART=<0|1|2>"
#endif
#include "kernels.h"
#include "nvp_launch.h"

#define NVP_CAT2_(a, b) a##b
#define NVP_CAT_(a, b) NVP_CAT2_(a, b)

#if MJLAB_NVP_PART == 0
hipError_t NVP_CAT_(mjlab_nvp_solve_, MJLAB_NVP)(const mjlab_model_t* m, const mjlab_data_t* d, int do_solve, int do_integrate, int flags, int lds_bytes,
                                                 hipStream_t st) {
  hipLaunchKernelGGL(k_solve_integrate<MJLAB_NVP>, dim3(m->size.nworld), dim3(64), (size_t)lds_bytes, st, *m, *d, do_solve, do_integrate, flags);
  return hipGetLastError();
}
hipError_t NVP_CAT_(mjlab_nvp_forward_, MJLAB_NVP)(const mjlab_model_t* m, const mjlab_data_t* d, int flags, int nsub, int lds_bytes, hipStream_t st) {
  (void)nsub;
  hipLaunchKernelGGL((k_substep<MJLAB_NVP, false>), dim3(m->size.nworld), dim3(64), (size_t)lds_bytes, st, *m, *d, flags, 1);
  return hipGetLastError();
}
#elif MJLAB_NVP_PART == 1
hipError_t NVP_CAT_(mjlab_nvp_step_, MJLAB_NVP)(const mjlab_model_t* m, const mjlab_data_t* d, int flags, int nsub, int lds_bytes, hipStream_t st) {
  hipLaunchKernelGGL((k_substep<MJLAB_NVP, true>), dim3(m->size.nworld), dim3(64), (size_t)lds_bytes, st, *m, *d, flags, nsub);
  return hipGetLastError();
}
hipError_t NVP_CAT_(mjlab_nvp_control_, MJLAB_NVP)(const mjlab_model_t* m, const mjlab_data_t* d, const mjlab_control_t* c, int fold, int lds_bytes,
                                                   hipStream_t st) {
  hipLaunchKernelGGL(k_control_step<MJLAB_NVP>, dim3(m->size.nworld), dim3(64), (size_t)lds_bytes, st, *m, *d, *c, fold);
  return hipGetLastError();
}
#else  // part 2: elliptic friction cones (stage_cone.h, kernels.h)
hipError_t NVP_CAT_(mjlab_nvp_cone_, MJLAB_NVP)(const mjlab_model_t* m, const mjlab_data_t* d, int flags, int lds_bytes, hipStream_t st) {
  hipLaunchKernelGGL(k_solve_cone<MJLAB_NVP>, dim3(m->size.nworld), dim3(64), (size_t)lds_bytes, st, *m, *d, flags);
  return hipGetLastError();
}
hipError_t NVP_CAT_(mjlab_nvp_forward_cone_, MJLAB_NVP)(const mjlab_model_t* m, const mjlab_data_t* d, int flags, int nsub, int lds_bytes, hipStream_t st) {
  (void)nsub;
  hipLaunchKernelGGL((k_substep_cone<MJLAB_NVP, false>), dim3(m->size.nworld), dim3(64), (size_t)lds_bytes, st, *m, *d, flags, 1);
  return hipGetLastError();
}
hipError_t NVP_CAT_(mjlab_nvp_step_cone_, MJLAB_NVP)(const mjlab_model_t* m, const mjlab_data_t* d, int flags, int nsub, int lds_bytes, hipStream_t st) {
  hipLaunchKernelGGL((k_substep_cone<MJLAB_NVP, true>), dim3(m->size.nworld), dim3(64), (size_t)lds_bytes, st, *m, *d, flags, nsub);
  return hipGetLastError();
}
// Diagnostic (mjlab_chol_selftest): the solve stage's factor + substitution pair of THIS padded size on caller-supplied symmetric
// positive definite matrices, one wave per matrix -- chol_factor_tiles + chol_solve_tiles where the solve stage uses the MFMA tiles, the
// LDS-broadcast column sweep elsewhere (chol_use_tiles).  In the cone unit so that the measured path's translation units stay as they are.
template <int NVP>
__global__ __launch_bounds__(64) void k_chol_selftest(const float* A, const float* b, float* x, const int n) {
  constexpr int LD = CholCfg<NVP>::LD, NB = CholCfg<NVP>::NB;
  __shared__ __attribute__((aligned(16))) float s_H[NVP * LD + NVP];
  float* s_invd = s_H + NVP * LD;
  const int w = blockIdx.x, lane = threadIdx.x;
  const float* Aw = A + (size_t)w * n * n;
  const float rhs = lane < n ? b[(size_t)w * n + lane] : 0.f;
  float sol;
  if constexpr (chol_use_tiles(NVP)) {
    f32x4 t[NB * (NB + 1) / 2];
    tiles_add_M<NVP, true, false>(t, nullptr, Aw, n, lane);
    chol_factor_tiles<NVP>(t, s_H, s_invd, lane);
    __syncthreads();
    sol = chol_solve_tiles<NVP>(s_H, s_invd, lane, rhs);
  } else {
    dense_global_to_lds(s_H, Aw, n, LD, lane, true);
    chol_pad_rows<NVP>(s_H, n, lane);
    chol_pad_diag<NVP>(s_H, n, lane);
    __syncthreads();
    chol_factor<NVP>(s_H, s_invd, n, lane);
    __syncthreads();
    sol = chol_solve<NVP>(s_H, s_invd, lane, rhs);
  }
  if (lane < n) x[(size_t)w * n + lane] = sol;
}
hipError_t NVP_CAT_(mjlab_nvp_chol_test_, MJLAB_NVP)(const float* A, const float* b, float* x, int n, int nbatch, hipStream_t st) {
  hipLaunchKernelGGL(k_chol_selftest<MJLAB_NVP>, dim3(nbatch), dim3(64), 0, st, A, b, x, n);
  return hipGetLastError();
}
hipError_t NVP_CAT_(mjlab_nvp_control_cone_, MJLAB_NVP)(const mjlab_model_t* m, const mjlab_data_t* d, const mjlab_control_t* c, int fold, int lds_bytes,
                                                        hipStream_t st) {
  hipLaunchKernelGGL(k_control_step_cone<MJLAB_NVP>, dim3(m->size.nworld), dim3(64), (size_t)lds_bytes, st, *m, *d, *c, fold);
  return hipGetLastError();
}
#endif
