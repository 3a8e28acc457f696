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
hipError_t NVP_CAT_(mjlab_nvp_control_cone_, MJLAB_NVP)(const mjlab_model_t* m, const mjlab_data_t* d, const mjlab_control_t* c, int fold, int lds_bytes,
                                                        hipStream_t st) {
  hipLaunchKernelGGL(k_control_step_cone<MJLAB_NVP>, dim3(m->size.nworld), dim3(64), (size_t)lds_bytes, st, *m, *d, *c, fold);
  return hipGetLastError();
}
#endif
