// rel_probe.hip -- which fp32 contraction pattern do the reference's jit-fused helpers (quat_inv, quat_mul, yaw_quat, quat_apply; NNC kernels
// built by hiprtc with its default -ffp-contract=fast) produce for MotionCommand._update_command's relative body poses?  One kernel with the
// candidate contraction sites switchable at run time (variant bits), driven by tools/experiments/rel_probe.py against the reference's own
// chain on the GPU.  Stand-alone: not part of the library.
//   bit 0  quat_mul: the five a*b + c sites as fma        bit 1  yaw_quat's atan2 arguments as fma
//   bit 2  quat_apply: vec + w * t as fma                 bit 3  cross products as fma(a, b, -(c * d))
//   bit 4  quat_inv's sum of squares pairwise             bit 5  cross products as fma(-c, d, a * b)
#include <hip/hip_runtime.h>
struct Quat { float w, x, y, z; };
#pragma clang fp contract(off)
__device__ __forceinline__ Quat quat_mul(const Quat a, const Quat b, const bool f) {
  const float ww = (a.z + a.x) * (b.x + b.y), yy = (a.w - a.y) * (b.w + b.z), zz = (a.w + a.y) * (b.w - b.z), xx = ww + yy + zz;
  Quat q;
  if (f) {
    const float qq = 0.5f * __builtin_fmaf(a.z - a.x, b.x - b.y, xx);
    q.w = __builtin_fmaf(a.z - a.y, b.y - b.z, qq - ww);
    q.x = __builtin_fmaf(a.x + a.w, b.x + b.w, qq - xx);
    q.y = __builtin_fmaf(a.w - a.x, b.y + b.z, qq - yy);
    q.z = __builtin_fmaf(a.z + a.y, b.w - b.x, qq - zz);
  } else {
    const float qq = 0.5f * (xx + (a.z - a.x) * (b.x - b.y));
    q.w = qq - ww + (a.z - a.y) * (b.y - b.z);
    q.x = qq - xx + (a.x + a.w) * (b.x + b.w);
    q.y = qq - yy + (a.w - a.x) * (b.y + b.z);
    q.z = qq - zz + (a.z + a.y) * (b.w - b.x);
  }
  return q;
}
__device__ __forceinline__ float det2(float a, float b, float c, float d, int mode) {  // a * b - c * d
  if (mode == 1) return __builtin_fmaf(a, b, -(c * d));
  if (mode == 2) return __builtin_fmaf(-c, d, a * b);
  return a * b - c * d;
}
__device__ __forceinline__ void cross3(float* o, const float* a, const float* b, int mode) {
  o[0] = det2(a[1], b[2], a[2], b[1], mode); o[1] = det2(a[2], b[0], a[0], b[2], mode); o[2] = det2(a[0], b[1], a[1], b[0], mode);
}
extern "C" __global__ void k_rel(const int n, const int nb, const float* apos_, const float* aquat_, const float* rpos_, const float* rquat_, const float* bpos_,
                                 const float* bquat_, float* out_pos, float* out_quat, const int variant, float* dbg) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * nb) return;
  const int w = i / nb;
  const float *apos = apos_ + 3 * w, *aq = aquat_ + 4 * w, *rp = rpos_ + 3 * w, *rq = rquat_ + 4 * w, *bpos = bpos_ + 3 * (size_t)i, *bq = bquat_ + 4 * (size_t)i;
  const bool v0 = variant & 1, v1 = variant & 2, v2 = variant & 4, v4 = variant & 16;
  const int cm = (variant & 8) ? 1 : (variant & 32) ? 2 : 0;
  const float s = v4 ? (aq[0] * aq[0] + aq[1] * aq[1]) + (aq[2] * aq[2] + aq[3] * aq[3]) : ((aq[0] * aq[0] + aq[1] * aq[1]) + aq[2] * aq[2]) + aq[3] * aq[3];
  const float n2 = fmaxf(s, 1e-9f);
  const Quat inv{aq[0] / n2, -aq[1] / n2, -aq[2] / n2, -aq[3] / n2};
  const Quat d = quat_mul(Quat{rq[0], rq[1], rq[2], rq[3]}, inv, v0);
  float y1, y2;
  if (v1) { y1 = 2.f * __builtin_fmaf(d.w, d.z, d.x * d.y); y2 = __builtin_fmaf(-2.f, __builtin_fmaf(d.y, d.y, d.z * d.z), 1.f); }
  else { y1 = 2.f * (d.w * d.z + d.x * d.y); y2 = 1.f - 2.f * (d.y * d.y + d.z * d.z); }
  const float yaw = atan2f(y1, y2);
  const float cw = cosf(yaw / 2.f), sz = sinf(yaw / 2.f);
  const float nrm = fmaxf(sqrtf(cw * cw + sz * sz), 1e-9f);
  const Quat dq{cw / nrm, 0.f / nrm, 0.f / nrm, sz / nrm};
  const Quat oq = quat_mul(dq, Quat{bq[0], bq[1], bq[2], bq[3]}, v0);
  if (dbg) {  // intermediates: inv (4), d (4), dq (4)
    float* o = dbg + 12 * (size_t)i;
    o[0] = inv.w; o[1] = inv.x; o[2] = inv.y; o[3] = inv.z; o[4] = d.w; o[5] = d.x; o[6] = d.y; o[7] = d.z; o[8] = dq.w; o[9] = dq.x; o[10] = dq.y; o[11] = dq.z;
  }
  out_quat[4 * (size_t)i] = oq.w; out_quat[4 * (size_t)i + 1] = oq.x; out_quat[4 * (size_t)i + 2] = oq.y; out_quat[4 * (size_t)i + 3] = oq.z;
  float rel[3], t[3], c[3], rot[3];
  for (int k = 0; k < 3; ++k) rel[k] = bpos[k] - apos[k];
  const float xyz[3] = {dq.x, dq.y, dq.z};
  cross3(t, xyz, rel, cm);
  for (int k = 0; k < 3; ++k) t[k] = t[k] * 2.f;
  cross3(c, xyz, t, cm);
  for (int k = 0; k < 3; ++k) rot[k] = (v2 ? __builtin_fmaf(dq.w, t[k], rel[k]) : rel[k] + dq.w * t[k]) + c[k];
  out_pos[3 * (size_t)i] = rp[0] + rot[0]; out_pos[3 * (size_t)i + 1] = rp[1] + rot[1]; out_pos[3 * (size_t)i + 2] = apos[2] + rot[2];
}
extern "C" int rel_probe(int n, int nb, const float* apos, const float* aquat, const float* rpos, const float* rquat, const float* bpos, const float* bquat,
                         float* out_pos, float* out_quat, int variant, float* dbg, void* stream) {
  hipLaunchKernelGGL(k_rel, dim3((n * nb + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, nb, apos, aquat, rpos, rquat, bpos, bquat, out_pos, out_quat, variant, dbg);
  return (int)hipGetLastError();
}
