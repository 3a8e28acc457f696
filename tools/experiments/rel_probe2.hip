// rel_probe2.hip -- stage by stage: quat_mul, yaw_quat, quat_apply of the reference's math helpers (jit-scripted; fused by NNC into kernels that
// hiprtc compiles with its default -ffp-contract=fast) against three renderings of the same expressions: mode 0 = no contraction,
// mode 1 = compiled by hipcc with contraction allowed (the compiler's own choice of fma sites), mode 2 = hand-placed fma sites.
// Driven by tools/experiments/rel_probe2.py.  Stand-alone: not part of the library.
#include <hip/hip_runtime.h>
namespace nofma {
#pragma clang fp contract(off)
__device__ __forceinline__ void quat_mul(const float* a, const float* b, float* o) {
  const float w1 = a[0], x1 = a[1], y1 = a[2], z1 = a[3], w2 = b[0], x2 = b[1], y2 = b[2], z2 = b[3];
  const float ww = (z1 + x1) * (x2 + y2), yy = (w1 - y1) * (w2 + z2), zz = (w1 + y1) * (w2 - z2), xx = ww + yy + zz;
  const float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
  o[0] = qq - ww + (z1 - y1) * (y2 - z2); o[1] = qq - xx + (x1 + w1) * (x2 + w2); o[2] = qq - yy + (w1 - x1) * (y2 + z2); o[3] = qq - zz + (z1 + y1) * (w2 - x2);
}
__device__ __forceinline__ void yaw_args(const float* q, float* o) {
  const float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  o[0] = 2 * (qw * qz + qx * qy); o[1] = 1 - 2 * (qy * qy + qz * qz);
}
__device__ __forceinline__ float det2(float a, float b, float c, float d) { return a * b - c * d; }
__device__ __forceinline__ float comb(float v, float w, float t, float c) { return v + w * t + c; }
}  // namespace nofma
namespace fast {
#pragma clang fp contract(fast)
__device__ __forceinline__ void quat_mul(const float* a, const float* b, float* o) {
  const float w1 = a[0], x1 = a[1], y1 = a[2], z1 = a[3], w2 = b[0], x2 = b[1], y2 = b[2], z2 = b[3];
  const float ww = (z1 + x1) * (x2 + y2), yy = (w1 - y1) * (w2 + z2), zz = (w1 + y1) * (w2 - z2), xx = ww + yy + zz;
  const float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
  o[0] = qq - ww + (z1 - y1) * (y2 - z2); o[1] = qq - xx + (x1 + w1) * (x2 + w2); o[2] = qq - yy + (w1 - x1) * (y2 + z2); o[3] = qq - zz + (z1 + y1) * (w2 - x2);
}
__device__ __forceinline__ void yaw_args(const float* q, float* o) {
  const float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  o[0] = 2 * (qw * qz + qx * qy); o[1] = 1 - 2 * (qy * qy + qz * qz);
}
__device__ __forceinline__ float det2(float a, float b, float c, float d) { return a * b - c * d; }
__device__ __forceinline__ float comb(float v, float w, float t, float c) { return v + w * t + c; }
}  // namespace fast
namespace hand {
#pragma clang fp contract(off)
__device__ __forceinline__ void quat_mul(const float* a, const float* b, float* o) {
  const float w1 = a[0], x1 = a[1], y1 = a[2], z1 = a[3], w2 = b[0], x2 = b[1], y2 = b[2], z2 = b[3];
  const float ww = (z1 + x1) * (x2 + y2), yy = (w1 - y1) * (w2 + z2), zz = (w1 + y1) * (w2 - z2), xx = ww + yy + zz;
  const float qq = 0.5f * __builtin_fmaf(z1 - x1, x2 - y2, xx);
  o[0] = __builtin_fmaf(z1 - y1, y2 - z2, qq - ww); o[1] = __builtin_fmaf(x1 + w1, x2 + w2, qq - xx); o[2] = __builtin_fmaf(w1 - x1, y2 + z2, qq - yy);
  o[3] = __builtin_fmaf(z1 + y1, w2 - x2, qq - zz);
}
__device__ __forceinline__ void yaw_args(const float* q, float* o) {
  const float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  o[0] = 2 * __builtin_fmaf(qw, qz, qx * qy); o[1] = __builtin_fmaf(-2.f, __builtin_fmaf(qy, qy, qz * qz), 1.f);
}
__device__ __forceinline__ float det2(float a, float b, float c, float d) { return __builtin_fmaf(a, b, -(c * d)); }
__device__ __forceinline__ float comb(float v, float w, float t, float c) { return __builtin_fmaf(w, t, v) + c; }
}  // namespace hand
#pragma clang fp contract(off)
#define DISPATCH(fn, ...) do { if (mode == 0) nofma::fn(__VA_ARGS__); else if (mode == 1) fast::fn(__VA_ARGS__); else hand::fn(__VA_ARGS__); } while (0)
extern "C" __global__ void k_quat_mul(int n, const float* a, const float* b, float* o, int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  DISPATCH(quat_mul, a + 4 * (size_t)i, b + 4 * (size_t)i, o + 4 * (size_t)i);
}
extern "C" __global__ void k_yaw_quat(int n, const float* q, float* o, int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float y[2];
  DISPATCH(yaw_args, q + 4 * (size_t)i, y);
  const float yaw = atan2f(y[0], y[1]);
  const float cw = cosf(yaw / 2), sz = sinf(yaw / 2);
  const float nrm = fmaxf(sqrtf(cw * cw + sz * sz), 1e-9f);
  o[4 * (size_t)i] = cw / nrm; o[4 * (size_t)i + 1] = 0.f / nrm; o[4 * (size_t)i + 2] = 0.f / nrm; o[4 * (size_t)i + 3] = sz / nrm;
}
// mode = cross mode + 3 * combination mode
extern "C" __global__ void k_quat_apply(int n, const float* q_, const float* v_, float* o, int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* q = q_ + 4 * (size_t)i; const float* v = v_ + 3 * (size_t)i;
  const int cm = mode % 3, km = mode / 3;
  auto det = [&](float a, float b, float c, float d) { return cm == 0 ? nofma::det2(a, b, c, d) : cm == 1 ? fast::det2(a, b, c, d) : hand::det2(a, b, c, d); };
  const float x = q[1], y = q[2], z = q[3];
  float t[3] = {det(y, v[2], z, v[1]), det(z, v[0], x, v[2]), det(x, v[1], y, v[0])};
  for (int k = 0; k < 3; ++k) t[k] = t[k] * 2;
  const float c[3] = {det(y, t[2], z, t[1]), det(z, t[0], x, t[2]), det(x, t[1], y, t[0])};
  for (int k = 0; k < 3; ++k) o[3 * (size_t)i + k] = km == 0 ? nofma::comb(v[k], q[0], t[k], c[k]) : km == 1 ? fast::comb(v[k], q[0], t[k], c[k]) : hand::comb(v[k], q[0], t[k], c[k]);
}
extern "C" int probe(int which, int n, const float* a, const float* b, float* o, int mode, void* stream) {
  const dim3 g((n + 255) / 256), blk(256);
  if (which == 0) hipLaunchKernelGGL(k_quat_mul, g, blk, 0, (hipStream_t)stream, n, a, b, o, mode);
  else if (which == 1) hipLaunchKernelGGL(k_yaw_quat, g, blk, 0, (hipStream_t)stream, n, a, o, mode);
  else hipLaunchKernelGGL(k_quat_apply, g, blk, 0, (hipStream_t)stream, n, a, b, o, mode);
  return (int)hipGetLastError();
}
// ---- enumeration of quat_mul's contraction sites (variant bits) and of yaw_quat's
extern "C" __global__ void k_quat_mul_enum(int n, const float* a_, const float* b_, float* o_, int v) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* a = a_ + 4 * (size_t)i; const float* b = b_ + 4 * (size_t)i; float* o = o_ + 4 * (size_t)i;
  const float w1 = a[0], x1 = a[1], y1 = a[2], z1 = a[3], w2 = b[0], x2 = b[1], y2 = b[2], z2 = b[3];
  const float Aw = z1 + x1, Bw = x2 + y2, Ay = w1 - y1, By = w2 + z2, Az = w1 + y1, Bz = w2 - z2;
  const float ww = Aw * Bw, yy = Ay * By, zz = Az * Bz;
  const int s1 = (v >> 3) % 3;
  float in = s1 == 0 ? ww + yy : s1 == 1 ? __builtin_fmaf(Aw, Bw, yy) : __builtin_fmaf(Ay, By, ww);
  const float xx = ((v / 24) & 1) ? __builtin_fmaf(Az, Bz, in) : in + zz;
  const float qq = 0.5f * ((v & 1) ? __builtin_fmaf(z1 - x1, x2 - y2, xx) : xx + (z1 - x1) * (x2 - y2));
  if (v == 99) {  // what the enumeration pointed at: xx = fma(Aw, Bw, yy) + zz; w, x with the last product contracted; y, z and qq plain
    const float xx9 = __builtin_fmaf(Aw, Bw, yy) + zz, qq9 = 0.5f * (xx9 + (z1 - x1) * (x2 - y2));
    o[0] = __builtin_fmaf(z1 - y1, y2 - z2, qq9 - ww); o[1] = __builtin_fmaf(x1 + w1, x2 + w2, qq9 - xx9);
    o[2] = (qq9 - yy) + (w1 - x1) * (y2 + z2); o[3] = (qq9 - zz) + (z1 + y1) * (w2 - x2);
    return;
  }
  const bool fo = v & 2, fi = v & 4;
  const float iw = fi ? __builtin_fmaf(-Aw, Bw, qq) : qq - ww, ix = qq - xx, iy = fi ? __builtin_fmaf(-Ay, By, qq) : qq - yy, iz = fi ? __builtin_fmaf(-Az, Bz, qq) : qq - zz;
  o[0] = fo ? __builtin_fmaf(z1 - y1, y2 - z2, iw) : iw + (z1 - y1) * (y2 - z2);
  o[1] = fo ? __builtin_fmaf(x1 + w1, x2 + w2, ix) : ix + (x1 + w1) * (x2 + w2);
  o[2] = fo ? __builtin_fmaf(w1 - x1, y2 + z2, iy) : iy + (w1 - x1) * (y2 + z2);
  o[3] = fo ? __builtin_fmaf(z1 + y1, w2 - x2, iz) : iz + (z1 + y1) * (w2 - x2);
}
extern "C" __global__ void k_yaw_enum(int n, const float* q_, float* o, int v) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* q = q_ + 4 * (size_t)i;
  const float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  const int m1 = v % 3, m2 = v / 3;
  const float s1 = m1 == 0 ? qw * qz + qx * qy : m1 == 1 ? __builtin_fmaf(qw, qz, qx * qy) : __builtin_fmaf(qx, qy, qw * qz);
  const float s2 = m2 == 0 ? qy * qy + qz * qz : m2 == 1 ? __builtin_fmaf(qy, qy, qz * qz) : __builtin_fmaf(qz, qz, qy * qy);
  const float yaw = atan2f(2 * s1, 1 - 2 * s2);
  const float cw = cosf(yaw / 2), sz = sinf(yaw / 2);
  const float nrm = fmaxf(sqrtf(cw * cw + sz * sz), 1e-9f);
  o[4 * (size_t)i] = cw / nrm; o[4 * (size_t)i + 1] = 0.f / nrm; o[4 * (size_t)i + 2] = 0.f / nrm; o[4 * (size_t)i + 3] = sz / nrm;
}
extern "C" int probe_enum(int which, int n, const float* a, const float* b, float* o, int v, void* stream) {
  const dim3 g((n + 255) / 256), blk(256);
  if (which == 0) hipLaunchKernelGGL(k_quat_mul_enum, g, blk, 0, (hipStream_t)stream, n, a, b, o, v);
  else hipLaunchKernelGGL(k_yaw_enum, g, blk, 0, (hipStream_t)stream, n, a, o, v);
  return (int)hipGetLastError();
}
// ---- a wider enumeration of quat_mul: sx = how xx = ww + yy + zz is formed (which product is added last: 0..2) x (first pair: plain / fma of
// the first / fma of the second: 0..2) x (last: plain / fma: 0..1); qf: qq's product contracted; xo: x's last product contracted;
// wm / ym / zm: (difference contracted ? 1 : 0) + 2 * (last product contracted ? 1 : 0)
extern "C" __global__ void k_quat_mul_wide(int n, const float* a_, const float* b_, float* o_, int sx, int qf, int xo, int wm, int ym, int zm) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* a = a_ + 4 * (size_t)i; const float* b = b_ + 4 * (size_t)i; float* o = o_ + 4 * (size_t)i;
  const float w1 = a[0], x1 = a[1], y1 = a[2], z1 = a[3], w2 = b[0], x2 = b[1], y2 = b[2], z2 = b[3];
  const float A[3] = {z1 + x1, w1 - y1, w1 + y1}, B[3] = {x2 + y2, w2 + z2, w2 - z2};  // ww, yy, zz
  const float P[3] = {A[0] * B[0], A[1] * B[1], A[2] * B[2]};
  const int last = sx % 3, pm = (sx / 3) % 3, lm = sx / 9;
  const int f0 = last == 0 ? 1 : 0, f1 = last == 2 ? 1 : 2;  // the first pair, in the source's order
  const float in = pm == 0 ? P[f0] + P[f1] : pm == 1 ? __builtin_fmaf(A[f0], B[f0], P[f1]) : __builtin_fmaf(A[f1], B[f1], P[f0]);
  const float xx = lm ? __builtin_fmaf(A[last], B[last], in) : in + P[last];
  const float qq = 0.5f * (qf ? __builtin_fmaf(z1 - x1, x2 - y2, xx) : xx + (z1 - x1) * (x2 - y2));
  auto comp = [&](int k, int m, float pa, float pb) {
    const float inner = (m & 1) ? __builtin_fmaf(-A[k], B[k], qq) : qq - P[k];
    return (m & 2) ? __builtin_fmaf(pa, pb, inner) : inner + pa * pb;
  };
  o[0] = comp(0, wm, z1 - y1, y2 - z2);
  o[1] = xo ? __builtin_fmaf(x1 + w1, x2 + w2, qq - xx) : (qq - xx) + (x1 + w1) * (x2 + w2);
  o[2] = comp(1, ym, w1 - x1, y2 + z2);
  o[3] = comp(2, zm, z1 + y1, w2 - x2);
}
extern "C" int probe_wide(int n, const float* a, const float* b, float* o, int sx, int qf, int xo, int wm, int ym, int zm, void* stream) {
  hipLaunchKernelGGL(k_quat_mul_wide, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, a, b, o, sx, qf, xo, wm, ym, zm);
  return (int)hipGetLastError();
}
