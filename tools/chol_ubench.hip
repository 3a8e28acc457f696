// Latency of one 36 x 36 LDL^T factorization alone, one wave per SIMD (GPU box):  the left-looking LDS-broadcast sweep (chol_factor)
// against the register-resident blocked factorization on MFMA tiles (chol_factor_tiles), each timed with the shader clock inside
// a kernel that does nothing else, plus a correctness check of both solves against a double-precision solve on the host.
//   hipcc -O3 -std=c++17 -ffp-contract=on --offload-arch=gfx950 tools/chol_ubench.hip -o gpurun_prof/chol_ubench && gpurun_prof/chol_ubench
#include "../mjlab_amd/csrc/kernels.h"

#include <cmath>
#include <vector>

constexpr int N = 36, NV = 35, LDh = CholCfg<N>::LD, REPS = 50;
struct Args { const float* M; const float* b; float* x; long long* cyc; };

__global__ __launch_bounds__(64, 4) void k_old(const Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *sH = smem, *sinv = sH + N * LDh, *sM = sinv + N;
  const int lane = threadIdx.x;
  glds_dense_to_packed(sM, a.M, NV, lane);
  for (int k = NV * (NV + 1) / 2 + lane; k < N * (N + 1) / 2; k += 64) sM[k] = 0.f;
  __syncthreads();
  long long tf = 0, ts = 0, tl = 0;
  float x = 0.f;
  for (int rep = 0; rep < REPS; ++rep) {
    long long t0 = clock64();
    packed_to_lds(sH, sM, NV, LDh, lane);
    chol_pad_rows<N>(sH, NV, lane);
    chol_pad_diag<N>(sH, NV, lane);
    __syncthreads();
    long long t1 = clock64();
    chol_factor<N>(sH, sinv, NV, lane);
    __syncthreads();
    long long t2 = clock64();
    x = chol_solve<N>(sH, sinv, lane, lane < NV ? a.b[lane] : 0.f);
    long long t3 = clock64();
    tl += t1 - t0; tf += t2 - t1; ts += t3 - t2;
    __syncthreads();
  }
  if (lane < NV) a.x[blockIdx.x * 64 + lane] = x;
  if (lane == 0) { a.cyc[blockIdx.x * 3] = tl / REPS; a.cyc[blockIdx.x * 3 + 1] = tf / REPS; a.cyc[blockIdx.x * 3 + 2] = ts / REPS; }
}
__global__ __launch_bounds__(64, 4) void k_new(const Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *sH = smem, *sinv = sH + N * LDh, *sM = sinv + N;
  const int lane = threadIdx.x;
  glds_dense_to_packed(sM, a.M, NV, lane);
  for (int k = NV * (NV + 1) / 2 + lane; k < N * (N + 1) / 2; k += 64) sM[k] = 0.f;
  __syncthreads();
  long long tf = 0, ts = 0, tl = 0;
  float x = 0.f;
  for (int rep = 0; rep < REPS; ++rep) {
    f32x4 t[CholT<N>::NT];
    long long t0 = clock64();
    tiles_add_M<N, false, false>(t, sM, a.M, NV, lane);
    __syncthreads();
    long long t1 = clock64();
    chol_factor_tiles<N>(t, sH, sinv, lane);
    __syncthreads();
    long long t2 = clock64();
    x = chol_solve_tiles<N>(sH, sinv, lane, lane < NV ? a.b[lane] : 0.f);
    long long t3 = clock64();
    tl += t1 - t0; tf += t2 - t1; ts += t3 - t2;
    __syncthreads();
  }
  if (lane < NV) a.x[blockIdx.x * 64 + lane] = x;
  if (lane == 0) { a.cyc[blockIdx.x * 3] = tl / REPS; a.cyc[blockIdx.x * 3 + 1] = tf / REPS; a.cyc[blockIdx.x * 3 + 2] = ts / REPS; }
}

int main() {
  std::vector<double> A(NV * NV, 0.0), B(NV * (NV + 5));
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
  for (auto& v : B) v = rnd();
  for (int i = 0; i < NV; ++i)
    for (int j = 0; j < NV; ++j) {
      double acc = i == j ? 0.05 : 0.0;
      for (int k = 0; k < NV + 5; ++k) acc += B[i * (NV + 5) + k] * B[j * (NV + 5) + k];
      A[i * NV + j] = acc;
    }
  std::vector<float> Mf(NV * NV), bf(NV);
  std::vector<double> b(NV);
  for (int i = 0; i < NV * NV; ++i) Mf[i] = (float)A[i];
  for (int i = 0; i < NV; ++i) { bf[i] = (float)rnd(); b[i] = bf[i]; }
  // reference solve (Gaussian elimination in double on the fp32-rounded matrix)
  std::vector<double> G(NV * NV), xr = b;
  for (int i = 0; i < NV * NV; ++i) G[i] = Mf[i];
  for (int k = 0; k < NV; ++k)
    for (int i = k + 1; i < NV; ++i) {
      const double f = G[i * NV + k] / G[k * NV + k];
      for (int j = k; j < NV; ++j) G[i * NV + j] -= f * G[k * NV + j];
      xr[i] -= f * xr[k];
    }
  for (int i = NV - 1; i >= 0; --i) {
    for (int j = i + 1; j < NV; ++j) xr[i] -= G[i * NV + j] * xr[j];
    xr[i] /= G[i * NV + i];
  }
  float *dM, *db, *dx; long long* dc;
  for (int nwg : {1024, 4096}) {
    hipMalloc(&dM, Mf.size() * 4); hipMalloc(&db, NV * 4); hipMalloc(&dx, nwg * 64 * 4); hipMalloc(&dc, nwg * 3 * 8);
    hipMemcpy(dM, Mf.data(), Mf.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, bf.data(), NV * 4, hipMemcpyHostToDevice);
    const size_t smem = (N * LDh + N + N * (N + 1) / 2 + 64) * 4;
    for (int which = 0; which < 2; ++which) {
      hipMemset(dx, 0, nwg * 64 * 4);
      Args a{dM, db, dx, dc};
      if (which == 0) k_old<<<nwg, 64, smem>>>(a); else k_new<<<nwg, 64, smem>>>(a);
      hipDeviceSynchronize();
      std::vector<float> x(nwg * 64); std::vector<long long> c(nwg * 3);
      hipMemcpy(x.data(), dx, x.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
      double err = 0, ref = 0, cl = 0, cf = 0, cs = 0;
      for (int w = 0; w < nwg; ++w) for (int i = 0; i < NV; ++i) { err = std::fmax(err, std::fabs(x[w * 64 + i] - xr[i])); ref = std::fmax(ref, std::fabs(xr[i])); }
      for (int w = 0; w < nwg; ++w) { cl += c[w * 3]; cf += c[w * 3 + 1]; cs += c[w * 3 + 2]; }
      printf("%s  %d waves: matrix -> factor input %7.0f  factor %7.0f  solve %7.0f cycles per call;  solve rel err %.2e (%s)\n", which ? "tiles (MFMA)    " : "sweep (LDS bcast)", nwg,
             cl / nwg, cf / nwg, cs / nwg, err / ref, hipGetErrorString(hipGetLastError()));
    }
  }
  return 0;
}
