// Calibration microbenchmarks for the two counter-derived numbers of bench.py's roofline object
// (/opt/skills/guides/MI355X_MICROARCH.md:154-162 "calibrate on a known byte count in
// your own access pattern", :291 "v_fma_f32 (wave64) 2 cyc (SIMD-32)").
//
//   * streaming kernels of KNOWN byte counts at 4 B / lane, 16 B / lane and in this library's own access pattern
//     (one 64-lane workgroup per world reading / writing short rows): what do FETCH_SIZE / WRITE_SIZE report?
//   * saturating VALU loops (v_fma_f32, v_pk_fma_f32, v_readlane + v_fma, DPP adds, LDS-broadcast + v_pk_fma = the
//     factor sweep's instruction mix, fp32 MFMA) at 1 / 2 / 4 waves per SIMD: cycles per wave64 instruction as a
//     wave sees them (s_memtime) and as SQ_INSTS_VALU / SQ_BUSY_CYCLES / SQ_ACTIVE_INST_VALU count them.
//
// Build (dev container, cross-compiles):  hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o gpurun_prof/ubench
// Run (GPU box): tools/ubench.sh <tag>   (plain run + the same rocprofv3 --pmc passes tools/gpu_round.sh uses)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

// ------------------------------------------------------------------ streaming kernels (known byte counts)
__global__ __launch_bounds__(256) void k_ub_read4(const float* __restrict__ src, float* __restrict__ out, size_t n) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += src[i];
  if (acc == 123.456f) out[0] = acc;  // never true for the fill pattern: the loads stay, nothing is written
}
__global__ __launch_bounds__(256) void k_ub_read16(const float4* __restrict__ src, float* __restrict__ out, size_t n4) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = src[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_ub_write4(float* __restrict__ dst, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}
__global__ __launch_bounds__(256) void k_ub_write16(float4* __restrict__ dst, size_t n4, float v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void k_ub_copy4(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_ub_copy16(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// This library's pattern: workgroup = one wave = one world; per field a row of `rowlen` floats (36: qpos-sized,
// 144 B = 1.125 cache lines, rows of consecutive worlds are adjacent), lanes >= rowlen idle; nfield fields that
// lie `fstride` floats apart (each field its own array).
__global__ __launch_bounds__(64) void k_ub_rows_read(const float* __restrict__ src, float* __restrict__ out, int rowlen, int nfield, size_t fstride) {
  const int w = blockIdx.x, lane = threadIdx.x;
  float acc = 0.f;
  if (lane < rowlen)
    for (int f = 0; f < nfield; ++f) acc += src[(size_t)f * fstride + (size_t)w * rowlen + lane];
  if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(64) void k_ub_rows_write(float* __restrict__ dst, int rowlen, int nfield, size_t fstride, float v) {
  const int w = blockIdx.x, lane = threadIdx.x;
  if (lane < rowlen)
    for (int f = 0; f < nfield; ++f) dst[(size_t)f * fstride + (size_t)w * rowlen + lane] = v;
}

// ------------------------------------------------------------------ VALU issue kernels
// Every wave runs `iters` iterations of a body of NINST instructions and records its own s_memtime span.
#define BODY16(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(8) INS(9) INS(10) INS(11) INS(12) INS(13) INS(14) INS(15)

__global__ __launch_bounds__(64) void k_ub_fma(long long* __restrict__ cyc, float* __restrict__ sink, int iters, float x, float y) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (float)threadIdx.x + i;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define INS(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
    BODY16(INS) BODY16(INS)
#undef INS
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  if (s == 123.456f) sink[0] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(64) void k_ub_fma_dep(long long* __restrict__ cyc, float* __restrict__ sink, int iters, float x, float y) {
  float a = (float)threadIdx.x;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define INS(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
    BODY16(INS) BODY16(INS)
#undef INS
  }
  const long long t1 = __builtin_readcyclecounter();
  if (a == 123.456f) sink[0] = a;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void k_ub_pkfma(long long* __restrict__ cyc, float* __restrict__ sink, int iters, float x, float y) {
  float2v a[16];
  const float2v xx = {x, x}, yy = {y, y};
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = float2v{(float)threadIdx.x + i, (float)i};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define INS(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(xx), "v"(yy));
    BODY16(INS) BODY16(INS)
#undef INS
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
  if (s == 123.456f) sink[0] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// substitution pattern: v_readlane_b32 s, v, k ; v_fma_f32 v, s, v, v   (16 pairs = 32 instructions)
__global__ __launch_bounds__(64) void k_ub_readlane_fma(long long* __restrict__ cyc, float* __restrict__ sink, int iters, float x) {
  float a[4], r = (float)threadIdx.x * x;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = (float)i;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define INS(i)                                                                   \
  {                                                                              \
    float s_;                                                                    \
    asm volatile("v_readlane_b32 %0, %1, " #i : "=s"(s_) : "v"(r));            \
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i & 3]) : "s"(s_), "v"(r)); \
  }
    BODY16(INS)
#undef INS
  }
  const long long t1 = __builtin_readcyclecounter();
  const float s = a[0] + a[1] + a[2] + a[3];
  if (s == 123.456f) sink[0] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// wave reduction pattern: v_add_f32 with DPP row shifts (the group16 / wave sums), 32 instructions on 4 chains
__global__ __launch_bounds__(64) void k_ub_dpp(long long* __restrict__ cyc, float* __restrict__ sink, int iters) {
  float a[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = (float)threadIdx.x + i;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define INS(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i & 3]));
    BODY16(INS) BODY16(INS)
#undef INS
  }
  const long long t1 = __builtin_readcyclecounter();
  const float s = a[0] + a[1] + a[2] + a[3];
  if (s == 123.456f) sink[0] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// factor-sweep pattern: one 128-bit LDS broadcast read feeds two v_pk_fma_f32 (8 reads + 16 pk_fma = 24 instructions)
__global__ __launch_bounds__(64) void k_ub_lds_pkfma(long long* __restrict__ cyc, float* __restrict__ sink, int iters) {
  __shared__ float4 row[64];
  row[threadIdx.x] = make_float4(1.f, 0.5f, 0.25f, 0.125f);
  __syncthreads();
  float2v a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = float2v{(float)threadIdx.x, (float)i};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    float4 q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = row[(it + j) & 63];  // all lanes, same address: broadcast; 8 reads in flight
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float2v lo = {q[j].x, q[j].y}, hi = {q[j].z, q[j].w};
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(lo), "v"(hi));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[(j + 4) & 7]) : "v"(hi), "v"(lo));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  if (s == 123.456f) sink[0] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
typedef float float4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k_ub_mfma(long long* __restrict__ cyc, float* __restrict__ sink, int iters, float x) {
  float4v acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = float4v{0.f, 0.f, 0.f, 0.f};
  const float av = (float)threadIdx.x * x, bv = x;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  if (s == 123.456f) sink[0] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// a VALU wave next to MFMA waves is a different question; here: v_fma and SALU interleaved 1:1 (address arithmetic mix)
__global__ __launch_bounds__(64) void k_ub_fma_salu(long long* __restrict__ cyc, float* __restrict__ sink, int iters, float x, float y, int k) {
  float a[16];
  int s0 = k;
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (float)threadIdx.x + i;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define INS(i)                                                                     \
  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y)); \
  asm volatile("s_add_u32 %0, %0, 3" : "+s"(s0) : : "scc");  /* SCC clobber declared: the loop's own s_cmp lives in SCC */
    BODY16(INS)
#undef INS
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = (float)s0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  if (s == 123.456f) sink[0] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static double mean_cycles(const long long* h, int n) {
  double s = 0;
  for (int i = 0; i < n; ++i) s += (double)h[i];
  return s / n;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);  // every result line reaches the log even if a later kernel is killed by a timeout
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  const size_t n = (size_t)1 << 28;  // 2^28 floats = 1 GiB per buffer: beyond the 256 MiB Infinity Cache
  float *a, *b, *sink;
  CK(hipMalloc(&a, n * 4));
  CK(hipMalloc(&b, n * 4));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 0, n * 4));
  CK(hipMemset(b, 0, n * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = quick ? 2 : 5;
  auto timed = [&](const char* name, double rd, double wr, auto launch) {
    launch();  // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("{\"kernel\": \"%s\", \"read_bytes\": %.0f, \"write_bytes\": %.0f, \"us\": %.2f, \"GBps\": %.1f}\n", name, rd, wr, ms * 1e3, (rd + wr) / (ms * 1e-3) / 1e9);
  };
  const int grid = 256 * 8;
  const double B = (double)n * 4;
  timed("k_ub_read4", B, 0, [&] { hipLaunchKernelGGL(k_ub_read4, dim3(grid), dim3(256), 0, 0, a, sink, n); });
  timed("k_ub_read16", B, 0, [&] { hipLaunchKernelGGL(k_ub_read16, dim3(grid), dim3(256), 0, 0, (const float4*)a, sink, n / 4); });
  timed("k_ub_write4", 0, B, [&] { hipLaunchKernelGGL(k_ub_write4, dim3(grid), dim3(256), 0, 0, b, n, 1.f); });
  timed("k_ub_write16", 0, B, [&] { hipLaunchKernelGGL(k_ub_write16, dim3(grid), dim3(256), 0, 0, (float4*)b, n / 4, 1.f); });
  timed("k_ub_copy4", B, B, [&] { hipLaunchKernelGGL(k_ub_copy4, dim3(grid), dim3(256), 0, 0, a, b, n); });
  timed("k_ub_copy16", B, B, [&] { hipLaunchKernelGGL(k_ub_copy16, dim3(grid), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4); });
  {
    // 4096 worlds x rows of 36 floats x 1024 fields that lie 256 Ki floats (1 MiB) apart: 604 MB, every byte touched once
    const int nworld = 4096, rowlen = 36, nfield = 1024;
    const size_t fstride = (size_t)1 << 18;
    const double rb = (double)nworld * rowlen * 4 * nfield;
    timed("k_ub_rows_read", rb, 0, [&] { hipLaunchKernelGGL(k_ub_rows_read, dim3(nworld), dim3(64), 0, 0, a, sink, rowlen, nfield, fstride); });
    timed("k_ub_rows_write", 0, rb, [&] { hipLaunchKernelGGL(k_ub_rows_write, dim3(nworld), dim3(64), 0, 0, b, rowlen, nfield, fstride, 2.f); });
  }
  long long* cyc;
  CK(hipMalloc(&cyc, 8192 * sizeof(long long)));
  std::vector<long long> h(8192);
  const int iters = quick ? 2000 : 20000;
  auto valu = [&](const char* name, int ninst_per_iter, int wps, auto launch) {
    const int nblk = 1024 * wps;
    launch(nblk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch(nblk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), cyc, nblk * sizeof(long long), hipMemcpyDeviceToHost));
    const double c = mean_cycles(h.data(), nblk), ninst = (double)ninst_per_iter * iters;
    printf("{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"inst_per_wave\": %.0f, \"cycles_per_wave\": %.0f, \"cycles_per_inst_wave\": %.3f, "
           "\"cycles_per_inst_simd\": %.3f, \"us\": %.1f, \"inst_per_us_per_simd\": %.1f}\n",
           name, wps, ninst, c, c / ninst, c / ninst / wps, ms * 1e3, ninst * wps / (ms * 1e3));
  };
  for (int wps : {1, 2, 4}) {
    if (quick && wps != 4) continue;  // counter passes: one configuration per kernel name (4 waves per SIMD, the product's occupancy)
    valu("k_ub_fma", 32, wps, [&](int g) { hipLaunchKernelGGL(k_ub_fma, dim3(g), dim3(64), 0, 0, cyc, sink, iters, 1.0001f, 0.5f); });
    valu("k_ub_fma_dep", 32, wps, [&](int g) { hipLaunchKernelGGL(k_ub_fma_dep, dim3(g), dim3(64), 0, 0, cyc, sink, iters, 1.0001f, 0.5f); });
    valu("k_ub_pkfma", 32, wps, [&](int g) { hipLaunchKernelGGL(k_ub_pkfma, dim3(g), dim3(64), 0, 0, cyc, sink, iters, 1.0001f, 0.5f); });
    valu("k_ub_readlane_fma", 32, wps, [&](int g) { hipLaunchKernelGGL(k_ub_readlane_fma, dim3(g), dim3(64), 0, 0, cyc, sink, iters, 1.0001f); });
    valu("k_ub_dpp", 32, wps, [&](int g) { hipLaunchKernelGGL(k_ub_dpp, dim3(g), dim3(64), 0, 0, cyc, sink, iters); });
    valu("k_ub_lds_pkfma", 24, wps, [&](int g) { hipLaunchKernelGGL(k_ub_lds_pkfma, dim3(g), dim3(64), 0, 0, cyc, sink, iters); });
    valu("k_ub_mfma", 8, wps, [&](int g) { hipLaunchKernelGGL(k_ub_mfma, dim3(g), dim3(64), 0, 0, cyc, sink, iters, 1.0001f); });
    valu("k_ub_fma_salu", 32, wps, [&](int g) { hipLaunchKernelGGL(k_ub_fma_salu, dim3(g), dim3(64), 0, 0, cyc, sink, iters, 1.0001f, 0.5f, 7); });
  }
  return 0;
}
