// How v_mfma_f32_16x16x4_f32 rounds (GPU box): D = A B + C for inputs whose exact result falls between two floats.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o gpurun_prof/mfma_probe && gpurun_prof/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float __attribute__((ext_vector_type(4))) f32x4;
struct Case { float a[4], b[4], c; };
__global__ void k(const Case* cs, float* out, int n) {
  const int lane = threadIdx.x, kk = lane >> 4;
  for (int i = 0; i < n; ++i) {
    const float a = cs[i].a[kk], b = cs[i].b[kk];
    f32x4 c = {cs[i].c, cs[i].c, cs[i].c, cs[i].c};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    if (lane == 0) out[i] = c[0];
  }
}
int main() {
  const float u = ldexpf(1.f, -23);  // ulp of 1.0
  Case cs[] = {
    {{1, 0, 0, 0}, {1.5f * u / 2, 0, 0, 0}, 1.f},      // 1 + 0.75 ulp: nearest 1 + ulp, toward zero 1
    {{1, 0, 0, 0}, {-u / 8, 0, 0, 0}, 1.f},            // 1 - ulp/8: nearest 1, toward zero 1 - ulp/2
    {{1, 1, 1, 1}, {u / 4, u / 4, u / 4, u / 4}, 1.f},  // four quarter-ulps: summed exactly 1 + ulp, one at a time 1
    {{1, 1, 0, 0}, {u / 2, u / 4, 0, 0}, 1.f},          // 1 + 0.75 ulp in two products
    {{3, 0, 0, 0}, {1.f + u, 0, 0, 0}, -3.f},          // fused multiply-add keeps 3 ulp; separately rounded product gives 4 ulp or 2 ulp
    {{1e-30f, 0, 0, 0}, {1e-10f, 0, 0, 0}, 0.f},       // 1e-40: a denormal result (flushed to 0?)
  };
  const int n = sizeof(cs) / sizeof(cs[0]);
  Case* d; float* o;
  hipMalloc(&d, sizeof(cs)); hipMalloc(&o, n * 4);
  hipMemcpy(d, cs, sizeof(cs), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o, n);
  float h[16];
  hipMemcpy(h, o, n * 4, hipMemcpyDeviceToHost);
  const char* what[] = {"1 + 0.75 ulp (one product)", "1 - ulp/8 (one product)", "1 + 4 x ulp/4", "1 + ulp/2 + ulp/4", "3 (1 + ulp) - 3 [fused: 3 ulp]", "1e-30 x 1e-10"};
  for (int i = 0; i < n; ++i) printf("%-34s -> %.9g  = 1 %+g ulp  (raw %a)\n", what[i], h[i], (h[i] - 1.f) / u, h[i]);
  return 0;
}
