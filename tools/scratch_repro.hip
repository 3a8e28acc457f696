// scratch_repro.hip -- is per-wave scratch (private segment) memory kept intact when kernels with DIFFERENT private-segment sizes run
// back to back on one queue?  Stand-alone check of the mechanism behind the round-5 "memory aperture violation" of the fused cone kernels
// built with register spills: no library code, no LDS, no spills -- only a private array that the compiler has to
// keep in scratch because it is indexed with a run-time rotation.  Every lane fills its array with (block, lane, index) tags, waits a
// while (so that every wave of the launch is resident at once), reads it back and counts mismatches.
//
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/scratch_repro tools/scratch_repro.hip && /tmp/scratch_repro
//   HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 /tmp/scratch_repro
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int N>
__global__ __launch_bounds__(64) void k_scratch(unsigned* bad, unsigned* first, int rot, long long spin) {
  unsigned a[N];
  const unsigned tag = ((unsigned)blockIdx.x << 16) | ((unsigned)threadIdx.x << 8);
  for (int i = 0; i < N; ++i) a[(i + rot) % N] = tag | (unsigned)i;
  const long long t0 = clock64();
  while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  unsigned nbad = 0;
  for (int i = 0; i < N; ++i) {
    const unsigned v = a[(i + rot) % N];
    if (v != (tag | (unsigned)i)) {
      if (!nbad && atomicAdd(bad + 1, 1u) == 0) { first[0] = tag | (unsigned)i; first[1] = v; }
      ++nbad;
    }
  }
  if (nbad) atomicAdd(bad, nbad);
}

static unsigned *d_bad, *d_first;
template <int N>
static unsigned run(int grid, int rot, long long spin, const char* what) {
  CHECK(hipMemset(d_bad, 0, 8));
  hipLaunchKernelGGL(k_scratch<N>, dim3(grid), dim3(64), 0, 0, d_bad, d_first, rot, spin);
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  unsigned h[2], f[2] = {0, 0};
  CHECK(hipMemcpy(h, d_bad, 8, hipMemcpyDeviceToHost));
  if (h[0]) CHECK(hipMemcpy(f, d_first, 8, hipMemcpyDeviceToHost));
  printf("  %-28s private %4d B/lane grid %5d: %u mismatching words in %u lanes", what, 4 * N, grid, h[0], h[1]);
  if (h[0]) printf("   first: expected %08x found %08x", f[0], f[1]);
  printf("\n");
  fflush(stdout);
  return h[0];
}

int main(int argc, char** argv) {
  const long long spin = argc > 1 ? atoll(argv[1]) : 200000;
  CHECK(hipMalloc(&d_bad, 8));
  CHECK(hipMalloc(&d_first, 8));
  unsigned total = 0;
  for (int grid : {8, 256, 4096}) {
    printf("grid %d\n", grid);
    total += run<100>(grid, 3, spin, "A: 416 B (the 36-dof kernel)");
    for (int k = 0; k < 3; ++k) total += run<80>(grid, 5, spin, "B: 336 B (the 32-dof kernel)");
    total += run<100>(grid, 7, spin, "A again");
    total += run<40>(grid, 1, spin, "C: 176 B");
    total += run<240>(grid, 2, spin, "D: 960 B");
    total += run<80>(grid, 9, spin, "B after D");
    total += run<64>(grid, 4, spin, "E: 272 B");
  }
  printf("TOTAL mismatching words: %u\n", total);
  return total ? 1 : 0;
}
