// launch_floor.hip -- what a LONE small launch costs on this box before it does anything: the floor under every
// latency-bound figure of the record (K1, the nine-target grid, one headline launch at a time).  HIP events around N
// back-to-back launches on one stream, as bench.py times a step.
//   empty          : a kernel that returns at once
//   d = k : every workgroup's wave follows k dependent pointers through a 256 MB buffer (each hop a cold HBM line and a
//           TLB entry) and stores one value.  Short launches are bounded by the CPU's launch rate, so the launch's own
//           floor on the GPU is the intercept of the long chains.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/launch_floor.hip -o scripts/ubench/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_empty() {}
__global__ void k_chain(const unsigned *__restrict__ next, unsigned start, unsigned *out, int depth) {
  unsigned p = start + blockIdx.x * 977u;
  for (int i = 0; i < depth; i++)
    p = next[p];
  if (out && threadIdx.x == 0)
    out[blockIdx.x] = p;
}

template <class F> static double time_us(F launch, int n) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 50; i++)
    launch(i);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < n; i++)
    launch(i);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return 1e3 * ms / n;
}

int main() {
  const size_t N = 64u << 20; // 256 MB of dwords: beyond the L2s and the Infinity Cache
  std::vector<unsigned> h(N);
  unsigned x = 12345u;
  for (size_t i = 0; i < N; i++) { // a random successor per element (not a permutation: chains are short)
    x ^= x << 13, x ^= x >> 17, x ^= x << 5;
    h[i] = x % (unsigned)N;
  }
  unsigned *next, *out;
  hipMalloc(&next, N * 4);
  hipMalloc(&out, 4096 * 4);
  hipMemcpy(next, h.data(), N * 4, hipMemcpyHostToDevice);
  const int n = 2000;
  printf("# us per launch, %d back-to-back launches on one stream between two events\n", n);
  for (int wg : {1, 9, 256}) {
    printf("workgroups %3d x 64 threads: empty %.2f | dependent HBM loads + one store:", wg,
           time_us([&](int) { hipLaunchKernelGGL(k_empty, dim3(wg), dim3(64), 0, 0); }, n));
    double t[6];
    const int depth[6] = {0, 1, 2, 4, 16, 64};
    for (int k = 0; k < 6; k++) {
      t[k] = time_us([&](int i) { hipLaunchKernelGGL(k_chain, dim3(wg), dim3(64), 0, 0, next, 7919u * i, out, depth[k]); }, n);
      printf("  d=%d %.2f", depth[k], t[k]);
    }
    printf("  => %.2f us per hop, %.2f us with no hop (extrapolated from d = 16, 64)\n", (t[5] - t[4]) / 48.0, t[4] - 16.0 * (t[5] - t[4]) / 48.0);
  }
  printf("workgroups 256 x 1024 threads: empty %.2f\n", time_us([&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(1024), 0, 0); }, n));
  return 0;
}
