// Micro-benchmark (gfx950): does ANY load flavour / memory type fetch less than a 128-byte line per lone dword?
// sparse_fetch.hip showed 54.7 G loads/s (= 7 TB/s of 128-byte lines) for non-temporal dword loads on hipMalloc memory.
// Here: stride-256 dword loads (no two loads share a line) with every cache-policy combination of the gfx950 global_load
// (default, nt, sc0, sc1, sc0 sc1, sc1 nt, sc0 sc1 nt) over (a) hipMalloc, (b) hipDeviceMallocUncached,
// (c) hipDeviceMallocFinegrained memory.  A flavour that moved 64- or 32-byte sectors would exceed 54.7 G loads/s.
// Build: hipcc --offload-arch=gfx950 -O3 sparse_policy.hip -o sparse_policy
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int POL> __device__ inline uint32_t ld(const uint8_t *p) {
  uint32_t v;
  if (POL == 0) asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if (POL == 1) asm volatile("global_load_dword %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  if (POL == 2) asm volatile("global_load_dword %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  if (POL == 3) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if (POL == 4) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  if (POL == 5) asm volatile("global_load_dword %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
  if (POL == 6) asm volatile("global_load_dword %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
  if (POL == 7) asm volatile("global_load_ubyte %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int POL> __global__ void __launch_bounds__(256) sparse(const uint8_t *base, uint64_t n, uint64_t stride, uint32_t *sink) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256 * 4) {
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint64_t j = i + (uint64_t)k * gridDim.x * 256;
      v[k] = ld<POL>(base + (j < n ? j : 0) * stride);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x12345678u)
    *sink = acc;
}

template <int POL> double run(const uint8_t *buf, uint64_t bytes, uint64_t stride, uint32_t *sink) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const uint64_t n = bytes / stride;
  double best = 0;
  for (int rep = 0; rep < 3; rep++) {
    float ms;
    CK(hipEventRecord(a));
    sparse<POL><<<256 * 16, 256>>>(buf + (rep & 1) * 4, n, stride, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    const double lps = n / (ms * 1e-3);
    if (lps > best) best = lps;
  }
  return best;
}

int main() {
  const uint64_t bytes = 8ull << 30;
  uint32_t *sink;
  CK(hipMalloc(&sink, 4));
  const char *mem_names[] = {"hipMalloc", "uncached", "finegrained"};
  const char *pol_names[] = {"default", "nt", "sc0", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt", "ubyte nt"};
  for (int m = 0; m < 3; m++) {
    uint8_t *buf = nullptr;
    hipError_t e = m == 0 ? hipMalloc(&buf, bytes)
                          : hipExtMallocWithFlags((void **)&buf, bytes, m == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
    if (e != hipSuccess) { printf("%s: allocation failed: %s\n", mem_names[m], hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    for (uint64_t stride : {256ull, 72ull}) {
      double r[8];
      r[0] = run<0>(buf, bytes, stride, sink); r[1] = run<1>(buf, bytes, stride, sink);
      r[2] = run<2>(buf, bytes, stride, sink); r[3] = run<3>(buf, bytes, stride, sink);
      r[4] = run<4>(buf, bytes, stride, sink); r[5] = run<5>(buf, bytes, stride, sink);
      r[6] = run<6>(buf, bytes, stride, sink); r[7] = run<7>(buf, bytes, stride, sink);
      for (int p = 0; p < 8; p++)
        printf("%-11s stride %3llu  %-10s : %7.2f G loads/s  (x128 B = %5.2f TB/s, x64 = %5.2f, x32 = %5.2f)\n", mem_names[m],
               (unsigned long long)stride, pol_names[p], r[p] / 1e9, r[p] * 128 / 1e12, r[p] * 64 / 1e12, r[p] * 32 / 1e12);
    }
    CK(hipFree(buf));
  }
  return 0;
}
