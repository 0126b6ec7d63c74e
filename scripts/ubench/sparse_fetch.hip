// Micro-benchmark (gfx950): what does ONE sparse 3-byte point sample cost the memory system?
// FETCH_SIZE is calibrated for wide coalesced streams only (MI355X_MICROARCH.md), so the bytes a lone dword load
// really moves are measured by time instead: N dword loads, one per `stride` bytes, over a region far beyond the
// 256 MB Infinity Cache, next to a 16-byte-per-lane streaming read of the same region.  If a lone dword costs a whole
// G-byte transfer then loads/s x G cannot exceed the streaming rate: G_eff = stream_bytes_per_s / loads_per_s is the
// largest granularity the sparse pattern can be moving (for strides >= G the loads share nothing).
// Build: hipcc --offload-arch=gfx950 -O3 sparse_fetch.hip -o sparse_fetch
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) sparse(const uint8_t *base, uint64_t n, uint64_t stride, uint32_t *sink) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256 * 4) {
    // four independent requests per thread in flight, like the frame kernel's samples
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint64_t j = i + (uint64_t)k * gridDim.x * 256;
      v[k] = j < n ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(base + j * stride)) : 0u;
    }
    acc += v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x12345678u)
    *sink = acc;
}

__global__ void __launch_bounds__(256) stream(const uint4 *base, uint64_t n16, uint32_t *sink) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
    const uint4 v = base[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u)
    *sink = acc;
}

int main() {
  const uint64_t bytes = 8ull << 30;
  uint8_t *buf;
  uint32_t *sink;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 1, bytes));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float ms;
  double stream_bps = 0;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(a));
    stream<<<256 * 16, 256>>>((const uint4 *)buf, bytes / 16, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    stream_bps = bytes / (ms * 1e-3);
  }
  printf("streaming read of 8 GiB (16 B per lane)        : %7.2f TB/s\n", stream_bps / 1e12);
  const uint64_t strides[] = {32, 64, 72, 128, 256, 512, 4096};
  for (uint64_t s : strides) {
    const uint64_t n = bytes / s;
    double best = 0;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(a));
      sparse<<<256 * 16, 256>>>(buf + (rep & 1) * 4, n, s, sink);
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms, a, b));
      const double lps = n / (ms * 1e-3);
      if (lps > best)
        best = lps;
    }
    printf("one dword every %4llu B (%9llu loads over 8 GiB): %7.2f G loads/s = %6.2f TB/s of %llu-byte spans; G_eff = "
           "%5.1f B per load at the streaming rate\n",
           (unsigned long long)s, (unsigned long long)n, best / 1e9, best * s / 1e12, (unsigned long long)s,
           stream_bps / best);
  }
  return 0;
}
