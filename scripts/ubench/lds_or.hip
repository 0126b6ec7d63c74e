// Micro-benchmark (gfx950): aligned ds_or_b32 (non-returning LDS atomic) vs ds_write_b8 / ds_write_b32 under
// realistic (variable-stride, conflicting) token addresses.  Build: hipcc --offload-arch=gfx950 -O3 lds_or.hip -o lds_or
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITERS 512
template <int KIND> __global__ void k(uint64_t *cycles, const uint32_t *addrs) {
  extern __shared__ unsigned char ring[];
  const uint32_t a = addrs[threadIdx.x];
  uint32_t v = 0x01020304u * (threadIdx.x & 3);
  __syncthreads();
  const uint64_t t0 = clock64();
  for (int i = 0; i < ITERS; i++) {
    if (KIND == 0) asm volatile("ds_write_b8 %0, %1" ::"v"(a), "v"(v) : "memory");
    if (KIND == 1) asm volatile("ds_write_b32 %0, %1" ::"v"(a & ~3u), "v"(v) : "memory");
    if (KIND == 2) asm volatile("ds_or_b32 %0, %1" ::"v"(a & ~3u), "v"(v) : "memory");
    if (KIND == 3) asm volatile("ds_or_b32 %0, %1\n ds_or_b32 %0, %1 offset:4\n ds_or_b32 %0, %1 offset:8\n ds_or_b32 %0, %1 offset:12\n ds_or_b32 %0, %1 offset:16" ::"v"(a & ~3u), "v"(v) : "memory");
    if (KIND == 4) { for (int j = 0; j < 5; j++) asm volatile("ds_write_b8 %0, %1\n ds_write_b8 %0, %1 offset:1\n ds_write_b8 %0, %1 offset:2\n ds_write_b8 %0, %1 offset:3" ::"v"(a + 4 * j), "v"(v) : "memory"); }
    if (KIND == 5) asm volatile("ds_or_b64 %0, %1" ::"v"(a & ~7u), "v"((uint64_t)v) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const uint64_t t1 = clock64();
  if (threadIdx.x == 0) cycles[0] = t1 - t0;
}
int main() {
  uint64_t *cyc; uint32_t *addrs; hipMalloc(&cyc, 64); hipMalloc(&addrs, 4096);
  const char *names[] = {"1x ds_write_b8", "1x ds_write_b32 (aligned)", "1x ds_or_b32 (aligned)", "5x ds_or_b32 (20-byte token)", "20x ds_write_b8 (20-byte token)", "1x ds_or_b64 (aligned)"};
  for (int pattern = 0; pattern < 2; pattern++) {
    uint32_t h[1024]; uint32_t pos = 0, seed = 12345;
    for (int i = 0; i < 1024; i++) { h[i] = pos; seed = seed * 1664525u + 1013904223u; pos += pattern ? 12 + (seed >> 24) % 9 : 20; }
    hipMemcpy(addrs, h, 4096, hipMemcpyHostToDevice);
    for (int kind = 0; kind < 6; kind++) {
      size_t smem = 24 * 1024 + 64;
      if (kind == 0) k<0><<<1, 1024, smem>>>(cyc, addrs); if (kind == 1) k<1><<<1, 1024, smem>>>(cyc, addrs);
      if (kind == 2) k<2><<<1, 1024, smem>>>(cyc, addrs); if (kind == 3) k<3><<<1, 1024, smem>>>(cyc, addrs);
      if (kind == 4) k<4><<<1, 1024, smem>>>(cyc, addrs); if (kind == 5) k<5><<<1, 1024, smem>>>(cyc, addrs);
      hipDeviceSynchronize(); uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("%s stride  %-34s: %8.1f cycles per iteration for 16 waves (%.2f per wave-instr-group)\n", pattern ? "12..20B random" : "20B fixed     ", names[kind], (double)c / ITERS, (double)c / ITERS / 16);
    }
  }
  return 0;
}
