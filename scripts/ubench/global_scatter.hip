// Micro-benchmark (gfx950): writing a frame's tokens to HBM.
//   A  "staged"  : every thread stores 2 tokens of ~19 bytes into an LDS buffer with byte stores, the workgroup drains the
//                  buffer with 16-byte coalesced non-temporal stores (what the frame kernel does today)
//   B  "direct"  : every thread writes its 2 tokens straight to HBM with exact-length unaligned stores
//                  (12 bytes as dwordx3, then dword / short / byte pieces by length)
//   C  "direct16": as B but each token written as one unaligned 16-byte store + one dword (over-writing is NOT allowed
//                  in the real kernel; this variant only measures what the store path could do at best)
// 256 workgroups x 1024 threads, 2048 tokens of 15..20 bytes per workgroup (~36 KB), like 1080p -> 80x24 truecolor.
// Build: hipcc --offload-arch=gfx950 -O3 global_scatter.hip -o global_scatter
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define TOK 2048
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
struct __attribute__((packed)) u128_u { u32x4 v; };
struct __attribute__((packed)) u96_u { u32x3 v; };
struct __attribute__((packed)) u32_u { uint32_t v; };
struct __attribute__((packed)) u16_u { uint16_t v; };

template <int KIND> __global__ void __launch_bounds__(1024) k(const uint32_t *offs, uint8_t *out, uint64_t stride, unsigned long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
  const int tid = threadIdx.x;
  const uint32_t *o = offs + (size_t)blockIdx.x * (TOK + 1);
  uint8_t *dst = out + (size_t)blockIdx.x * stride;
  const unsigned long long t0 = clock64();
#pragma unroll
  for (int kk = 0; kk < 2; kk++) {
    const int i = tid + kk * 1024;
    const uint32_t a = o[i], len = o[i + 1] - a;
    const uint32_t w = 0x30313233u + i;
    if (KIND == 0) {
      for (uint32_t j = 0; j < 20; j++) /* up to 20 byte stores, predicated like the kernel's fields */
        if (j < len) asm volatile("ds_write_b8 %0, %1" ::"v"(a + j), "v"(w >> (8 * (j & 3))) : "memory");
    } else if (KIND == 1) {
      uint8_t *p = dst + a;
      u32x3 x = {w, w + 1, w + 2};
      ((u96_u *)p)->v = x; /* len >= 14 */
      uint32_t r = len - 12; /* 3..8 */
      uint8_t *q = p + 12;
      if (r >= 4) { ((u32_u *)q)->v = w; q += 4; r -= 4; }
      if (r >= 4) { ((u32_u *)q)->v = w; q += 4; r -= 4; }
      if (r >= 2) { ((u16_u *)q)->v = (uint16_t)w; q += 2; r -= 2; }
      if (r >= 1) *q = (uint8_t)w;
    } else {
      uint8_t *p = dst + a;
      u32x4 x = {w, w + 1, w + 2, w + 3};
      ((u128_u *)p)->v = x;
      ((u32_u *)(p + len - 4))->v = w;
    }
  }
  if (KIND == 0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const uint32_t total = o[TOK];
    for (uint32_t g = tid * 16u; g + 16u <= total; g += 16u * 1024u) {
      const u32x4 v = *(const u32x4 *)(ring + g);
      __builtin_nontemporal_store(v, (u32x4 *)(dst + g));
    }
  }
  if (tid == 0) cyc[blockIdx.x] = clock64() - t0;
}

int main() {
  const int nb = 256;
  uint32_t *h = (uint32_t *)malloc((size_t)nb * (TOK + 1) * 4);
  uint32_t seed = 777;
  uint32_t mx = 0;
  for (int b = 0; b < nb; b++) {
    uint32_t pos = 0;
    for (int i = 0; i <= TOK; i++) { h[b * (TOK + 1) + i] = pos; seed = seed * 1664525u + 1013904223u; pos += 15 + (seed >> 24) % 6; }
    if (pos > mx) mx = pos;
  }
  const uint64_t stride = (mx + 64 + 15) & ~15ull;
  uint32_t *offs; uint8_t *out; unsigned long long *cyc;
  hipMalloc(&offs, (size_t)nb * (TOK + 1) * 4); hipMalloc(&out, nb * stride); hipMalloc(&cyc, nb * 8);
  hipMemcpy(offs, h, (size_t)nb * (TOK + 1) * 4, hipMemcpyHostToDevice);
  const char *names[] = {"A staged (LDS byte stores + 16-byte drain)", "B direct exact-length unaligned stores", "C direct 16+4 bytes (upper bound of the store path)"};
  for (int kind = 0; kind < 3; kind++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t smem = kind == 0 ? 48 * 1024 : 0;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      for (int it = 0; it < 50; it++) {
        if (kind == 0) { hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024); k<0><<<nb, 1024, smem>>>(offs, out, stride, cyc); }
        if (kind == 1) k<1><<<nb, 1024, 0>>>(offs, out, stride, cyc);
        if (kind == 2) k<2><<<nb, 1024, 0>>>(offs, out, stride, cyc);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[256]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    double mean = 0; for (int b = 0; b < nb; b++) mean += c[b]; mean /= nb;
    printf("%-52s: %7.2f us per launch, %8.0f cycles per workgroup (%.1f MB written)\n", names[kind], ms / 50 * 1e3, mean, nb * (double)mx / 1e6);
  }
  return 0;
}
