// Micro-benchmark (gfx950): cost and correctness of misaligned LDS stores and of sparse global gathers.
// Build: hipcc --offload-arch=gfx950 -O3 lds_unaligned.hip -o lds_unaligned ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define ITERS 512

template <int KIND> __global__ void lds_store(uint64_t *cycles, uint32_t *check, int mis, int stride) {
  extern __shared__ unsigned char ring[];
  const uint32_t lane = threadIdx.x;
  const uint32_t addr = lane * stride + mis;
  uint32_t v = 0x04030201u + lane;
  __syncthreads();
  const uint64_t t0 = clock64();
  for (int i = 0; i < ITERS; i++) {
    if (KIND == 0) { // four byte stores
      asm volatile("ds_write_b8 %0, %1\n ds_write_b8 %0, %2 offset:1\n ds_write_b8 %0, %3 offset:2\n ds_write_b8 %0, %4 offset:3"
                   :: "v"(addr), "v"(v), "v"(v >> 8), "v"(v >> 16), "v"(v >> 24) : "memory");
    } else if (KIND == 1) { // one (possibly misaligned) dword store
      asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory");
    } else if (KIND == 2) { // byte stores using d16_hi: no shifts needed for bytes 0 and 2
      asm volatile("ds_write_b8 %0, %1\n ds_write_b8_d16_hi %0, %1 offset:2" :: "v"(addr), "v"(v) : "memory");
    } else if (KIND == 3) { // two b16 stores
      asm volatile("ds_write_b16 %0, %1\n ds_write_b16_d16_hi %0, %1 offset:2" :: "v"(addr), "v"(v) : "memory");
    } else { // 8-byte store
      uint64_t vv = ((uint64_t)v << 32) | v;
      asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(vv) : "memory");
    }
    v += 0x01010101u;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const uint64_t t1 = clock64();
  __syncthreads();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  uint32_t r = 0;
  for (int k = 0; k < 4; k++) r |= (uint32_t)ring[addr + k] << (8 * k);
  check[blockIdx.x * blockDim.x + threadIdx.x] = r - (0x04030201u + lane + (ITERS - 1) * 0x01010101u);
}

// gather: each lane reads one RGB pixel `pitch` bytes apart, as 3 byte loads or 1 unaligned dword load
template <int KIND> __global__ void gather(const uint8_t *src, size_t bytes, int pitch, uint32_t *out, uint64_t *cycles) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  const uint64_t t0 = clock64();
  for (int it = 0; it < 8; it++) {
    size_t a = ((gid * 8 + it) * (size_t)pitch + 3) % (bytes - 8);
    a -= a % 3; a += 3;
    const uint8_t *p = src + a;
    if (KIND == 0) {
      acc += (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    } else {
      uint32_t v; __builtin_memcpy(&v, p - 1, 4);
      acc += v >> 8;
    }
  }
  out[gid] = acc;
  const uint64_t t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  uint64_t *cyc; uint32_t *chk;
  hipMalloc(&cyc, 4096 * 8); hipMalloc(&chk, 1024 * 1024 * 4);
  const char *names[] = {"4x ds_write_b8", "ds_write_b32", "b8 + b8_d16_hi (2 bytes)", "2x ds_write_b16", "ds_write_b64"};
  for (int stride : {20, 4, 1}) for (int mis = 0; mis < 4; mis++) for (int kind = 0; kind < 5; kind++) {
    const int threads = 1024;
    hipMemset(chk, 0xFF, threads * 4);
    size_t smem = (size_t)threads * stride + 64;
    if (kind == 0) lds_store<0><<<1, threads, smem>>>(cyc, chk, mis, stride);
    if (kind == 1) lds_store<1><<<1, threads, smem>>>(cyc, chk, mis, stride);
    if (kind == 2) lds_store<2><<<1, threads, smem>>>(cyc, chk, mis, stride);
    if (kind == 3) lds_store<3><<<1, threads, smem>>>(cyc, chk, mis, stride);
    if (kind == 4) lds_store<4><<<1, threads, smem>>>(cyc, chk, mis, stride);
    hipDeviceSynchronize();
    uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    std::vector<uint32_t> h(threads); hipMemcpy(h.data(), chk, threads * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < threads; i++) bad += h[i] != 0;
    printf("LDS stride %2d mis %d %-28s: %7.1f cyc/iter (16 waves)  last-writer check: %s\n", stride, mis, names[kind], (double)c / ITERS,
           stride >= 8 ? (kind == 2 ? "n/a" : (bad ? "MISMATCH" : "ok")) : "n/a(overlap)");
  }
  const size_t bytes = 1ull << 30; uint8_t *src; hipMalloc(&src, bytes); hipMemset(src, 7, bytes);
  uint32_t *out; hipMalloc(&out, 256 * 1024 * 4);
  for (int pitch : {72, 29, 3, 6000}) for (int kind = 0; kind < 2; kind++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (kind == 0) gather<0><<<256, 1024>>>(src, bytes, pitch, out, cyc); else gather<1><<<256, 1024>>>(src, bytes, pitch, out, cyc);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> c(256); hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto x : c) mean += x; mean /= 256;
    printf("gather pitch %5d %-22s: %8.1f us kernel, %9.0f cycles/block (8 px per lane, 1024 lanes)\n", pitch,
           kind ? "1 unaligned dword" : "3 byte loads", ms * 1e3, mean);
  }
  return 0;
}
