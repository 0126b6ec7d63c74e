// rows_floor.hip -- what the memory system gives the TRAFFIC of a render launch when nothing but the traffic is left:
// the floor under the HBM-bound workloads of the record (DESIGN 4.0 "Roofline").  A launch of the render reads every
// byte of the source rows it samples (the samples of a row are closer together than a 128-byte line: 72 B apart for
// 1080p -> 80 columns, 57.6 B for 4K -> 200, 28.8 B for 4K -> 400) and writes its frames; this kernel moves exactly those
// bytes -- the sampled rows of each frame with coalesced 16-byte loads, the frame's output bytes with 16-byte
// non-temporal stores -- from one workgroup per frame, four launches in flight, a fresh set of frames every launch
// (sets far beyond the 256 MB Infinity Cache), and reports TB/s.  Forms: rows + frames (the render's mix), rows only,
// frames only.
// Build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-result scripts/ubench/rows_floor.hip -o scripts/ubench/rows_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// form: bit 0 = read the rows, bit 1 = write the frame
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
    k_rows(const uint8_t *__restrict__ src, size_t frame_bytes, int row_bytes, int src_h, int rows, uint8_t *__restrict__ out,
           size_t out_stride, int out_bytes, int form, uint32_t *__restrict__ sink) {
  const int f = blockIdx.x, tid = threadIdx.x;
  const uint8_t *fr = src + (size_t)f * frame_bytes;
  uint8_t *o = out + (size_t)f * out_stride;
  const int row_v = row_bytes / 16, out_v = out_bytes / 16;
  const uint32_t yr = (uint32_t)(((uint64_t)src_h << 16) / (uint32_t)rows) + 1u; // the sampler's row rule (image.c:293-294)
  u4 acc = {0u, 0u, 0u, 0u};
  // four rows' loads (all in flight together), then those rows' share of the frame's bytes: the render's interleaving
  // of reads and writes, with the memory-level parallelism of its request-ahead
  for (int r0 = 0; r0 < rows; r0 += 4) {
    const int r1 = r0 + 4 < rows ? r0 + 4 : rows;
    if (form & 1) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int r = r0 + k < rows ? r0 + k : rows - 1;
        uint32_t sy = ((uint32_t)r * yr) >> 16;
        sy = sy < (uint32_t)src_h - 1u ? sy : (uint32_t)src_h - 1u;
        const u4 *row = reinterpret_cast<const u4 *>(fr + (size_t)sy * row_bytes);
        if (r0 + k < rows)
          for (int v = tid; v < row_v; v += THREADS)
            acc ^= __builtin_nontemporal_load(row + v);
      }
    }
    if (form & 2) {
      // (shares cut at 128-byte lines: a wave's 1 KB store that straddles lines costs a quarter of the write rate, below)
      const int v0 = (int)((long long)out_v * r0 / rows) & ~7, v1 = r1 == rows ? out_v : (int)((long long)out_v * r1 / rows) & ~7;
      const u4 val = {(uint32_t)r0, (uint32_t)tid, (uint32_t)f, 0x20202020u};
      for (int v = v0 + tid; v < v1; v += THREADS)
        __builtin_nontemporal_store(val, reinterpret_cast<u4 *>(o) + v);
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u && sink)
    sink[f] = 1u;
}

// the write side alone, as plainly as it can be asked: `total` bytes, G workgroups, each 16-byte store coalesced with its
// neighbours; NT = non-temporal
template <bool NT> __global__ void __launch_bounds__(512) k_fill(u4 *__restrict__ out, size_t n_v, int per_wg_contig) {
  const u4 val = {(uint32_t)blockIdx.x, (uint32_t)threadIdx.x, 7u, 0x20202020u};
  if (per_wg_contig) { // every workgroup owns one contiguous share (the render's layout: a frame per workgroup)
    const size_t share = (n_v + gridDim.x - 1) / gridDim.x, v0 = share * blockIdx.x, v1 = v0 + share < n_v ? v0 + share : n_v;
    for (size_t v = v0 + threadIdx.x; v < v1; v += 512)
      if (NT) __builtin_nontemporal_store(val, out + v); else out[v] = val;
  } else { // grid-stride: the whole grid sweeps the buffer front to back
    for (size_t v = (size_t)blockIdx.x * 512 + threadIdx.x; v < n_v; v += (size_t)gridDim.x * 512)
      if (NT) __builtin_nontemporal_store(val, out + v); else out[v] = val;
  }
}

// the write side under the render's constraint -- 256 frames, each a contiguous slot written front to back -- with the
// two freedoms a kernel has: G workgroups share a frame (interleaved 8 KB pieces: the frame's front is G x 8 KB wide and
// 256 / G... frames are open at a time), and a thread issues BURST stores back to back before anything else
template <int BURST> __global__ void __launch_bounds__(512) k_frames(u4 *__restrict__ out, size_t frame_v, int frames, int G) {
  const u4 val = {(uint32_t)blockIdx.x, (uint32_t)threadIdx.x, 7u, 0x20202020u};
  const int groups = gridDim.x / G, grp = blockIdx.x / G, p = blockIdx.x % G;
  for (int f = grp; f < frames; f += groups) {
    u4 *o = out + (size_t)f * frame_v;
    for (size_t v0 = (size_t)p * 512 * BURST; v0 < frame_v; v0 += (size_t)G * 512 * BURST) {
#pragma unroll
      for (int b = 0; b < BURST; b++) {
        const size_t v = v0 + (size_t)b * 512 + threadIdx.x;
        if (v < frame_v)
          __builtin_nontemporal_store(val, o + v);
      }
    }
  }
}

struct Shape {
  const char *name;
  int src_w, src_h, rows, out_bytes;
};

int main(int argc, char **argv) {
  const int frames = 256, streams_n = argc > 1 ? atoi(argv[1]) : 4, launches = 400;
  const Shape shapes[] = {
      {"1080p -> 80x24 truecolor (24 rows of 5760 B, 36 KB out)", 1920, 1080, 24, 35952},
      {"1080p -> 80x24 ansi-256  (24 rows of 5760 B, 22 KB out)", 1920, 1080, 24, 22288},
      {"4K -> 200x60 truecolor   (60 rows of 11520 B, 225 KB out)", 3840, 2160, 60, 224768},
      {"4K -> 400x120 half block (240 rows of 11520 B, 1.85 MB out)", 3840, 2160, 240, 1845408},
  };
  hipStream_t st[8];
  for (int i = 0; i < streams_n; i++)
    hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
  uint32_t *sink;
  hipMalloc(&sink, frames * 4);
  printf("# 256 frames per launch, one workgroup of 512 threads per frame, %d launches in flight, %d launches timed; "
         "TB/s of the bytes named\n", streams_n, launches);
  for (const Shape &s : shapes) {
    const size_t frame_bytes = (size_t)s.src_w * s.src_h * 3;
    const int sets = (int)(((size_t)40 << 30) / (frame_bytes * frames)); // as many sets as fit 40 GB, 2..12
    const int nsets = sets < 2 ? 2 : (sets > 12 ? 12 : sets);
    const size_t out_stride = ((size_t)s.out_bytes + 4095) & ~(size_t)4095;
    uint8_t *src, *out;
    if (hipMalloc(&src, frame_bytes * frames * nsets) != hipSuccess || hipMalloc(&out, out_stride * frames * nsets) != hipSuccess) {
      printf("%s: allocation failed\n", s.name);
      return 1;
    }
    hipMemset(src, 0x5a, frame_bytes * frames * nsets);
    hipMemset(out, 0, out_stride * frames * nsets);
    const double rd = (double)s.rows * s.src_w * 3 * frames, wr = (double)(s.out_bytes / 16 * 16) * frames;
    printf("%s, %d input sets: rows %.1f MB + frames %.1f MB per launch\n", s.name, nsets, rd / 1e6, wr / 1e6);
    for (int form : {3, 1, 2}) {
      auto run = [&](int n) {
        for (int i = 0; i < n; i++) {
          const int set = i % nsets;
          hipLaunchKernelGGL(k_rows<512>, dim3(frames), dim3(512), 0, st[i % streams_n], src + frame_bytes * frames * set, frame_bytes,
                             s.src_w * 3, s.src_h, s.rows, out + out_stride * frames * set, out_stride, s.out_bytes, form, sink);
        }
      };
      run(40);
      hipDeviceSynchronize();
      const auto t0 = std::chrono::steady_clock::now();
      run(launches);
      hipDeviceSynchronize();
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / launches;
      const double bytes = (form & 1 ? rd : 0.0) + (form & 2 ? wr : 0.0);
      printf("   %-14s %8.2f us per launch = %5.2f TB/s\n", form == 3 ? "rows + frames" : form == 1 ? "rows only" : "frames only", us,
             bytes / us / 1e6);
    }
    hipFree(src);
    hipFree(out);
  }
  // ---- the write side alone: 472 MB (a 4K -> 400x120 launch's frames), fresh buffer every launch
  {
    const size_t bytes = (size_t)1845408 / 16 * 16 * 256, n_v = bytes / 16;
    const int nsets = 6;
    u4 *buf;
    hipMalloc(&buf, bytes * nsets);
    printf("write side alone, %.1f MB per launch, %d launches in flight\n", bytes / 1e6, streams_n);
    auto timeit = [&](const char *name, auto launch) {
      for (int i = 0; i < 12; i++) launch(i);
      hipDeviceSynchronize();
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 120; i++) launch(i);
      hipDeviceSynchronize();
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 120;
      printf("   %-58s %8.2f us = %5.2f TB/s\n", name, us, bytes / us / 1e6);
    };
    for (int g : {256, 1024, 4096}) {
      char nm[128];
      snprintf(nm, sizeof nm, "%d workgroups, contiguous share each, non-temporal", g);
      timeit(nm, [&](int i) { hipLaunchKernelGGL(k_fill<true>, dim3(g), dim3(512), 0, st[i % streams_n], buf + n_v * (i % nsets), n_v, 1); });
      snprintf(nm, sizeof nm, "%d workgroups, contiguous share each, plain stores", g);
      timeit(nm, [&](int i) { hipLaunchKernelGGL(k_fill<false>, dim3(g), dim3(512), 0, st[i % streams_n], buf + n_v * (i % nsets), n_v, 1); });
      snprintf(nm, sizeof nm, "%d workgroups, grid-stride sweep, non-temporal", g);
      timeit(nm, [&](int i) { hipLaunchKernelGGL(k_fill<true>, dim3(g), dim3(512), 0, st[i % streams_n], buf + n_v * (i % nsets), n_v, 0); });
      snprintf(nm, sizeof nm, "%d workgroups, grid-stride sweep, plain stores", g);
      timeit(nm, [&](int i) { hipLaunchKernelGGL(k_fill<false>, dim3(g), dim3(512), 0, st[i % streams_n], buf + n_v * (i % nsets), n_v, 0); });
    }
    for (size_t frame_v : {(size_t)1845408 / 16, (size_t)1845504 / 16})
    for (int G : {1, 4, 256}) {
      char nm[128];
      snprintf(nm, sizeof nm, "256 frame slots of %zu B, %3d workgroups per frame (%3d frames open), burst 1", frame_v * 16, G, 256 / G);
      timeit(nm, [&](int i) { hipLaunchKernelGGL(k_frames<1>, dim3(256), dim3(512), 0, st[i % streams_n], buf + n_v * (i % nsets), frame_v, 256, G); });
      snprintf(nm, sizeof nm, "256 frame slots of %zu B, %3d workgroups per frame (%3d frames open), burst 8", frame_v * 16, G, 256 / G);
      timeit(nm, [&](int i) { hipLaunchKernelGGL(k_frames<8>, dim3(256), dim3(512), 0, st[i % streams_n], buf + n_v * (i % nsets), frame_v, 256, G); });
    }
    timeit("hipMemsetAsync", [&](int i) { hipMemsetAsync(buf + n_v * (i % nsets), 0x20, bytes, st[i % streams_n]); });
    hipFree(buf);
  }
  return 0;
}
