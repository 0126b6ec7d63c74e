/*
 * stream_kernels.hpp -- the two full-frame passes of the client display path as stand-alone HBM-streaming
 * kernels: apply_color_filter (lib/video/rgba/color_filter.c:274-345) and the
 * x/y flips of session_display_convert_to_ascii (src/common/session/display.c:546-600).
 *
 * The render kernel does not need them -- achip_frame_t.ops folds both into its sampler, touching ~2 K
 * pixels instead of 2 M -- but callers that want the transformed IMAGE get it at memory speed: every byte
 * is read once and written once with 16-byte accesses (16 RGB24 pixels = 48 bytes = three uint4 per
 * thread); these are the genuinely bandwidth-bound neighbours of the path.
 */
#pragma once

#include "render_kernels.hpp"

namespace achip {

/* one RGB24 pixel through the filter; `ops` as in achip_frame_t.ops */
__device__ inline uint32_t tint_rgb(uint32_t r, uint32_t g, uint32_t b, uint32_t ops) {
  return tint_pixel(r | (g << 8) | (b << 16), ops);
}

/* 48 bytes = 16 pixels held in 12 dwords: apply `fn` to every pixel in place */
template <class F> __device__ inline void map_16_pixels(uint32_t (&w)[12], F fn) {
#pragma unroll
  for (int q = 0; q < 4; q++) { /* 4 pixels = 3 dwords: R0G0B0R1 G1B1R2G2 B2R3G3B3 */
    uint32_t &a = w[3 * q], &b = w[3 * q + 1], &c = w[3 * q + 2];
    const uint32_t p0 = fn(a & 0x00FFFFFFu);
    const uint32_t p1 = fn((a >> 24) | ((b & 0xFFFFu) << 8));
    const uint32_t p2 = fn((b >> 16) | ((c & 0xFFu) << 16));
    const uint32_t p3 = fn(c >> 8);
    a = p0 | (p1 << 24);
    b = (p1 >> 8) | (p2 << 16);
    c = (p2 >> 16) | (p3 << 8);
  }
}

/* in-place tint of a tightly packed RGB24 buffer of `nbytes` (multiple of 3), base 16-byte aligned */
__global__ void __launch_bounds__(256) tint_stream_kernel(uint8_t *__restrict__ px, uint64_t nbytes, uint32_t ops) {
  const uint64_t groups = nbytes / 48u;
  for (uint64_t gidx = (uint64_t)blockIdx.x * 256u + threadIdx.x; gidx < groups; gidx += (uint64_t)gridDim.x * 256u) {
    uint4 *p = reinterpret_cast<uint4 *>(px + gidx * 48u);
    uint4 v0 = p[0], v1 = p[1], v2 = p[2];
    uint32_t w[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
    map_16_pixels(w, [ops](uint32_t rgb) { return tint_pixel(rgb, ops); });
    p[0] = make_uint4(w[0], w[1], w[2], w[3]);
    p[1] = make_uint4(w[4], w[5], w[6], w[7]);
    p[2] = make_uint4(w[8], w[9], w[10], w[11]);
  }
  /* tail (< 16 pixels) */
  const uint64_t tail0 = groups * 48u;
  for (uint64_t o = tail0 + 3u * ((uint64_t)blockIdx.x * 256u + threadIdx.x); o + 2u < nbytes;
       o += 3u * (uint64_t)gridDim.x * 256u) {
    const uint32_t r = tint_rgb(px[o], px[o + 1], px[o + 2], ops);
    px[o] = (uint8_t)r;
    px[o + 1] = (uint8_t)(r >> 8);
    px[o + 2] = (uint8_t)(r >> 16);
  }
}

/* generic (strided / unaligned) variant: one pixel per thread */
__global__ void __launch_bounds__(256)
    tint_pixels_kernel(uint8_t *__restrict__ px, int w, int h, int stride, uint32_t ops) {
  const uint64_t total = (uint64_t)w * (uint64_t)h;
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256u) {
    const uint32_t y = (uint32_t)(i / (uint32_t)w), x = (uint32_t)(i - (uint64_t)y * (uint32_t)w);
    uint8_t *p = px + (size_t)y * (size_t)stride + (size_t)x * 3u;
    const uint32_t r = tint_rgb(p[0], p[1], p[2], ops);
    p[0] = (uint8_t)r;
    p[1] = (uint8_t)(r >> 8);
    p[2] = (uint8_t)(r >> 16);
  }
}

/* out-of-place flip of a tightly packed image whose rows are a multiple of 48 bytes (w % 16 == 0):
 * group g of source row y goes, pixel-reversed when flip_x, to the mirrored group of the target row */
__global__ void __launch_bounds__(256)
    flip_stream_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int h, uint32_t ops) {
  const uint32_t gpr = (uint32_t)w / 16u; /* groups per row */
  const uint64_t groups = (uint64_t)gpr * (uint64_t)h;
  for (uint64_t gidx = (uint64_t)blockIdx.x * 256u + threadIdx.x; gidx < groups; gidx += (uint64_t)gridDim.x * 256u) {
    const uint32_t y = (uint32_t)(gidx / gpr), g = (uint32_t)(gidx - (uint64_t)y * gpr);
    const uint4 *p = reinterpret_cast<const uint4 *>(src + ((size_t)y * gpr + g) * 48u);
    const uint4 v0 = p[0], v1 = p[1], v2 = p[2];
    uint32_t a[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
    uint32_t o[12];
    if (ops & ACHIP_OP_FLIP_X) {
      /* pixel k of the output group = pixel 15-k of the input group */
      uint32_t px[16];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        px[4 * q + 0] = a[3 * q] & 0x00FFFFFFu;
        px[4 * q + 1] = (a[3 * q] >> 24) | ((a[3 * q + 1] & 0xFFFFu) << 8);
        px[4 * q + 2] = (a[3 * q + 1] >> 16) | ((a[3 * q + 2] & 0xFFu) << 16);
        px[4 * q + 3] = a[3 * q + 2] >> 8;
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t p0 = px[15 - 4 * q], p1 = px[14 - 4 * q], p2 = px[13 - 4 * q], p3 = px[12 - 4 * q];
        o[3 * q] = p0 | (p1 << 24);
        o[3 * q + 1] = (p1 >> 8) | (p2 << 16);
        o[3 * q + 2] = (p2 >> 16) | (p3 << 8);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 12; k++)
        o[k] = a[k];
    }
    const uint32_t ty = (ops & ACHIP_OP_FLIP_Y) ? (uint32_t)h - 1u - y : y;
    const uint32_t tg = (ops & ACHIP_OP_FLIP_X) ? gpr - 1u - g : g;
    uint4 *q = reinterpret_cast<uint4 *>(dst + ((size_t)ty * gpr + tg) * 48u);
    q[0] = make_uint4(o[0], o[1], o[2], o[3]);
    q[1] = make_uint4(o[4], o[5], o[6], o[7]);
    q[2] = make_uint4(o[8], o[9], o[10], o[11]);
  }
}

/* generic flip: one pixel per thread, any width / stride */
__global__ void __launch_bounds__(256)
    flip_pixels_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int h, int src_stride,
                       int dst_stride, uint32_t ops) {
  const uint64_t total = (uint64_t)w * (uint64_t)h;
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256u) {
    const uint32_t y = (uint32_t)(i / (uint32_t)w), x = (uint32_t)(i - (uint64_t)y * (uint32_t)w);
    const uint32_t sx = (ops & ACHIP_OP_FLIP_X) ? (uint32_t)w - 1u - x : x;
    const uint32_t sy = (ops & ACHIP_OP_FLIP_Y) ? (uint32_t)h - 1u - y : y;
    const uint8_t *s = src + (size_t)sy * (size_t)src_stride + (size_t)sx * 3u;
    uint8_t *d = dst + (size_t)y * (size_t)dst_stride + (size_t)x * 3u;
    d[0] = s[0];
    d[1] = s[1];
    d[2] = s[2];
  }
}


/* ------------------------------------------------------------------------------------------- */
/* Compaction of a rendered slab ("prefer gathering compacted per-rank buffers ... lengths first").  The     */
/* render kernels leave frame i at slab + i*stride (stride = the worst case: 44.5 KB for 80x24 truecolor, where real    */
/* video is 2-4 KB); what a consumer on the other side of PCIe or xGMI wants is the bytes that are used -- the reference */
/* ships exactly frame_size bytes per client (lib/network/acip/server.c:190-222).  Frame i goes to                      */
/* dst + off[i], off[i] = sum over j < i of round16(len[j]): frame starts stay 16-byte aligned, so every byte moves in  */
/* uint4 accesses (<= 15 bytes of padding per frame).  dst may be device memory or PINNED HOST memory mapped into the   */
/* device: the kernel's stores then ARE the transfer -- exact length, no second DMA, no host round trip for a size.     */
/* Workgroup (i, y) handles slice y of frame i; every workgroup recomputes the prefix it needs from len[] (n <= a few   */
/* thousand L2-resident words), so there is no inter-workgroup hand-off.  A frame whose length is a render error code   */
/* (>= 0xFFFFFFF0) takes no room.  A frame whose last 16-byte group would end beyond dst_capacity is not copied (the      */
/* caller sees off[n] > dst_capacity).                                                                                  */
/* ------------------------------------------------------------------------------------------- */
__device__ inline uint32_t pack_len_ok(uint32_t l) { return l >= 0xFFFFFFF0u ? 0u : l; }

__global__ void __launch_bounds__(256)
    pack_frames_kernel(const uint8_t *__restrict__ slab, uint64_t stride, const uint32_t *__restrict__ len, int n,
                       uint8_t *__restrict__ dst, uint64_t dst_capacity, uint64_t *__restrict__ off_out,
                       uint32_t *__restrict__ len_out) {
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = (int)blockIdx.x;
  uint32_t *wsum = lds_ptr<uint32_t>(0); /* 4 wave totals x {below i, all} in units of 16 bytes */
  /* 16-byte groups in front of frame i, and (workgroup (n-1, 0) only needs it) of the whole slab */
  uint32_t below = 0, all = 0;
  for (int j = tid; j < n; j += 256) {
    const uint32_t g = (pack_len_ok(len[j]) + 15u) >> 4;
    below += j < i ? g : 0u;
    all += g;
  }
  below = wave_read_lane(wave_inclusive_scan(below), 63);
  all = wave_read_lane(wave_inclusive_scan(all), 63);
  if (lane == 0) {
    wsum[wave] = below;
    wsum[4 + wave] = all;
  }
  __syncthreads();
  const uint64_t off = 16ull * ((uint64_t)wsum[0] + wsum[1] + wsum[2] + wsum[3]);
  const uint64_t total = 16ull * ((uint64_t)wsum[4] + wsum[5] + wsum[6] + wsum[7]);
  const uint32_t l = pack_len_ok(len[i]);
  if (blockIdx.y == 0 && tid == 0) {
    if (off_out) {
      off_out[i] = off;
      if (i == n - 1)
        off_out[n] = total;
    }
    if (len_out)
      len_out[i] = len[i]; /* error codes travel as they are */
  }
  /* slice y of the frame's 16-byte groups; the last group may carry up to 15 stale bytes of the slot behind the frame's
   * end (the slot is stride >= round16(len) wide), which the padding rule allows -- so the frame fits only if its last
   * WHOLE group does: no store ever lands behind dst_capacity, whatever its alignment */
  const uint32_t groups = (l + 15u) >> 4;
  if (off + 16ull * groups > dst_capacity)
    return;
  const uint32_t per = (groups + gridDim.y - 1u) / gridDim.y;
  const uint32_t g0 = blockIdx.y * per, g1 = min(groups, g0 + per);
  const uint4 *src4 = reinterpret_cast<const uint4 *>(slab + (uint64_t)i * stride);
  uint4 *dst4 = reinterpret_cast<uint4 *>(dst + off);
  /* thread t takes the group that lies t groups behind a 128-byte line boundary of the DESTINATION address (frames are
   * packed at 16-byte granularity: without this every wave's 1 KB store would straddle lines, which the memory side
   * takes a quarter slower: profiles/r04_rows_floor.txt); at most seven threads sit out the first trip */
  const uint32_t shift = (uint32_t)((uintptr_t)dst4 >> 4) & 7u;
  uint32_t a = ((g0 + shift) & ~7u) + (uint32_t)tid; /* group index + shift */
  for (; a + 256u < g1 + shift; a += 512u) { /* two groups per trip: both loads in flight */
    const uint32_t g = a - shift;            /* (wraps below zero only where a < g0 + shift: skipped) */
    const bool first = a >= g0 + shift;
    uint4 x = make_uint4(0u, 0u, 0u, 0u);
    if (first)
      x = src4[g];
    const uint4 y = src4[g + 256u];
    if (first)
      dst4[g] = x;
    dst4[g + 256u] = y;
  }
  if (a >= g0 + shift && a < g1 + shift)
    dst4[a - shift] = src4[a - shift];
}


/* ------------------------------------------------------------------------------------------- */
/* Ingest of the SAMPLED rows only (frame_table_publish_rows).  The renderer point-samples out_h of a      */
/* frame's src_h rows (image.c:293-325): 24 of 1080 for an 80x24 target -- 138 KB of a 6.2 MB frame.  The host packs     */
/* those rows behind a table of their indices and sends ONE DMA; this kernel puts them where the frame's descriptor       */
/* expects them (the buffer keeps the full frame's layout, so plans and their output do not change).                      */
/* staged = [n_rows x u32 row index, padded to 16 bytes][n_rows x row_bytes].  Workgroup (x, r) copies slice x of row r.   */
/* ------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
    scatter_rows_kernel(const uint8_t *__restrict__ staged, uint32_t n_rows, uint32_t row_bytes, uint8_t *__restrict__ frame,
                        uint64_t frame_pitch) {
  const uint32_t r = blockIdx.y;
  if (r >= n_rows)
    return;
  const uint32_t table = (n_rows * 4u + 15u) & ~15u;
  const uint32_t row = reinterpret_cast<const uint32_t *>(staged)[r];
  const uint8_t *src = staged + table + (uint64_t)r * row_bytes;
  uint8_t *dst = frame + (uint64_t)row * frame_pitch;
  const uint32_t i0 = blockIdx.x * 256u + threadIdx.x, step = gridDim.x * 256u;
  if ((((uintptr_t)src | (uintptr_t)dst) & 15u) == 0u) {
    const uint32_t groups = row_bytes >> 4;
    for (uint32_t g = i0; g < groups; g += step)
      reinterpret_cast<uint4 *>(dst)[g] = reinterpret_cast<const uint4 *>(src)[g];
    for (uint32_t b = (groups << 4) + i0; b < row_bytes; b += step)
      dst[b] = src[b];
  } else {
    for (uint32_t b = i0; b < row_bytes; b += step)
      dst[b] = src[b];
  }
}


/* The same for a whole tick's clients in ONE launch (frame_table_publish_rows_batch): the staged block starts with one
 * 32-byte record per client {frame pointer, offset of its block, rows, frame row bytes, columns}; workgroup (x, r, c)
 * copies slice x of row r of client c.  columns = 0: the block is [row table][rows], whole rows are copied; otherwise it
 * is [row table][column table][rows x columns pixels] -- only the pixels the targets sample were staged (a frame at most
 * half as wide as its source) -- and every pixel goes to its place in the full-geometry frame. */
struct scatter_client_t {
  uint64_t frame;
  uint32_t off, n_rows, row_bytes, n_cols, _pad[2];
};
__global__ void __launch_bounds__(256) scatter_rows_batch_kernel(const uint8_t *__restrict__ staged, uint32_t n_clients) {
  const uint32_t c = blockIdx.z;
  if (c >= n_clients)
    return;
  const scatter_client_t cl = reinterpret_cast<const scatter_client_t *>(staged)[c];
  const uint32_t r = blockIdx.y;
  if (r >= cl.n_rows)
    return;
  const uint32_t table = (cl.n_rows * 4u + 15u) & ~15u;
  const uint8_t *blk = staged + cl.off;
  const uint32_t row = reinterpret_cast<const uint32_t *>(blk)[r];
  uint8_t *dst = reinterpret_cast<uint8_t *>(cl.frame) + (uint64_t)row * cl.row_bytes;
  const uint32_t i0 = blockIdx.x * 256u + threadIdx.x, step = gridDim.x * 256u;
  if (cl.n_cols) {
    const uint32_t *cols = reinterpret_cast<const uint32_t *>(blk + table);
    const uint8_t *src = blk + table + ((cl.n_cols * 4u + 15u) & ~15u) + (uint64_t)r * cl.n_cols * 3u;
    for (uint32_t i = i0; i < cl.n_cols; i += step) {
      uint8_t *d = dst + (uint64_t)cols[i] * 3u;
      d[0] = src[3u * i], d[1] = src[3u * i + 1u], d[2] = src[3u * i + 2u];
    }
    return;
  }
  const uint8_t *src = blk + table + (uint64_t)r * cl.row_bytes;
  if ((((uintptr_t)src | (uintptr_t)dst) & 15u) == 0u) {
    const uint32_t groups = cl.row_bytes >> 4;
    for (uint32_t g = i0; g < groups; g += step)
      reinterpret_cast<uint4 *>(dst)[g] = reinterpret_cast<const uint4 *>(src)[g];
    for (uint32_t b = (groups << 4) + i0; b < cl.row_bytes; b += step)
      dst[b] = src[b];
  } else {
    for (uint32_t b = i0; b < cl.row_bytes; b += step)
      dst[b] = src[b];
  }
}

} // namespace achip
