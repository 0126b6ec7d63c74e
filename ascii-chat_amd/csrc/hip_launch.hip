/*
 * hip_launch.hip -- plain-C launchers of the gfx950 kernels (the frame kernel itself is instantiated per
 * geometry in render_inst.hip).
 * Built only with hipcc --offload-arch=gfx950; there is no host/CPU variant of these entry points.
 */
#include "hip_launch.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>

#include "render_inst.h"
#include "render_stream.hpp"
#include "crc_kernels.hpp"
#include "render_variants.h"

extern "C" int achip_launch_render(int mode, int variant, int has_composite, const achip_frame_t *frames_dev, int n_frames,
                                   const achip_lut_t *lut_dev, uint8_t *out, uint64_t out_stride, uint32_t *out_len,
                                   unsigned long long *prof, int parts, int rows_per_part,
                                   unsigned long long *part_sync, uint32_t epoch, const achip_uniform_t *uniform,
                                   void *stream) {
  if (n_frames <= 0)
    return (int)hipSuccess;
  if (parts < 1 || (parts > 1 && (!part_sync || rows_per_part < 1)))
    return (int)hipErrorInvalidValue;
  if (ACHIP_IS_ROWS_VARIANT(variant)) { /* wave-autonomous kernel of the run-structured modes (render_rows.hpp) */
    /* (the profiled entry points' per-wave stamps do not exist here: prof is ignored) */
    /* a frame's blocks shared out over `parts` workgroups: the PARTS geometry only (rows_per_part means nothing here) */
    if (parts != 1 && (!ACHIP_ROWS_VARIANT_PARTS(variant) || parts > 64 || epoch == 0u))
      return (int)hipErrorInvalidValue;
    const achip_partsdev_t ps = {parts, epoch, part_sync};
    switch (variant) {
#define X(id, W, C)                                                                                                    \
  case id:                                                                                                             \
    return achipk_render_rinst_launch_##id(mode, has_composite, frames_dev, n_frames, lut_dev, out, out_stride, out_len, \
                                          uniform, nullptr, &ps, stream);
      ACHIP_ROWS_VARIANTS(X)
#undef X
    }
    return (int)hipErrorInvalidValue;
  }
  if (ACHIP_IS_STREAM_VARIANT(variant)) { /* wave-autonomous kernel: per-cell modes (render_stream.hpp) */
    if (parts != 1) { /* a frame's blocks shared out over `parts` workgroups (rows_per_part means nothing here) */
      if (variant != 18 || parts > 64 || epoch == 0u)
        return (int)hipErrorInvalidValue;
      const achip_partsdev_t ps = {parts, epoch, part_sync};
      return achipk_render_sinst_parts_launch_18(mode, has_composite, frames_dev, n_frames, lut_dev, out, out_stride, out_len, uniform,
                                                prof, &ps, stream);
    }
    switch (variant) {
#define X(id, W, C)                                                                                                    \
  case id:                                                                                                             \
    return achipk_render_sinst_launch_##id(mode, has_composite, frames_dev, n_frames, lut_dev, out, out_stride, out_len, \
                                          uniform, prof, nullptr, stream);
      ACHIP_STREAM_VARIANTS(X)
#undef X
    }
    return (int)hipErrorInvalidValue;
  }
  switch (variant) { /* one translation unit per geometry: render_inst.hip */
#define X(id, B, C, R)                                                                                                 \
  case id:                                                                                                             \
    return achipk_render_inst_launch_##id(mode, has_composite, frames_dev, n_frames, lut_dev, out, out_stride, out_len, \
                                         prof, parts, rows_per_part, part_sync, epoch, uniform, stream);
    ACHIP_VARIANTS(X)
#undef X
  }
  return (int)hipErrorInvalidValue;
}

/* whole-frame launch of a per-cell mode with the frame CRC-32C riding the drain (stream geometries 16 / 17) */
extern "C" int achip_launch_render_crc(int mode, int variant, int has_composite, const achip_frame_t *frames_dev,
                                       int n_frames, const achip_lut_t *lut_dev, uint8_t *out, uint64_t out_stride,
                                       uint32_t *out_len, const achip_wire_t *wire, const achip_uniform_t *uniform,
                                       unsigned long long *prof, void *stream) {
  if (n_frames <= 0)
    return (int)hipSuccess;
  if (!wire || !wire->crc)
    return (int)hipErrorInvalidValue;
  switch (variant) {
#define X(id, W, C)                                                                                                    \
  case id:                                                                                                             \
    return achipk_render_sinst_launch_##id(mode, has_composite, frames_dev, n_frames, lut_dev, out, out_stride, out_len, \
                                          uniform, prof, wire, stream);
    ACHIP_STREAM_VARIANTS(X)
#undef X
#define X(id, W, C)                                                                                                    \
  case id:                                                                                                             \
    return achipk_render_rinst_launch_##id(mode, has_composite, frames_dev, n_frames, lut_dev, out, out_stride, out_len,  \
                                          uniform, wire, nullptr, stream);
    ACHIP_ROWS_VARIANTS(X)
#undef X
  }
  return (int)hipErrorInvalidValue;
}
extern "C" int achip_launch_render_pack(int mode, int variant, const achip_frame_t *frames_dev, int n_frames,
                                        const achip_lut_t *lut_dev, uint64_t bound, uint32_t *out_len, const achip_wire_t *wire,
                                        const achip_uniform_t *uniform, const achip_packdev_t *pack, void *stream) {
  if (n_frames <= 0)
    return (int)hipSuccess;
  if (bound > (uint64_t)ACHIP_PACK_FRAME_CAP)
    return (int)hipErrorInvalidValue;
  /* 1024 threads while every frame has a CU to itself (variant 16: one block per wave), 512-thread workgroups otherwise */
  return variant == 16 ? achipk_render_sinst_pack_launch_16(mode, frames_dev, n_frames, lut_dev, bound, out_len, uniform, wire, pack, stream)
                       : achipk_render_sinst_pack_launch_17(mode, frames_dev, n_frames, lut_dev, bound, out_len, uniform, wire, pack, stream);
}
extern "C" int achip_pack_frame_cap(void) { return ACHIP_PACK_FRAME_CAP; }
/* exact-length truecolor frames of any size in ONE launch (render_stream.hpp LF): stream geometries 16 / 17 */
extern "C" int achipk_render_sinst_lenfirst_launch_16(const achip_frame_t *, int, const achip_lut_t *, uint64_t, uint32_t *, const achip_uniform_t *, const achip_packdev_t *, void *);
extern "C" int achipk_render_sinst_lenfirst_launch_17(const achip_frame_t *, int, const achip_lut_t *, uint64_t, uint32_t *, const achip_uniform_t *, const achip_packdev_t *, void *);
extern "C" int achip_launch_render_length_first(int variant, const achip_frame_t *frames_dev, int n_frames, const achip_lut_t *lut_dev,
                                                uint64_t bound, uint32_t *out_len, const achip_uniform_t *uniform,
                                                const achip_packdev_t *pack, void *stream) {
  if (n_frames <= 0)
    return (int)hipSuccess;
  return variant == 16 ? achipk_render_sinst_lenfirst_launch_16(frames_dev, n_frames, lut_dev, bound, out_len, uniform, pack, stream)
                       : achipk_render_sinst_lenfirst_launch_17(frames_dev, n_frames, lut_dev, bound, out_len, uniform, pack, stream);
}
#ifdef ACHIP_ALL_GEOMETRIES
extern "C" int achip_variant_has_crc(int variant) { return variant == 16 || variant == 17 || (ACHIP_IS_ROWS_VARIANT(variant) && variant != 26 && !ACHIP_ROWS_VARIANT_WIDE(variant) && !ACHIP_ROWS_VARIANT_PARTS(variant)); }
#else
extern "C" int achip_variant_has_crc(int variant) { return variant == 16 || variant == 17; }
#endif
/* ... and whether riding the drain beats a second pass over the slab there (measured, profiles/r03_rows_kernel.txt): yes
 * for the per-cell modes' stream kernel (+2.4 us against +8.3 us per 256-frame step); no for the rows kernel, whose
 * per-slice checksum chains cost more than the stand-alone kernel's pass (+12 against +8 us on 80x24 half blocks, +240
 * against +99 us on 400x120) -- there the fused form only runs when asked for (asciichat_hip_plan_set_fused_crc) */
extern "C" int achip_variant_crc_pays(int variant) { return variant == 16 || variant == 17; }

extern "C" int achip_launch_packets_from_crc(const uint32_t *len_dev, const uint32_t *crc_dev, const uint32_t *dims_dev,
                                             int n, uint8_t *hdr_out, uint32_t *pkt_crc_out, void *stream) {
  if (n <= 0)
    return (int)hipSuccess;
  hipLaunchKernelGGL(achip::crc_packets_kernel, dim3((unsigned)(n + 255) / 256u), dim3(256), 0,
                     static_cast<hipStream_t>(stream), len_dev, crc_dev, dims_dev, n, hdr_out, pkt_crc_out);
  return (int)hipGetLastError();
}

extern "C" int achip_variant_block(int variant) {
  switch (variant) {
#define X(id, W, C)                                                                                                    \
  case id:                                                                                                             \
    return 64 * W;
    ACHIP_STREAM_VARIANTS(X)
    ACHIP_ROWS_VARIANTS(X)
#undef X
#define X(id, B, C, R)                                                                                                 \
  case id:                                                                                                             \
    return B;
    ACHIP_VARIANTS(X)
#undef X
  }
  return -1;
}

extern "C" int achip_variant_cap(int variant) {
  switch (variant) { /* stream geometries: cells per FRAME (rows need not fit anything) */
#define X(id, W, C)                                                                                                    \
  case id:                                                                                                             \
    return ACHIP_STREAM_MAXBLK * 64 * C;
    ACHIP_STREAM_VARIANTS(X)
#undef X
#define X(id, W, C) /* rows geometries: cells of the widest padded row -- a block's, or W segments of a block each */   \
  case id:                                                                                                             \
    return !ACHIP_ROWS_VARIANT_WIDE(id) ? 64 * C : 64 * C * W < ACHIP_ROWS_WIDE_MAX_ROW ? 64 * C * W : ACHIP_ROWS_WIDE_MAX_ROW;
    ACHIP_ROWS_VARIANTS(X)
#undef X
#define X(id, B, C, R)                                                                                                 \
  case id:                                                                                                             \
    return C;
    ACHIP_VARIANTS(X)
#undef X
  }
  return -1;
}

extern "C" int achip_variant_lds_bytes(int mode, int variant) {
  switch (variant) {
#define X(id, W, C)                                                                                                    \
  case id:                                                                                                             \
    return achipk_render_sinst_lds_##id(mode);
    ACHIP_STREAM_VARIANTS(X)
#undef X
#define X(id, W, C)                                                                                                    \
  case id:                                                                                                             \
    return achipk_render_rinst_lds_##id(mode);
    ACHIP_ROWS_VARIANTS(X)
#undef X
#define X(id, B, C, R)                                                                                                 \
  case id:                                                                                                             \
    return achipk_render_inst_lds_##id(mode);
    ACHIP_VARIANTS(X)
#undef X
  }
  return -1;
}

extern "C" int achip_launch_resize(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh, void *stream) {
  const uint32_t xr = (uint32_t)((((uint64_t)sw << 16) / (uint64_t)dw) + 1u);
  const uint32_t yr = (uint32_t)((((uint64_t)sh << 16) / (uint64_t)dh) + 1u);
  const uint64_t total = (uint64_t)dw * (uint64_t)dh;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 2048u)
    blocks = 2048u;
  hipLaunchKernelGGL(achip::resize_nn_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, sw, sh,
                     3 * sw, dst, dw, dh, xr, yr);
  return (int)hipGetLastError();
}

extern "C" int achip_launch_resize_batch(const achip_resize_batch_t *batch, void *stream) {
  if (!batch || batch->n <= 0)
    return (int)hipSuccess;
  if (batch->n > ACHIP_RESIZE_BATCH_MAX)
    return (int)hipErrorInvalidValue;
  achip_resize_batch_t b = *batch;
  uint64_t most = 1;
  for (int k = 0; k < b.n; k++) {
    achip_resize_item_t *it = &b.item[k];
    if (!it->src || !it->dst || it->sw <= 0 || it->sh <= 0 || it->dw <= 0 || it->dh <= 0)
      return (int)hipErrorInvalidValue;
    it->x_ratio = (uint32_t)((((uint64_t)it->sw << 16) / (uint64_t)it->dw) + 1u);
    it->y_ratio = (uint32_t)((((uint64_t)it->sh << 16) / (uint64_t)it->dh) + 1u);
    const uint64_t total = (uint64_t)it->dw * (uint64_t)it->dh;
    if (total > most)
      most = total;
  }
  unsigned blocks = (unsigned)((most + 255) / 256);
  if (blocks > 2048u)
    blocks = 2048u;
  hipLaunchKernelGGL(achip::resize_nn_batch_kernel, dim3(blocks, (unsigned)b.n), dim3(256), 0,
                     static_cast<hipStream_t>(stream), b);
  return (int)hipGetLastError();
}

extern "C" int achip_launch_composite(const achip_composite_t *comp_dev, int canvas_w, int canvas_h, uint8_t *dst,
                                      void *stream) {
  const uint64_t total = (uint64_t)canvas_w * (uint64_t)canvas_h;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 2048u)
    blocks = 2048u;
  hipLaunchKernelGGL(achip::composite_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), comp_dev,
                     dst);
  return (int)hipGetLastError();
}

/* ---- display-path streaming passes (stream_kernels.hpp) -------------------------------------------- */
#include "stream_kernels.hpp"

static unsigned stream_blocks(uint64_t items) {
  uint64_t b = (items + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 256u * 16u ? 256u * 16u : b)); /* <= 16 workgroups per CU, grid-stride beyond */
}

extern "C" int achip_launch_tint(uint8_t *px, int w, int h, int stride, uint32_t ops, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (stride == 3 * w && (reinterpret_cast<uintptr_t>(px) & 15u) == 0) {
    const uint64_t nbytes = (uint64_t)w * (uint64_t)h * 3u;
    hipLaunchKernelGGL(achip::tint_stream_kernel, dim3(stream_blocks(nbytes / 48u + 1)), dim3(256), 0, s, px, nbytes,
                       ops);
  } else {
    hipLaunchKernelGGL(achip::tint_pixels_kernel, dim3(stream_blocks((uint64_t)w * (uint64_t)h)), dim3(256), 0, s, px,
                       w, h, stride, ops);
  }
  return (int)hipGetLastError();
}

extern "C" int achip_launch_flip(const uint8_t *src, uint8_t *dst, int w, int h, int src_stride, int dst_stride,
                                 uint32_t ops, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = w % 16 == 0 && src_stride == 3 * w && dst_stride == 3 * w &&
                   ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
  if (vec)
    hipLaunchKernelGGL(achip::flip_stream_kernel, dim3(stream_blocks((uint64_t)(w / 16) * (uint64_t)h)), dim3(256), 0,
                       s, src, dst, w, h, ops);
  else
    hipLaunchKernelGGL(achip::flip_pixels_kernel, dim3(stream_blocks((uint64_t)w * (uint64_t)h)), dim3(256), 0, s, src,
                       dst, w, h, src_stride, dst_stride, ops);
  return (int)hipGetLastError();
}

namespace achip {
__global__ void __launch_bounds__(64) comp_poke_kernel(achip_composite_t *comp, achip_comp_poke_t poke) {
  if (threadIdx.x < 9u)
    comp->s[threadIdx.x].src = poke.src[threadIdx.x];
}
} // namespace achip
extern "C" int achip_launch_comp_poke(achip_composite_t *comp_dev, const achip_comp_poke_t *poke, void *stream) {
  hipLaunchKernelGGL(achip::comp_poke_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), comp_dev, *poke);
  return (int)hipGetLastError();
}

/* rows of a frame, staged behind their index table, to their places in the frame buffer (frame_table_publish_rows) */
extern "C" int achip_launch_scatter_rows(const uint8_t *staged_dev, uint32_t n_rows, uint32_t row_bytes, uint8_t *frame_dev,
                                         uint64_t frame_pitch, void *stream) {
  if (n_rows == 0u)
    return (int)hipSuccess;
  const unsigned slices = row_bytes > 16384u ? 4u : row_bytes > 4096u ? 2u : 1u;
  hipLaunchKernelGGL(achip::scatter_rows_kernel, dim3(slices, n_rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                     staged_dev, n_rows, row_bytes, frame_dev, frame_pitch);
  return (int)hipGetLastError();
}

/* ... of a whole tick's clients in one launch: staged_dev = [n_clients x 32-byte records][per client: index table, rows] */
extern "C" int achip_launch_scatter_rows_batch(const uint8_t *staged_dev, uint32_t n_clients, uint32_t max_rows,
                                               uint32_t max_row_bytes, void *stream) {
  if (n_clients == 0u || max_rows == 0u)
    return (int)hipSuccess;
  if (max_rows > 65535u || n_clients > 65535u)
    return (int)hipErrorInvalidValue;
  const unsigned slices = max_row_bytes > 16384u ? 4u : max_row_bytes > 4096u ? 2u : 1u;
  hipLaunchKernelGGL(achip::scatter_rows_batch_kernel, dim3(slices, max_rows, n_clients), dim3(256), 0,
                     static_cast<hipStream_t>(stream), staged_dev, n_clients);
  return (int)hipGetLastError();
}

/* compaction of a rendered slab (stream_kernels.hpp: pack_frames_kernel).  Workgroups per frame: enough slices that the
 * launch has a few workgroups per CU whatever the batch size, never slices below 4 KB */
extern "C" int achip_launch_pack(const uint8_t *slab, uint64_t stride, const uint32_t *len_dev, int n, uint8_t *dst,
                                 uint64_t dst_capacity, uint64_t *off_out, uint32_t *len_out, void *stream) {
  if (n <= 0)
    return (int)hipSuccess;
  if (n > 65535 * 16)
    return (int)hipErrorInvalidValue;
  unsigned slices = (unsigned)((1024 + n - 1) / n);
  const unsigned max_slices = (unsigned)((stride + 4095u) / 4096u);
  if (slices > max_slices)
    slices = max_slices;
  if (slices < 1u)
    slices = 1u;
  if (slices > 64u)
    slices = 64u;
  hipLaunchKernelGGL(achip::pack_frames_kernel, dim3((unsigned)n, slices), dim3(256), 64, static_cast<hipStream_t>(stream),
                     slab, stride, len_dev, n, dst, dst_capacity, off_out, len_out);
  return (int)hipGetLastError();
}

/* CRC-32C of n buffers + optional wire headers (crc_kernels.hpp).  Buffers up to 128 KB are checksummed by one
 * workgroup each, which also finishes them; larger ones are cut into 64 KB spans and finished by a second
 * kernel, which needs `partial` = n * achip_crc_parts(max_len) u32 of device scratch. */
/* rounds of 4 KB per span: 64 KB spans, 16 KB ones while the call has so few of them that the GPU is not full (a lone
 * 540 KB frame: nine workgroups walking sixteen rounds each; as thirty-three of four rounds the pass is a third shorter --
 * and sixteen 256 KB frames 20.0 -> 15.5 us, sixty-four 19.2 against 24.3; from ~512 spans of 64 KB on the larger spans win:
 * 256 x 1.8 MB 141 against 179 us; scripts/gpu_crc_sweep.py, profiles/r04_wire_audit.txt) */
static int crc_span_rounds(uint32_t max_len, int n) {
  static long few = -1; /* ASCIICHAT_HIP_CRC_SMALL_SPANS (diagnostics, read once): 16 KB spans up to this many 64 KB ones */
  if (few < 0) {
    const char *e = getenv("ASCIICHAT_HIP_CRC_SMALL_SPANS");
    few = e && e[0] ? atol(e) : 512;
  }
  const uint64_t spans64 = ((uint64_t)max_len + 65535u) / 65536u;
  return (uint64_t)(n > 0 ? n : 1) * spans64 <= (uint64_t)few ? 4 : 16;
}

extern "C" int achip_crc_parts(uint32_t max_len, int n) {
  /* ASCIICHAT_HIP_CRC_FRAME_MAX (diagnostics, read once): buffers up to this many bytes take the one-workgroup kernel */
  static long forced = -1;
  if (forced < 0) {
    const char *e = getenv("ASCIICHAT_HIP_CRC_FRAME_MAX");
    forced = e && e[0] ? atol(e) : 0;
  }
  const uint64_t span_bytes = (uint64_t)crc_span_rounds(max_len, n) * 4096u;
  const int spans = (int)(((uint64_t)max_len + span_bytes - 1u) / span_bytes);
  if (forced > 0)
    return max_len <= (uint32_t)forced ? 1 : spans;
  if (max_len <= 32u * 4096u)
    return 1;
  /* above 128 KB, by measurement (scripts/gpu_crc_sweep.py, profiles/r04_wire_audit.txt; us, one call at a time): the
   * one-workgroup kernel takes 4.5 + 58 per MB of the longest buffer, whatever the count up to a workgroup per CU (it is
   * bound by its LDS look-ups: 256 KB 19, 1 MB 62, 1.8 MB 106-141); spans + the finish kernel take a fixed ~15 and 0.25
   * per MB of ALL buffers (256 x 1.8 MB: 138).  Until round 4's audit every buffer above 128 KB went to the spans, and
   * those cost 31 us however little they checksummed (a barrier-fenced tree of bit-serial multiplications in every span
   * and one thread's chain of them in the finish kernel): a lone 200x60 truecolor frame's checksum took 31 us behind a
   * 7 us render. */
  const uint64_t mb16 = ((uint64_t)max_len + 65535u) >> 16;                 /* longest buffer, in 64 KB          */
  const uint64_t waves = ((uint64_t)(n > 0 ? n : 1) + 255u) / 256u;         /* rounds of a workgroup per CU     */
  const uint64_t t_frame = 45u * 16u + 580u * mb16 * waves;                 /* 0.1 us * 16                      */
  const uint64_t t_spans = 150u * 16u + 25u * mb16 * (uint64_t)(n > 0 ? n : 1) / 10u;
  return t_frame <= t_spans ? 1 : spans;
}

/* the prebuilt tables of the checksum kernels (crc_math.hpp: crc_frame_tables_init_kernel<BLOCK>: slicing tables, the Horner
 * table of a BLOCK-thread workgroup, the power tables), one image per device and BLOCK */
template <int BLOCK> static hipError_t frame_crc_tables(const uint4 **out) {
  constexpr int MAX_DEVICES = 16;
  static std::mutex mu;
  static uint32_t *tab[MAX_DEVICES] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess)
    return e;
  if (dev < 0 || dev >= MAX_DEVICES)
    return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(mu);
  if (!tab[dev]) {
    uint32_t *t = nullptr;
    e = hipMalloc(reinterpret_cast<void **>(&t), (size_t)ACHIP_FRAME_CRC_TAB_BYTES);
    if (e != hipSuccess)
      return e;
    hipLaunchKernelGGL((achip::crc_frame_tables_init_kernel<BLOCK>), dim3(1), dim3(256), ACHIP_FRAME_CRC_TAB_BYTES, nullptr, t);
    e = hipGetLastError();
    if (e == hipSuccess)
      e = hipDeviceSynchronize();
    if (e != hipSuccess) {
      (void)hipFree(t);
      return e;
    }
    tab[dev] = t;
  }
  *out = reinterpret_cast<const uint4 *>(tab[dev]);
  return hipSuccess;
}

/* Builds this device's table images NOW (plan_create calls it): the first checksum / wire / pack call of a process would
 * otherwise allocate, launch on the null stream and synchronise the device in the middle of a tick -- or inside a stream
 * capture, where all three are illegal */
extern "C" int achip_launch_warm_crc_tables(void) {
  const uint4 *tab = nullptr;
  hipError_t e = frame_crc_tables<1024>(&tab);
  if (e == hipSuccess)
    e = frame_crc_tables<256>(&tab);
  return (int)e;
}

/* pack != NULL: the same pass also compacts the slab (crc_kernels.hpp COPY instantiations) */
/* counters != NULL (n zero words the caller owns: a plan's): the span form finishes its frames in the SAME launch */
static int launch_crc32c(const uint8_t *base, uint64_t stride, const uint32_t *len_dev, uint32_t fixed_len, uint32_t max_len,
                         int n, uint32_t *partial, uint32_t *counters, const uint32_t *dims_dev, uint32_t *crc_out, uint8_t *hdr_out,
                         uint32_t *pkt_crc_out, const achip::CrcPack *pack, hipStream_t s, const uint64_t *at = nullptr) {
  /* at != NULL (never with pack): frame i lies at base + at[i] instead of base + i * stride */
  const achip::CrcPack in_place = {nullptr, 0, const_cast<uint64_t *>(at), nullptr};
  const int parts = achip_crc_parts(max_len, n);
  if (parts == 1) { /* 1024 threads per frame; every workgroup runs only the rounds its own frame needs */
    const uint4 *tab = nullptr;
    const hipError_t te = frame_crc_tables<1024>(&tab);
    if (te != hipSuccess)
      return (int)te;
    if (pack)
      hipLaunchKernelGGL((achip::crc32c_frame_kernel<1024, true>), dim3((unsigned)n), dim3(1024), (size_t)achip::CrcLds::bytes,
                         s, base, stride, len_dev, fixed_len, n, dims_dev, crc_out, hdr_out, pkt_crc_out, *pack, tab);
    else
      hipLaunchKernelGGL((achip::crc32c_frame_kernel<1024, false>), dim3((unsigned)n), dim3(1024), (size_t)achip::CrcLds::bytes,
                         s, base, stride, len_dev, fixed_len, n, dims_dev, crc_out, hdr_out, pkt_crc_out, in_place, tab);
    return (int)hipGetLastError();
  }
  const int rounds = crc_span_rounds(max_len, n); /* 64 KB (or, for a handful of buffers, 16 KB) spans of 256-thread workgroups */
  const uint64_t v_bytes = (uint64_t)parts * (uint64_t)rounds * 4096u;
  const uint4 *tab256 = nullptr;
  const hipError_t te256 = frame_crc_tables<256>(&tab256);
  if (te256 != hipSuccess)
    return (int)te256;
  static const achip::CrcSpanPows cp16 = achip::crc_span_pows(16u * 4096u), cp4 = achip::crc_span_pows(4u * 4096u); /* once per process */
  /* the one-launch form only for SMALL calls: an arrival is a release + acquire at agent scope -- an L2 write-back and
   * invalidate -- and thousands of workgroups doing that to each other cost far more than the launch they save (128 frames
   * of 320x90: 94 -> 140 us; a lone frame: 18.6 -> 16 us) */
  if ((uint64_t)n * (uint64_t)parts > 128u)
    counters = nullptr;
  achip::CrcFinish fin;
  fin.counters = counters;
  fin.cp = rounds == 16 ? cp16 : cp4;
  fin.xinv_v = achip::crc_pow(achip::CRC_XINV8, v_bytes);
  fin.dims = dims_dev;
  fin.crc_out = crc_out;
  fin.hdr_out = hdr_out;
  fin.pkt_crc_out = pkt_crc_out;
  if (pack)
    hipLaunchKernelGGL(achip::crc32c_span_kernel<true>, dim3((unsigned)n * (unsigned)parts), dim3(256),
                       (size_t)achip::CrcLds::bytes, s, base, stride, len_dev, fixed_len, n, parts, rounds, partial, tab256, fin, *pack);
  else
    hipLaunchKernelGGL(achip::crc32c_span_kernel<false>, dim3((unsigned)n * (unsigned)parts), dim3(256),
                       (size_t)achip::CrcLds::bytes, s, base, stride, len_dev, fixed_len, n, parts, rounds, partial, tab256, fin, in_place);
  if (!counters)
    hipLaunchKernelGGL(achip::crc32c_finish_kernel, dim3((unsigned)n), dim3(64), (size_t)ACHIP_FRAME_CRC_TAB_BYTES, s, partial, parts, fin.cp,
                       fin.xinv_v, len_dev, fixed_len, n, dims_dev, crc_out, hdr_out, pkt_crc_out, tab256);
  return (int)hipGetLastError();
}

extern "C" int achip_launch_crc32c(const uint8_t *base, uint64_t stride, const uint32_t *len_dev, uint32_t fixed_len,
                                   uint32_t max_len, int n, uint32_t *partial, uint32_t *counters, const uint32_t *dims_dev,
                                   uint32_t *crc_out, uint8_t *hdr_out, uint32_t *pkt_crc_out, void *stream) {
  return launch_crc32c(base, stride, len_dev, fixed_len, max_len, n, partial, counters, dims_dev, crc_out, hdr_out, pkt_crc_out, nullptr,
                       static_cast<hipStream_t>(stream));
}

/* ... of frames that lie packed at base + at[i] (the exact-length forms of the render) */
extern "C" int achip_launch_crc32c_at(const uint8_t *base, const uint64_t *at, const uint32_t *len_dev, uint32_t max_len, int n,
                                      uint32_t *partial, uint32_t *counters, const uint32_t *dims_dev, uint32_t *crc_out, uint8_t *hdr_out,
                                      uint32_t *pkt_crc_out, void *stream) {
  return launch_crc32c(base, 0, len_dev, 0u, max_len, n, partial, counters, dims_dev, crc_out, hdr_out, pkt_crc_out, nullptr,
                       static_cast<hipStream_t>(stream), at);
}

/* checksums + headers + compaction in ONE pass over the slab: frame i also goes to dst + off[i] (pack_frames' layout) */
extern "C" int achip_launch_crc32c_pack(const uint8_t *base, uint64_t stride, const uint32_t *len_dev, uint32_t max_len, int n,
                                        uint32_t *partial, uint32_t *counters, const uint32_t *dims_dev, uint32_t *crc_out, uint8_t *hdr_out,
                                        uint32_t *pkt_crc_out, uint8_t *dst, uint64_t dst_capacity, uint64_t *off_out,
                                        uint32_t *len_out, void *stream) {
  const achip::CrcPack pack = {dst, dst_capacity, off_out, len_out};
  return launch_crc32c(base, stride, len_dev, 0u, max_len, n, partial, counters, dims_dev, crc_out, hdr_out, pkt_crc_out, &pack,
                       static_cast<hipStream_t>(stream));
}
