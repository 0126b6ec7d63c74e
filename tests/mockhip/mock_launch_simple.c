/*
 * mock_launch_simple.c -- hip_launch.hip's entry points WITHOUT kernels.  TESTS ONLY (see mock_hip.c).
 * For sanitizer runs of the host C (ThreadSanitizer cannot follow the emulator's fibers): a "render" writes a line that is
 * a pure function of what the real sampler would read -- "F<fnv of the samples> <out_w>x<out_h> m<mode>\n" -- so a caller
 * that is handed another caller's frame, a torn descriptor or a half-staged image gets a different string.
 */
#include <stdio.h>
#include <string.h>

#include "achip_host.h"
#include "hip_launch.h"
#include "render_variants.h"

enum { MOCK_OK = 0, MOCK_INVALID = 1, MOCK_UNSUPPORTED = 801 };
#define ACHIP_STREAM_MAXBLK 2048 /* render_stream.hpp (a C++ header) */

static uint32_t sample_hash(const achip_frame_t *f) {
  uint32_t h = 2166136261u;
  const size_t stride = f->src_stride ? (size_t)f->src_stride : (size_t)f->src_w * 3u;
  for (uint32_t y = 0; y < (uint32_t)f->out_h; y++) {
    uint32_t sy = (uint32_t)(((uint64_t)y * f->y_ratio) >> 16);
    if (sy > (uint32_t)f->src_h - 1u)
      sy = (uint32_t)f->src_h - 1u;
    if (f->ops & ACHIP_OP_FLIP_Y)
      sy = (uint32_t)f->src_h - 1u - sy;
    for (uint32_t x = 0; x < (uint32_t)f->out_w; x++) {
      uint32_t sx = (uint32_t)(((uint64_t)x * f->x_ratio) >> 16);
      if (sx > (uint32_t)f->src_w - 1u)
        sx = (uint32_t)f->src_w - 1u;
      if (f->ops & ACHIP_OP_FLIP_X)
        sx = (uint32_t)f->src_w - 1u - sx;
      const uint8_t *p = f->src + (size_t)sy * stride + (size_t)sx * 3u;
      for (int c = 0; c < 3; c++)
        h = (h ^ p[c]) * 16777619u;
    }
  }
  return h;
}

int achip_launch_render(int mode, int variant, int has_composite, const achip_frame_t *frames, int n, const achip_lut_t *lut,
                        uint8_t *out, uint64_t stride, uint32_t *out_len, unsigned long long *phase_cycles, int parts,
                        int rows_per_part, unsigned long long *part_sync, uint32_t epoch, const achip_uniform_t *uniform,
                        void *stream) {
  (void)variant, (void)has_composite, (void)lut, (void)phase_cycles, (void)parts, (void)rows_per_part, (void)part_sync, (void)epoch,
      (void)stream;
  for (int i = 0; i < n; i++) {
    achip_frame_t f = frames[i];
    if (uniform && uniform->enabled) { /* the batch's common descriptor travels by value, as in the kernels */
      f = uniform->f;
      f.src = uniform->f.src + (int64_t)i * uniform->src_pitch;
    }
    if (!f.src || f.out_w <= 0 || f.out_h <= 0) {
      out_len[i] = ACHIP_LEN_BADDESC;
      continue;
    }
    char line[96];
    const int len = snprintf(line, sizeof line, "F%08x %dx%d+%d+%d m%d\n", sample_hash(&f), f.out_w, f.out_h, f.pad_left, f.pad_top, mode);
    if ((uint64_t)len + 1u > stride) {
      out_len[i] = ACHIP_LEN_OVERFLOW;
      continue;
    }
    memcpy(out + (size_t)i * stride, line, (size_t)len + 1u);
    out_len[i] = (uint32_t)len;
  }
  return MOCK_OK;
}

int achip_launch_render_crc(int mode, int variant, int has_composite, const achip_frame_t *frames_dev, int n_frames,
                            const achip_lut_t *lut_dev, uint8_t *out, uint64_t out_stride, uint32_t *out_len,
                            const achip_wire_t *wire, const achip_uniform_t *uniform, unsigned long long *prof, void *stream) {
  (void)mode, (void)variant, (void)has_composite, (void)frames_dev, (void)n_frames, (void)lut_dev, (void)out, (void)out_stride,
      (void)out_len, (void)wire, (void)uniform, (void)prof, (void)stream;
  return MOCK_UNSUPPORTED;
}
int achip_launch_render_pack(int mode, int variant, const achip_frame_t *frames_dev, int n_frames, const achip_lut_t *lut_dev, uint64_t bound,
                             uint32_t *out_len, const achip_wire_t *wire, const achip_uniform_t *uniform,
                             const achip_packdev_t *pack, void *stream) {
  (void)mode, (void)variant, (void)frames_dev, (void)n_frames, (void)lut_dev, (void)bound, (void)out_len, (void)wire, (void)uniform, (void)pack,
      (void)stream;
  return MOCK_UNSUPPORTED;
}
int achip_launch_render_length_first(int variant, const achip_frame_t *frames_dev, int n_frames, const achip_lut_t *lut_dev, uint64_t bound,
                                     uint32_t *out_len, const achip_uniform_t *uniform, const achip_packdev_t *pack, void *stream) {
  (void)variant, (void)frames_dev, (void)n_frames, (void)lut_dev, (void)bound, (void)out_len, (void)uniform, (void)pack, (void)stream;
  return MOCK_UNSUPPORTED;
}
int achip_launch_crc32c_at(const uint8_t *base, const uint64_t *at, const uint32_t *len, uint32_t max, int n, uint32_t *partial,
                           uint32_t *counters, const uint32_t *dims, uint32_t *crc, uint8_t *hdr, uint32_t *pkt, void *s) {
  (void)base, (void)at, (void)len, (void)max, (void)n, (void)partial, (void)counters, (void)dims, (void)crc, (void)hdr, (void)pkt, (void)s;
  return MOCK_UNSUPPORTED;
}
int achip_pack_frame_cap(void) { return 0; } /* no such kernels here: plans keep the two-pass forms */
int achip_launch_packets_from_crc(const uint32_t *a, const uint32_t *b, const uint32_t *c, int n, uint8_t *h, uint32_t *p, void *s) {
  (void)a, (void)b, (void)c, (void)n, (void)h, (void)p, (void)s;
  return MOCK_UNSUPPORTED;
}
int achip_launch_resize(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh, void *stream) {
  (void)src, (void)sw, (void)sh, (void)dst, (void)dw, (void)dh, (void)stream;
  return MOCK_UNSUPPORTED;
}
int achip_launch_resize_batch(const achip_resize_batch_t *b, void *s) {
  (void)b, (void)s;
  return MOCK_UNSUPPORTED;
}
int achip_launch_comp_poke(achip_composite_t *c, const achip_comp_poke_t *p, void *s) {
  (void)c, (void)p, (void)s;
  return MOCK_UNSUPPORTED;
}
int achip_launch_composite(const achip_composite_t *c, int w, int h, uint8_t *d, void *s) {
  (void)c, (void)w, (void)h, (void)d, (void)s;
  return MOCK_UNSUPPORTED;
}
int achip_launch_tint(uint8_t *px, int w, int h, int stride, uint32_t ops, void *s) {
  (void)px, (void)w, (void)h, (void)stride, (void)ops, (void)s;
  return MOCK_UNSUPPORTED;
}
int achip_launch_flip(const uint8_t *src, uint8_t *dst, int w, int h, int ss, int ds, uint32_t ops, void *s) {
  (void)src, (void)dst, (void)w, (void)h, (void)ss, (void)ds, (void)ops, (void)s;
  return MOCK_UNSUPPORTED;
}
int achip_launch_pack(const uint8_t *slab, uint64_t stride, const uint32_t *len, int n, uint8_t *dst, uint64_t cap, uint64_t *off,
                      uint32_t *len_out, void *s) {
  (void)slab, (void)stride, (void)len, (void)n, (void)dst, (void)cap, (void)off, (void)len_out, (void)s;
  return MOCK_UNSUPPORTED;
}
/* the row / pixel scatter of ingest as plain loops over the same staged layout (stream_kernels.hpp) */
int achip_launch_scatter_rows(const uint8_t *st, uint32_t n_rows, uint32_t row_bytes, uint8_t *frame, uint64_t pitch, void *s) {
  (void)s;
  const uint32_t table = (n_rows * 4u + 15u) & ~15u;
  for (uint32_t r = 0; r < n_rows; r++)
    memcpy(frame + (uint64_t)((const uint32_t *)st)[r] * pitch, st + table + (uint64_t)r * row_bytes, row_bytes);
  return MOCK_OK;
}
int achip_launch_scatter_rows_batch(const uint8_t *st, uint32_t n, uint32_t mr, uint32_t mb, void *s) {
  (void)mr, (void)mb, (void)s;
  for (uint32_t c = 0; c < n; c++) {
    struct {
      uint64_t frame;
      uint32_t off, n_rows, row_bytes, n_cols, pad[2];
    } cl;
    memcpy(&cl, st + 32u * c, 32);
    const uint8_t *blk = st + cl.off;
    const uint32_t tr = (cl.n_rows * 4u + 15u) & ~15u, tc = (cl.n_cols * 4u + 15u) & ~15u;
    const uint32_t *rows = (const uint32_t *)blk, *cols = (const uint32_t *)(blk + tr);
    for (uint32_t r = 0; r < cl.n_rows; r++) {
      uint8_t *dst = (uint8_t *)(uintptr_t)cl.frame + (uint64_t)rows[r] * cl.row_bytes;
      if (!cl.n_cols) {
        memcpy(dst, blk + tr + (uint64_t)r * cl.row_bytes, cl.row_bytes);
        continue;
      }
      const uint8_t *src = blk + tr + tc + (uint64_t)r * cl.n_cols * 3u;
      for (uint32_t i = 0; i < cl.n_cols; i++)
        memcpy(dst + (uint64_t)cols[i] * 3u, src + 3u * i, 3);
    }
  }
  return MOCK_OK;
}
int achip_launch_crc32c(const uint8_t *base, uint64_t stride, const uint32_t *len, uint32_t fixed, uint32_t max, int n, uint32_t *partial,
                        uint32_t *counters, const uint32_t *dims, uint32_t *crc, uint8_t *hdr, uint32_t *pkt, void *s) {
  (void)base, (void)stride, (void)len, (void)fixed, (void)max, (void)n, (void)partial, (void)counters, (void)dims, (void)crc, (void)hdr, (void)pkt, (void)s;
  return MOCK_UNSUPPORTED;
}
int achip_launch_crc32c_pack(const uint8_t *base, uint64_t stride, const uint32_t *len, uint32_t max, int n, uint32_t *partial,
                             uint32_t *counters, const uint32_t *dims, uint32_t *crc, uint8_t *hdr, uint32_t *pkt, uint8_t *dst, uint64_t cap,
                             uint64_t *off, uint32_t *len_out, void *s) {
  (void)base, (void)stride, (void)len, (void)max, (void)n, (void)partial, (void)counters, (void)dims, (void)crc, (void)hdr, (void)pkt, (void)dst,
      (void)cap, (void)off, (void)len_out, (void)s;
  return MOCK_UNSUPPORTED;
}
int achip_crc_parts(uint32_t max_len, int n) {
  (void)n;
  (void)max_len;
  return 1;
}
int achip_variant_has_crc(int v) {
  (void)v;
  return 0;
}
int achip_variant_crc_pays(int v) {
  (void)v;
  return 0;
}
int achip_launch_warm_crc_tables(void) { return 0; } /* (the mock's kernels build their tables per launch) */
int achip_variant_block(int variant) {
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
int achip_variant_cap(int variant) {
  switch (variant) {
#define X(id, W, C)                                                                                                    \
  case id:                                                                                                             \
    return ACHIP_STREAM_MAXBLK * 64 * C;
    ACHIP_STREAM_VARIANTS(X)
#undef X
#define X(id, W, C)                                                                                                    \
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
int achip_variant_lds_bytes(int mode, int variant) {
  (void)mode, (void)variant;
  return 0;
}
