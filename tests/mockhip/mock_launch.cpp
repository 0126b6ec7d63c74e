/*
 * mock_launch.cpp -- hip_launch.hip's entry points on the CPU fiber emulator.  TESTS ONLY (see mock_hip.c).
 * Every launch runs the product kernel (tests/hipemu/emu_driver.cpp includes the unmodified kernel headers) at the call,
 * one launch at a time (the emulator's launch state is global: a mutex), so "streams" are trivially ordered.  Not
 * modelled: RCCL (comm.c's collectives) and stream capture.
 */
#include "../hipemu/emu_driver.cpp"

#include <mutex>

#include "hip_launch.h"

static std::mutex g_emu_mu;
enum { MOCK_OK = 0, MOCK_INVALID = 1 /* hipErrorInvalidValue */, MOCK_UNSUPPORTED = 801 /* hipErrorNotSupported */ };

extern "C" int achip_launch_render(int mode, int variant, int has_composite, const achip_frame_t *frames_dev, int n_frames,
                                   const achip_lut_t *lut_dev, uint8_t *out, uint64_t out_stride, uint32_t *out_len,
                                   unsigned long long *phase_cycles, int parts, int rows_per_part, unsigned long long *part_sync,
                                   uint32_t epoch, const achip_uniform_t *uniform, void *stream) {
  (void)phase_cycles, (void)stream;
  if (n_frames <= 0)
    return MOCK_OK;
  if ((variant == 26 || ACHIP_ROWS_VARIANT_WIDE(variant) || ACHIP_ROWS_VARIANT_PARTS(variant)) && has_composite) /* as the product's launcher: the sixteen-wave rows geometry and the segment geometries carry the fast sampler only (render_rows_inst.hip) */
    return MOCK_INVALID;
  std::lock_guard<std::mutex> lock(g_emu_mu);
  const int was = emu_set_uniform(uniform && uniform->enabled ? 1 : 0);
  emu_set_parts(parts > 1 ? parts : 1, rows_per_part, part_sync, epoch);
  const int rc = emu_render_batch(mode, variant, frames_dev, n_frames, lut_dev, out, out_stride, out_len);
  emu_set_parts(1, 0, nullptr, 1);
  (void)emu_set_uniform(was);
  return rc == 0 ? MOCK_OK : MOCK_INVALID;
}

extern "C" int achip_launch_resize(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh, void *stream) {
  (void)stream;
  std::lock_guard<std::mutex> lock(g_emu_mu);
  emu_resize_nn(src, sw, sh, dst, dw, dh, achip_nn_ratio(sw, dw), achip_nn_ratio(sh, dh));
  return MOCK_OK;
}

extern "C" int achip_launch_scatter_rows(const uint8_t *staged, uint32_t n_rows, uint32_t row_bytes, uint8_t *frame,
                                         uint64_t frame_pitch, void *stream) {
  (void)stream;
  if (n_rows == 0u)
    return MOCK_OK;
  std::lock_guard<std::mutex> lock(g_emu_mu);
  hipemu::launch(dim3(row_bytes > 4096u ? 2u : 1u, n_rows), dim3(256), 0,
                 [&] { achip::scatter_rows_kernel(staged, n_rows, row_bytes, frame, frame_pitch); });
  return MOCK_OK;
}

extern "C" int achip_launch_scatter_rows_batch(const uint8_t *staged, uint32_t n_clients, uint32_t max_rows,
                                               uint32_t max_row_bytes, void *stream) {
  (void)stream;
  if (n_clients == 0u || max_rows == 0u)
    return MOCK_OK;
  std::lock_guard<std::mutex> lock(g_emu_mu);
  emu_scatter_rows_batch(staged, n_clients, max_rows, max_row_bytes > 4096u ? 2 : 1);
  return MOCK_OK;
}

extern "C" int achip_launch_pack(const uint8_t *slab, uint64_t stride, const uint32_t *len, int n, uint8_t *dst, uint64_t cap,
                                 uint64_t *off_out, uint32_t *len_out, void *stream) {
  (void)stream;
  if (n <= 0)
    return MOCK_OK;
  std::lock_guard<std::mutex> lock(g_emu_mu);
  emu_pack(slab, stride, len, n, dst, cap, off_out, len_out, 2);
  return MOCK_OK;
}

/* the wire stage: stand-alone CRC + headers, headers from known CRCs, and the render with the CRC riding its drain */
extern "C" int achip_crc_parts(uint32_t max_len, int n) { /* (the emulated launcher sizes its own span registers) */
  (void)n;
  return max_len <= 32u * 4096u ? 1 : (int)(((uint64_t)max_len + 65535u) / 65536u);
}
extern "C" int achip_launch_crc32c(const uint8_t *base, uint64_t stride, const uint32_t *len, uint32_t fixed_len, uint32_t max_len,
                                   int n, uint32_t *partial, uint32_t *counters, const uint32_t *dims, uint32_t *crc_out, uint8_t *hdr_out,
                                   uint32_t *pkt_out, void *stream) {
  (void)partial, (void)counters, (void)stream;
  if (n <= 0)
    return MOCK_OK;
  std::lock_guard<std::mutex> lock(g_emu_mu);
  emu_crc32c(base, stride, len, fixed_len, max_len, n, 0, 0, dims, crc_out, hdr_out, pkt_out);
  return MOCK_OK;
}
extern "C" int achip_launch_crc32c_pack(const uint8_t *base, uint64_t stride, const uint32_t *len, uint32_t max_len, int n,
                                        uint32_t *partial, uint32_t *counters, const uint32_t *dims, uint32_t *crc_out, uint8_t *hdr_out,
                                        uint32_t *pkt_out, uint8_t *dst, uint64_t cap, uint64_t *off_out, uint32_t *len_out,
                                        void *stream) {
  (void)partial, (void)counters, (void)stream;
  if (n <= 0)
    return MOCK_OK;
  std::lock_guard<std::mutex> lock(g_emu_mu);
  emu_crc32c_pack(base, stride, len, max_len, n, 0, 0, dims, crc_out, hdr_out, pkt_out, dst, cap, off_out, len_out);
  return MOCK_OK;
}
extern "C" int achip_launch_packets_from_crc(const uint32_t *len, const uint32_t *crc, const uint32_t *dims, int n, uint8_t *hdr_out,
                                             uint32_t *pkt_out, void *stream) {
  (void)stream;
  if (n <= 0)
    return MOCK_OK;
  std::lock_guard<std::mutex> lock(g_emu_mu);
  hipemu::launch(dim3((unsigned)(n + 255) / 256u), dim3(256), 0,
                 [&] { achip::crc_packets_kernel(len, crc, dims, n, hdr_out, pkt_out); });
  return MOCK_OK;
}
extern "C" int achip_variant_has_crc(int variant) { return variant == 16 || variant == 17 || (ACHIP_IS_ROWS_VARIANT(variant) && !ACHIP_ROWS_VARIANT_WIDE(variant) && !ACHIP_ROWS_VARIANT_PARTS(variant)); }
extern "C" int achip_variant_crc_pays(int variant) { return variant == 16 || variant == 17; }
extern "C" int achip_launch_render_crc(int mode, int variant, int has_composite, const achip_frame_t *frames, int n,
                                       const achip_lut_t *lut, uint8_t *out, uint64_t stride, uint32_t *out_len,
                                       const achip_wire_t *wire, const achip_uniform_t *uniform, unsigned long long *prof, void *stream) {
  (void)has_composite, (void)uniform, (void)prof, (void)stream;
  if (n <= 0)
    return MOCK_OK;
  if (!wire || !wire->crc)
    return MOCK_INVALID;
  std::lock_guard<std::mutex> lock(g_emu_mu);
  const int rc = ACHIP_IS_ROWS_VARIANT(variant)
                     ? emu_render_rows_crc(mode, variant, frames, n, lut, out, stride, out_len, wire->crc, wire->dims, wire->hdr, wire->pkt_crc)
                     : emu_render_stream_crc(mode, variant, frames, n, lut, out, stride, out_len, wire->crc, wire->dims, wire->hdr, wire->pkt_crc);
  return rc == 0 ? MOCK_OK : MOCK_INVALID;
}

extern "C" int emu_render_stream_pack(int mode, int variant, const achip_frame_t *frames, int n, const achip_lut_t *lut,
                                      uint64_t stride, uint32_t *len, uint32_t *crc, const uint32_t *dims, uint8_t *hdr,
                                      uint32_t *pkt, uint8_t *dst, uint64_t capacity, uint64_t *off_out, uint32_t *len_out,
                                      unsigned long long *cursor);
extern "C" int achip_pack_frame_cap(void) { return 48 * 1024; }
extern "C" int achip_launch_render_pack(int mode, int variant, const achip_frame_t *frames, int n, const achip_lut_t *lut, uint64_t bound,
                                        uint32_t *out_len, const achip_wire_t *wire, const achip_uniform_t *uniform,
                                        const achip_packdev_t *pack, void *stream) {
  (void)uniform, (void)stream;
  if (n <= 0)
    return MOCK_OK;
  if (!pack || !pack->dst || !pack->cursor || bound > 48 * 1024 || (wire && !wire->crc))
    return MOCK_INVALID;
  std::lock_guard<std::mutex> lock(g_emu_mu);
  const int rc = emu_render_stream_pack(mode, variant == 16 ? 16 : 17, frames, n, lut, bound, out_len, wire ? wire->crc : nullptr,
                                        wire ? wire->dims : nullptr, wire ? wire->hdr : nullptr, wire ? wire->pkt_crc : nullptr,
                                        pack->dst, pack->capacity, pack->off_out, pack->len_out, pack->cursor);
  return rc == 0 ? MOCK_OK : MOCK_INVALID;
}

extern "C" int emu_render_stream_lenfirst(int variant, const achip_frame_t *frames, int n, const achip_lut_t *lut, uint64_t stride,
                                          uint32_t *len, uint8_t *dst, uint64_t capacity, uint64_t *off_out, uint32_t *len_out,
                                          unsigned long long *cursor);
extern "C" int achip_launch_render_length_first(int variant, const achip_frame_t *frames, int n, const achip_lut_t *lut, uint64_t bound,
                                                uint32_t *out_len, const achip_uniform_t *uniform, const achip_packdev_t *pack,
                                                void *stream) {
  (void)uniform, (void)stream;
  if (n <= 0)
    return MOCK_OK;
  if (!pack || !pack->dst || !pack->cursor)
    return MOCK_INVALID;
  std::lock_guard<std::mutex> lock(g_emu_mu);
  const int rc = emu_render_stream_lenfirst(variant == 16 ? 16 : 17, frames, n, lut, bound, out_len, pack->dst, pack->capacity, pack->off_out,
                                            pack->len_out, pack->cursor);
  return rc == 0 ? MOCK_OK : MOCK_INVALID;
}
/* checksums of frames that lie packed at base + at[i]: frame by frame through the emulated stand-alone kernels */
extern "C" int achip_launch_crc32c_at(const uint8_t *base, const uint64_t *at, const uint32_t *len, uint32_t max_len, int n, uint32_t *partial,
                                      uint32_t *counters, const uint32_t *dims, uint32_t *crc_out, uint8_t *hdr_out, uint32_t *pkt_out,
                                      void *stream) {
  (void)partial, (void)counters, (void)stream;
  std::lock_guard<std::mutex> lock(g_emu_mu);
  for (int i = 0; i < n; i++)
    emu_crc32c(base + (len[i] < 0xFFFFFFF0u ? at[i] : 0), 0, len + i, 0, max_len, 1, 0, 0, dims ? dims + 2 * i : nullptr, crc_out + i,
               hdr_out ? hdr_out + 24 * i : nullptr, pkt_out ? pkt_out + i : nullptr);
  return MOCK_OK;
}

/* image-space passes and composites, with the launchers' own choice between the vector and the per-pixel kernels */
extern "C" int achip_launch_tint(uint8_t *px, int w, int h, int stride, uint32_t ops, void *stream) {
  (void)stream;
  std::lock_guard<std::mutex> lock(g_emu_mu);
  if (stride == 3 * w && (reinterpret_cast<uintptr_t>(px) & 15u) == 0)
    emu_tint(px, w, h, stride, ops, 1);
  else
    emu_tint(px, w, h, stride, ops, 0);
  return MOCK_OK;
}
extern "C" int achip_launch_flip(const uint8_t *src, uint8_t *dst, int w, int h, int src_stride, int dst_stride, uint32_t ops,
                                 void *stream) {
  (void)stream;
  std::lock_guard<std::mutex> lock(g_emu_mu);
  const bool vec = w % 16 == 0 && src_stride == 3 * w && dst_stride == 3 * w &&
                   ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
  if (vec)
    emu_flip(src, dst, w, h, ops, 1);
  else
    hipemu::launch(dim3(3), dim3(256), 0, [&] { achip::flip_pixels_kernel(src, dst, w, h, src_stride, dst_stride, ops); });
  return MOCK_OK;
}
extern "C" int achip_launch_composite(const achip_composite_t *comp, int canvas_w, int canvas_h, uint8_t *dst, void *stream) {
  (void)canvas_w, (void)canvas_h, (void)stream;
  std::lock_guard<std::mutex> lock(g_emu_mu);
  emu_composite(comp, dst);
  return MOCK_OK;
}
extern "C" int achip_launch_resize_batch(const achip_resize_batch_t *b, void *stream) {
  (void)stream;
  if (!b || b->n <= 0)
    return MOCK_OK;
  std::lock_guard<std::mutex> lock(g_emu_mu);
  emu_resize_batch(b);
  return MOCK_OK;
}
extern "C" int achip_launch_comp_poke(achip_composite_t *comp, const achip_comp_poke_t *poke, void *stream) {
  (void)stream;
  for (int k = 0; k < 9; k++) /* comp_poke_kernel (hip_launch.hip): nine pointers */
    comp->s[k].src = poke->src[k];
  return MOCK_OK;
}

/* geometry facts of hip_launch.hip, from the same table (render_variants.h) */
extern "C" int achip_launch_warm_crc_tables(void) { return 0; } /* (the mock's kernels build their tables per launch) */
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
extern "C" int achip_variant_lds_bytes(int, int) { return 0; }
