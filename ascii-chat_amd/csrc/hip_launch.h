/* hip_launch.h -- C interface between the host shim (C) and the hipcc-compiled kernels. */
#ifndef ACHIP_HIP_LAUNCH_H
#define ACHIP_HIP_LAUNCH_H

#include <stdint.h>

#include "achip_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* all return a hipError_t as int (0 = hipSuccess); `stream` is a hipStream_t */
/* has_composite: some frame of the batch samples a virtual composite (achip_frame_t.comp != NULL) */
int achip_launch_render(int mode, int variant, int has_composite, const achip_frame_t *frames_dev, int n_frames,
                        const achip_lut_t *lut_dev, uint8_t *out, uint64_t out_stride, uint32_t *out_len,
                        unsigned long long *phase_cycles /* NULL, or 8 u64 per frame (diagnostics) */,
                        int parts /* workgroups per frame (1 = whole frame per workgroup) */, int rows_per_part,
                        unsigned long long *part_sync /* n_frames*parts u64, zeroed once; NULL when parts == 1 */,
                        uint32_t epoch /* differs from launch to launch on the same part_sync */,
                        const achip_uniform_t *uniform /* NULL, or the batch's common descriptor (achip_frames_uniform) */,
                        void *stream);
/* the same for a whole-frame launch of a per-cell mode in a stream geometry that carries the fused frame CRC
 * (achip_variant_has_crc): wire->crc[i] = asciichat_crc32(frame i), 0 for a frame that did not fit its slot; with
 * wire->hdr / wire->pkt_crc also the 24-byte packet headers and the CRCs of header || frame */
int achip_launch_render_crc(int mode, int variant, int has_composite, const achip_frame_t *frames_dev, int n_frames,
                            const achip_lut_t *lut_dev, uint8_t *out, uint64_t out_stride, uint32_t *out_len,
                            const achip_wire_t *wire, const achip_uniform_t *uniform, unsigned long long *prof,
                            void *stream);
/* whole-frame launch of a per-cell FOREGROUND mode (truecolor with an all-ASCII palette, 256, 16; single sources) that
 * writes the frames at their exact lengths itself (the PACK instantiations of stream geometries 16 / 17): no slab -- `bound` only
 * limits a frame's length and must not exceed achip_pack_frame_cap().  wire = NULL: no checksums. */
int achip_launch_render_pack(int mode, int variant /* 16: 1024-thread workgroups, else 512 */, const achip_frame_t *frames_dev,
                             int n_frames, const achip_lut_t *lut_dev, uint64_t bound, uint32_t *out_len, const achip_wire_t *wire, const achip_uniform_t *uniform,
                             const achip_packdev_t *pack, void *stream);
int achip_launch_crc32c_at(const uint8_t *base, const uint64_t *at, const uint32_t *len_dev, uint32_t max_len, int n, uint32_t *partial,
                           uint32_t *counters, const uint32_t *dims_dev, uint32_t *crc_out, uint8_t *hdr_out, uint32_t *pkt_crc_out,
                           void *stream);
/* exact-length truecolor frames of any size in one launch (render_stream.hpp LF; stream geometries 16 / 17) */
int achip_launch_render_length_first(int variant, const achip_frame_t *frames_dev, int n_frames, const achip_lut_t *lut_dev, uint64_t bound,
                                     uint32_t *out_len, const achip_uniform_t *uniform, const achip_packdev_t *pack, void *stream);
int achip_pack_frame_cap(void); /* bytes of one frame those instantiations can stage; 0 = not available in this build */
int achip_variant_has_crc(int variant);
int achip_variant_crc_pays(int variant); /* the fused form is the faster one: what plans pick by themselves */
/* 24-byte ascii_frame_packet_t headers and header || frame CRCs from lengths + frame CRCs that are already known */
int achip_launch_packets_from_crc(const uint32_t *len_dev, const uint32_t *crc_dev, const uint32_t *dims_dev, int n,
                                  uint8_t *hdr_out, uint32_t *pkt_crc_out, void *stream);
int achip_launch_resize(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh, void *stream);
/* up to ACHIP_RESIZE_BATCH_MAX resizes in ONE launch (src / dst / sizes filled in; the ratios are computed here) */
int achip_launch_resize_batch(const achip_resize_batch_t *batch, void *stream);
/* comp_dev->s[k].src = poke->src[k] for k < 9 (stream-ordered, one wave) */
int achip_launch_comp_poke(achip_composite_t *comp_dev, const achip_comp_poke_t *poke, void *stream);
int achip_launch_composite(const achip_composite_t *comp_dev, int canvas_w, int canvas_h, uint8_t *dst, void *stream);

/* display-path streaming passes (stream_kernels.hpp); ops as in achip_frame_t.ops */
int achip_launch_tint(uint8_t *px, int w, int h, int stride, uint32_t ops, void *stream);
int achip_launch_flip(const uint8_t *src, uint8_t *dst, int w, int h, int src_stride, int dst_stride, uint32_t ops,
                      void *stream);

/* compacted copy of a slab: frame i -> dst + off[i], off[i] = sum_{j<i} round16(len[j]); off_out (n + 1 entries, [n] =
 * total) and len_out (n) may be NULL; dst / off_out / len_out may be device memory or mapped pinned host memory */
int achip_launch_pack(const uint8_t *slab, uint64_t stride, const uint32_t *len_dev, int n, uint8_t *dst,
                      uint64_t dst_capacity, uint64_t *off_out, uint32_t *len_out, void *stream);

/* staged_dev = [n_rows x u32 row index, padded to 16][n_rows x row_bytes]: row i goes to frame_dev + index[i] * frame_pitch */
int achip_launch_scatter_rows(const uint8_t *staged_dev, uint32_t n_rows, uint32_t row_bytes, uint8_t *frame_dev,
                              uint64_t frame_pitch, void *stream);

/* staged_dev = [n_clients x {u64 frame, u32 off, u32 n_rows, u32 row_bytes, 12 bytes pad}][per client at `off`: index
 * table padded to 16, rows]: every client's rows to their places in ITS frame buffer, one launch */
int achip_launch_scatter_rows_batch(const uint8_t *staged_dev, uint32_t n_clients, uint32_t max_rows, uint32_t max_row_bytes,
                                    void *stream);

/* wire stage (crc_kernels.hpp): CRC-32C of n buffers at base + i*stride (len_dev[i] bytes, or fixed_len when
 * len_dev == NULL; every length <= max_len) and, when hdr_out != NULL, the 24-byte ascii_frame_packet_t headers
 * (dims_dev = n x {width, height}) and the CRC of header || frame.  partial: n * achip_crc_parts(max_len, n) u32 of
 * device scratch, unused (may be NULL) when achip_crc_parts(max_len, n) == 1. */
int achip_crc_parts(uint32_t max_len, int n);
int achip_launch_crc32c(const uint8_t *base, uint64_t stride, const uint32_t *len_dev, uint32_t fixed_len,
                        uint32_t max_len, int n, uint32_t *partial, uint32_t *counters, const uint32_t *dims_dev, uint32_t *crc_out,
                        uint8_t *hdr_out, uint32_t *pkt_crc_out, void *stream);
/* counters: NULL, or n device words that are zero between launches (a plan's own): the span form then finishes its frames in
 * the same launch -- the last span of a frame to arrive combines the registers -- instead of a second kernel */

/* the same pass also compacting the slab: frame i to dst + off[i], off[i] = sum of round16(len[j]), j < i (achip_launch_pack's layout) */
int achip_launch_crc32c_pack(const uint8_t *base, uint64_t stride, const uint32_t *len_dev, uint32_t max_len, int n,
                             uint32_t *partial, uint32_t *counters, const uint32_t *dims_dev, uint32_t *crc_out, uint8_t *hdr_out,
                             uint32_t *pkt_crc_out, uint8_t *dst, uint64_t dst_capacity, uint64_t *off_out, uint32_t *len_out,
                             void *stream);

int achip_launch_warm_crc_tables(void); /* the checksum kernels' table images of the current device, built eagerly */
int achip_variant_block(int variant); /* threads per workgroup, -1 for an unknown id */
int achip_variant_cap(int variant);   /* cells per chunk                               */
int achip_variant_lds_bytes(int mode, int variant);

#ifdef __cplusplus
}
#endif
#endif
