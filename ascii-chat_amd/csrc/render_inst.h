/* render_inst.h -- entry points of the per-geometry translation units (render_inst.hip, -DACHIP_INST=id). */
#ifndef ACHIP_RENDER_INST_H
#define ACHIP_RENDER_INST_H

#include <stdint.h>

#include "achip_types.h"
#include "render_variants.h"

#ifdef __cplusplus
extern "C" {
#endif

#define X(id, B, C, R)                                                                                                 \
  int achip_render_inst_launch_##id(int mode, int comp, const achip_frame_t *frames, int n, const achip_lut_t *lut,    \
                                    uint8_t *out, uint64_t stride, uint32_t *len, unsigned long long *prof, int parts, \
                                    int rows_per_part, unsigned long long *part_sync, uint32_t epoch,                  \
                                    const achip_uniform_t *uniform, void *stream);                                     \
  int achip_render_inst_lds_##id(int mode);
ACHIP_VARIANTS(X)
#undef X

/* the stream-kernel geometries (render_stream_inst.hip, -DACHIP_SINST=id) */
#define X(id, W, C)                                                                                                    \
  int achip_render_sinst_launch_##id(int mode, int comp, const achip_frame_t *frames, int n, const achip_lut_t *lut,   \
                                     uint8_t *out, uint64_t stride, uint32_t *len, const achip_uniform_t *uniform,     \
                                     unsigned long long *prof, const achip_wire_t *wire, void *stream);                                                                    \
  int achip_render_sinst_lds_##id(int mode);
ACHIP_STREAM_VARIANTS(X)
#undef X

/* the PACK instantiations of stream geometries 16 and 17 (render_stream_inst.hip with -DACHIP_SINST=16 / 17) */
int achip_render_sinst_pack_launch_16(int mode, const achip_frame_t *frames, int n, const achip_lut_t *lut, uint64_t stride,
                                      uint32_t *len, const achip_uniform_t *uniform, const achip_wire_t *wire,
                                      const achip_packdev_t *pack, void *stream);
int achip_render_sinst_pack_launch_17(int mode, const achip_frame_t *frames, int n, const achip_lut_t *lut, uint64_t stride,
                                      uint32_t *len, const achip_uniform_t *uniform, const achip_wire_t *wire,
                                      const achip_packdev_t *pack, void *stream);

/* the PARTS instantiations of stream geometry 18 (a frame's blocks shared out over ps->parts workgroups, 2..64) */
int achip_render_sinst_parts_launch_18(int mode, int comp, const achip_frame_t *frames, int n, const achip_lut_t *lut,
                                       uint8_t *out, uint64_t stride, uint32_t *len, const achip_uniform_t *uniform,
                                       unsigned long long *prof, const achip_partsdev_t *ps, void *stream);

/* the rows-kernel geometries (render_rows_inst.hip, -DACHIP_RINST=id): run-structured modes, whole frames */
#define X(id, W, C)                                                                                                    \
  int achip_render_rinst_launch_##id(int mode, int comp, const achip_frame_t *frames, int n, const achip_lut_t *lut,   \
                                     uint8_t *out, uint64_t stride, uint32_t *len, const achip_uniform_t *uniform,     \
                                     const achip_wire_t *wire, void *stream);                                          \
  int achip_render_rinst_lds_##id(int mode);
ACHIP_ROWS_VARIANTS(X)
#undef X

#ifdef __cplusplus
}
#endif
#endif
