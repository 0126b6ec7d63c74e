/* render_inst.h -- entry points of the per-geometry translation units (render_inst.hip, -DACHIP_INST=id).  Their names
 * start with achipk_, not achip_: they are the library's own plumbing and stay local (exports.map). */
#ifndef ACHIP_RENDER_INST_H
#define ACHIP_RENDER_INST_H

#include <stdint.h>

#include "achip_types.h"
#include "render_variants.h"

#ifdef __cplusplus
extern "C" {
#endif

/* four translation units per geometry: _p0 = modes 0..2 (mono, truecolor / 256-colour foreground), _p1 = 3, 4 (16-colour
 * foreground, truecolor background), _p2 = 5..7 (the coloured half-block modes), _p3 = 8, 9 (mono half blocks, dither) */
#define ACHIP_INST_PART_OF(m) ((m) <= 2 ? 0 : (m) <= 4 ? 1 : (m) <= 7 ? 2 : 3)
#define ACHIP_INST_ARGS                                                                                                \
  int mode, int comp, const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out, uint64_t stride,       \
      uint32_t *len, unsigned long long *prof, int parts, int rows_per_part, unsigned long long *part_sync,            \
      uint32_t epoch, const achip_uniform_t *uniform, void *stream
#define ACHIP_INST_PASS mode, comp, frames, n, lut, out, stride, len, prof, parts, rows_per_part, part_sync, epoch, uniform, stream
#define X(id, B, C, R)                                                                                                 \
  int achipk_render_inst_launch_##id##_p0(ACHIP_INST_ARGS);                                                             \
  int achipk_render_inst_launch_##id##_p1(ACHIP_INST_ARGS);                                                             \
  int achipk_render_inst_launch_##id##_p2(ACHIP_INST_ARGS);                                                             \
  int achipk_render_inst_launch_##id##_p3(ACHIP_INST_ARGS);                                                             \
  int achipk_render_inst_lds_##id##_p0(int mode);                                                                       \
  int achipk_render_inst_lds_##id##_p1(int mode);                                                                       \
  int achipk_render_inst_lds_##id##_p2(int mode);                                                                       \
  int achipk_render_inst_lds_##id##_p3(int mode);                                                                       \
  static inline int achipk_render_inst_launch_##id(ACHIP_INST_ARGS) {                                                   \
    switch (ACHIP_INST_PART_OF(mode)) {                                                                                \
    case 0: return achipk_render_inst_launch_##id##_p0(ACHIP_INST_PASS);                                                \
    case 1: return achipk_render_inst_launch_##id##_p1(ACHIP_INST_PASS);                                                \
    case 2: return achipk_render_inst_launch_##id##_p2(ACHIP_INST_PASS);                                                \
    default: return achipk_render_inst_launch_##id##_p3(ACHIP_INST_PASS);                                               \
    }                                                                                                                  \
  }                                                                                                                    \
  static inline int achipk_render_inst_lds_##id(int mode) {                                                             \
    switch (ACHIP_INST_PART_OF(mode)) {                                                                                \
    case 0: return achipk_render_inst_lds_##id##_p0(mode);                                                              \
    case 1: return achipk_render_inst_lds_##id##_p1(mode);                                                              \
    case 2: return achipk_render_inst_lds_##id##_p2(mode);                                                              \
    default: return achipk_render_inst_lds_##id##_p3(mode);                                                             \
    }                                                                                                                  \
  }
ACHIP_VARIANTS(X)
#undef X

/* the stream-kernel geometries (render_stream_inst.hip, -DACHIP_SINST=id) */
#define X(id, W, C)                                                                                                    \
  int achipk_render_sinst_launch_##id(int mode, int comp, const achip_frame_t *frames, int n, const achip_lut_t *lut,   \
                                     uint8_t *out, uint64_t stride, uint32_t *len, const achip_uniform_t *uniform,     \
                                     unsigned long long *prof, const achip_wire_t *wire, void *stream);                                                                    \
  int achipk_render_sinst_lds_##id(int mode);
ACHIP_STREAM_VARIANTS(X)
#undef X

/* the PACK instantiations of stream geometries 16 and 17 (render_stream_inst.hip with -DACHIP_SINST=16 / 17) */
int achipk_render_sinst_pack_launch_16(int mode, const achip_frame_t *frames, int n, const achip_lut_t *lut, uint64_t stride,
                                      uint32_t *len, const achip_uniform_t *uniform, const achip_wire_t *wire,
                                      const achip_packdev_t *pack, void *stream);
int achipk_render_sinst_pack_launch_17(int mode, const achip_frame_t *frames, int n, const achip_lut_t *lut, uint64_t stride,
                                      uint32_t *len, const achip_uniform_t *uniform, const achip_wire_t *wire,
                                      const achip_packdev_t *pack, void *stream);

/* the PARTS instantiations of stream geometry 18 (a frame's blocks shared out over ps->parts workgroups, 2..64) */
int achipk_render_sinst_parts_launch_18(int mode, int comp, const achip_frame_t *frames, int n, const achip_lut_t *lut,
                                       uint8_t *out, uint64_t stride, uint32_t *len, const achip_uniform_t *uniform,
                                       unsigned long long *prof, const achip_partsdev_t *ps, void *stream);

/* the rows-kernel geometries (render_rows_inst.hip, -DACHIP_RINST=id -DACHIP_RMODE=mode): run-structured modes, whole
 * frames; one translation unit per (geometry, mode) */
#define ACHIP_RINST_ARGS                                                                                               \
  int mode, int comp, const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out, uint64_t stride,       \
      uint32_t *len, const achip_uniform_t *uniform, const achip_wire_t *wire, const achip_partsdev_t *ps, void *stream
#define ACHIP_RINST_MODES(Y, id) Y(id, 0) Y(id, 5) Y(id, 6) Y(id, 7) Y(id, 8) /* mono, the four half-block modes */
#define Y(id, m)                                                                                                       \
  int achipk_render_rinst_launch_##id##_m##m(ACHIP_RINST_ARGS);                                                         \
  int achipk_render_rinst_lds_##id##_m##m(int mode);
#define X(id, W, C) ACHIP_RINST_MODES(Y, id)
ACHIP_ROWS_VARIANTS(X)
#undef X
#undef Y
#define Y(id, m)                                                                                                       \
  case m:                                                                                                              \
    return achipk_render_rinst_launch_##id##_m##m(mode, comp, frames, n, lut, out, stride, len, uniform, wire, ps, stream);
#define Z(id, m)                                                                                                       \
  case m:                                                                                                              \
    return achipk_render_rinst_lds_##id##_m##m(mode);
#define X(id, W, C)                                                                                                    \
  static inline int achipk_render_rinst_launch_##id(ACHIP_RINST_ARGS) {                                                 \
    switch (mode) { ACHIP_RINST_MODES(Y, id) }                                                                         \
    return 1; /* hipErrorInvalidValue */                                                                               \
  }                                                                                                                    \
  static inline int achipk_render_rinst_lds_##id(int mode) {                                                            \
    switch (mode) { ACHIP_RINST_MODES(Z, id) }                                                                         \
    return -1;                                                                                                         \
  }
ACHIP_ROWS_VARIANTS(X)
#undef X
#undef Y
#undef Z

#ifdef __cplusplus
}
#endif
#endif
