/*
 * achip_host.h -- host-side (plain C, no GPU) planning logic of the render path: what the reference
 * does around its hot loops before/after pixels are touched -- aspect fit, padding sizes, sampling
 * ratios, glyph tables, mode dispatch, output bounds.  Shared by the C-ABI shim and the tests.
 */
#ifndef ACHIP_HOST_H
#define ACHIP_HOST_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <sys/types.h>

#include "achip_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* aspect_ratio(), lib/util/aspect_ratio.c:69-91 -- exported with the reference's own name & signature */
void aspect_ratio(const ssize_t img_w, const ssize_t img_h, const ssize_t width, const ssize_t height,
                  const bool stretch, ssize_t *out_width, ssize_t *out_height);

/* build_utf8_luminance_cache + build_utf8_ramp64_cache (common.c:380-490) into the device layout.
 * Returns 0, or -1 for a NULL/empty palette. */
int achip_lut_build(const char *palette_chars, achip_lut_t *lut);

/* image_print_with_capabilities dispatch (ascii.c:955-1002 + sgr.c:413-436, x86 SIMD_SUPPORT build).
 * TRUECOLOR+BACKGROUND maps to the Floyd-Steinberg 16-colour renderer, as in the reference. */
int achip_mode_from_caps(int color_level, int render_mode);

/* Fill one frame descriptor the way ascii_convert_with_capabilities (ascii.c:194-387) sizes things:
 * aspect fit BEFORE half-block doubling, padding only when use_aspect && wants_padding.
 * Returns 0 on success, -1 where the reference returns NULL (bad dims, resized image > 3840x2160). */
int achip_frame_setup(achip_frame_t *f, const uint8_t *src_dev, int src_w, int src_h, ssize_t width, ssize_t height,
                      int render_mode, bool wants_padding, bool use_aspect, bool stretch);

/* Descriptor for rendering an image as-is (image_print_* / rgb_to_*_halfblocks_scalar on an already sized image). */
int achip_frame_identity(achip_frame_t *f, const uint8_t *src_dev, int w, int h);

/* Fold the display path's pre-passes into a frame descriptor (session_display_convert_to_ascii,
 * src/common/session/display.c:546-623): flips apply only to images larger than 1x1 (display.c:549);
 * color_filter is the reference's color_filter_t (0 none, 1 black .. 11 yellow; 12 = rainbow is a
 * post-pass on the ANSI string there and is rejected here).  Returns 0, or -1 for an unknown filter. */
int achip_frame_set_display_ops(achip_frame_t *f, bool flip_x, bool flip_y, int color_filter);
/* ACHIP_MODE_16_DITHER_BG frames: pick which exported form of the dithered renderer the frame follows --
 * image_print_16color_dithered_with_background(img, use_background, pal) (foreground.c:752-846), or with
 * ramp_glyph (and !use_background) image_print_16color_dithered(img, pal) (foreground.c:650-750).
 * Default (never called) = use_background, what image_print_color_simd dispatches to (sgr.c:429-430). */
/* Fill `u` for frames[0..n): enabled iff every field but `src` is equal and src[i] == src[0] + i * pitch for one pitch
 * (n == 1: always) -- frames with sources of their own, or frames that all sample the SAME composite (no source, pitch 0:
 * the grid's target clients).  Returns u->enabled. */
int achip_frames_uniform(const achip_frame_t *frames, int n, achip_uniform_t *u);
int achip_frame_set_dither_style(achip_frame_t *f, bool use_background, bool ramp_glyph);
/* COLOR_FILTER_RAINBOW of the display path: the colour of color_filter_calculate_rainbow(time_seconds)
 * (lib/video/rgba/color_filter.c:169-243, float HSV walk with a luminance floor) ... */
void achip_rainbow_color(float time_seconds, uint8_t *r, uint8_t *g, uint8_t *b);
/* ... and the frame rendered as if rainbow_replace_ansi_colors(result, time_seconds) had been run over its output
 * (color_filter.c:348-408).  Replaces any tint set by achip_frame_set_display_ops (the reference applies one or the
 * other, display.c:611,639); flips are kept. */
int achip_frame_set_rainbow(achip_frame_t *f, float time_seconds);

/* 16.16 nearest-neighbour ratio, image.c:293-294 */
uint32_t achip_nn_ratio(int src, int dst);

/* Upper bound (bytes, excluding the NUL) of one rendered frame; multiple of 16 when rounded by the caller. */
size_t achip_out_bound(int mode, const achip_frame_t *f);

/* The in-memory camera frame blob [u32 BE width][u32 BE height][RGB24 pixels] that the server keeps per client
 * (video_frame_get_latest).  Validates it as the reference does -- exact = false: collect_video_sources
 * (src/server/stream.c:330-372: size >= 8 + 3wh, extra bytes ignored); exact = true: the IMAGE_FRAME receive
 * handler (src/server/protocol.c:784-815: size == 8 + 3wh) -- with dimensions in 1..3840 x 1..2160
 * (image_validate_dimensions, lib/util/image.c:100-113).  Returns 0 and the parsed fields, or ACHIP_BLOB_*. */
#define ACHIP_BLOB_SHORT (-1) /* shorter than a header and one pixel        */
#define ACHIP_BLOB_DIMS (-2)  /* zero or oversized dimensions               */
#define ACHIP_BLOB_SIZE (-3)  /* byte count does not match the dimensions   */
int achip_frame_blob_parse(const void *blob, size_t size, bool exact, uint32_t *width, uint32_t *height,
                           const uint8_t **pixels);

/* Launch geometry for a batch.  A batch with fewer frames than the GPU has CUs leaves most of it idle when one
 * workgroup renders a whole frame, so such frames are cut into `parts` bands of text rows rendered by separate
 * workgroups (each band exactly one chunk of its geometry; the bands learn their output offset from each other
 * on the device).
 *   variant_caps[v]  cells per chunk of kernel geometry v (render_variants.h)
 *   n_cus            compute units this launch can count on: the device's, divided by the number of launches the
 *                    caller keeps in flight on separate streams (asciichat_hip_plan_set_concurrency)
 *   split_request    0 = automatic, < 0 = never split, > 0 = this many text rows per band
 *   forced_variant   >= 0: use this geometry (tuning), -1: choose
 * Outputs the geometry id, bands per frame (1 = no split) and text rows per band.  Returns 0, or -1 when a
 * padded row does not fit the geometry. */
long achip_max_cells(const achip_frame_t *frames, int n_frames);
/* what a launch states in the ACHIP_UNIFORM_MAX_CELLS field for geometry `variant`: the largest frame's cells (stream
 * geometries) or the most blocks any frame has (rows geometries: whole text rows per block) */
long achip_uniform_extent(int mode, int variant, const achip_frame_t *frames, int n_frames);
/* false for a descriptor the kernels' 32-bit source offsets (and 24-bit row-stride multiply) cannot address */
bool achip_frame_extent_ok(const achip_frame_t *f);
int achip_choose_geometry(int mode, const achip_frame_t *frames, int n_frames, bool palette_ascii_only,
                          const int *variant_caps, int n_cus, int split_request, int forced_variant, int *variant,
                          int *parts, int *rows_per_part);
/* true when every glyph of the palette is a single byte < 0x80 */
bool achip_palette_ascii_only(const char *palette_chars);

/* calculate_optimal_grid_layout (src/server/stream.c:523-651) */
void achip_grid_layout(const int *src_w, const int *src_h, int n, int term_w, int term_h, int *cols, int *rows);

/* create_multi_source_composite geometry (stream.c:664-779): fills comp for n (<= 9 used) sources. */
void achip_composite_setup(achip_composite_t *comp, const uint8_t *const *src_dev, const int *src_w, const int *src_h,
                           int n, int term_w, int term_h);

#ifdef __cplusplus
}
#endif
#endif
