/*
 * asciichat_oracle.h -- CPU restatement of ascii-chat's image->ASCII render path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link,
 * import or execute it, and only as the checker / reported CPU baseline.
 *
 * This is a sequential plain-C restatement of the reference algorithm, written
 * from the behaviour of the reference sources cited per function (paths are
 * relative to the upstream tree, zfogg/ascii-chat @ 2026-07-23).  No reference
 * source is copied or compiled into this file.
 *
 * Pinning (see oracle/README.md and tests/test_oracle_pins.py): the reference
 * cannot be built in this image without writing stand-ins for un-vendored
 * third-party headers (sokol_time.h, uthash.h), which this project's rules
 * forbid, so the oracle is pinned against (1) every known-answer value the
 * reference's own unit tests hold for this path and (2) the whole-frame
 * length + FNV-1a-32 anchors of the reference's output recorded in SURVEY.md
 * section 8(c) / Appendix B.
 */
#ifndef ASCIICHAT_ORACLE_H
#define ASCIICHAT_ORACLE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* include/ascii-chat/platform/terminal.h:578-589, 660-667 */
enum { ORC_COLOR_AUTO = -1, ORC_COLOR_NONE = 0, ORC_COLOR_16 = 1, ORC_COLOR_256 = 2, ORC_COLOR_TRUECOLOR = 3 };
enum { ORC_RENDER_FOREGROUND = 0, ORC_RENDER_BACKGROUND = 1, ORC_RENDER_HALF_BLOCK = 2 };

/* include/ascii-chat/video/ascii/common.h:467-472 (only the fields the path reads) */
typedef struct {
  uint8_t bytes[4];
  uint8_t len;
} orc_glyph_t;

/* utf8_palette_cache_t, lib/video/ascii/common.c:380-490 */
typedef struct {
  orc_glyph_t cache[256];  /* luminance -> glyph            (build_utf8_luminance_cache) */
  orc_glyph_t cache64[64]; /* 6-bit luminance bucket -> glyph (build_utf8_ramp64_cache)   */
  uint8_t ramp[64];        /* char_index_ramp[0..63]                                      */
  int char_count;
} orc_palette_t;

/* ---- scalar helpers ---------------------------------------------------- */
uint32_t orc_fnv1a32(const void *data, size_t n);
int orc_luma(int r, int g, int b);                       /* foreground.c:93              */
uint8_t orc_rgb_to_256(uint8_t r, uint8_t g, uint8_t b); /* ansi.c:360-379               */
uint8_t orc_rgb_to_16(uint8_t r, uint8_t g, uint8_t b);  /* ansi.c:437-477               */
bool orc_rep_is_profitable(uint32_t run);                /* output_buffer.c:148-155      */
int orc_digits_u32(uint32_t v);                          /* util/number.h:62-82          */
/* append_truecolor_fg/bg (ansi.c:143-193), 256/16-colour SGR strings (ansi.c:326-419):
 * write into dst, return number of bytes written. */
int orc_sgr_truecolor(char *dst, int bg, uint8_t r, uint8_t g, uint8_t b);
int orc_sgr_256(char *dst, int bg, uint8_t idx);
int orc_sgr_16(char *dst, int bg, uint8_t idx);

int orc_palette_build(const char *chars, orc_palette_t *pal); /* common.c:380-490        */

/* aspect_ratio(), lib/util/aspect_ratio.c:18-91 (float, ROUND = (int)(0.5f + x)) */
void orc_aspect_ratio(long img_w, long img_h, long width, long height, bool stretch, long *out_w, long *out_h);

/* image_resize_interpolation, lib/video/rgba/image.c:267-328 (nearest neighbour, 16.16) */
void orc_resize_nn(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh);

/* ---- renderers: RGB24 tightly packed in, malloc'd NUL-terminated string out ---- */
char *orc_print_mono(const uint8_t *rgb, int w, int h, const char *palette, size_t *len);          /* foreground.c:27-138  */
char *orc_print_truecolor_fg(const uint8_t *rgb, int w, int h, const char *palette, size_t *len);  /* foreground.c:195-308 */
char *orc_print_256_fg(const uint8_t *rgb, int w, int h, const char *palette, size_t *len);        /* foreground.c:433-509 */
char *orc_print_16_fg(const uint8_t *rgb, int w, int h, const char *palette, size_t *len);         /* foreground.c:535-624 */
char *orc_print_truecolor_bg(const uint8_t *rgb, int w, int h, const char *palette, size_t *len);  /* background.c:17-84   */
char *orc_print_16_dithered(const uint8_t *rgb, int w, int h, bool use_background, const char *palette,
                            size_t *len);                                                          /* foreground.c:752-846 */
char *orc_print_16_dithered_fg(const uint8_t *rgb, int w, int h, const char *palette, size_t *len); /* foreground.c:650-750 */
char *orc_halfblock_truecolor(const uint8_t *rgb, int w, int h, size_t *len);                      /* halfblock.c:48-165   */
char *orc_halfblock_256(const uint8_t *rgb, int w, int h, size_t *len);                            /* halfblock.c:416-524  */
char *orc_halfblock_16(const uint8_t *rgb, int w, int h, size_t *len);                             /* halfblock.c:297-405  */
char *orc_halfblock_mono(const uint8_t *rgb, int w, int h, size_t *len);                           /* halfblock.c:184-286  */

/* image_print_with_capabilities dispatcher, ascii.c:955-1002 + sgr.c:413-436 (x86 SIMD_SUPPORT build) */
char *orc_print_with_caps(const uint8_t *rgb, int w, int h, int color_level, int render_mode, const char *palette,
                          size_t *len);

char *orc_pad_width(const char *frame, size_t pad_left);  /* ascii.c:457-517 */
char *orc_pad_height(const char *frame, size_t pad_top);  /* ascii.c:902-941 */

/* ascii_convert_with_capabilities, ascii.c:194-387 */
char *orc_convert_with_caps(const uint8_t *rgb, int src_w, int src_h, long width, long height, int color_level,
                            int render_mode, bool wants_padding, bool use_aspect, bool stretch, const char *palette,
                            size_t *len);
/* ascii_convert, ascii.c:72-191; option_render_mode stands for GET_OPTION(render_mode) */
char *orc_convert(const uint8_t *rgb, int src_w, int src_h, long width, long height, bool color, bool use_aspect,
                  bool stretch, const char *palette, int option_render_mode, size_t *len);

/* client display path pre-passes (src/common/session/display.c:546-623):
 * flips on a full copy, then apply_color_filter (lib/video/rgba/color_filter.c:246-345) on another copy */
void orc_flip(uint8_t *rgb, int w, int h, bool flip_x, bool flip_y);                 /* display.c:563-590 */
int orc_color_filter(uint8_t *rgb, int w, int h, int stride, int color_filter);      /* color_filter.c:274-345 */
char *orc_display_convert(const uint8_t *rgb, int src_w, int src_h, long width, long height, int color_level,
                          int render_mode, bool wants_padding, bool use_aspect, bool stretch, const char *palette,
                          bool flip_x, bool flip_y, int color_filter, size_t *len);

/* COLOR_FILTER_RAINBOW: lib/video/rgba/color_filter.c:169-243 (colour of the moment, float) and :348-408 (every
 * ESC[38;2;..m of a finished frame rewritten with it; NULL when the frame holds none, as the reference) */
void orc_rainbow_color(float time_seconds, uint8_t rgb[3]);
char *orc_rainbow_replace(const char *frame, float time_seconds, size_t *out_len);

/* ---- uncalled leftovers: lib/video/ascii/rle.c:13-162, frame_validator.c:13-80 ---- */
char *orc_expand_rle(const char *in, size_t n, size_t *out_len);
char *orc_compress_rle(const char *in, size_t n, size_t *out_len);
int orc_frame_validate_integrity(const char *d, size_t n);
size_t orc_frame_get_valid_end(const char *d, size_t n);

/* ---- ingest ---------------------------------------------------------------- */
/* the frame-blob checks of collect_video_sources (src/server/stream.c:330-372; exact = 0) and of the IMAGE_FRAME
 * receive handler (src/server/protocol.c:784-815; exact = 1).  Returns 1 when the reference accepts the blob
 * (and its dimensions), 0 when it skips / rejects it. */
int orc_frame_blob_accept(const void *blob, size_t size, int exact, uint32_t *w, uint32_t *h);

/* ---- wire stage after render ------------------------------------------------ */
/* asciichat_crc32_sw, lib/network/crc32.c:171-189: CRC-32C (Castagnoli), bit by bit */
uint32_t orc_crc32c(const void *data, size_t n);
/* acip_send_ascii_frame's header, lib/network/acip/server.c:186-214: 24 bytes in network byte order;
 * returns the CRC that packet_send_via_transport (lib/network/acip/send.c:59-69) computes over header+frame */
uint32_t orc_ascii_frame_packet(const void *frame, size_t n, uint32_t width, uint32_t height, uint8_t hdr[24]);

/* ascii_create_grid, ascii.c:602-885 */
typedef struct {
  const char *frame_data;
  size_t frame_size;
} orc_frame_source_t;
char *orc_create_grid(const orc_frame_source_t *sources, int n, int width, int height, size_t *out_size);

/* server pixel-space composite, src/server/stream.c:523-651 (layout) and :664-779 (composite) */
void orc_grid_layout(const int *src_w, const int *src_h, int n, int term_w, int term_h, int *cols, int *rows);
/* returns malloc'd (term_w) x (2*term_h) RGB24 canvas */
uint8_t *orc_composite(const uint8_t *const *src, const int *src_w, const int *src_h, int n, int term_w, int term_h,
                       int *out_w, int *out_h);

#ifdef __cplusplus
}
#endif
#endif
