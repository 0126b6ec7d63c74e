/*
 * asciichat_render.h -- the DROP-IN layer of libasciichat_hip.so.
 *
 * These are the reference's own libasciichat entry points for the image->ASCII render path, with the
 * same names, C signatures, struct layouts, ownership and NULL/error behaviour, so that the callers
 * listed in SURVEY.md section 8(b) (src/server/stream.c:841, src/common/session/display.c:632,
 * src/common/session/host.c:696,710, src/web/mirror.c:209, tests/unit/video/ascii_test.c) link against
 * this library unmodified.  Each declaration cites the reference declaration it replaces
 * (paths relative to the reference tree, include/ascii-chat/...).
 *
 * Ownership (ascii.h:42): every returned char* is an individual malloc() block that the caller
 * free()s.  Inputs are borrowed, never retained, never mutated.  All entry points are re-entrant;
 * each calling thread gets its own HIP stream and pinned staging block.
 *
 * Behavioural difference by design: there is no CPU renderer in this library.  Without a HIP device
 * the render entry points return NULL (asciichat_hip_last_error() tells why).
 */
#ifndef ASCIICHAT_RENDER_H
#define ASCIICHAT_RENDER_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <sys/types.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- common/error_codes.h:51-109 ------------------------------------------------------------ */
typedef enum {
  ASCIICHAT_OK = 0,
  ERROR_GENERAL = 1,
  ERROR_MEMORY = 3,
  ERROR_TERMINAL = 25,
  ERROR_NOT_SUPPORTED = 30,
  ERROR_BUFFER = 81,
  ERROR_INVALID_STATE = 85,
  ERROR_INVALID_PARAM = 86
} asciichat_error_t;

/* ---- video/rgba/image.h:81-148 -------------------------------------------------------------- */
typedef struct {
  uint8_t r, g, b;
} __attribute__((packed)) rgb_pixel_t;

typedef enum { IMAGE_ALLOC_SIMD = 0, IMAGE_ALLOC_POOL = 1 } image_alloc_method_t;

typedef struct {
  int w;
  int h;
  rgb_pixel_t *pixels;
  uint8_t alloc_method;
} image_t;

#define IMAGE_MAX_WIDTH 3840
#define IMAGE_MAX_HEIGHT 2160
#define IMAGE_MAX_PIXELS_SIZE (IMAGE_MAX_WIDTH * IMAGE_MAX_HEIGHT * sizeof(rgb_pixel_t))

/* ---- platform/terminal.h:578-589, 593-628, 660-667, 707-738 ----------------------------------- */
typedef enum {
  TERM_COLOR_AUTO = -1,
  TERM_COLOR_NONE = 0,
  TERM_COLOR_16 = 1,
  TERM_COLOR_256 = 2,
  TERM_COLOR_TRUECOLOR = 3
} terminal_color_mode_t;

typedef enum { COLOR_FILTER_NONE = 0, COLOR_FILTER_RAINBOW = 12, COLOR_FILTER_COUNT } color_filter_t;

typedef enum { RENDER_MODE_FOREGROUND = 0, RENDER_MODE_BACKGROUND = 1, RENDER_MODE_HALF_BLOCK = 2 } render_mode_t;

typedef struct {
  terminal_color_mode_t color_level;
  uint32_t capabilities;
  uint32_t color_count;
  bool utf8_support;
  bool detection_reliable;
  render_mode_t render_mode;
  char term_type[64];
  char colorterm[64];
  bool wants_background;
  int palette_type;
  char palette_custom[64];
  uint8_t desired_fps;
  color_filter_t color_filter;
  bool wants_padding;
  size_t pad_height;
} terminal_capabilities_t;

/* ---- video/ascii/palette.h:161-197 ------------------------------------------------------------ */
#define PALETTE_CHARS_STANDARD "   ...',;:clodxkO0KXNWM"
#define PALETTE_CHARS_BLOCKS "   ░░▒▒▓▓██"
#define PALETTE_CHARS_DIGITAL "   -=≡≣▰▱◼"
#define PALETTE_CHARS_MINIMAL "   .-+*#"
#define PALETTE_CHARS_COOL "   ▁▂▃▄▅▆▇█"

/* ---- video/ascii/common.h:467-472, 498-512, 134 ------------------------------------------------ */
typedef struct {
  uint8_t width;
  uint8_t utf8_bytes[4];
  uint8_t byte_len;
  uint8_t _pad;
} utf8_char_t;

/* utf8_palette_cache_t (common.h:474-490): callers read the three leading tables; the reference's bookkeeping
 * tail (access counters, eviction heap index, third-party hash handle) is private there and opaque here. */
typedef struct utf8_palette_cache {
  utf8_char_t cache[256];
  utf8_char_t cache64[64];
  uint8_t char_index_ramp[256];
  uint64_t _private[40];
} utf8_palette_cache_t;

/* get_utf8_palette_cache (common.c:270-377): borrowed, thread-safe, built once per palette string. */
utf8_palette_cache_t *get_utf8_palette_cache(const char *ascii_chars);

void build_utf8_luminance_cache(const char *ascii_chars, utf8_char_t cache[256]);
void build_utf8_ramp64_cache(const char *ascii_chars, utf8_char_t cache64[64], uint8_t char_index_ramp[256]);
void ascii_simd_init(void);
extern char g_default_luminance_palette[256]; /* filled by ascii_simd_init (common.c:576-604) */

/* ---- video/ascii/ascii.h:172-174, 213-215, 230, 325, 344, 358-361, 400 -------------------------- */
char *ascii_convert(image_t *original, const ssize_t width, const ssize_t height, const bool color,
                    const bool _aspect_ratio, const bool stretch, const char *palette_chars,
                    const char luminance_palette[256]);
char *ascii_convert_with_capabilities(image_t *original, const ssize_t width, const ssize_t height,
                                      const terminal_capabilities_t *caps, const bool use_aspect_ratio,
                                      const bool stretch, const char *palette_chars);
/* ADDITIVE, not in the reference (declared here because it shares the reference's types): the same call with the frame left
 * in the caller's buffer, NUL-terminated -- no malloc per frame.  ASCIICHAT_OK and strlen in *out_len (may be NULL);
 * ERROR_BUFFER when frame + NUL do not fit (*out_len = what it needs, nothing usable in `out`); otherwise the code of the
 * condition under which ascii_convert_with_capabilities returns NULL (ascii.c:198-212,256-265). */
asciichat_error_t ascii_convert_with_capabilities_into(image_t *original, const ssize_t width, const ssize_t height,
                                                       const terminal_capabilities_t *caps, const bool use_aspect_ratio,
                                                       const bool stretch, const char *palette_chars, char *out,
                                                       size_t out_capacity, size_t *out_len);
char *image_print_with_capabilities(const image_t *image, const terminal_capabilities_t *caps, const char *palette);
char *ascii_pad_frame_width(const char *frame, size_t pad_left);
char *ascii_pad_frame_height(const char *frame, size_t pad_top);

typedef struct {
  const char *frame_data;
  size_t frame_size;
} ascii_frame_source_t;
char *ascii_create_grid(ascii_frame_source_t *sources, int source_count, int width, int height, size_t *out_size);

/* ascii_convert() reads GET_OPTION(render_mode) (ascii.c:138,152); the options registry is out of
 * scope, so the value is held here.  Default RENDER_MODE_FOREGROUND. */
void asciichat_hip_set_option_render_mode(render_mode_t mode);

/* ---- video/rgba/image.h (prototypes after :195) -------------------------------------------------- */
image_t *image_new(size_t width, size_t height);
void image_destroy(image_t *p);
image_t *image_new_from_pool(size_t width, size_t height);
void image_destroy_to_pool(image_t *image);
void image_clear(image_t *p);
image_t *image_new_copy(const image_t *source);
void image_resize(const image_t *s, image_t *d);
void image_resize_interpolation(const image_t *source, image_t *dest);

/* ---- video/ascii/scalar/foreground.h, background.h, sgr.c:413 ------------------------------------- */
char *image_print(const image_t *p, const char *palette);
char *image_print_color(const image_t *p, const char *palette);
char *image_print_256color(const image_t *image, const char *palette);
char *image_print_16color(const image_t *image, const char *palette);
char *image_print_color_background(const image_t *p, const char *palette);
char *image_print_color_simd(image_t *image, bool use_background_mode, bool use_256color, const char *ascii_chars);
/* include/ascii-chat/video/rgba/image.h:462,488 (lib/video/ascii/scalar/foreground.c:650-750, 752-846):
 * Floyd-Steinberg 16-colour renderers; (.., true, ..) is what image_print_color_simd dispatches to */
char *image_print_16color_dithered(const image_t *image, const char *palette);
char *image_print_16color_dithered_with_background(const image_t *image, bool use_background, const char *palette);

/* ---- video/ascii/scalar/halfblock.h ------------------------------------------------------------- */
char *rgb_to_truecolor_halfblocks_scalar(const uint8_t *rgb, int width, int height, int stride_bytes);
char *rgb_to_256color_halfblocks_scalar(const uint8_t *rgb, int width, int height, int stride_bytes,
                                        const char *palette);
char *rgb_to_16color_halfblocks_scalar(const uint8_t *rgb, int width, int height, int stride_bytes,
                                       const char *palette);
char *rgb_to_halfblocks_scalar(const uint8_t *rgb, int width, int height, int stride_bytes, const char *palette);

/* ---- video/terminal/ansi.h: scalar colour helpers and the RLE context ------------------------------ */
typedef enum { ANSI_MODE_FOREGROUND = 0, ANSI_MODE_BACKGROUND, ANSI_MODE_FOREGROUND_BACKGROUND } ansi_color_mode_t;
typedef struct {
  char *buffer;
  size_t capacity;
  size_t length;
  ansi_color_mode_t mode;
  bool first_pixel;
  uint8_t last_r, last_g, last_b;
} ansi_rle_context_t;

uint8_t rgb_to_256color(uint8_t r, uint8_t g, uint8_t b);
uint8_t rgb_to_16color(uint8_t r, uint8_t g, uint8_t b);
void get_16color_rgb(uint8_t color_index, uint8_t *r, uint8_t *g, uint8_t *b);
/* ansi.h:62-67, 272: one pixel of the Floyd-Steinberg 16-colour pass against a caller-held width x height error buffer
 * (NULL = no dithering) */
typedef struct {
  int r;
  int g;
  int b;
} rgb_error_t;
uint8_t rgb_to_16color_dithered(int r, int g, int b, int x, int y, int width, int height, rgb_error_t *error_buffer);
char *append_truecolor_fg(char *dst, uint8_t r, uint8_t g, uint8_t b);
char *append_truecolor_bg(char *dst, uint8_t r, uint8_t g, uint8_t b);
char *append_truecolor_fg_bg(char *dst, uint8_t fg_r, uint8_t fg_g, uint8_t fg_b, uint8_t bg_r, uint8_t bg_g,
                             uint8_t bg_b);
char *append_256color_fg(char *dst, uint8_t color_index);
char *append_256color_bg(char *dst, uint8_t color_index);
char *append_16color_fg(char *dst, uint8_t color_index);
char *append_16color_bg(char *dst, uint8_t color_index);
void ansi_rle_init(ansi_rle_context_t *ctx, char *buffer, size_t capacity, ansi_color_mode_t mode);
void ansi_rle_add_pixel(ansi_rle_context_t *ctx, uint8_t r, uint8_t g, uint8_t b, char ascii_char);
void ansi_rle_finish(ansi_rle_context_t *ctx);

/* ---- video/ascii/output_buffer.h:84-88 and helpers -------------------------------------------------- */
typedef struct {
  char *buf;
  size_t len;
  size_t cap;
} outbuf_t;
void ob_reserve(outbuf_t *ob, size_t need);
void ob_putc(outbuf_t *ob, char c);
void ob_write(outbuf_t *ob, const char *s, size_t n);
void ob_term(outbuf_t *ob);
void ob_u8(outbuf_t *ob, uint8_t v);
void ob_u32(outbuf_t *ob, uint32_t v);
void emit_set_fg(outbuf_t *ob, uint8_t r, uint8_t g, uint8_t b);
void emit_set_bg(outbuf_t *ob, uint8_t r, uint8_t g, uint8_t b);
void emit_reset(outbuf_t *ob);
bool rep_is_profitable(uint32_t runlen);
void emit_rep(outbuf_t *ob, uint32_t extra);

/* ---- video/ascii/rle.h:61,87 and video/ascii/frame_validator.h:20,32 (exported by the reference,
 * called from nowhere in its tree; host string utilities on finished frames) ------------------------------------ */
char *ansi_expand_rle(const char *input, size_t input_len);   /* ESC[Nb -> N copies of the last printable character */
char *ansi_compress_rle(const char *input, size_t input_len); /* runs of one byte >= 6 long -> byte ESC[<n-1>b          */
bool frame_validate_integrity(const char *frame_data, size_t frame_size); /* ends right after its last ESC[0m ?       */
size_t frame_get_valid_end(const char *frame_data, size_t frame_size);

/* ---- video/rgba/color_filter.h:129,155 (COLOR_FILTER_RAINBOW of the display path, src/common/session/display.c:639-650,
 * src/web/mirror.c:223).  The batch path folds this into the emission (achip_frame_set_rainbow); these are the
 * reference's host-string forms.  rainbow_replace_ansi_colors returns NULL when the frame holds no ESC[38;2;..m. */
void color_filter_calculate_rainbow(float time, uint8_t *r, uint8_t *g, uint8_t *b);
char *rainbow_replace_ansi_colors(const char *ansi_string, float time_seconds);

/* ---- util/aspect_ratio.h ---------------------------------------------------------------------------- */
void aspect_ratio(const ssize_t img_w, const ssize_t img_h, const ssize_t width, const ssize_t height,
                  const bool stretch, ssize_t *out_width, ssize_t *out_height);

/* ---- buffer_pool.h:41-50 and API ------------------------------------------------------------------------
 * Same header-magic free contract (buffer_pool_free(NULL, p, size) works on any block).  New: objects
 * larger than BUFFER_POOL_MAX_SINGLE_SIZE -- i.e. every 1080p/4K frame, which the reference sends to the
 * malloc fallback (lib/buffer_pool.c:122-143) -- come from a PINNED, device-mapped size class, so frames
 * allocated with image_new_from_pool() are read by the GPU in place (zero-copy over PCIe) instead of
 * being staged.  */
#define BUFFER_POOL_MAX_BYTES (337 * 1024 * 1024)
#define BUFFER_POOL_SHRINK_DELAY_NS 5000000000ULL
#define BUFFER_POOL_MIN_SIZE 64
#define BUFFER_POOL_MAX_SINGLE_SIZE (4 * 1024 * 1024)
#define BUFFER_POOL_PINNED_MAX_BYTES (1024ull * 1024 * 1024) /* cap of the pinned frame class (new) */

typedef struct buffer_pool buffer_pool_t;
buffer_pool_t *buffer_pool_create(size_t max_bytes, uint64_t shrink_delay_ns);
void buffer_pool_destroy(buffer_pool_t *pool);
void *buffer_pool_alloc(buffer_pool_t *pool, size_t size);
void buffer_pool_free(buffer_pool_t *pool, const void *data, size_t size);
void buffer_pool_shrink(buffer_pool_t *pool);
void buffer_pool_get_stats(buffer_pool_t *pool, size_t *current_bytes, size_t *used_bytes, size_t *free_bytes);
void buffer_pool_init_global(void);
void buffer_pool_cleanup_global(void);
buffer_pool_t *buffer_pool_get_global(void);
/* new: how many live blocks are pinned+device-mapped, and whether a pointer lies in one */
size_t buffer_pool_pinned_blocks(buffer_pool_t *pool);
bool buffer_pool_is_pinned(const void *data);

#ifdef __cplusplus
}
#endif
#endif
