/*
 * achip_types.h -- plain-C descriptor structs shared by the host shim (C), the HIP kernels and the
 * tests.  "achip" = asciichat-hip.  Everything here is POD and lives in HBM (or pinned host memory)
 * exactly as laid out below.
 */
#ifndef ACHIP_TYPES_H
#define ACHIP_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Render modes: one per reference renderer on the path (SURVEY.md section 8a rows PM..HM). */
enum {
  ACHIP_MODE_MONO = 0,    /* image_print                         foreground.c:27-138   */
  ACHIP_MODE_TRUE_FG = 1, /* image_print_color (+ansi_rle_*)     foreground.c:195-308  */
  ACHIP_MODE_256_FG = 2,  /* image_print_256color                foreground.c:433-509  */
  ACHIP_MODE_16_FG = 3,   /* image_print_16color                 foreground.c:535-624  */
  ACHIP_MODE_TRUE_BG = 4, /* image_print_color_background        background.c:17-84    */
  ACHIP_MODE_HB_TRUE = 5, /* rgb_to_truecolor_halfblocks_scalar  halfblock.c:48-165    */
  ACHIP_MODE_HB_256 = 6,  /* rgb_to_256color_halfblocks_scalar   halfblock.c:416-524   */
  ACHIP_MODE_HB_16 = 7,   /* rgb_to_16color_halfblocks_scalar    halfblock.c:297-405   */
  ACHIP_MODE_HB_MONO = 8, /* rgb_to_halfblocks_scalar            halfblock.c:184-286   */
  ACHIP_MODE_16_DITHER_BG = 9, /* image_print_16color_dithered_with_background(.., true, ..)  foreground.c:752-846
                                  (what TRUECOLOR + RENDER_MODE_BACKGROUND dispatches to, sgr.c:429-430) */
  ACHIP_MODE_COUNT = 10
};

/* One source of a pixel-space grid composite (create_multi_source_composite, src/server/stream.c:664-779). */
typedef struct {
  const uint8_t *src; /* RGB24, tightly packed; NULL = empty cell */
  int32_t src_w, src_h;
  int32_t src_stride;           /* bytes per source row (>= 3*src_w)          */
  int32_t _pad0;
  int32_t tile_w, tile_h;       /* contain-fitted size inside the cell        */
  int32_t org_x, org_y;         /* canvas position of tile pixel (0,0)        */
  uint32_t x_ratio, y_ratio;    /* 16.16 nearest-neighbour ratios src -> tile */
} achip_comp_src_t;

typedef struct {
  int32_t canvas_w, canvas_h; /* W x 2H pixels */
  int32_t cols, rows;
  int32_t cell_w, cell_h;
  int32_t n_src; /* <= 9 */
  int32_t _pad;
  achip_comp_src_t s[9];
} achip_composite_t;

/* One frame of a batch.  Sampling is the reference's nearest-neighbour rule (image.c:293-325):
 * sx = min((x * x_ratio) >> 16, src_w - 1). */
typedef struct {
  const uint8_t *src;            /* RGB24 source frame (device-visible); ignored when comp != NULL */
  const achip_composite_t *comp; /* optional: sample a virtual composite canvas instead of src     */
  int32_t src_w, src_h;          /* source (or canvas) size in pixels                              */
  int32_t out_w, out_h;          /* sampled size: text columns x pixel rows (2 per text row in half-block) */
  int32_t pad_left, pad_top;     /* ascii_pad_frame_width / _height folded into the emission       */
  uint32_t x_ratio, y_ratio;     /* ((src << 16) / out) + 1                                         */
  int32_t src_stride;            /* bytes per source row; 0 = tightly packed (3*src_w)              */
  uint32_t ops;                  /* display-path pre-passes folded into the sampler: ACHIP_OP_* | tint << 8 */
} achip_frame_t;

/* A launch whose descriptors differ only in their source pointer, and there by a constant pitch (a batch of equally
 * sized client frames in one slab, or a single frame): the common descriptor travels in the kernel arguments and
 * workgroup i reads frame `f` with src = f.src + i * src_pitch -- no dependent descriptor fetch in front of the
 * first gather.  enabled == 0: descriptors are read from the device array as usual. */
typedef struct {
  achip_frame_t f;
  int64_t src_pitch;
  uint32_t enabled;
  uint32_t flags; /* ACHIP_UNIFORM_*: launch-wide facts that travel with the kernel arguments even when enabled == 0 */
} achip_uniform_t;
/* Where the stream kernel's CRC instantiation leaves the wire stage's results (device pointers; by value in the kernel
 * arguments).  crc is required; hdr / pkt_crc (24-byte ascii_frame_packet_t headers, CRC of header || frame) and dims
 * ({width, height} per frame for the headers) are optional. */
typedef struct {
  uint32_t *crc;
  const uint32_t *dims;
  uint8_t *hdr;
  uint32_t *pkt_crc;
} achip_wire_t;
/* Where a PACK instantiation of the stream kernel leaves its frames (device pointers; by value in the kernel arguments):
 * frames at their exact lengths, back to back from dst in the order in which they finish, every start 16-byte aligned.
 * off_out (n + 1 entries, [n] = total bytes) and len_out (n: the length, or the render error code of a frame that takes
 * no room) may be NULL.  cursor: two 64-bit words owned by the plan, zero between launches ([0] next free byte, [1]
 * workgroups that have reported); dst / off_out / len_out may be device memory or the device alias of mapped host memory. */
typedef struct {
  uint8_t *dst;
  uint64_t capacity;
  uint64_t *off_out;
  uint32_t *len_out;
  unsigned long long *cursor;
} achip_packdev_t;
/* PARTS instantiations of the stream kernel (a frame's blocks shared out over `parts` workgroups; by value in the kernel
 * arguments): sync = n_frames * parts 64-bit words owned by the caller, {epoch:32, bytes:32}; a workgroup publishes the
 * bytes of its blocks under the launch's epoch, its successors add up what the parts in front of them published.  Words
 * never need clearing: every launch on them takes a new epoch (never 0). */
typedef struct {
  int parts;
  uint32_t epoch;
  unsigned long long *sync;
} achip_partsdev_t;
/* Several nearest-neighbour resizes in one launch (the grid path resizes every source a rank owns per tick): by value in
 * the kernel arguments, workgroup (x, k) works on entry k. */
#define ACHIP_RESIZE_BATCH_MAX 16
typedef struct {
  const uint8_t *src;
  uint8_t *dst;
  int32_t sw, sh, dw, dh;
  uint32_t x_ratio, y_ratio; /* ((s << 16) / d) + 1, image.c:282-283 */
} achip_resize_item_t;
typedef struct {
  achip_resize_item_t item[ACHIP_RESIZE_BATCH_MAX];
  int32_t n, _pad;
} achip_resize_batch_t;
/* new source pointers for the (<= 9) placed tiles of a device-resident composite descriptor, by value in the kernel
 * arguments of a one-wave launch (the single-GPU grid path: the render samples the clients' frames directly) */
typedef struct {
  const uint8_t *src[9];
} achip_comp_poke_t;
#define ACHIP_UNIFORM_PALETTE_ASCII 1u /* every glyph of the launch's palette is a single byte < 0x80 */
/* bits 31..8: cells ((pad_left + out_w) * out_h) of the launch's largest frame, 0 = not stated.  The stream kernel
 * sizes its per-block LDS words from it (a frame with more cells than stated is refused: ACHIP_LEN_BADDESC). */
#define ACHIP_UNIFORM_MAX_CELLS_SHIFT 8
#define ACHIP_UNIFORM_MAX_CELLS(cells) ((uint32_t)((cells) > 0 && (cells) < (1l << 24) ? (cells) : 0) << ACHIP_UNIFORM_MAX_CELLS_SHIFT)

/* achip_frame_t.ops: the client display path flips the frame and applies a monochrome tint on full-frame
 * copies before rendering (src/common/session/display.c:546-623, lib/video/rgba/color_filter.c:246-345).
 * Both commute with nearest-neighbour sampling, so here they are an index map and a per-sample map. */
#define ACHIP_OP_FLIP_X 1u      /* sample column src_w-1-x                                   */
#define ACHIP_OP_FLIP_Y 2u      /* sample row    src_h-1-y                                   */
#define ACHIP_OP_TINT 4u        /* grey = (77R+150G+29B)>>8, channel = tint*grey/255          */
#define ACHIP_OP_TINT_ON_WHITE 8u /* foreground_on_bg filters: (tint*(255-grey) + 255*grey)/255 */
#define ACHIP_OP_TINT_SHIFT 8   /* bits 31..8: tint colour 0xBBGGRR                           */
/* ACHIP_MODE_16_DITHER_BG only: the two exported foreground-only forms of the dithered renderer */
#define ACHIP_OP_DITHER_FG 16u   /* image_print_16color_dithered_with_background(.., false, ..): one fg SGR per
                                    cell, glyph cache[Y]  (foreground.c:809-819)                */
#define ACHIP_OP_DITHER_RAMP 32u /* image_print_16color_dithered: as above with the glyph taken through the
                                    64-entry ramp, cache[ramp[Y>>2]]  (foreground.c:712-723)    */
#define ACHIP_OP_DITHER_MASK 48u
/* rainbow_replace_ansi_colors (lib/video/rgba/color_filter.c:348-408; display.c:639-650, web/mirror.c:223) folded
 * into the emission: every `ESC[38;2;..m` the frame would carry is written with the colour in bits 31..8 instead
 * (the decision WHETHER to emit one still follows the pixels).  Exclusive with ACHIP_OP_TINT. */
#define ACHIP_OP_FG_OVERRIDE 64u

/* Glyph tables of one palette (utf8_palette_cache_t restated, common.c:380-490).  A glyph is its
 * UTF-8 bytes packed little-endian in a u32; its length follows from the lead byte. */
typedef struct {
  uint32_t glyph[256];  /* cache[Y]                 */
  uint32_t glyph64[64]; /* cache64[i]               */
  uint8_t ramp[64];     /* char_index_ramp[0..63]   */
  uint32_t flags;       /* ACHIP_LUT_*              */
} achip_lut_t;
#define ACHIP_LUT_MULTIBYTE 1u /* some glyph is a multi-byte UTF-8 sequence */

#define ACHIP_LEN_OVERFLOW 0xFFFFFFFFu /* out_len[] value when a frame did not fit its slab slot */
#define ACHIP_LEN_BADDESC 0xFFFFFFFEu  /* out_len[] value for an unsupported descriptor          */

#ifdef __cplusplus
}
#endif
#endif
