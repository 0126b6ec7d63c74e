/* achip_host.c -- see achip_host.h.  Plain C11, no GPU calls. */
#include "achip_host.h"
#include "internal.h"

#include <math.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

/* ROUND(), include/ascii-chat/util/math.h:53; result floored at MIN_DIMENSION (aspect_ratio.c:18-36) */
static ssize_t round_dim(float v) {
  int r = (int)(0.5f + v);
  return r > 0 ? r : 1;
}

void aspect_ratio(const ssize_t img_w, const ssize_t img_h, const ssize_t width, const ssize_t height,
                  const bool stretch, ssize_t *out_width, ssize_t *out_height) {
  if (!out_width || !out_height)
    return;
  if (img_w <= 0 || img_h <= 0) {
    *out_width = 1;
    *out_height = 1;
    return;
  }
  if (stretch) {
    *out_width = width;
    *out_height = height;
    return;
  }
  /* terminal cells are twice as tall as wide (CHAR_ASPECT 2.0f); float on purpose: parity with the reference */
  const float cell = 2.0f;
  ssize_t w_fit = round_dim((float)height * (float)img_w / (float)img_h * cell);
  ssize_t h_fit = round_dim(((float)width / cell) * (float)img_h / (float)img_w);
  if (w_fit <= width) {
    *out_width = w_fit;
    *out_height = height;
  } else {
    *out_width = width;
    *out_height = h_fit;
  }
  if (*out_width <= 0)
    *out_width = 1;
  if (*out_height <= 0)
    *out_height = 1;
}

static int utf8_seq_len(unsigned char c) {
  if ((c & 0xE0) == 0xC0)
    return 2;
  if ((c & 0xF0) == 0xE0)
    return 3;
  if ((c & 0xF8) == 0xF0)
    return 4;
  return 1;
}

int achip_lut_build(const char *palette_chars, achip_lut_t *lut) {
  if (!palette_chars || !lut || palette_chars[0] == '\0')
    return -1;
  memset(lut, 0, sizeof(*lut));
  uint32_t packed[256];
  int count = 0;
  size_t len = strlen(palette_chars), pos = 0;
  while (pos < len && count < 255) {
    int n = utf8_seq_len((unsigned char)palette_chars[pos]);
    uint32_t g = 0;
    for (int k = 0; k < n && pos + (size_t)k < len; k++)
      g |= (uint32_t)(unsigned char)palette_chars[pos + (size_t)k] << (8 * k);
    packed[count++] = g;
    pos += (size_t)n;
  }
  for (int i = 0; i < 256; i++) {
    int ci = count > 1 ? (i * (count - 1) + 127) / 255 : 0;
    if (ci >= count)
      ci = count - 1;
    lut->glyph[i] = packed[ci];
  }
  for (int i = 0; i < 64; i++) {
    int ci = count > 1 ? (i * (count - 1) + 31) / 63 : 0;
    if (ci >= count)
      ci = count - 1;
    lut->ramp[i] = (uint8_t)ci;
    lut->glyph64[i] = packed[ci];
  }
  for (int i = 0; i < 256; i++) /* the kernels take a one-byte-per-glyph fast path when nothing is multi-byte */
    if ((lut->glyph[i] & 0xFFu) >= 128u || (lut->glyph64[i & 63] & 0xFFu) >= 128u)
      lut->flags |= ACHIP_LUT_MULTIBYTE;
  return 0;
}

int achip_mode_from_caps(int color_level, int render_mode) {
  if (render_mode == 2) { /* RENDER_MODE_HALF_BLOCK */
    switch (color_level) {
    case 3:
      return ACHIP_MODE_HB_TRUE;
    case 2:
      return ACHIP_MODE_HB_256;
    case 1:
      return ACHIP_MODE_HB_16;
    default:
      return ACHIP_MODE_HB_MONO;
    }
  }
  switch (color_level) {
  case 3:
    return render_mode == 1 ? ACHIP_MODE_16_DITHER_BG : ACHIP_MODE_TRUE_FG; /* BACKGROUND -> dithered (sgr.c:429) */
  case 2:
    return ACHIP_MODE_256_FG;
  case 1:
    return ACHIP_MODE_16_FG;
  default:
    return ACHIP_MODE_MONO;
  }
}

/* tint colours and modes of lib/video/rgba/color_filter.c:24-150 (index = color_filter_t) */
static const struct {
  uint8_t r, g, b, on_white;
} k_filters[12] = {{0, 0, 0, 0},     {0, 0, 0, 1},     {255, 255, 255, 0}, {0, 255, 65, 0},  {255, 0, 255, 0}, {255, 0, 170, 0},
                   {255, 136, 0, 0}, {0, 221, 221, 0}, {0, 255, 255, 0},   {255, 182, 193, 0}, {255, 51, 51, 0}, {255, 235, 153, 0}};

int achip_frame_set_display_ops(achip_frame_t *f, bool flip_x, bool flip_y, int color_filter) {
  if (!f || color_filter < 0 || color_filter >= 12)
    return -1;
  uint32_t ops = 0;
  if (f->src_w > 1 && f->src_h > 1) { /* display.c:549 */
    if (flip_x)
      ops |= ACHIP_OP_FLIP_X;
    if (flip_y)
      ops |= ACHIP_OP_FLIP_Y;
  }
  if (color_filter != 0) {
    ops |= ACHIP_OP_TINT | (k_filters[color_filter].on_white ? ACHIP_OP_TINT_ON_WHITE : 0u);
    ops |= ((uint32_t)k_filters[color_filter].r | ((uint32_t)k_filters[color_filter].g << 8) |
            ((uint32_t)k_filters[color_filter].b << 16))
           << ACHIP_OP_TINT_SHIFT;
  }
  f->ops = (f->ops & ACHIP_OP_DITHER_MASK) | ops;
  return 0;
}

/* color_filter.c:169-243: hue = 360 * (t mod 3.5 s) / 3.5 s walked through the six HSV sextants at S = V = 1, then
 * white is added until the BT.709 luminance reaches 120.  float throughout, like the reference. */
void achip_rainbow_color(float time_seconds, uint8_t *r, uint8_t *g, uint8_t *b) {
  const float period = 3.5f;
  const float phase = fmodf(time_seconds, period) / period;
  const float hue = phase * 360.0f;
  const float h = hue / 60.0f;
  const int sextant = (int)floorf(h);
  const float rise = h - (float)sextant, fall = 1.0f - rise;
  const uint8_t up = (uint8_t)(rise * 255.0f + 0.5f), down = (uint8_t)(fall * 255.0f + 0.5f);
  /* per sextant: which channel is full, which one moves (up or down), which one is zero */
  static const uint8_t k_full[6] = {0, 1, 1, 2, 2, 0}, k_move[6] = {1, 0, 2, 1, 0, 2}, k_rising[6] = {1, 0, 1, 0, 1, 0};
  uint8_t c[3] = {255, 0, 0}; /* negative times leave the switch through its default: red */
  const int s = sextant % 6;
  if (s >= 0) {
    c[0] = c[1] = c[2] = 0;
    c[k_full[s]] = 255;
    c[k_move[s]] = k_rising[s] ? up : down;
  }
  const float floor_lum = 120.0f;
  const float lum = 0.2126f * c[0] + 0.7152f * c[1] + 0.0722f * c[2];
  if (lum < floor_lum) {
    const float boost = (floor_lum - lum) / 3.0f;
    for (int k = 0; k < 3; k++)
      c[k] = (uint8_t)fminf(255.0f, c[k] + boost);
  }
  *r = c[0];
  *g = c[1];
  *b = c[2];
}

int achip_frame_set_rainbow(achip_frame_t *f, float time_seconds) {
  if (!f)
    return -1;
  uint8_t r, g, b;
  achip_rainbow_color(time_seconds, &r, &g, &b);
  f->ops &= ACHIP_OP_FLIP_X | ACHIP_OP_FLIP_Y | ACHIP_OP_DITHER_MASK;
  f->ops |= ACHIP_OP_FG_OVERRIDE | (((uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16)) << ACHIP_OP_TINT_SHIFT);
  return 0;
}

int achip_frames_uniform(const achip_frame_t *frames, int n, achip_uniform_t *u) {
  if (!u)
    return 0;
  memset(u, 0, sizeof(*u));
  /* (composite frames too, round 5: the grid's target clients share ONE descriptor -- no source of their own, the same
   * composite -- and a launch that carries it in its arguments spares every workgroup the dependent load of its entry) */
  if (!frames || n <= 0 || (!frames[0].src && !frames[0].comp) || (frames[0].src && frames[0].comp))
    return 0;
  const int64_t pitch = n > 1 ? (int64_t)((intptr_t)frames[1].src - (intptr_t)frames[0].src) : 0;
  for (int i = 1; i < n; i++) {
    achip_frame_t a = frames[i];
    if ((int64_t)((intptr_t)a.src - (intptr_t)frames[0].src) != pitch * i)
      return 0;
    a.src = frames[0].src;
    if (memcmp(&a, &frames[0], sizeof(a)) != 0) /* descriptors are memset by achip_frame_setup: padding is zero */
      return 0;
  }
  u->f = frames[0];
  u->src_pitch = pitch;
  u->enabled = 1;
  return 1;
}

int achip_frame_set_dither_style(achip_frame_t *f, bool use_background, bool ramp_glyph) {
  if (!f || (use_background && ramp_glyph)) /* the ramp glyph only exists in the foreground-only function */
    return -1;
  f->ops &= ~ACHIP_OP_DITHER_MASK;
  if (!use_background)
    f->ops |= ACHIP_OP_DITHER_FG | (ramp_glyph ? ACHIP_OP_DITHER_RAMP : 0u);
  return 0;
}

uint32_t achip_nn_ratio(int src, int dst) { return (uint32_t)((((uint64_t)src << 16) / (uint64_t)dst) + 1u); }

int achip_frame_identity(achip_frame_t *f, const uint8_t *src_dev, int w, int h) {
  if (!f || w <= 0 || h <= 0)
    return -1;
  memset(f, 0, sizeof(*f));
  f->src = src_dev;
  f->src_w = w;
  f->src_h = h;
  f->out_w = w;
  f->out_h = h;
  f->x_ratio = achip_nn_ratio(w, w);
  f->y_ratio = achip_nn_ratio(h, h);
  return 0;
}

int achip_frame_setup(achip_frame_t *f, const uint8_t *src_dev, int src_w, int src_h, ssize_t width, ssize_t height,
                      int render_mode, bool wants_padding, bool use_aspect, bool stretch) {
  if (!f)
    return -1;
  if (src_w <= 0 || src_w > 10000 || src_h <= 0 || src_h > 10000) /* ascii.c:204 */
    return -1;
  ssize_t rw = width, rh = height;
  if (use_aspect)
    aspect_ratio(src_w, src_h, rw, rh, stretch, &rw, &rh);
  const ssize_t text_w = rw, text_h = rh;
  if (render_mode == 2)
    rh *= 2; /* ascii.c:230-232 */
  ssize_t pad_w = 0, pad_h = 0;
  if (use_aspect && wants_padding) { /* ascii.c:238-244 */
    pad_w = width > text_w ? (width - text_w) / 2 : 0;
    pad_h = height > text_h ? (height - text_h) / 2 : 0;
  }
  /* image_new() of the resized image: image_validate_dimensions, lib/util/image.c:100-113 */
  if (rw <= 0 || rh <= 0 || rw > 3840 || rh > 2160)
    return -1;
  memset(f, 0, sizeof(*f));
  f->src = src_dev;
  f->src_w = src_w;
  f->src_h = src_h;
  f->out_w = (int32_t)rw;
  f->out_h = (int32_t)rh;
  f->pad_left = (int32_t)pad_w;
  f->pad_top = (int32_t)pad_h;
  f->x_ratio = achip_nn_ratio(src_w, (int)rw);
  f->y_ratio = achip_nn_ratio(src_h, (int)rh);
  return 0;
}

/* ---- staging of host images (drop-in layer): only what the sampler will read crosses PCIe ------------------------- */
static int g_stage_columns = -1; /* ASCIICHAT_HIP_STAGE_COLUMNS=0: sampled rows only (diagnostics / A-B) */

size_t achip_stage_extent(const achip_frame_t *f, int *w, int *h) {
  int columns = __atomic_load_n(&g_stage_columns, __ATOMIC_RELAXED);
  if (columns < 0) { /* idempotent: every thread that gets here stores the same value */
    const char *e = getenv("ASCIICHAT_HIP_STAGE_COLUMNS");
    columns = !(e && e[0] == '0');
    __atomic_store_n(&g_stage_columns, columns, __ATOMIC_RELAXED);
  }
  const int cy = f->out_h < f->src_h;
  const int cx = columns && f->out_w > 0 && f->out_w <= 3840 && 2 * (long)f->out_w <= (long)f->src_w;
  *w = cx ? f->out_w : f->src_w;
  *h = cy ? f->out_h : f->src_h;
  return cx || cy ? (size_t)*w * (size_t)*h * 3u : 0;
}

void achip_stage_gather(const achip_frame_t *f, const uint8_t *host_px, uint8_t *dst, achip_frame_t *d) {
  int w, h;
  (void)achip_stage_extent(f, &w, &h);
  const int cx = w != f->src_w, cy = h != f->src_h;
  const size_t stride = f->src_stride ? (size_t)f->src_stride : (size_t)f->src_w * 3u;
  const size_t row_bytes = (size_t)w * 3u;
  uint32_t off[3840]; /* a resized image is at most 3840 wide (achip_frame_setup; image.c:100-113) */
  if (cx) { /* byte offset of every sampled column, once (image.c:293-312; a horizontal flip is folded in) */
    for (int x = 0; x < w; x++) {
      uint32_t sx = (uint32_t)(((uint64_t)(uint32_t)x * f->x_ratio) >> 16);
      if (sx > (uint32_t)f->src_w - 1u)
        sx = (uint32_t)f->src_w - 1u;
      if (f->ops & ACHIP_OP_FLIP_X)
        sx = (uint32_t)f->src_w - 1u - sx;
      off[x] = sx * 3u;
    }
  }
  for (int y = 0; y < h; y++) {
    uint32_t sy = (uint32_t)y;
    if (cy) {
      sy = (uint32_t)(((uint64_t)(uint32_t)y * f->y_ratio) >> 16);
      if (sy > (uint32_t)f->src_h - 1u)
        sy = (uint32_t)f->src_h - 1u;
      if (f->ops & ACHIP_OP_FLIP_Y)
        sy = (uint32_t)f->src_h - 1u - sy;
    }
    const uint8_t *row = host_px + (size_t)sy * stride;
    uint8_t *o = dst + (size_t)y * row_bytes;
    if (!cx) {
      memcpy(o, row, row_bytes);
    } else {
      /* four bytes in, four bytes out per sample (the fourth is overwritten by the next sample): one load and one store
       * instead of three of each.  The load may reach one byte into the next source row, so the image's last row -- nothing
       * is known to lie behind it -- and every row's last sample (its fourth byte would land in the next row of dst, or behind
       * dst) go byte by byte. */
      int x = 0;
      if (sy + 1u < (uint32_t)f->src_h)
        for (; x + 1 < w; x++, o += 3) {
          uint32_t v;
          memcpy(&v, row + off[x], 4);
          memcpy(o, &v, 4);
        }
      for (; x < w; x++, o += 3) {
        const uint8_t *p = row + off[x];
        o[0] = p[0], o[1] = p[1], o[2] = p[2];
      }
    }
  }
  if (cx) {
    d->src_w = w;
    d->x_ratio = 1u << 16; /* sampled column x = column x of the compacted image */
    d->ops &= ~ACHIP_OP_FLIP_X;
  }
  if (cy) {
    d->src_h = h;
    d->y_ratio = 1u << 16;
    d->ops &= ~ACHIP_OP_FLIP_Y;
  }
  d->src_stride = (int32_t)row_bytes;
}

/* ---- ingest (frame_table.c): what a set of render targets reads of a source frame ---------------------------------- */
/* the source rows that the targets sample (the sampler's own rule: render_stream.hpp stream_request / render_kernels.hpp
 * sample_frame_raw -- sy = min((y * y_ratio) >> 16, src_h - 1), mirrored under ACHIP_OP_FLIP_Y), ascending, unique.
 * Targets are a tick's render descriptors: those set up for sources of another height belong to other clients and are
 * passed over.  Returns the number of rows, or -1 when no target describes an h-row source. */
int achip_sampled_rows(const achip_frame_t *targets, int n_targets, uint32_t h, uint32_t *rows_out, uint8_t *mark) {
  memset(mark, 0, h);
  int matched = 0;
  for (int i = 0; i < n_targets; i++) {
    const achip_frame_t *f = &targets[i];
    if (f->comp || (uint32_t)f->src_h != h || f->out_h <= 0)
      continue;
    matched++;
    for (uint32_t y = 0; y < (uint32_t)f->out_h; y++) {
      uint32_t sy = (uint32_t)(((uint64_t)y * f->y_ratio) >> 16);
      if (sy > h - 1u)
        sy = h - 1u;
      if (f->ops & ACHIP_OP_FLIP_Y)
        sy = h - 1u - sy;
      mark[sy] = 1;
    }
  }
  if (!matched)
    return -1;
  int n = 0;
  for (uint32_t r = 0; r < h; r++)
    if (mark[r])
      rows_out[n++] = r;
  return n;
}

/* ... and the source columns (sx = min((x * x_ratio) >> 16, src_w - 1), mirrored under ACHIP_OP_FLIP_X) of the targets that
 * describe h-row sources; -1 when one of those is not w columns wide (its columns on this frame are unknown: whole rows) */
static int sampled_cols(const achip_frame_t *targets, int n_targets, uint32_t w, uint32_t h, uint32_t *cols_out, uint8_t *mark) {
  memset(mark, 0, w);
  for (int i = 0; i < n_targets; i++) {
    const achip_frame_t *f = &targets[i];
    if (f->comp || (uint32_t)f->src_h != h || f->out_h <= 0)
      continue;
    if ((uint32_t)f->src_w != w || f->out_w <= 0)
      return -1;
    for (uint32_t x = 0; x < (uint32_t)f->out_w; x++) {
      uint32_t sx = (uint32_t)(((uint64_t)x * f->x_ratio) >> 16);
      if (sx > w - 1u)
        sx = w - 1u;
      if (f->ops & ACHIP_OP_FLIP_X)
        sx = w - 1u - sx;
      mark[sx] = 1;
    }
  }
  int n = 0;
  for (uint32_t c = 0; c < w; c++)
    if (mark[c])
      cols_out[n++] = c;
  return n;
}

/* what the targets read of a w x h frame: the sampled rows, and the sampled columns when they are at most half of the
 * frame's (n_cols = 0 otherwise: whole rows are staged -- a per-pixel gather would cost more than it saves).  A tick's
 * clients mostly share one geometry: the set is built once per distinct (w, h). */
void achip_sample_set_free(achip_sample_set_t *S) {
  free(S->rows);
  free(S->cols);
  memset(S, 0, sizeof(*S));
}
/* 0, -1 (a target does not describe this frame) or -2 (memory) */
int achip_sample_set_build(achip_sample_set_t *S, const achip_frame_t *targets, int n_targets, uint32_t w, uint32_t h) {
  memset(S, 0, sizeof(*S));
  S->w = w;
  S->h = h;
  S->rows = (uint32_t *)malloc((size_t)h * sizeof(uint32_t));
  S->cols = (uint32_t *)malloc((size_t)w * sizeof(uint32_t));
  uint8_t *mark = (uint8_t *)malloc(w > h ? w : h);
  int rc = 0;
  if (!S->rows || !S->cols || !mark) {
    rc = -2;
  } else {
    S->n_rows = achip_sampled_rows(targets, n_targets, h, S->rows, mark);
    if (S->n_rows < 0)
      rc = -1;
    else {
      const int nc = sampled_cols(targets, n_targets, w, h, S->cols, mark);
      S->n_cols = nc > 0 && 2u * (uint32_t)nc <= w ? nc : 0;
    }
  }
  free(mark);
  if (rc)
    achip_sample_set_free(S);
  return rc;
}
size_t achip_sample_set_block_bytes(const achip_sample_set_t *S) {
  const size_t tr = ((size_t)S->n_rows * 4u + 15u) & ~(size_t)15;
  if (!S->n_cols)
    return tr + (((size_t)S->n_rows * S->w * 3u + 15u) & ~(size_t)15);
  return tr + (((size_t)S->n_cols * 4u + 15u) & ~(size_t)15) + (((size_t)S->n_rows * (size_t)S->n_cols * 3u + 15u) & ~(size_t)15);
}
/* [row table][column table, if any][rows or pixels] of one frame into blk */
void achip_sample_set_pack(const achip_sample_set_t *S, const uint8_t *pixels, uint8_t *blk) {
  const size_t tr = ((size_t)S->n_rows * 4u + 15u) & ~(size_t)15, row_bytes = (size_t)S->w * 3u;
  memcpy(blk, S->rows, (size_t)S->n_rows * 4u);
  if (!S->n_cols) {
    for (int r = 0; r < S->n_rows; r++)
      memcpy(blk + tr + (size_t)r * row_bytes, pixels + (size_t)S->rows[r] * row_bytes, row_bytes);
    return;
  }
  const size_t tc = ((size_t)S->n_cols * 4u + 15u) & ~(size_t)15;
  memcpy(blk + tr, S->cols, (size_t)S->n_cols * 4u);
  uint8_t *o = blk + tr + tc;
  for (int r = 0; r < S->n_rows; r++) {
    const uint8_t *row = pixels + (size_t)S->rows[r] * row_bytes;
    for (int c = 0; c < S->n_cols; c++, o += 3) {
      const uint8_t *px = row + (size_t)S->cols[c] * 3u;
      o[0] = px[0], o[1] = px[1], o[2] = px[2];
    }
  }
}

size_t achip_out_bound(int mode, const achip_frame_t *f) {
  const bool hb = mode >= ACHIP_MODE_HB_TRUE && mode <= ACHIP_MODE_HB_MONO;
  const size_t rows = hb ? ((size_t)f->out_h + 1) / 2 : (size_t)f->out_h;
  size_t cell; /* worst-case bytes a single cell can own (a REP head replaces >= 5 suppressed cells) */
  switch (mode) {
  case ACHIP_MODE_MONO:
    cell = 4;
    break;
  case ACHIP_MODE_TRUE_FG:
    cell = 19 + 4;
    break;
  case ACHIP_MODE_256_FG:
    cell = 11 + 4;
    break;
  case ACHIP_MODE_16_FG:
    cell = 5 + 4;
    break;
  case ACHIP_MODE_TRUE_BG:
    cell = 19 + 19 + 4;
    break;
  case ACHIP_MODE_HB_TRUE:
    cell = 19 + 19 + 3;
    break;
  case ACHIP_MODE_HB_256:
    cell = 11 + 11 + 3;
    break;
  case ACHIP_MODE_HB_16:
    cell = 5 + 6 + 3;
    break;
  case ACHIP_MODE_16_DITHER_BG:
    cell = 6 + 5 + 4;
    break;
  default:
    cell = 3;
    break;
  }
  /* per row: padding + cells + one REP tail (ESC [ dddd b) + reset + newline; per frame: pad_top + final reset */
  const size_t row = (size_t)f->pad_left + (size_t)f->out_w * cell + 7 + 4 + 4 + 1;
  return (size_t)f->pad_top + rows * row + 8;
}

bool achip_palette_ascii_only(const char *palette_chars) {
  if (!palette_chars)
    return false;
  for (const unsigned char *p = (const unsigned char *)palette_chars; *p; p++)
    if (*p >= 0x80)
      return false;
  return true;
}

/* the stream-kernel family (render_variants.h: ACHIP_STREAM_VARIANTS; render_stream.hpp: ACHIP_STREAM_MAXBLK) */
#define ACHIP_HOST_STREAM_FIRST 16
#define ACHIP_HOST_STREAM_MAXBLK 2048

/* The kernels address a source with 32-bit byte offsets (a frame is at most tens of MB): a descriptor whose last pixel
 * lies 4 GiB or more behind its first -- only possible with an absurd explicit row stride -- is refused on the host. */
bool achip_frame_extent_ok(const achip_frame_t *f) {
  if (!f || f->comp)
    return true; /* composites carry their own (tile-sized) sources */
  const uint64_t stride = f->src_stride > 0 ? (uint64_t)f->src_stride : 3ull * (uint64_t)(f->src_w > 0 ? f->src_w : 0);
  if (f->src_stride < 0)
    return false;
  const uint64_t extent = (uint64_t)(f->src_h > 0 ? f->src_h - 1 : 0) * stride + 3ull * (uint64_t)(f->src_w > 0 ? f->src_w : 0);
  return extent < 0xFFFFFFF0ull && stride < (1ull << 24);
}

/* cells ((pad_left + out_w) * out_h) of the largest frame: what ACHIP_UNIFORM_MAX_CELLS carries to the stream kernel */
long achip_max_cells(const achip_frame_t *frames, int n_frames) {
  long max_cells = 0;
  for (int i = 0; i < n_frames; i++) {
    const long c = (long)(frames[i].pad_left + frames[i].out_w) * (long)frames[i].out_h;
    if (c > max_cells)
      max_cells = c;
  }
  return max_cells;
}

/* the rows-kernel family (render_variants.h: ACHIP_ROWS_VARIANTS): a block is a whole number of text rows */
#define ACHIP_HOST_ROWS_FIRST 24
#ifndef ACHIP_ROWS_WIDE_CPL
#define ACHIP_ROWS_WIDE_CPL 5 /* (render_variants.h) */
#endif
#ifndef ACHIP_ROWS_PARTS_CPL
#define ACHIP_ROWS_PARTS_CPL 2 /* (render_variants.h) */
#endif
static int rows_variant_cpl(int variant) {
  return variant == 24 || variant == 26 ? 7 : variant == 25 ? 4 : variant == 27 || variant == 29 ? ACHIP_ROWS_WIDE_CPL
         : variant == 31 ? ACHIP_ROWS_PARTS_CPL : variant == 32 || variant == 28 ? 2 : variant == 30 || variant == 33 || variant == 34 ? 1 : 0;
}
/* the geometries that share a frame's blocks out over workgroups (render_variants.h ACHIP_ROWS_VARIANT_PARTS): fast sampler only */
static bool rows_variant_parts(int variant) { return variant == 31 || variant == 32 || variant == 33 || variant == 34; }
/* the geometries whose blocks are SEGMENTS of a row (render_variants.h ACHIP_ROWS_VARIANT_WIDE; render_rows.hpp WIDE): rows of
 * at most `waves` segments of 64 * cpl cells, and of at most ACHIP_ROWS_WIDE_MAX_ROW cells */
#define ACHIP_HOST_ROWS_WIDE_MAX_ROW 4096
static int rows_variant_wide_waves(int variant) { return variant == 27 ? 16 : variant == 29 ? 8 : variant == 30 || variant == 32 ? 4 : variant == 34 ? 2 : 0; }
/* segments of a padded row of wp cells (the kernel's own arithmetic: equal widths, the last one shorter, never empty) */
static long rows_wide_segments(long wp, int cpl) {
  if (wp <= 0)
    return 0;
  const long n0 = (wp + 64L * cpl - 1) / (64L * cpl), segw = (wp + n0 - 1) / n0;
  return (wp + segw - 1) / segw;
}
/* the widest padded row a rows geometry takes */
static int rows_variant_max_row(int variant) {
  const int cpl = rows_variant_cpl(variant), waves = rows_variant_wide_waves(variant);
  if (!waves)
    return 64 * cpl;
  return 64 * cpl * waves < ACHIP_HOST_ROWS_WIDE_MAX_ROW ? 64 * cpl * waves : ACHIP_HOST_ROWS_WIDE_MAX_ROW;
}

/* ASCIICHAT_HIP_ROWS_PARTS (diagnostics, read once): 1 = the run-structured modes' small launches are never shared out over
 * workgroups of the rows kernel, N = over this many where the CUs allow; ASCIICHAT_HIP_ROWS_PARTS_WIDE=0: not the rows of
 * 129-512 cells (geometry 32), which keep their row bands then; =2: geometry 32 for the 256- / 16-colour half blocks too */
/* (read once per process; relaxed atomics: every thread that finds -1 computes the same value) */
static int env_int_once(_Atomic int *cache, const char *name) {
  int v = atomic_load_explicit(cache, memory_order_relaxed);
  if (v < 0) {
    const char *e = getenv(name);
    v = e && e[0] ? atoi(e) : 0;
    if (v < 0)
      v = 0;
    atomic_store_explicit(cache, v, memory_order_relaxed);
  }
  return v;
}
static int rows_parts_forced(void) {
  static _Atomic int forced = -1;
  return env_int_once(&forced, "ASCIICHAT_HIP_ROWS_PARTS");
}
static int rows_parts_wide_level(void) { /* 0: never; 1 (default): the modes it was measured ahead for; 2 (diagnostics): all five */
  static _Atomic int on = -1;
  int v = atomic_load_explicit(&on, memory_order_relaxed);
  if (v < 0) {
    const char *e = getenv("ASCIICHAT_HIP_ROWS_PARTS_WIDE");
    v = e && e[0] == '0' ? 0 : e && e[0] == '2' ? 2 : 1;
    atomic_store_explicit(&on, v, memory_order_relaxed);
  }
  return v;
}

/* what the ACHIP_UNIFORM_MAX_CELLS field of a launch carries: cells of the largest frame for the stream geometries,
 * BLOCKS of the frame with the most blocks for the rows geometries (rows / (64 * CPL / row width), rounded up; the
 * geometries that cut rows into segments: rows x segments) */
long achip_uniform_extent(int mode, int variant, const achip_frame_t *frames, int n_frames) {
  const int cpl = rows_variant_cpl(variant);
  if (!cpl)
    return achip_max_cells(frames, n_frames);
  const bool hb = mode >= ACHIP_MODE_HB_TRUE && mode <= ACHIP_MODE_HB_MONO;
  long most = 0;
  for (int i = 0; i < n_frames; i++) {
    const long wp = (long)frames[i].pad_left + frames[i].out_w;
    const long rows = hb ? ((long)frames[i].out_h + 1) / 2 : frames[i].out_h;
    const long rpb = wp > 0 ? (64L * cpl) / wp : 0;
    const long blocks = rows_variant_wide_waves(variant) ? rows * rows_wide_segments(wp, cpl) : rpb > 0 ? (rows + rpb - 1) / rpb : 0;
    if (blocks > most)
      most = blocks;
  }
  return most;
}

int achip_choose_geometry(int mode, const achip_frame_t *frames, int n_frames, bool palette_ascii_only,
                          const int *variant_caps, int n_cus, int split_request, int forced_variant, int *variant,
                          int *parts, int *rows_per_part) {
  const bool hb = mode >= ACHIP_MODE_HB_TRUE && mode <= ACHIP_MODE_HB_MONO;
  int max_wp = 0, max_rows = 0;
  for (int i = 0; i < n_frames; i++) {
    const int wp = frames[i].pad_left + frames[i].out_w;
    const int rows = hb ? (frames[i].out_h + 1) / 2 : frames[i].out_h;
    if (wp > max_wp)
      max_wp = wp;
    if (rows > max_rows)
      max_rows = rows;
  }
  *parts = 1;
  *rows_per_part = max_rows > 0 ? max_rows : 1;
  if (n_cus < 1)
    n_cus = 256;
  /* Whole-frame geometries (measured on MI355X, profiles/r01_split_sweep.txt, r01_overlap.txt): up to ~1.5
   * workgroups per CU the 1024-thread x 2-cell geometry (4) has the shortest per-frame latency chain; with more
   * frames than that in flight -- one large launch, or several launches kept in flight on separate streams, in which
   * case the caller passes its share of the CUs -- two or three 512-thread workgroups per CU overlap each other's
   * gather (HBM-bound) and token (latency-bound) phases instead.  Not the half-block modes: their kernels need more
   * than 128 VGPRs in the small geometries and lose more than the overlap gains. */
  /* The per-cell renderers (truecolor / 256 / 16 foreground, truecolor background; truecolor-fg with an all-ASCII
   * palette) have no run structure: whole-frame launches of them take the wave-autonomous stream kernel
   * (render_stream.hpp), whose waves gather, tokenise and drain independently -- no chunk, so no row-width limit,
   * only a bound on the cells of a frame.  ACHIP_STREAM_* below restate render_variants.h / render_stream.hpp. */
  const long max_cells = achip_max_cells(frames, n_frames);
  int max_src_w = 0;
  bool general_sampler = false, dense = true; /* composites / 1x1 sources; every source IS the image its target samples */
  for (int i = 0; i < n_frames; i++) {
    if (!frames[i].comp && frames[i].src_w > max_src_w)
      max_src_w = frames[i].src_w;
    general_sampler |= frames[i].comp != NULL || (long)frames[i].src_w * (long)frames[i].src_h == 1;
    dense &= !frames[i].comp && frames[i].src_w == frames[i].out_w && frames[i].src_h == frames[i].out_h;
  }
  /* truecolor foreground with a palette that holds multi-byte glyphs (round 6: the stream kernel's instantiation of its own,
   * render_stream.hpp ACHIP_STREAM_MODE_TRUE_FG_U8): whole frames of single sources in geometries 16 / 17, never shared out
   * over workgroups (its RLE state crosses blocks through LDS words); composites and 1x1 sources stay with the phase kernel */
  const bool u8_true = mode == ACHIP_MODE_TRUE_FG && !palette_ascii_only;
  const bool cell_mode = mode == ACHIP_MODE_256_FG || mode == ACHIP_MODE_16_FG || mode == ACHIP_MODE_TRUE_BG ||
                         (mode == ACHIP_MODE_TRUE_FG && (palette_ascii_only || !general_sampler));
  const bool stream_forced = forced_variant >= ACHIP_HOST_STREAM_FIRST && forced_variant < ACHIP_HOST_ROWS_FIRST;
  /* cells a block owns: 64 per lane slot, minus the ghost slot of truecolor-fg (render_stream.hpp: SLds::EFF) */
  const int ghost = mode == ACHIP_MODE_TRUE_FG ? 1 : 0;
  if (stream_forced) {
    const int cpl = forced_variant == 19 || forced_variant == 20 ? 1 : 2;
    if (!cell_mode || forced_variant > 20 || max_cells > (long)ACHIP_HOST_STREAM_MAXBLK * (64 * cpl - ghost))
      return -1;
    if (u8_true && forced_variant != 16 && forced_variant != 17 && forced_variant != 20)
      return -1;
    *variant = forced_variant;
    return 0; /* whole frames only */
  }
  /* The run-structured renderers (mono, half blocks) start a run at every row's first cell, so a whole number of text
   * rows is a self-contained block that ONE wave can take through the path (render_rows.hpp): whole-frame launches of
   * them take that kernel whenever the widest padded row fits a block (64 * CPL cells). */
  /* (... and whose sources are at most 21 845 pixels wide: render_rows.hpp keeps a sample's byte offset in 16 bits) */
  const bool run_mode = (mode == ACHIP_MODE_MONO || hb) && max_src_w <= 21845;
  if (forced_variant >= ACHIP_HOST_ROWS_FIRST) {
    const int cpl = rows_variant_cpl(forced_variant);
    if (!run_mode || !cpl || max_wp > rows_variant_max_row(forced_variant) || achip_uniform_extent(mode, forced_variant, frames, n_frames) > ACHIP_HOST_STREAM_MAXBLK)
      return -1;
    if ((forced_variant == 26 || rows_variant_wide_waves(forced_variant) || rows_variant_parts(forced_variant)) && general_sampler) /* (geometry 26, the segment geometries and the shared-out one carry the fast sampler only) */
      return -1;
    *variant = forced_variant;
    return 0; /* whole frames only */
  }
  /* automatic: whenever whole frames are launched anyway (the split policy below decides that first) */
  /* automatic splitting only when whole frames would leave CUs idle: below 3/4 frame per CU -- the half-block modes below
   * half a frame per CU (round 4 audit: 128 frames of 120x40, one launch at a time: mono half blocks 23.9 us in three
   * bands each against 19.8 whole, 256 colours 29.3 against 26.8; 64 frames: 14.0 against 28.4 the other way) */
  const bool whole_auto = split_request == 0 && (4 * n_frames >= 3 * n_cus || (hb && 2 * n_frames >= n_cus));
  const bool may_split = mode != ACHIP_MODE_16_DITHER_BG && !(mode == ACHIP_MODE_TRUE_FG && !palette_ascii_only) &&
                         split_request >= 0 && max_rows > 1 && !whole_auto;
  /* frames small enough for ONE block per wave of a wave-autonomous geometry are never cut into row bands, however few
   * they are: a lone 80x24 frame takes 6.3 us through stream geometry 16 against 8.4 us as 24 bands of the phase kernel
   * whose workgroups hand their offsets to each other through global memory, a lone mono frame 7.0 against 8.1 us through
   * rows geometry 25 (profiles/r04_small_batch_variants.txt); the nine 160x48 targets of the grid -- four blocks per
   * wave -- stay with the bands (9.6 against 20 us) */
  /* Small launches of the per-cell modes -- a lone frame, the nine targets of a grid -- share a frame's blocks out over
   * four-wave workgroups (render_stream.hpp PARTS, geometry 18: one wave per SIMD, one block per wave where the CUs
   * allow), as long as every workgroup of the launch has a CU to itself (they hand their byte counts to each other
   * through memory, so all of them must be resident): one workgroup per frame queues all of the frame's waves on the
   * four SIMDs of ONE CU (profiles/r04_lone_frame_timeline.txt). */
  if (forced_variant < 0 && cell_mode && !u8_true && split_request == 0 && max_cells <= (long)ACHIP_HOST_STREAM_MAXBLK * (128 - ghost)) {
    const long nblk = (max_cells + (128 - ghost) - 1) / (128 - ghost);
    /* four blocks per workgroup (a wave each), but up to sixteen workgroups for a small frame (a lone 80x24 frame: 5.8 us
     * in 16 parts of one block, 5.9 in four; more than that loses again: 4K -> 200x60 in 64 parts 7.2 us against 6.9 in 24,
     * the grid's nine targets 8.6 in 28 against 8.0 in 16 -- the first block of a part polls every part in front of it) */
    long np = (nblk + 3) / 4;
    if (np < 16)
      np = nblk < 16 ? nblk : 16;
    if (np > 64)
      np = 64;
    if (np * n_frames > n_cus)
      np = n_cus / n_frames;
    { /* ASCIICHAT_HIP_STREAM_PARTS (diagnostics, read once): 1 = never, N = this many where the CUs allow */
      static _Atomic int forced_cache = -1;
      const int forced = env_int_once(&forced_cache, "ASCIICHAT_HIP_STREAM_PARTS");
      if (forced >= 1 && forced <= 64 && (long)forced * n_frames <= n_cus)
        np = forced;
    }
    /* only while every wave has at most ONE block: a workgroup can publish its byte count only when its LAST block is
     * scanned, so with a second round per wave every part waits for the part in front of it to finish its first -- the
     * parts run one after the other (sixteen 200x60 frames in 16 parts of six blocks: 53 us against 9.9 as row bands; 128
     * 80x24 frames in two parts each: 11.7 against 7.0 whole; profiles/r04_small_batch_parts.txt) */
    if (np >= 2 && (nblk + np - 1) / np <= 4) {
      *variant = 18;
      *parts = (int)np;
      *rows_per_part = 1;
      return 0;
    }
  }
  const bool one_block_per_wave = cell_mode && split_request == 0 && max_cells <= 16 * (128 - ghost);
  /* (round 4 audit: from half a frame per CU on, frames of at most three blocks per wave of the sixteen-wave geometry go
   * whole too -- 128 frames of 120x40, one launch at a time: 12.9 us whole against 16.8 as three bands each) */
  /* (round 6, after the stream kernel's lean loop -- scripts/gpu_truecolor_half_cu.py, profiles/r06_whole_from_three_eighths.txt,
   * one launch at a time, 1080p sources: the row bands' phase kernel is now the slower one per cell, so whole frames win well below
   * 3/4 frame per CU.  Truecolor foreground from single sources (the lean loop), any size, from 3/8 frame per CU on: 96 frames of
   * 120x40 9.7 us against 16.0 as bands, 160x45 12.7 / 16.1, 200x60 19.9 / 25.7, 238x70 25.9 / 29.7, 320x90 43.1 / 43.7 -- and
   * 160 frames 13.7 / 26.3, 20.8 / 31.7, 26.9 / 43.5, 44.0 / 69.1; at 80 frames the bands are still ahead from 200 columns on
   * (16.8 / 19.8).  The other per-cell modes, frames of at most four blocks per wave (160x45), from 3/8 too: 256 colours 96 frames
   * of 120x40 10.1 / 13.4, 160x45 13.0 / 13.6, 128 frames 13.3 / 14.5; larger frames keep their bands up to 3/4 (200x60, 128
   * frames: 22.6 whole against 18.8)) */
  const bool lean_true = mode == ACHIP_MODE_TRUE_FG && palette_ascii_only && !general_sampler;
  /* (... and truecolor frames of at most four blocks per wave already from 5/16: 80 frames of 120x40 9.7 against 11.0, 160x45
   * 12.6 against 15.9; at 64 frames 160x45 is behind its bands, 12.6 against 11.6) */
  const bool four_blocks = max_cells <= 4 * 16 * (128 - ghost);
  const bool few_blocks = cell_mode && split_request == 0 &&
                          ((8 * n_frames >= 3 * n_cus && (lean_true || four_blocks)) || (16 * n_frames >= 5 * n_cus && lean_true && four_blocks));
  if (forced_variant < 0 && cell_mode && (!may_split || max_wp > variant_caps[0] || one_block_per_wave || few_blocks) &&
      max_cells <= (long)ACHIP_HOST_STREAM_MAXBLK * (128 - ghost)) {
    /* measured (profiles/r02_stream_sweep.txt): 1024 threads x 2 cells -- one block per wave for a 1080p -> 80x24
     * frame -- is the shortest single launch while every frame has a CU to itself; with more frames than CUs in
     * flight (a large batch, or several launches kept in flight: the caller passes its share of the CUs), or frames
     * of several blocks per wave, 512-thread workgroups pack better (four launches in flight, us per step, 16 vs 17: 1080p -> 80x24 truecolor
     * 10.9 vs 8.2, ANSI-256 7.0 vs 6.6, 4K -> 200x60 equal within noise) */
    /* round 4 audit (scripts/gpu_policy_audit.py, profiles/r04_policy_audit.txt: six terminal sizes x five batch sizes x
     * four modes, every geometry forced in turn): the rule is the FRAMES PER CU, not the blocks per wave -- with at most a
     * frame per CU of the plan's share a CU holds ONE workgroup either way, and sixteen waves fill it better than eight
     * however many blocks each walks (256 frames, one launch at a time: 120x40 truecolor 17.3 vs 22.3 us, 200x60 34.1 vs
     * 43.4, 320x90 71.7 vs 94.2; 64 frames with four launches in flight: 20.9 vs 26.1 at 200x60); above that two
     * 512-thread workgroups share a CU and win or tie (256 frames, four in flight: 35.7 vs 36.9 at 200x60). */
    /* (up to TWO frames per CU: 128 frames at a share of 64 CUs, 200x60: 23.5 vs 29.6 us; at three they tie, at four
     * the 512-thread workgroups win by 3-10 %) */
    /* (round 6's last audit, AUDIT_BATCHES=320..2048 of scripts/gpu_policy_audit.py, profiles/r06_whole_from_three_eighths.txt: with
     * the GPU to itself, between one and two frames per CU, frames of more than one block per wave pack better as 512-thread
     * workgroups -- 512 frames of 120x40 21.2-22.9 us against 28.1-29.2, 160x45 26.6-28.4 against 34.3-35.1; 320 / 384 / 512 frames of 200x60 truecolor 36.3 / 37.5 / 40.8 us against 41.3 / 42.3 / 49.9, 320x90 80.1-86.4 against
     * 87.0-96.7, 640x90 150.8-162.2 against 168.3-182.5, 256 colours 13-22 %; 80x24 frames level; at a share of 64 CUs two frames per
     * CU stay level between the two) */
    const bool big_alone = n_cus > 128 && n_frames > n_cus && max_cells > 16L * (128 - ghost);
    *variant = n_frames <= 2 * n_cus && !big_alone ? 16 : 17;
    return 0;
  }
  /* (the coloured half-block modes only from a frame per four CUs on: their tokens are long -- two SGRs and a three-byte glyph
   * per cell -- and a rows-kernel block is ONE wave taking its cells through the path four at a time, so with few frames the
   * row bands of the phase kernel, a thread per cell, are shorter: a lone 80x24 half-block truecolor frame 9.0 us as bands
   * against 12.6 whole, eight of them 9.7 against 12.9, sixty-four in four bands each 10.2 against 13.3, ninety-six 17.9 against
   * 13.4; profiles/r04_small_run_modes.txt) */
  const bool short_tokens = mode == ACHIP_MODE_MONO || mode == ACHIP_MODE_HB_MONO;
  /* Small launches of the run-structured modes (round 6) -- a lone mono frame, a few half-block frames --
   * share a frame's blocks out over four-wave workgroups of the rows kernel (render_rows.hpp PARTS, geometry 31: 128-cell
   * blocks -- ONE text row at 80 or 120 columns --, one block per wave) as the per-cell modes do above, while every workgroup
   * of the launch has a CU to itself.  Measured (scripts/gpu_r6_k.sh, profiles/r06_small_rows_parts.txt; one launch after the
   * other, us per launch): a lone 80x24 mono frame 6.5 against 7.6 whole on geometry 25 and 7.5-8.2 as row bands; a lone
   * half-block truecolor frame 8.3 against 9.0-9.4 as bands (11.6 whole); eight frames 7.4 / 9.2 against 8.0 / 9.7.  With
   * 256-cell blocks (three rows each, two workgroups) nothing is gained: the shorter block shortens the chain, not the idle
   * SIMDs. */
  /* (rows of 129-256 cells as ONE row per block of four cell slots were measured too -- scripts/gpu_r6_n.sh: a lone 160x48 mono
   * frame 8.2 us against the row bands' 8.1-8.3, half-block truecolor 11.1 against 9.8-10.0, 200x60 12.6 against 10.1 -- and
   * lost: the block is what a wave walks alone, and 160 cells are twice 80.  Such rows are cut into segments instead: below) */
  if (forced_variant < 0 && run_mode && !general_sampler && may_split && split_request == 0 && max_wp <= 64 * ACHIP_ROWS_PARTS_CPL) {
    const long nblk = achip_uniform_extent(mode, 31, frames, n_frames);
    long np = (nblk + 3) / 4;
    if (np > 64)
      np = 64;
    if (np * n_frames > n_cus)
      np = n_cus / n_frames;
    {
      const int forced = rows_parts_forced();
      if (forced >= 1 && forced <= 64 && (long)forced * n_frames <= n_cus)
        np = forced;
    }
    if (np >= 2 && nblk > 0 && (nblk + np - 1) / np <= 4) {
      *variant = 31;
      *parts = (int)np;
      *rows_per_part = 1;
      return 0;
    }
  }
  /* ... and rows of 129-512 cells cut into at most four segments of at most 128 cells, WHOLE rows per four-wave workgroup
   * (geometry 32 = render_rows.hpp WIDE + PARTS: the segments of a row talk through LDS words, so a row never leaves its
   * workgroup; the workgroups hand their byte counts on as above), a segment per wave: two rows per workgroup at two
   * segments a row (160x48: 24 workgroups for a mono frame), one at three or four.  Only while every wave has ONE block.
   * Measured against the row bands, interleaved (scripts/gpu_r6_q.sh, profiles/r06_small_rows_parts.txt visit Q; 1 / 4 / 10
   * frames of 160x48, 200x60, 256x30, 300x40, 400x30, us per launch): mono 8.0-8.2 against 8.5 (160x48), 7.3-7.6 against
   * 8.3-8.5 (256x30), 3-11 % ahead everywhere; mono half blocks 2-10 % ahead; truecolor half blocks level to 6 % ahead (9.8
   * against 9.9, 9.1 against 9.7 at 256x30; never more than 0.5 % behind); the 256- / 16-colour half blocks 6-20 % BEHIND
   * (byte-built tokens of one wave against a thread per cell: 9.7 against 9.0, 11.3 against 9.4) -- those keep their bands. */
  const bool parts_wide_mode = short_tokens || mode == ACHIP_MODE_HB_TRUE || rows_parts_wide_level() == 2;
  if (forced_variant < 0 && run_mode && parts_wide_mode && !general_sampler && may_split && split_request == 0 && rows_parts_wide_level() != 0 &&
      max_wp > 64 * ACHIP_ROWS_PARTS_CPL && max_wp <= rows_variant_max_row(32)) {
    const long nseg = rows_wide_segments(max_wp, rows_variant_cpl(32));
    const long rpw = nseg > 0 ? 4 / nseg : 0; /* rows of a workgroup */
    long np = rpw > 0 ? (max_rows + rpw - 1) / rpw : 0;
    if (np > 64)
      np = 64;
    if (np * n_frames > n_cus)
      np = n_cus / n_frames;
    {
      const int forced = rows_parts_forced();
      if (forced >= 1 && forced <= 64 && (long)forced * n_frames <= n_cus)
        np = forced;
    }
    if (np >= 2 && ((max_rows + np - 1) / np) * nseg <= 4 && achip_uniform_extent(mode, 32, frames, n_frames) <= ACHIP_HOST_STREAM_MAXBLK) {
      *variant = 32;
      *parts = (int)np;
      *rows_per_part = 1;
      return 0;
    }
  }
  if (forced_variant < 0 && run_mode && may_split && split_request == 0 && max_wp <= 256 &&
      (short_tokens || 4 * n_frames > n_cus) && achip_uniform_extent(mode, 25, frames, n_frames) <= 8) {
    *variant = 25; /* one block of whole rows per wave of its eight: see above */
    return 0;
  }
  if (forced_variant < 0 && run_mode && !may_split && max_wp <= 64 * 7) {
    /* the geometry whose blocks waste fewer lane slots on this row width; the smaller one on a tie.  Measured
     * (profiles/r03_rows_kernel.txt, 256 frames per launch): 1080p -> 80x24 half blocks 15.7 us per step with four
     * launches in flight / 22.6 one at a time (phase kernel: 22.6 / 24.1); 4K -> 400x120 half blocks 212 us with four in
     * flight (phase: 244) but 308 one at a time (phase: 256) -- its 7-slot blocks run at eight waves per CU, so ONE launch
     * of at most a frame per CU stays with the phase kernel's sixteen. */
    const int used4 = max_wp <= 256 ? (256 / max_wp) * max_wp : 0, used7 = (448 / max_wp) * max_wp;
    int v = used4 * 448 >= used7 * 256 ? 25 : 24;
    /* (round 4, the audit with launches in flight through bench.py's schedule: above two frames per CU of the share the
     * four-slot geometry wins whatever the slots it wastes -- its workgroups need fewer registers and more of them share a
     * CU: 256 mono frames of 160x45 at a share of 64 CUs 14.3 against 17.3 us, 200x60 20.7 against 24.7, half-block
     * truecolor 38.4 against 40.8; at two frames per CU the seven-slot one is still ahead, 9.0 against 10.6) */
    if (short_tokens && max_wp <= 256 && n_frames > 2 * n_cus) /* (mono and mono half blocks, 200x60: 29.0 against 35.3; the coloured
                                                                   half-block modes differ by -14 .. +5 % and keep the slot rule) */
      v = 25;
    /* (round 5, word-built SGRs: truecolor half blocks from full-frame sources on a shared GPU as well -- 256 frames at a
     * share of 64 CUs, 1080p -> 160x45 35.8 against 39.0 us, 200x60 53.8 against 57.5, 4K -> 200x60 76.4 against 79.7; from
     * dense sources the seven-slot geometry keeps its 0-7 %: profiles/r05_policy_audit_hb.txt) */
    if (mode == ACHIP_MODE_HB_TRUE && !dense && n_cus <= 128 && max_wp <= 256 && n_frames > n_cus)
      v = 25;
    /* (round 4 audit: the coloured half-block modes at no more than a frame per CU only while a wave has ONE block --
     * 256 frames of 120x40, one launch at a time: 42.9 us against the phase kernel's 37.5, 238x70 104 against 95; 80x24,
     * four blocks a frame, 21.2 against 23.1.  Mono keeps the rows kernel: 238x70 47 against 52.) */
    const int ext = achip_uniform_extent(mode, v, frames, n_frames);
    /* (... the half-block mono mode as well: 256 frames of 238x70 64 against 53; above a frame per CU the rows kernel
     * wins for all of them -- 4K -> 400x120, 128 frames at a share of 64 CUs, bench.py's schedule: 108 against 118;
     * mono rows wider than the four-slot geometry take the seven-slot one at any count: 256 frames of 320x90 79 against 88) */
    const bool mono = mode == ACHIP_MODE_MONO;
    /* (... and on a GPU that is SHARED -- the caller keeps launches in flight and passed a share of at most half the CUs --
     * mono frames of several blocks per wave follow the half-block rule too: 64 frames of 238x70 at a share of 64 CUs 13.8 us
     * on the phase kernel against 17.3, 320x90 22.6 against 27.1; with the GPU to itself the rows kernel stays ahead, above) */
    /* (the half-block modes' frames of one block per wave as well: 64 frames of 80x24 at a share of 64 CUs 5.2-5.9 us on the
     * phase kernel against 6.2-6.9, all four of them) */
    const bool shared_gpu = n_cus <= 128;
    /* (round 5: ... or the seven-slot geometry as ONE sixteen-wave workgroup per frame (26), which fills a CU the way two
     * eight-wave workgroups of two launches do.  First measured for dense sources and rows wider than the four-slot
     * geometry (256 frames of 400x240 -> 400x120 truecolor half blocks, one launch at a time: 195.9 us against the phase
     * kernel's 241.8 and 258.8 on geometry 24; profiles/r05_rows_sixteen_waves_ab.txt); with the truecolor SGRs built as
     * words (render_kernels.hpp word_sgr) it is ahead of the phase kernel for every launch of at most a frame per CU of
     * the share whose frames are big enough to give each of the sixteen waves a block -- profiles/r05_policy_audit_hb.txt,
     * 256 / 128 frames one launch at a time, 64 with four launches in flight:
     *   truecolor, dense sources   120x40 25.3 against 31.2 us, 200x60 46.6 / 64.4, 320x90 (in flight) 26.9 / 37.5
     *   truecolor, 1080p sources   160x45 44.2 / 49.9, 200x60 63.1 / 69.9, 320x90 124.7 / 154.6; 120x40 level
     *   truecolor, 4K sources      320x90 and 400x120 up to three quarters of a frame per CU (400x120, 192 frames: 219
     *                              against 245; 256 frames: 274 against 255 -- the gather, HBM-bound, is what sixteen
     *                              independent waves do worse than the phase kernel's thousand threads) and in flight
     *   256 / 16 colours           (byte-built tokens) rows beyond the four-slot geometry, and from 160 columns on when the
     *                              sources are dense or the launch is at most three quarters of a frame per CU (7-26 %; a
     *                              full frame per CU from 1080p sources: -6 .. +6 %, left with the phase kernel) */
    /* (... and the short-token modes, profiles/r05_policy_audit_mono.txt -- mono, whole frames (from three quarters of a
     * frame per CU on; below that its row bands win): 256 frames one launch at a time, 1080p -> 200x60 32.2 against 38.1 us,
     * 320x90 59.5 against 81.9, dense 160x45 18.8 against 25.0; 64 frames at a share of 64 CUs: 200x60 7.9 against 10.0,
     * 320x90 15.6 against 22.6; mono half blocks from dense sources 7-28 %, from full frames only beyond the four-slot
     * geometry: 320x90 76.7 against 86.6, 200x60 48.0 against 41.5 the other way) */
    if (!general_sampler && n_frames <= n_cus && achip_uniform_extent(mode, 26, frames, n_frames) <= ACHIP_HOST_STREAM_MAXBLK) {
      const bool big_src = max_src_w > 1920;
      bool take26;
      /* (round 6, the audit again with this round's forms among the automatic choices, profiles/r06_policy_audit_run_modes.txt:
       * from sources up to 1080p already from 120 columns -- 192 mono frames of 120x40 17.8 against 19.7 us, 64 at a share of
       * 64 CUs 4.6 against 5.3, truecolor half blocks 128 frames 28.1 against 30.2; 256 colours from 200 columns at a full
       * frame per CU too -- 238x70 70.5 against 81.6; mono half blocks from 220 columns -- 128 frames of 238x70 45.2 against 51.2) */
      if (mode == ACHIP_MODE_HB_TRUE || mode == ACHIP_MODE_MONO)
        /* (mono from 4K sources at a full frame per CU too -- round 6's last audit, 256 frames one launch at a time: 320x90 69.2 against
         * 90.7 us on geometry 24 and 87.2 on the phase kernel, 400x120 104.2 against 145.4 / 137.1; truecolor half blocks there are level:
         * BASELINE configs[4] 242-246 against the phase kernel's 248) */
        take26 = dense ? max_wp >= 120 : !big_src ? max_wp >= 120
                 : (max_wp > 256 && (shared_gpu || 4 * n_frames <= 3 * n_cus || mode == ACHIP_MODE_MONO));
      else if (mode == ACHIP_MODE_HB_MONO) /* (rows beyond the four-slot geometry on a shared GPU too -- round 6's last audit: 64 frames of
                                             * 320x90 at a share of 64 CUs 17.6 us against the phase kernel's 22.2, geometry 24 27.9) */
        take26 = dense ? max_wp >= 120 : ((!shared_gpu && max_wp > 220) || max_wp > 256);
      else /* (... from dense sources already from 120 columns, audited again after quant16's diet, profiles/r06_policy_audit_hb16.txt:
              * 128-256 frames of 120x40, one launch at a time, 256 colours 22.9-23.5 against 26.5-27.2 us, 16 colours 24.5-25.4 against
              * 27.4-27.7; 64 frames at a share of 64 CUs 6.6 / 7.1 against 7.4 / 7.7) */
        take26 = max_wp >= 200 || (max_wp >= 160 && (dense || 4 * n_frames <= 3 * n_cus)) || (dense && max_wp >= 120);
      /* (... but not from 4K sources from 3/4 frame per CU on while the four-slot geometry holds the rows: 192 / 256 frames of 4K -> 200x60,
       * 256 colours 70.8 / 86.0 us against the phase kernel's 65.5 / 79.5, 16 colours 73.1 / 87.8 against 66.8 / 78.1 -- the last audit) */
      if ((mode == ACHIP_MODE_HB_256 || mode == ACHIP_MODE_HB_16) && big_src && max_wp <= 256 && 4 * n_frames >= 3 * n_cus)
        take26 = false;
      if (take26) {
        *variant = 26;
        return 0;
      }
    }
    /* (round 6's last audits: the 256- / 16-colour and mono half blocks of one block per wave take the rows kernel at no more than a frame
     * per CU from dense sources only -- from full-frame sources the phase kernel is 8-10 % ahead there: 128 / 192 / 256 frames of 1080p ->
     * 80x24, 256 colours 16.0 / 17.3 / 20.2 us against 17.4 / 18.7 / 21.0, 16 colours 16.0 / 17.3 / 19.7 against 17.3 / 19.1 / 21.8, mono
     * half blocks 12.4 against 13.4; truecolor half blocks the other way, 18.3 against 17.5) */
    const bool take = n_frames > n_cus ? true : mono ? !(shared_gpu && ext > 8)
                                                     : (ext <= 8 && !shared_gpu && (dense || mode == ACHIP_MODE_HB_TRUE));
    if ((v == 25 || n_frames > n_cus || (mono && max_wp > 256)) && ext <= ACHIP_HOST_STREAM_MAXBLK && take) {
      *variant = v;
      return 0;
    }
  }
  /* Rows beyond one block of the rows kernel (448 cells; round 6): cut into segments, a segment per wave (render_rows.hpp
   * WIDE) -- whole-frame launches of single sources; rows of up to 4096 cells as sixteen-wave workgroups (27), with more
   * than a frame per CU two eight-wave workgroups per CU (29) while a row is at most eight segments. */
  /* (audited over four row widths x seven batch sizes x {mono, truecolor half blocks} x {dense, 1080p sources} x {one launch
   * at a time, four plans in flight}, every geometry forced in turn: profiles/r06_policy_audit_wide.txt -- ahead of the phase
   * kernel by 15-45 % everywhere but one corner, mono on a shared GPU at up to two frames per CU of the share, where the
   * 512-thread phase geometry is 5-12 % ahead: 128 frames of 1000x40 at a share of 64 CUs 47.6 against 53.5 us) */
  const bool wide_corner = mode == ACHIP_MODE_MONO && n_cus <= 128 && n_frames > n_cus && n_frames <= 2 * n_cus && max_wp <= variant_caps[1];
  if (forced_variant < 0 && run_mode && !may_split && !general_sampler && max_wp > 64 * 7 && !wide_corner) {
    const int v = n_frames > n_cus && max_wp <= rows_variant_max_row(29) ? 29 : 27;
    if (max_wp <= rows_variant_max_row(v) && achip_uniform_extent(mode, v, frames, n_frames) <= ACHIP_HOST_STREAM_MAXBLK) {
      *variant = v;
      return 0;
    }
  }
  if (forced_variant >= 0) {
    if (max_wp > variant_caps[forced_variant])
      return -1;
    if (hb && (forced_variant == 1 || forced_variant == 2))
      return -1; /* no half-block instantiations in the 512- / 256-thread geometries (render_inst.hip: has_mode) */
    *variant = forced_variant;
  } else if (!hb && n_frames > (3 * n_cus) / 2 && max_wp <= variant_caps[1]) {
    *variant = 1;
  } else if (max_wp <= variant_caps[4]) {
    *variant = 4;
  } else if (max_wp <= variant_caps[0]) {
    *variant = 0;
  } else {
    return -1;
  }
  /* Row bands.  Never for the serial dither, nor for truecolor-fg with multi-byte glyphs (its RLE state would
   * have to be searched backwards across bands).  Automatic splitting only when whole frames would leave CUs
   * idle (fewer than 3/4 frame per CU): at one frame per CU and above it costs more than it gains. */
  const bool splittable = mode != ACHIP_MODE_16_DITHER_BG && !(mode == ACHIP_MODE_TRUE_FG && !palette_ascii_only) &&
                          split_request >= 0 && max_rows > 1;
  if (!splittable || whole_auto)
    return 0;
  if (forced_variant >= 0 && forced_variant != 1 && forced_variant != 2 && forced_variant != 4)
    return 0; /* row-band kernels exist for the geometries this policy picks (render_inst.hip: HAS_SPLIT) */
  /* a band is exactly one chunk: at most cap / wp rows for every frame of the batch */
  const int band_cap = forced_variant >= 0 ? variant_caps[forced_variant] : variant_caps[4];
  if (max_wp > band_cap)
    return 0;
  int rpp_max = band_cap / max_wp;
  if (rpp_max > max_rows)
    rpp_max = max_rows;
  int rpp;
  if (split_request > 0) {
    rpp = split_request;
  } else { /* about one band per CU and never more bands than CUs unless the chunk size forces it: a launch of 288 bands
            * takes two turns on 256 CUs (24 half-block frames: 17.0 us in twelve bands each against 9.5 in eight) */
    const int want = n_cus / n_frames > 1 ? n_cus / n_frames : 1;
    rpp = (max_rows + want - 1) / want;
  }
  if (rpp > rpp_max)
    rpp = rpp_max;
  if (rpp < 1)
    rpp = 1;
  const int np = (max_rows + rpp - 1) / rpp;
  if (np <= 1)
    return 0;
  if (forced_variant < 0) {
    /* geometry of a band: the half-block kernels need > 128 VGPRs in the small geometries (half the waves per
     * CU), so they always take the 1024-thread one; the others take it while all bands fit the GPU at once and
     * otherwise the geometry whose chunk matches the band, several of which share a CU */
    const long blocks = (long)n_frames * np;
    const int cells = rpp * max_wp;
    if (hb || blocks <= n_cus)
      *variant = 4;
    else if (cells > variant_caps[2])
      *variant = 1;
    else
      *variant = variant_caps[2] >= cells ? 2 : 1; /* (geometry 2 is not in the default build: render_variants.h) */
  }
  *parts = np;
  *rows_per_part = rpp;
  return 0;
}

int achip_frame_blob_parse(const void *blob, size_t size, bool exact, uint32_t *width, uint32_t *height,
                           const uint8_t **pixels) {
  const uint8_t *b = (const uint8_t *)blob;
  if (!b || size < 8u + 3u) /* stream.c:330: a header and at least one pixel */
    return ACHIP_BLOB_SHORT;
  const uint32_t w = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3]; /* NET_TO_HOST_U32 */
  const uint32_t h = ((uint32_t)b[4] << 24) | ((uint32_t)b[5] << 16) | ((uint32_t)b[6] << 8) | b[7];
  /* stream.c:334 (0 < w <= 4096, 0 < h <= 2160) and image_validate_dimensions (lib/util/image.c:100-113:
   * <= IMAGE_MAX_WIDTH x IMAGE_MAX_HEIGHT = 3840 x 2160), which is the tighter of the two */
  if (w == 0u || h == 0u || w > 4096u || h > 2160u || w > 3840u)
    return ACHIP_BLOB_DIMS;
  const size_t expect = 8u + (size_t)w * (size_t)h * 3u; /* image_calc_rgb_size cannot overflow at these sizes */
  if (exact ? size != expect : size < expect) /* protocol.c:812 receive check / stream.c:363 collect check */
    return ACHIP_BLOB_SIZE;
  if (width)
    *width = w;
  if (height)
    *height = h;
  if (pixels)
    *pixels = b + 8;
  return 0;
}

void achip_grid_layout(const int *src_w, const int *src_h, int n, int term_w, int term_h, int *cols, int *rows) {
  if (n <= 0) {
    *cols = 0;
    *rows = 0;
    return;
  }
  if (n == 1) {
    *cols = 1;
    *rows = 1;
    return;
  }
  const float cell_ratio = 2.0f;
  float mean_aspect = 0.0f;
  int counted = 0;
  for (int i = 0; i < n; i++) {
    if (src_w[i] > 0 && src_h[i] > 0) {
      mean_aspect += (float)src_w[i] / (float)src_h[i];
      counted++;
    }
  }
  if (counted > 0)
    mean_aspect /= counted;
  else
    mean_aspect = 1.6f;
  int pick_c = 1, pick_r = n;
  float pick_u = 0.0f;
  for (int c = 1; c <= n; c++) {
    const int r = (n + c - 1) / c;
    if (c * r - n > c)
      continue;
    const int cw = term_w / c, ch = term_h / r;
    if (cw < 20 || ch < 10)
      continue;
    float used = 0.0f;
    for (int i = 0; i < n; i++) {
      const float cell_visual = (float)cw / ((float)ch * cell_ratio);
      int fw, fh;
      if (mean_aspect > cell_visual) {
        fw = cw;
        fh = (int)((cw / mean_aspect) / cell_ratio);
      } else {
        fh = ch;
        fw = (int)(ch * cell_ratio * mean_aspect);
      }
      if (fw > cw)
        fw = cw;
      if (fh > ch)
        fh = ch;
      used += fw * fh;
    }
    const float u = used / (float)(cw * ch * n);
    if (u > pick_u) {
      pick_u = u;
      pick_c = c;
      pick_r = r;
    }
  }
  *cols = pick_c;
  *rows = pick_r;
}

#define ACHIP_GRID_MAX_SOURCES 64 /* the layout counts every client with video; only the first nine are placed */

void achip_composite_setup(achip_composite_t *comp, const uint8_t *const *src_dev, const int *src_w, const int *src_h,
                           int n, int term_w, int term_h) {
  memset(comp, 0, sizeof(*comp));
  int cols, rows;
  /* The layout is chosen for the sources that HAVE video -- the reference passes sources_with_video, and averages
   * the aspect ratio over the sources with an image (stream.c:525-558, 671); a client without video gets no cell. */
  int have_w[ACHIP_GRID_MAX_SOURCES], have_h[ACHIP_GRID_MAX_SOURCES], have = 0;
  for (int i = 0; i < n && have < ACHIP_GRID_MAX_SOURCES; i++)
    if (src_dev[i] && src_w[i] > 0 && src_h[i] > 0) {
      have_w[have] = src_w[i];
      have_h[have] = src_h[i];
      have++;
    }
  achip_grid_layout(have_w, have_h, have, term_w, term_h, &cols, &rows);
  comp->canvas_w = term_w;
  comp->canvas_h = term_h * 2;
  comp->cols = cols;
  comp->rows = rows;
  if (cols <= 0 || rows <= 0)
    return;
  comp->cell_w = comp->canvas_w / cols;
  comp->cell_h = comp->canvas_h / rows;
  int placed = 0;
  for (int i = 0; i < n && placed < 9; i++) {
    if (!src_dev[i] || src_w[i] <= 0 || src_h[i] <= 0)
      continue;
    achip_comp_src_t *s = &comp->s[placed];
    const int row = placed / cols, col = placed % cols;
    placed++;
    const float sa = (float)src_w[i] / (float)src_h[i];
    const float ca = (float)comp->cell_w / (float)comp->cell_h;
    int tw, th;
    if (sa > ca) {
      tw = comp->cell_w;
      th = (int)((comp->cell_w / sa) + 0.5f);
    } else {
      th = comp->cell_h;
      tw = (int)((comp->cell_h * sa) + 0.5f);
    }
    if (tw <= 0 || th <= 0)
      continue; /* leaves s->src NULL: an empty cell */
    s->src = src_dev[i];
    s->src_w = src_w[i];
    s->src_h = src_h[i];
    s->src_stride = 3 * src_w[i];
    s->tile_w = tw;
    s->tile_h = th;
    s->org_x = col * comp->cell_w + (comp->cell_w - tw) / 2;
    s->org_y = row * comp->cell_h + (comp->cell_h - th) / 2;
    s->x_ratio = achip_nn_ratio(src_w[i], tw);
    s->y_ratio = achip_nn_ratio(src_h[i], th);
  }
  comp->n_src = placed;
}
