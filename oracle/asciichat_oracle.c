/*
 * asciichat_oracle.c -- CPU restatement of ascii-chat's render path (see header).
 * TEST INFRASTRUCTURE ONLY -- never linked into the product library.
 */
#include "asciichat_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* growable byte sink (stands for outbuf_t, output_buffer.c:22-70)            */
/* ------------------------------------------------------------------------- */
typedef struct {
  char *p;
  size_t n, cap;
} sink_t;

static void sk_need(sink_t *s, size_t extra) {
  if (s->n + extra + 1 <= s->cap)
    return;
  size_t c = s->cap ? s->cap : 4096;
  while (c < s->n + extra + 1)
    c += c / 2;
  s->p = (char *)realloc(s->p, c);
  s->cap = c;
}
static void sk_byte(sink_t *s, int b) {
  sk_need(s, 1);
  s->p[s->n++] = (char)b;
}
static void sk_mem(sink_t *s, const void *m, size_t k) {
  sk_need(s, k);
  memcpy(s->p + s->n, m, k);
  s->n += k;
}
static char *sk_finish(sink_t *s, size_t *len) {
  sk_need(s, 0);
  s->p[s->n] = '\0';
  if (len)
    *len = s->n;
  return s->p;
}

/* ------------------------------------------------------------------------- */
uint32_t orc_fnv1a32(const void *data, size_t n) {
  const uint8_t *p = (const uint8_t *)data;
  uint32_t h = 2166136261u;
  for (size_t i = 0; i < n; i++) {
    h ^= p[i];
    h *= 16777619u;
  }
  return h;
}

/* L1: Y = (77R + 150G + 29B + 128) >> 8, constants common.h:80-86; result is already in [0,255] */
int orc_luma(int r, int g, int b) { return (77 * r + 150 * g + 29 * b + 128) >> 8; }

/* ansi.c:360-379 */
uint8_t orc_rgb_to_256(uint8_t r, uint8_t g, uint8_t b) {
  int avg = (r + g + b) / 3;
  int spread = abs(r - avg) + abs(g - avg) + abs(b - avg);
  if (spread < 30)
    return (uint8_t)(232 + (avg * 23) / 255);
  return (uint8_t)(16 + 36 * ((r * 5) / 255) + 6 * ((g * 5) / 255) + ((b * 5) / 255));
}

/* the 16 fixed RGBs, ansi.c:442-459 */
static const uint8_t k_ansi16[16][3] = {{0, 0, 0},       {128, 0, 0},   {0, 128, 0},   {128, 128, 0},
                                        {0, 0, 128},     {128, 0, 128}, {0, 128, 128}, {192, 192, 192},
                                        {128, 128, 128}, {255, 0, 0},   {0, 255, 0},   {255, 255, 0},
                                        {0, 0, 255},     {255, 0, 255}, {0, 255, 255}, {255, 255, 255}};

/* ansi.c:437-477: first minimum of squared distance */
uint8_t orc_rgb_to_16(uint8_t r, uint8_t g, uint8_t b) {
  int best = 0, best_d = INT_MAX;
  for (int i = 0; i < 16; i++) {
    int dr = r - k_ansi16[i][0], dg = g - k_ansi16[i][1], db = b - k_ansi16[i][2];
    int d = dr * dr + dg * dg + db * db;
    if (d < best_d) {
      best_d = d;
      best = i;
    }
  }
  return (uint8_t)best;
}

int orc_digits_u32(uint32_t v) {
  int d = 1;
  while (v >= 10u) {
    v /= 10u;
    d++;
  }
  return d;
}

/* output_buffer.c:148-155 */
bool orc_rep_is_profitable(uint32_t run) {
  if (run <= 2)
    return false;
  uint32_t k = run - 1;
  return k > (uint32_t)(orc_digits_u32(k) + 3);
}

static int put_dec(char *dst, uint32_t v) {
  char tmp[10];
  int n = 0;
  do {
    tmp[n++] = (char)('0' + v % 10u);
    v /= 10u;
  } while (v);
  for (int i = 0; i < n; i++)
    dst[i] = tmp[n - 1 - i];
  return n;
}

/* ESC[38;2;R;G;Bm / ESC[48;2;R;G;Bm, decimals without leading zeros (init_dec3, common.c:546-570) */
int orc_sgr_truecolor(char *dst, int bg, uint8_t r, uint8_t g, uint8_t b) {
  char *p = dst;
  memcpy(p, bg ? "\033[48;2;" : "\033[38;2;", 7);
  p += 7;
  p += put_dec(p, r);
  *p++ = ';';
  p += put_dec(p, g);
  *p++ = ';';
  p += put_dec(p, b);
  *p++ = 'm';
  return (int)(p - dst);
}

/* ESC[38;5;Nm / ESC[48;5;Nm, ansi.c:326-357 */
int orc_sgr_256(char *dst, int bg, uint8_t idx) {
  char *p = dst;
  memcpy(p, bg ? "\033[48;5;" : "\033[38;5;", 7);
  p += 7;
  p += put_dec(p, idx);
  *p++ = 'm';
  return (int)(p - dst);
}

/* fg 30-37 / 90-97, bg 40-47 / 100-107, ansi.c:384-435; out-of-range index -> 7 (fg) / 0 (bg) */
int orc_sgr_16(char *dst, int bg, uint8_t idx) {
  if (idx >= 16)
    idx = bg ? 0 : 7;
  int code = bg ? (idx < 8 ? 40 + idx : 100 + (idx - 8)) : (idx < 8 ? 30 + idx : 90 + (idx - 8));
  char *p = dst;
  *p++ = '\033';
  *p++ = '[';
  p += put_dec(p, (uint32_t)code);
  *p++ = 'm';
  return (int)(p - dst);
}

static void sk_reset(sink_t *s) { sk_mem(s, "\033[0m", 4); }

/* emit_rep, output_buffer.c:157-164 */
static void sk_rep(sink_t *s, uint32_t extra) {
  char t[16];
  int n = 0;
  t[n++] = '\033';
  t[n++] = '[';
  n += put_dec(t + n, extra);
  t[n++] = 'b';
  sk_mem(s, t, (size_t)n);
}

/* ------------------------------------------------------------------------- */
/* L2 + L3: palette -> glyph tables (common.c:380-490)                        */
/* ------------------------------------------------------------------------- */
int orc_palette_build(const char *chars, orc_palette_t *pal) {
  if (!chars || !pal || !chars[0])
    return -1;
  memset(pal, 0, sizeof(*pal));
  const char *start[256];
  int blen[256];
  int n = 0;
  const char *p = chars;
  const char *end = chars + strlen(chars);
  while (*p && n < 255) {
    unsigned char c = (unsigned char)*p;
    int l = 1;
    if ((c & 0xE0) == 0xC0)
      l = 2;
    else if ((c & 0xF0) == 0xE0)
      l = 3;
    else if ((c & 0xF8) == 0xF0)
      l = 4;
    start[n] = p;
    blen[n] = l;
    n++;
    /* a truncated trailing sequence walks past the NUL in the reference (UB); stop at the terminator here */
    p = (p + l <= end) ? p + l : end;
  }
  pal->char_count = n;
  for (int i = 0; i < 256; i++) {
    int ci = n > 1 ? (i * (n - 1) + 127) / 255 : 0;
    if (ci >= n)
      ci = n - 1;
    pal->cache[i].len = (uint8_t)blen[ci];
    for (int k = 0; k < blen[ci] && start[ci] + k < end; k++)
      pal->cache[i].bytes[k] = (uint8_t)start[ci][k];
  }
  for (int i = 0; i < 64; i++) {
    int ci = n > 1 ? (i * (n - 1) + 31) / 63 : 0;
    if (ci >= n)
      ci = n - 1;
    pal->ramp[i] = (uint8_t)ci;
    pal->cache64[i].len = (uint8_t)blen[ci];
    for (int k = 0; k < blen[ci] && start[ci] + k < end; k++)
      pal->cache64[i].bytes[k] = (uint8_t)start[ci][k];
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* A1: aspect_ratio (aspect_ratio.c:18-91), float arithmetic on purpose       */
/* ------------------------------------------------------------------------- */
static long round_pos(float x) { /* ROUND(), util/math.h:53 */
  int r = (int)(0.5f + x);
  return r > 0 ? r : 1;
}

void orc_aspect_ratio(long img_w, long img_h, long width, long height, bool stretch, long *out_w, long *out_h) {
  if (img_w <= 0 || img_h <= 0) {
    *out_w = 1;
    *out_h = 1;
    return;
  }
  if (stretch) {
    *out_w = width;
    *out_h = height;
    return;
  }
  const float cell_aspect = 2.0f; /* CHAR_ASPECT */
  long w_from_h = round_pos((float)height * (float)img_w / (float)img_h * cell_aspect);
  long h_from_w = round_pos(((float)width / cell_aspect) * (float)img_h / (float)img_w);
  if (w_from_h <= width) {
    *out_w = w_from_h;
    *out_h = height;
  } else {
    *out_w = width;
    *out_h = h_from_w;
  }
  if (*out_w <= 0)
    *out_w = 1;
  if (*out_h <= 0)
    *out_h = 1;
}

/* ------------------------------------------------------------------------- */
/* R1: nearest-neighbour resize (image.c:267-328)                             */
/* ------------------------------------------------------------------------- */
void orc_resize_nn(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh) {
  if (!src || !dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0)
    return;
  const uint32_t xr = (uint32_t)((((uint64_t)sw << 16) / (uint64_t)dw) + 1);
  const uint32_t yr = (uint32_t)((((uint64_t)sh << 16) / (uint64_t)dh) + 1);
  for (int y = 0; y < dh; y++) {
    uint32_t sy = ((uint32_t)y * yr) >> 16;
    if (sy >= (uint32_t)sh)
      sy = (uint32_t)sh - 1;
    const uint8_t *srow = src + (size_t)sy * (size_t)sw * 3;
    uint8_t *drow = dst + (size_t)y * (size_t)dw * 3;
    for (int x = 0; x < dw; x++) {
      uint32_t sx = ((uint32_t)x * xr) >> 16;
      if (sx >= (uint32_t)sw)
        sx = (uint32_t)sw - 1;
      drow[3 * x + 0] = srow[3 * sx + 0];
      drow[3 * x + 1] = srow[3 * sx + 1];
      drow[3 * x + 2] = srow[3 * sx + 2];
    }
  }
}

/* ------------------------------------------------------------------------- */
/* PM: monochrome (foreground.c:27-138).  Note the double mapping: */
/* the run key is ramp[Y>>2] and the glyph is cache64[that key].               */
/* ------------------------------------------------------------------------- */
char *orc_print_mono(const uint8_t *rgb, int w, int h, const char *palette, size_t *len) {
  orc_palette_t pal;
  if (!rgb || w <= 0 || h <= 0 || orc_palette_build(palette, &pal) != 0)
    return NULL;
  sink_t s = {0};
  for (int y = 0; y < h; y++) {
    const uint8_t *row = rgb + (size_t)y * (size_t)w * 3;
    int x = 0;
    while (x < w) {
      uint8_t key = pal.ramp[orc_luma(row[3 * x], row[3 * x + 1], row[3 * x + 2]) >> 2];
      int j = x + 1;
      while (j < w && pal.ramp[orc_luma(row[3 * j], row[3 * j + 1], row[3 * j + 2]) >> 2] == key)
        j++;
      uint32_t run = (uint32_t)(j - x);
      /* a palette of more than 64 characters makes the reference index cache64[] past its 64 entries (undefined
       * behaviour there: foreground.c:93-102 reads whatever follows the table); product and oracle clamp instead */
      const orc_glyph_t *g = &pal.cache64[key < 64 ? key : 63];
      sk_mem(&s, g->bytes, g->len);
      if (orc_rep_is_profitable(run)) {
        sk_rep(&s, run - 1);
      } else {
        for (uint32_t k = 1; k < run; k++)
          sk_mem(&s, g->bytes, g->len);
      }
      x = j;
    }
    if (y != h - 1)
      sk_byte(&s, '\n');
  }
  return sk_finish(&s, len);
}

/* ------------------------------------------------------------------------- */
/* PT: truecolor foreground (foreground.c:195-308 + ansi_rle_*, ansi.c:248-314) */
/* ------------------------------------------------------------------------- */
char *orc_print_truecolor_fg(const uint8_t *rgb, int w, int h, const char *palette, size_t *len) {
  orc_palette_t pal;
  if (!rgb || orc_palette_build(palette, &pal) != 0)
    return NULL;
  sink_t s = {0};
  /* RLE state: persists across rows, touched only by single-byte ASCII glyph pixels */
  bool first = true;
  int lr = 0xFF, lg = 0xFF, lb = 0xFF;
  char t[24];
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      const uint8_t *px = rgb + ((size_t)y * (size_t)w + (size_t)x) * 3;
      int r = px[0], g = px[1], b = px[2];
      const orc_glyph_t *gl = &pal.cache[orc_luma(r, g, b)];
      if (gl->len == 1 && gl->bytes[0] < 128) {
        if (first || r != lr || g != lg || b != lb) {
          sk_mem(&s, t, (size_t)orc_sgr_truecolor(t, 0, (uint8_t)r, (uint8_t)g, (uint8_t)b));
          lr = r;
          lg = g;
          lb = b;
          first = false;
        }
        sk_byte(&s, gl->bytes[0]);
      } else {
        /* multi-byte glyph: SGR every time, RLE state untouched (foreground.c:281-296) */
        sk_mem(&s, t, (size_t)orc_sgr_truecolor(t, 0, (uint8_t)r, (uint8_t)g, (uint8_t)b));
        sk_mem(&s, gl->bytes, gl->len);
      }
    }
    if (y != h - 1)
      sk_byte(&s, '\n');
  }
  sk_reset(&s); /* ansi_rle_finish */
  return sk_finish(&s, len);
}

/* P256 (foreground.c:433-509) */
char *orc_print_256_fg(const uint8_t *rgb, int w, int h, const char *palette, size_t *len) {
  orc_palette_t pal;
  if (!rgb || w <= 0 || h <= 0 || orc_palette_build(palette, &pal) != 0)
    return NULL;
  sink_t s = {0};
  char t[16];
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      const uint8_t *px = rgb + ((size_t)y * (size_t)w + (size_t)x) * 3;
      sk_mem(&s, t, (size_t)orc_sgr_256(t, 0, orc_rgb_to_256(px[0], px[1], px[2])));
      const orc_glyph_t *gl = &pal.cache[orc_luma(px[0], px[1], px[2])];
      sk_mem(&s, gl->bytes, gl->len);
    }
    sk_reset(&s);
    if (y < h - 1)
      sk_byte(&s, '\n');
  }
  return sk_finish(&s, len);
}

/* P16 (foreground.c:535-624): glyph = cache[ramp[Y>>2]] -- 256-entry table indexed by the ramp (quirk) */
char *orc_print_16_fg(const uint8_t *rgb, int w, int h, const char *palette, size_t *len) {
  orc_palette_t pal;
  if (!rgb || w <= 0 || h <= 0 || orc_palette_build(palette, &pal) != 0)
    return NULL;
  sink_t s = {0};
  char t[16];
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      const uint8_t *px = rgb + ((size_t)y * (size_t)w + (size_t)x) * 3;
      sk_mem(&s, t, (size_t)orc_sgr_16(t, 0, orc_rgb_to_16(px[0], px[1], px[2])));
      const orc_glyph_t *gl = &pal.cache[pal.ramp[orc_luma(px[0], px[1], px[2]) >> 2]];
      sk_mem(&s, gl->bytes, gl->len);
    }
    sk_reset(&s);
    if (y < h - 1)
      sk_byte(&s, '\n');
  }
  return sk_finish(&s, len);
}

/* PB (background.c:17-84) */
char *orc_print_truecolor_bg(const uint8_t *rgb, int w, int h, const char *palette, size_t *len) {
  orc_palette_t pal;
  if (!rgb || orc_palette_build(palette, &pal) != 0)
    return NULL;
  sink_t s = {0};
  char t[24];
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      const uint8_t *px = rgb + ((size_t)y * (size_t)w + (size_t)x) * 3;
      int Y = orc_luma(px[0], px[1], px[2]);
      sk_mem(&s, t, (size_t)orc_sgr_truecolor(t, 1, px[0], px[1], px[2]));
      if (Y < 128)
        sk_mem(&s, "\033[38;2;255;255;255m", 19);
      else
        sk_mem(&s, "\033[38;2;0;0;0m", 13);
      sk_mem(&s, pal.cache[Y].bytes, pal.cache[Y].len);
    }
    sk_reset(&s);
    if (y < h - 1)
      sk_byte(&s, '\n');
  }
  return sk_finish(&s, len);
}

/* PD (foreground.c:752-846 + rgb_to_16color_dithered, ansi.c:511-583): Floyd-Steinberg, int errors,
 * C truncating division; error uses the UNclamped accumulated value. */
static char *dithered16(const uint8_t *rgb, int w, int h, bool use_background, bool ramp_glyph, const char *palette,
                        size_t *len) {
  orc_palette_t pal;
  if (!rgb || w <= 0 || h <= 0 || orc_palette_build(palette, &pal) != 0)
    return NULL;
  int *err = (int *)calloc((size_t)w * (size_t)h * 3, sizeof(int));
  sink_t s = {0};
  char t[16];
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      const uint8_t *px = rgb + ((size_t)y * (size_t)w + (size_t)x) * 3;
      int *e = err + ((size_t)y * (size_t)w + (size_t)x) * 3;
      int v[3] = {px[0] + e[0], px[1] + e[1], px[2] + e[2]};
      e[0] = e[1] = e[2] = 0;
      uint8_t c[3];
      for (int k = 0; k < 3; k++)
        c[k] = (uint8_t)(v[k] < 0 ? 0 : (v[k] > 255 ? 255 : v[k]));
      uint8_t idx = orc_rgb_to_16(c[0], c[1], c[2]);
      for (int k = 0; k < 3; k++) {
        int q = v[k] - (int)k_ansi16[idx][k];
        if (x + 1 < w)
          err[((size_t)y * w + (x + 1)) * 3 + k] += (q * 7) / 16;
        if (y + 1 < h) {
          if (x - 1 >= 0)
            err[((size_t)(y + 1) * w + (x - 1)) * 3 + k] += (q * 3) / 16;
          err[((size_t)(y + 1) * w + x) * 3 + k] += (q * 5) / 16;
          if (x + 1 < w)
            err[((size_t)(y + 1) * w + (x + 1)) * 3 + k] += (q * 1) / 16;
        }
      }
      if (use_background) {
        int bl = (k_ansi16[idx][0] * 77 + k_ansi16[idx][1] * 150 + k_ansi16[idx][2] * 29) / 256;
        sk_mem(&s, t, (size_t)orc_sgr_16(t, 1, idx));
        sk_mem(&s, t, (size_t)orc_sgr_16(t, 0, (uint8_t)(bl < 127 ? 15 : 0)));
      } else {
        sk_mem(&s, t, (size_t)orc_sgr_16(t, 0, idx));
      }
      /* glyph: the with_background function uses cache[Y] (foreground.c:819); the fg-only function at
       * foreground.c:650-750 uses cache[ramp[Y>>2]] (:719-723) -- not reachable from any dispatcher. */
      const int Y = orc_luma(px[0], px[1], px[2]);
      const orc_glyph_t *gl = &pal.cache[ramp_glyph ? pal.ramp[Y >> 2] : Y];
      sk_mem(&s, gl->bytes, gl->len);
    }
    sk_reset(&s);
    if (y < h - 1)
      sk_byte(&s, '\n');
  }
  free(err);
  return sk_finish(&s, len);
}
/* image_print_16color_dithered_with_background, foreground.c:752-846 */
char *orc_print_16_dithered(const uint8_t *rgb, int w, int h, bool use_background, const char *palette, size_t *len) {
  return dithered16(rgb, w, h, use_background, false, palette, len);
}
/* image_print_16color_dithered, foreground.c:650-750 */
char *orc_print_16_dithered_fg(const uint8_t *rgb, int w, int h, const char *palette, size_t *len) {
  return dithered16(rgb, w, h, false, true, palette, len);
}

/* ------------------------------------------------------------------------- */
/* half-block renderers (halfblock.c)                                          */
/* ------------------------------------------------------------------------- */
static const char k_upper_half[3] = {(char)0xE2, (char)0x96, (char)0x80}; /* U+2580 */

typedef struct {
  uint8_t t[3], b[3];
} hb_cell_t;

static hb_cell_t hb_fetch(const uint8_t *rgb, int w, int h, int x, int y) {
  hb_cell_t c;
  const uint8_t *pt = rgb + ((size_t)y * (size_t)w + (size_t)x) * 3;
  memcpy(c.t, pt, 3);
  if (y + 1 < h)
    memcpy(c.b, pt + (size_t)w * 3, 3);
  else
    memcpy(c.b, pt, 3); /* odd height: bottom duplicates top (halfblock.c:81-88) */
  return c;
}

static void hb_glyph_run(sink_t *s, const char *glyph3, uint32_t run) {
  sk_mem(s, glyph3, 3);
  if (orc_rep_is_profitable(run)) {
    sk_rep(s, run - 1);
  } else {
    for (uint32_t k = 1; k < run; k++)
      sk_mem(s, glyph3, 3);
  }
}

/* HT (halfblock.c:48-165) */
char *orc_halfblock_truecolor(const uint8_t *rgb, int w, int h, size_t *len) {
  sink_t s = {0};
  if (w <= 0 || h <= 0 || !rgb)
    return sk_finish(&s, len);
  char t[24];
  for (int y = 0; y < h; y += 2) {
    int cf[3] = {-1, -1, -1}, cb[3] = {-1, -1, -1};
    int x = 0;
    while (x < w) {
      hb_cell_t c = hb_fetch(rgb, w, h, x, y);
      int j = x + 1;
      for (; j < w; j++) {
        hb_cell_t d = hb_fetch(rgb, w, h, j, y);
        if (memcmp(&c, &d, sizeof(c)) != 0)
          break;
      }
      uint32_t run = (uint32_t)(j - x);
      bool transparent = !(c.t[0] | c.t[1] | c.t[2] | c.b[0] | c.b[1] | c.b[2]);
      if (transparent) {
        if (cf[0] != -1 || cf[1] != -1 || cf[2] != -1 || cb[0] != -1 || cb[1] != -1 || cb[2] != -1) {
          sk_reset(&s);
          cf[0] = cf[1] = cf[2] = cb[0] = cb[1] = cb[2] = -1;
        }
        for (uint32_t k = 0; k < run; k++)
          sk_byte(&s, ' ');
      } else {
        if (cf[0] != c.t[0] || cf[1] != c.t[1] || cf[2] != c.t[2]) {
          sk_mem(&s, t, (size_t)orc_sgr_truecolor(t, 0, c.t[0], c.t[1], c.t[2]));
          cf[0] = c.t[0], cf[1] = c.t[1], cf[2] = c.t[2];
        }
        if (cb[0] != c.b[0] || cb[1] != c.b[1] || cb[2] != c.b[2]) {
          sk_mem(&s, t, (size_t)orc_sgr_truecolor(t, 1, c.b[0], c.b[1], c.b[2]));
          cb[0] = c.b[0], cb[1] = c.b[1], cb[2] = c.b[2];
        }
        hb_glyph_run(&s, k_upper_half, run);
      }
      x = j;
    }
    sk_reset(&s);
    if (y + 2 < h)
      sk_byte(&s, '\n');
  }
  return sk_finish(&s, len);
}

/* H256 / H16 share one skeleton (halfblock.c:297-405, 416-524): runs on the quantised pair,
 * transparency on the run head's RAW rgb. */
static char *hb_indexed(const uint8_t *rgb, int w, int h, int use256, size_t *len) {
  sink_t s = {0};
  if (w <= 0 || h <= 0 || !rgb)
    return sk_finish(&s, len);
  char t[16];
  for (int y = 0; y < h; y += 2) {
    int cur_f = -1, cur_b = -1;
    int x = 0;
    while (x < w) {
      hb_cell_t c = hb_fetch(rgb, w, h, x, y);
      uint8_t qf = use256 ? orc_rgb_to_256(c.t[0], c.t[1], c.t[2]) : orc_rgb_to_16(c.t[0], c.t[1], c.t[2]);
      uint8_t qb = use256 ? orc_rgb_to_256(c.b[0], c.b[1], c.b[2]) : orc_rgb_to_16(c.b[0], c.b[1], c.b[2]);
      int j = x + 1;
      for (; j < w; j++) {
        hb_cell_t d = hb_fetch(rgb, w, h, j, y);
        uint8_t f2 = use256 ? orc_rgb_to_256(d.t[0], d.t[1], d.t[2]) : orc_rgb_to_16(d.t[0], d.t[1], d.t[2]);
        uint8_t b2 = use256 ? orc_rgb_to_256(d.b[0], d.b[1], d.b[2]) : orc_rgb_to_16(d.b[0], d.b[1], d.b[2]);
        if (f2 != qf || b2 != qb)
          break;
      }
      uint32_t run = (uint32_t)(j - x);
      bool transparent = !(c.t[0] | c.t[1] | c.t[2] | c.b[0] | c.b[1] | c.b[2]);
      if (transparent) {
        if (cur_f != -1 || cur_b != -1) {
          sk_reset(&s);
          cur_f = cur_b = -1;
        }
        for (uint32_t k = 0; k < run; k++)
          sk_byte(&s, ' ');
      } else {
        if (cur_f != qf) {
          sk_mem(&s, t, (size_t)(use256 ? orc_sgr_256(t, 0, qf) : orc_sgr_16(t, 0, qf)));
          cur_f = qf;
        }
        if (cur_b != qb) {
          sk_mem(&s, t, (size_t)(use256 ? orc_sgr_256(t, 1, qb) : orc_sgr_16(t, 1, qb)));
          cur_b = qb;
        }
        hb_glyph_run(&s, k_upper_half, run);
      }
      x = j;
    }
    sk_reset(&s);
    if (y + 2 < h)
      sk_byte(&s, '\n');
  }
  return sk_finish(&s, len);
}

char *orc_halfblock_256(const uint8_t *rgb, int w, int h, size_t *len) { return hb_indexed(rgb, w, h, 1, len); }
char *orc_halfblock_16(const uint8_t *rgb, int w, int h, size_t *len) { return hb_indexed(rgb, w, h, 0, len); }

/* HM (halfblock.c:184-286): luminance weights 76/150/29 with no rounding term */
char *orc_halfblock_mono(const uint8_t *rgb, int w, int h, size_t *len) {
  static const char shades[4][3] = {{(char)0xE2, (char)0x96, (char)0x91},
                                    {(char)0xE2, (char)0x96, (char)0x92},
                                    {(char)0xE2, (char)0x96, (char)0x93},
                                    {(char)0xE2, (char)0x96, (char)0x88}};
  sink_t s = {0};
  if (w <= 0 || h <= 0 || !rgb)
    return sk_finish(&s, len);
  for (int y = 0; y < h; y += 2) {
    int x = 0;
    while (x < w) {
      hb_cell_t c = hb_fetch(rgb, w, h, x, y);
      int j = x + 1;
      for (; j < w; j++) {
        hb_cell_t d = hb_fetch(rgb, w, h, j, y);
        if (memcmp(&c, &d, sizeof(c)) != 0)
          break;
      }
      uint32_t run = (uint32_t)(j - x);
      uint8_t lt = (uint8_t)((c.t[0] * 76 + c.t[1] * 150 + c.t[2] * 29) >> 8);
      uint8_t lb = (uint8_t)((c.b[0] * 76 + c.b[1] * 150 + c.b[2] * 29) >> 8);
      if (lt < 16 && lb < 16) {
        for (uint32_t k = 0; k < run; k++)
          sk_byte(&s, ' ');
      } else {
        hb_glyph_run(&s, shades[lt >> 6], run);
      }
      x = j;
    }
    if (y + 2 < h)
      sk_byte(&s, '\n');
  }
  return sk_finish(&s, len);
}

/* E3 + D1: dispatcher as an x86-64 build (SIMD_SUPPORT defined) takes it */
char *orc_print_with_caps(const uint8_t *rgb, int w, int h, int color_level, int render_mode, const char *palette,
                          size_t *len) {
  if (!rgb || !palette)
    return NULL;
  if (render_mode == ORC_RENDER_HALF_BLOCK) {
    switch (color_level) {
    case ORC_COLOR_TRUECOLOR:
      return orc_halfblock_truecolor(rgb, w, h, len);
    case ORC_COLOR_256:
      return orc_halfblock_256(rgb, w, h, len);
    case ORC_COLOR_16:
      return orc_halfblock_16(rgb, w, h, len);
    default:
      return orc_halfblock_mono(rgb, w, h, len);
    }
  }
  switch (color_level) {
  case ORC_COLOR_TRUECOLOR:
    if (render_mode == ORC_RENDER_BACKGROUND)
      return orc_print_16_dithered(rgb, w, h, true, palette, len); /* sgr.c:429-430 */
    return orc_print_truecolor_fg(rgb, w, h, palette, len);
  case ORC_COLOR_256:
    return orc_print_256_fg(rgb, w, h, palette, len);
  case ORC_COLOR_16:
    return orc_print_16_fg(rgb, w, h, palette, len);
  default:
    return orc_print_mono(rgb, w, h, palette, len);
  }
}

/* W1 (ascii.c:457-517) */
char *orc_pad_width(const char *frame, size_t pad_left) {
  if (!frame)
    return NULL;
  size_t n = strlen(frame);
  if (pad_left == 0) {
    char *c = (char *)malloc(n + 1);
    memcpy(c, frame, n + 1);
    return c;
  }
  size_t lines = 1;
  for (size_t i = 0; i < n; i++)
    if (frame[i] == '\n')
      lines++;
  char *out = (char *)malloc(n + lines * pad_left + 1);
  char *p = out;
  bool bol = true;
  for (size_t i = 0; i < n; i++) {
    if (bol) {
      memset(p, ' ', pad_left);
      p += pad_left;
      bol = false;
    }
    *p++ = frame[i];
    if (frame[i] == '\n')
      bol = true;
  }
  *p = '\0';
  return out;
}

/* W2 (ascii.c:902-941) */
char *orc_pad_height(const char *frame, size_t pad_top) {
  if (!frame)
    return NULL;
  size_t n = strlen(frame);
  char *out = (char *)malloc(n + pad_top + 1);
  memset(out, '\n', pad_top);
  memcpy(out + pad_top, frame, n + 1);
  return out;
}

/* image_validate_dimensions, lib/util/image.c:100-113 (IMAGE_MAX_WIDTH/HEIGHT 3840x2160) */
static bool dims_ok(long w, long h) { return w > 0 && h > 0 && w <= 3840 && h <= 2160; }

static char *finish_padded(char *ascii, size_t pad_w, size_t pad_h, size_t *len) {
  if (!ascii)
    return NULL;
  if (ascii[0] == '\0') { /* "returned empty string" -> NULL (ascii.c:174-180, 345-352) */
    free(ascii);
    return NULL;
  }
  char *a = orc_pad_width(ascii, pad_w);
  free(ascii);
  char *b = orc_pad_height(a, pad_h);
  free(a);
  if (len)
    *len = strlen(b);
  return b;
}

/* E2 (ascii.c:194-387) */
char *orc_convert_with_caps(const uint8_t *rgb, int src_w, int src_h, long width, long height, int color_level,
                            int render_mode, bool wants_padding, bool use_aspect, bool stretch, const char *palette,
                            size_t *len) {
  if (!rgb || src_w <= 0 || src_w > 10000 || src_h <= 0 || src_h > 10000)
    return NULL;
  long rw = width, rh = height;
  if (use_aspect)
    orc_aspect_ratio(src_w, src_h, rw, rh, stretch, &rw, &rh);
  long out_w = rw, out_h = rh;
  if (render_mode == ORC_RENDER_HALF_BLOCK)
    rh *= 2;
  size_t pad_w = 0, pad_h = 0;
  if (use_aspect && wants_padding) {
    pad_w = (size_t)(width > out_w ? (width - out_w) / 2 : 0);
    pad_h = (size_t)(height > out_h ? (height - out_h) / 2 : 0);
  }
  if (rw <= 0 || rh <= 0 || !dims_ok(rw, rh))
    return NULL;
  uint8_t *rs = (uint8_t *)calloc((size_t)rw * (size_t)rh, 3);
  orc_resize_nn(rgb, src_w, src_h, rs, (int)rw, (int)rh);
  char *ascii = orc_print_with_caps(rs, (int)rw, (int)rh, color_level, render_mode, palette, NULL);
  free(rs);
  return finish_padded(ascii, pad_w, pad_h, len);
}

/* E1 (ascii.c:72-191) */
char *orc_convert(const uint8_t *rgb, int src_w, int src_h, long width, long height, bool color, bool use_aspect,
                  bool stretch, const char *palette, int option_render_mode, size_t *len) {
  if (!rgb || !palette || !palette[0])
    return NULL;
  long rw = width, rh = height;
  if (use_aspect)
    orc_aspect_ratio(src_w, src_h, rw, rh, stretch, &rw, &rh);
  size_t pad_w = 0, pad_h = 0;
  if (use_aspect) {
    pad_w = (size_t)(width > rw ? (width - rw) / 2 : 0);
    pad_h = (size_t)(height > rh ? (height - rh) / 2 : 0);
  }
  if (rw <= 0 || rh <= 0 || !dims_ok(rw, rh))
    return NULL;
  uint8_t *rs = (uint8_t *)calloc((size_t)rw * (size_t)rh, 3);
  orc_resize_nn(rgb, src_w, src_h, rs, (int)rw, (int)rh);
  char *ascii;
  if (color) {
    if (option_render_mode == ORC_RENDER_HALF_BLOCK)
      ascii = orc_halfblock_truecolor(rs, (int)rw, (int)rh, NULL);
    else if (option_render_mode == ORC_RENDER_BACKGROUND)
      ascii = orc_print_16_dithered(rs, (int)rw, (int)rh, true, palette, NULL);
    else
      ascii = orc_print_truecolor_fg(rs, (int)rw, (int)rh, palette, NULL);
  } else {
    ascii = orc_print_mono(rs, (int)rw, (int)rh, palette, NULL);
  }
  free(rs);
  return finish_padded(ascii, pad_w, pad_h, len);
}

/* ------------------------------------------------------------------------- */
/* display-path pre-passes                                                     */
/* ------------------------------------------------------------------------- */
void orc_flip(uint8_t *rgb, int w, int h, bool flip_x, bool flip_y) {
  if (!rgb || !(flip_x || flip_y) || w <= 1 || h <= 1) /* display.c:549 */
    return;
  if (flip_x) {
    for (int y = 0; y < h; y++) {
      uint8_t *row = rgb + (size_t)y * w * 3;
      for (int x = 0; x < w / 2; x++) {
        uint8_t t[3];
        memcpy(t, row + 3 * x, 3);
        memcpy(row + 3 * x, row + 3 * (w - 1 - x), 3);
        memcpy(row + 3 * (w - 1 - x), t, 3);
      }
    }
  }
  if (flip_y) {
    uint8_t *tmp = (uint8_t *)malloc((size_t)w * 3);
    for (int y = 0; y < h / 2; y++) {
      uint8_t *a = rgb + (size_t)y * w * 3, *b = rgb + (size_t)(h - 1 - y) * w * 3;
      memcpy(tmp, a, (size_t)w * 3);
      memcpy(a, b, (size_t)w * 3);
      memcpy(b, tmp, (size_t)w * 3);
    }
    free(tmp);
  }
}

/* the registry of color_filter.c:24-150: {r, g, b, foreground_on_bg}, index = color_filter_t */
static const uint8_t k_filter_tint[12][4] = {{0, 0, 0, 0},     {0, 0, 0, 1},     {255, 255, 255, 0}, {0, 255, 65, 0},
                                             {255, 0, 255, 0}, {255, 0, 170, 0}, {255, 136, 0, 0},   {0, 221, 221, 0},
                                             {0, 255, 255, 0}, {255, 182, 193, 0}, {255, 51, 51, 0}, {255, 235, 153, 0}};

int orc_color_filter(uint8_t *rgb, int w, int h, int stride, int color_filter) {
  if (!rgb || w <= 0 || h <= 0 || stride <= 0)
    return -1;
  if (color_filter == 0)
    return 0;
  if (color_filter < 0 || color_filter >= 12)
    return -1; /* 12 = rainbow: never reaches this function from the display path (display.c:611); its string pass is
                * orc_rainbow_replace */
  const uint8_t *t = k_filter_tint[color_filter];
  for (int y = 0; y < h; y++) {
    uint8_t *row = rgb + (size_t)y * (size_t)stride;
    for (int x = 0; x < w; x++) {
      uint8_t *px = row + 3 * x;
      const unsigned gray = (77u * px[0] + 150u * px[1] + 29u * px[2]) >> 8; /* rgb_to_grayscale, color_filter.h:172 */
      for (int k = 0; k < 3; k++)
        px[k] = (uint8_t)(t[3] ? (t[k] * (255u - gray) + 255u * gray) / 255u : (t[k] * gray) / 255u);
    }
  }
  return 0;
}

char *orc_display_convert(const uint8_t *rgb, int src_w, int src_h, long width, long height, int color_level,
                          int render_mode, bool wants_padding, bool use_aspect, bool stretch, const char *palette,
                          bool flip_x, bool flip_y, int color_filter, size_t *len) {
  if (!rgb || src_w <= 0 || src_h <= 0)
    return NULL;
  const size_t bytes = (size_t)src_w * (size_t)src_h * 3;
  uint8_t *copy = (uint8_t *)malloc(bytes);
  memcpy(copy, rgb, bytes);
  orc_flip(copy, src_w, src_h, flip_x, flip_y);
  if (color_filter != 0 && color_filter != 12)
    orc_color_filter(copy, src_w, src_h, src_w * 3, color_filter);
  char *out = orc_convert_with_caps(copy, src_w, src_h, width, height, color_level, render_mode, wants_padding,
                                    use_aspect, stretch, palette, len);
  free(copy);
  return out;
}

/* ------------------------------------------------------------------------- */
/* G1: text-space grid (ascii.c:527-885)                                       */
/* ------------------------------------------------------------------------- */
static int csi_skip(const char *d, int n, int i) { /* i points at ESC '[' ; returns index after the final byte */
  i += 2;
  while (i < n) {
    char c = d[i++];
    if (c >= '@' && c <= '~')
      break;
  }
  return i;
}

static int visual_width(const char *d, int n) {
  int vw = 0, i = 0;
  while (i < n) {
    if (d[i] == '\033' && i + 1 < n && d[i + 1] == '[') {
      i = csi_skip(d, n, i);
    } else {
      vw++;
      i++;
    }
  }
  return vw;
}

static int truncate_visual(const char *d, int n, int target) {
  int vw = 0, i = 0;
  while (i < n && vw < target) {
    if (d[i] == '\033' && i + 1 < n && d[i + 1] == '[') {
      i = csi_skip(d, n, i);
    } else {
      vw++;
      i++;
    }
  }
  return i;
}

static char *blank_canvas(int width, int height, size_t *size) {
  size_t total = (size_t)width * (size_t)height + (size_t)height + 1;
  char *c = (char *)malloc(total);
  memset(c, ' ', total - 1);
  c[total - 1] = '\0';
  for (int r = 0; r < height; r++)
    c[(size_t)r * (size_t)(width + 1) + (size_t)width] = '\n';
  *size = total;
  return c;
}

char *orc_create_grid(const orc_frame_source_t *sources, int n, int width, int height, size_t *out_size) {
  if (!sources || n <= 0 || width <= 0 || height <= 0 || !out_size)
    return NULL;

  if (n == 1) {
    size_t total;
    char *res = blank_canvas(width, height, &total);
    const char *sd = sources[0].frame_data;
    int ss = (int)sources[0].frame_size;
    *out_size = total - 1;
    if (!sd || ss <= 0)
      return res;
    int lines = 0;
    for (int i = 0; i < ss; i++)
      if (sd[i] == '\n')
        lines++;
    int vpad = (height - lines) / 2;
    if (vpad < 0)
      vpad = 0;
    int row = vpad, pos = 0;
    while (pos < ss && row < height) {
      int ls = pos, ll = 0;
      while (pos < ss && sd[pos] != '\n') {
        ll++;
        pos++;
      }
      int vw = visual_width(sd + ls, ll);
      int hpad = (width - vw) / 2;
      if (hpad < 0)
        hpad = 0;
      size_t dst = (size_t)row * (size_t)(width + 1) + (size_t)hpad;
      int copy = truncate_visual(sd + ls, ll, width - hpad);
      if (copy > 0 && dst + (size_t)copy < total)
        memcpy(res + dst, sd + ls, (size_t)copy);
      if (pos < ss && sd[pos] == '\n')
        pos++;
      row++;
    }
    return res;
  }

  /* layout search (ascii.c:712-769), float32 on purpose */
  float best = -1.0f;
  int bc = 1, br = n;
  for (int tc = 1; tc <= n; tc++) {
    int tr = (int)ceil((double)n / tc);
    if (tc * tr - n > n / 2)
      continue;
    int cw = (width - (tc - 1)) / tc;
    int ch = (height - (tr - 1)) / tr;
    if (cw < 10 || ch < 3)
      continue;
    float cell_aspect = ((float)cw / (float)ch) / 2.0f;
    float a = 1.0f - fabsf(logf(cell_aspect));
    if (a < 0)
      a = 0;
    float u = (float)n / (float)(tc * tr);
    float score = (n == 2) ? a * 0.9f + u * 0.1f : a * 0.7f + u * 0.3f;
    if (tc == tr)
      score += 0.05f;
    if (score > best) {
      best = score;
      bc = tc;
      br = tr;
    }
  }
  int cw = (width - (bc - 1)) / bc;
  int ch = (height - (br - 1)) / br;
  if (cw < 10 || ch < 3) {
    char *res = (char *)malloc(sources[0].frame_size + 1);
    if (sources[0].frame_data && sources[0].frame_size > 0) {
      memcpy(res, sources[0].frame_data, sources[0].frame_size);
      res[sources[0].frame_size] = '\0';
      *out_size = sources[0].frame_size;
    } else {
      res[0] = '\0';
      *out_size = 0;
    }
    return res;
  }

  size_t total;
  char *mix = blank_canvas(width, height, &total);
  for (int s = 0; s < n; s++) {
    int gr = s / bc, gc = s % bc;
    int r0 = gr * (ch + 1), c0 = gc * (cw + 1);
    const char *sd = sources[s].frame_data;
    int ss = (int)sources[s].frame_size;
    int srow = 0, pos = 0;
    while (pos < ss && srow < ch && r0 + srow < height) {
      int ls = pos;
      while (pos < ss && sd[pos] != '\n')
        pos++;
      int ll = pos - ls;
      int copy = truncate_visual(sd + ls, ll, cw);
      int vw = visual_width(sd + ls, copy);
      if (copy > 0 && c0 + vw <= width) {
        /* raw bytes are copied, so escape-laden lines may overrun the cell in byte space (reference behaviour) */
        size_t at = (size_t)(r0 + srow) * (size_t)(width + 1) + (size_t)c0;
        if ((size_t)copy <= total - at) /* SAFE_MEMCPY(dest, mixed_size - mixed_pos, ..): a copy that would leave the
                                           canvas is refused as a whole (lib/platform/posix/system.c:653-666) */
          memcpy(mix + at, sd + ls, (size_t)copy);
      }
      if (pos < ss && sd[pos] == '\n')
        pos++;
      srow++;
    }
    if (gc < bc - 1 && c0 + cw < width) {
      for (int r = r0; r < r0 + ch && r < height; r++) {
        size_t idx = (size_t)r * (size_t)(width + 1) + (size_t)(c0 + cw);
        if (idx < total - 1)
          mix[idx] = '|';
      }
    }
    if (gr < br - 1 && r0 + ch < height) {
      for (int c = c0; c < c0 + cw && c < width; c++) {
        size_t idx = (size_t)(r0 + ch) * (size_t)(width + 1) + (size_t)c;
        if (idx < total - 1)
          mix[idx] = '_';
      }
      if (gc < bc - 1 && c0 + cw < width) {
        size_t idx = (size_t)(r0 + ch) * (size_t)(width + 1) + (size_t)(c0 + cw);
        if (idx < total - 1)
          mix[idx] = '+';
      }
    }
  }
  mix[total - 1] = '\0'; /* the reference's strlen (ascii.c:883) is undefined when a paste took the terminator with it;
                            outside that case this line changes nothing */
  *out_size = strlen(mix);
  return mix;
}

/* ------------------------------------------------------------------------- */
/* C1 + C2: pixel-space composite (src/server/stream.c:523-779)                */
/* ------------------------------------------------------------------------- */
void orc_grid_layout(const int *src_w, const int *src_h, int n, int term_w, int term_h, int *cols, int *rows) {
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
  const float cell_char_aspect = 2.0f;
  float avg = 0.0f;
  for (int i = 0; i < n; i++)
    avg += (float)src_w[i] / (float)src_h[i];
  avg /= n;
  int bc = 1, br = n;
  float best = 0.0f;
  for (int c = 1; c <= n; c++) {
    int r = (n + c - 1) / c;
    if (c * r - n > c)
      continue;
    int cw = term_w / c, ch = term_h / r;
    if (cw < 20 || ch < 10)
      continue;
    float used = 0.0f;
    int cell_area = cw * ch;
    for (int i = 0; i < n; i++) {
      float cva = (float)cw / ((float)ch * cell_char_aspect);
      int fw, fh;
      if (avg > cva) {
        fw = cw;
        fh = (int)((cw / avg) / cell_char_aspect);
      } else {
        fh = ch;
        fw = (int)(ch * cell_char_aspect * avg);
      }
      if (fw > cw)
        fw = cw;
      if (fh > ch)
        fh = ch;
      used += fw * fh;
    }
    float util = used / (float)(cell_area * n);
    if (util > best) {
      best = util;
      bc = c;
      br = r;
    }
  }
  *cols = bc;
  *rows = br;
}

uint8_t *orc_composite(const uint8_t *const *src, const int *src_w, const int *src_h, int n, int term_w, int term_h,
                       int *out_w, int *out_h) {
  int cols, rows;
  /* calculate_optimal_grid_layout (stream.c:523-651) sees sources_with_video cells and averages the aspect of the
   * sources that have an image (:546-552): a client without video takes no cell and does not move the others'. */
  int vw[64], vh[64], nv = 0;
  for (int i = 0; i < n && nv < 64; i++)
    if (src[i]) {
      vw[nv] = src_w[i];
      vh[nv] = src_h[i];
      nv++;
    }
  orc_grid_layout(vw, vh, nv, term_w, term_h, &cols, &rows);
  int cw_px = term_w, ch_px = term_h * 2;
  uint8_t *canvas = (uint8_t *)calloc((size_t)cw_px * (size_t)ch_px, 3);
  *out_w = cw_px;
  *out_h = ch_px;
  if (cols <= 0 || rows <= 0)
    return canvas;
  for (int i = 0, vi = 0; i < n && vi < 9; i++) {
    if (!src[i])
      continue;
    int row = vi / cols, col = vi % cols;
    vi++;
    int cell_w = cw_px / cols, cell_h = ch_px / rows;
    float sa = (float)src_w[i] / (float)src_h[i];
    float ca = (float)cell_w / (float)cell_h;
    int tw, th;
    if (sa > ca) {
      tw = cell_w;
      th = (int)((cell_w / sa) + 0.5f);
    } else {
      th = cell_h;
      tw = (int)((cell_h * sa) + 0.5f);
    }
    if (tw <= 0 || th <= 0)
      continue; /* the reference would fail image_new_from_pool(0, ..) here */
    uint8_t *tile = (uint8_t *)malloc((size_t)tw * (size_t)th * 3);
    orc_resize_nn(src[i], src_w[i], src_h[i], tile, tw, th);
    int x0 = col * cell_w, y0 = row * cell_h;
    int xp = (cell_w - tw) / 2, yp = (cell_h - th) / 2;
    for (int y = 0; y < th; y++) {
      for (int x = 0; x < tw; x++) {
        int dx = x0 + xp + x, dy = y0 + yp + y;
        if (dx < x0 || dx > x0 + cell_w - 1 || dy < y0 || dy > y0 + cell_h - 1)
          continue;
        if (dx < 0 || dx >= cw_px || dy < 0 || dy >= ch_px)
          continue;
        memcpy(canvas + ((size_t)dy * cw_px + dx) * 3, tile + ((size_t)y * tw + x) * 3, 3);
      }
    }
    free(tile);
  }
  return canvas;
}

/* ------------------------------------------------------------------------------------------- */
/* wire stage: CRC-32C and the ASCII frame packet header                                         */
/* ------------------------------------------------------------------------------------------- */
static uint32_t crc32c_update(uint32_t crc, const uint8_t *p, size_t n) {
  for (size_t i = 0; i < n; i++) { /* lib/network/crc32.c:177-186 */
    crc ^= p[i];
    for (int j = 0; j < 8; j++)
      crc = (crc & 1u) ? (crc >> 1) ^ 0x82F63B78u : crc >> 1;
  }
  return crc;
}

uint32_t orc_crc32c(const void *data, size_t n) { return ~crc32c_update(0xFFFFFFFFu, (const uint8_t *)data, n); }

static void be32(uint8_t *p, uint32_t v) {
  p[0] = (uint8_t)(v >> 24);
  p[1] = (uint8_t)(v >> 16);
  p[2] = (uint8_t)(v >> 8);
  p[3] = (uint8_t)v;
}

uint32_t orc_ascii_frame_packet(const void *frame, size_t n, uint32_t width, uint32_t height, uint8_t hdr[24]) {
  be32(hdr + 0, width); /* server.c:208-214 */
  be32(hdr + 4, height);
  be32(hdr + 8, (uint32_t)n);
  be32(hdr + 12, 0);
  be32(hdr + 16, orc_crc32c(frame, n));
  be32(hdr + 20, 0);
  uint32_t c = crc32c_update(0xFFFFFFFFu, hdr, 24); /* the payload send.c checksums is header followed by frame */
  return ~crc32c_update(c, (const uint8_t *)frame, n);
}

/* ------------------------------------------------------------------------------------------- */
/* ingest: the camera frame blob                                                                 */
/* ------------------------------------------------------------------------------------------- */
int orc_frame_blob_accept(const void *blob, size_t size, int exact, uint32_t *w, uint32_t *h) {
  const uint8_t *p = (const uint8_t *)blob;
  if (!exact && !(size > 0 && size >= 4 * 2 + 3)) /* stream.c:330 */
    return 0;
  if (exact && size < 8) /* protocol.c reads the two header words first */
    return 0;
  const uint32_t pw = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
  const uint32_t ph = ((uint32_t)p[4] << 24) | ((uint32_t)p[5] << 16) | ((uint32_t)p[6] << 8) | p[7];
  if (!exact && (pw == 0 || ph == 0 || pw > 4096 || ph > 2160)) /* stream.c:334 */
    return 0;
  if (pw == 0 || ph == 0 || pw > 3840 || ph > 2160) /* image_validate_dimensions, lib/util/image.c:100-113 */
    return 0;
  const size_t rgb = (size_t)pw * (size_t)ph * 3; /* image_calc_rgb_size */
  if (exact && rgb > (size_t)3840 * 2160 * 3)     /* image_validate_buffer_size, protocol.c:804 */
    return 0;
  const size_t expect = 8 + rgb;
  if (exact ? size != expect : size < expect) /* protocol.c:812 / stream.c:363 */
    return 0;
  *w = pw;
  *h = ph;
  return 1;
}

/* ------------------------------------------------------------------------------------------- */
/* uncalled leftovers: REP expansion / compression of a finished frame, frame integrity check    */
/* (written independently of the product's hostutil.c: byte-at-a-time state machines)            */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
  char *p;
  size_t n, cap;
} obuf_t;
static void obuf_put(obuf_t *o, const char *s, size_t k) {
  if (o->n + k + 1 > o->cap) {
    o->cap = (o->n + k + 1) * 2 + 64;
    o->p = (char *)realloc(o->p, o->cap);
  }
  memcpy(o->p + o->n, s, k);
  o->n += k;
  o->p[o->n] = 0;
}

char *orc_expand_rle(const char *in, size_t n, size_t *out_len) {
  if (!in || !n)
    return NULL;
  obuf_t o = {0};
  char last[4] = {' ', 0, 0, 0};
  size_t last_n = 1, i = 0;
  while (i < n) {
    if (in[i] == 27 && i + 1 < n && in[i + 1] == '[') { /* rle.c:25-52 */
      size_t j = i + 2;
      unsigned long param = 0;
      for (; j < n && ((in[j] >= '0' && in[j] <= '9') || in[j] == ';'); j++)
        param = in[j] == ';' ? 0 : (unsigned long)(uint32_t)(param * 10 + (unsigned long)(in[j] - '0'));
      if (j >= n)
        break; /* truncated sequence: nothing more is written */
      if (in[j] == 'b' && (uint32_t)param > 0) {
        for (uint32_t r = 0; r < (uint32_t)param; r++)
          obuf_put(&o, last, last_n);
      } else {
        obuf_put(&o, in + i, j + 1 - i);
      }
      i = j + 1;
      continue;
    }
    unsigned char c = (unsigned char)in[i]; /* rle.c:53-80 */
    size_t k = 1;
    if (c >= 0xC0 && c < 0xE0)
      k = 2;
    else if (c >= 0xE0 && c < 0xF0)
      k = 3;
    else if (c >= 0xF0 && c < 0xF8)
      k = 4;
    if (k > n - i)
      k = n - i;
    obuf_put(&o, in + i, k);
    if (c >= 0x20 && c != 0x7F) {
      memcpy(last, in + i, k);
      last_n = k;
    }
    i += k;
  }
  if (!o.p)
    obuf_put(&o, "", 0);
  *out_len = o.n;
  return o.p;
}

char *orc_compress_rle(const char *in, size_t n, size_t *out_len) {
  if (!in || !n)
    return NULL;
  obuf_t o = {0};
  size_t i = 0;
  while (i < n) {
    if (in[i] == 27 && i + 1 < n && in[i + 1] == '[') { /* rle.c:103-117: sequences pass through */
      size_t j = i + 2;
      while (j < n && ((in[j] >= '0' && in[j] <= '9') || in[j] == ';'))
        j++;
      if (j < n)
        j++;
      obuf_put(&o, in + i, j - i);
      i = j;
      continue;
    }
    const int c = (signed char)in[i]; /* rle.c:119-121 compares a plain char, signed on x86-64 */
    if (c < 0x20 || c == 0x7F) {
      obuf_put(&o, in + i, 1);
      i++;
      continue;
    }
    size_t run = 0;
    while (i + run < n && in[i + run] == in[i])
      run++;
    obuf_put(&o, in + i, 1);
    if (run > 1 && orc_rep_is_profitable((uint32_t)run)) {
      char t[16];
      int k = snprintf(t, sizeof t, "\033[%ub", (unsigned)(run - 1));
      obuf_put(&o, t, (size_t)k);
    } else {
      for (size_t q = 1; q < run; q++)
        obuf_put(&o, in + i, 1);
    }
    i += run;
  }
  if (!o.p)
    obuf_put(&o, "", 0);
  *out_len = o.n;
  return o.p;
}

static long last_reset(const char *d, size_t n) {
  for (long i = (long)n - 4; i >= 0; i--)
    if (d[i] == 27 && d[i + 1] == '[' && d[i + 2] == '0' && d[i + 3] == 'm')
      return i;
  return -1;
}
int orc_frame_validate_integrity(const char *d, size_t n) {
  if (!d || n == 0)
    return 0;
  const long p = n >= 4 ? last_reset(d, n) : -1;
  return p >= 0 && (size_t)p + 4 == n;
}
size_t orc_frame_get_valid_end(const char *d, size_t n) {
  if (!d || n < 4)
    return n;
  const long p = last_reset(d, n);
  return p < 0 ? n : (size_t)p + 4;
}

/* ------------------------------------------------------------------------- */
/* rainbow filter (color_filter.c:169-243, 348-408)                            */
/* ------------------------------------------------------------------------- */
void orc_rainbow_color(float time_seconds, uint8_t rgb[3]) {
  float phase = fmodf(time_seconds, 3.5f) / 3.5f;
  float hue = phase * 360.0f;
  float h = hue / 60.0f;
  int i = (int)floorf(h);
  float f = h - (float)i;
  float q = 1.0f - f;
  uint8_t r = 255, g = 0, b = 0;
  switch (i % 6) {
  case 0: r = 255; g = (uint8_t)(f * 255.0f + 0.5f); b = 0; break;
  case 1: r = (uint8_t)(q * 255.0f + 0.5f); g = 255; b = 0; break;
  case 2: r = 0; g = 255; b = (uint8_t)(f * 255.0f + 0.5f); break;
  case 3: r = 0; g = (uint8_t)(q * 255.0f + 0.5f); b = 255; break;
  case 4: r = (uint8_t)(f * 255.0f + 0.5f); g = 0; b = 255; break;
  case 5: r = 255; g = 0; b = (uint8_t)(q * 255.0f + 0.5f); break;
  default: break;
  }
  float lum = 0.2126f * r + 0.7152f * g + 0.0722f * b;
  if (lum < 120.0f) {
    float boost = (120.0f - lum) / 3.0f;
    r = (uint8_t)fminf(255.0f, r + boost);
    g = (uint8_t)fminf(255.0f, g + boost);
    b = (uint8_t)fminf(255.0f, b + boost);
  }
  rgb[0] = r;
  rgb[1] = g;
  rgb[2] = b;
}

char *orc_rainbow_replace(const char *frame, float time_seconds, size_t *out_len) {
  static const char k_lead[] = "\033[38;2;";
  if (!frame || !strstr(frame, k_lead))
    return NULL;
  uint8_t c[3];
  orc_rainbow_color(time_seconds, c);
  char code[32];
  int code_len = snprintf(code, sizeof code, "\033[38;2;%d;%d;%dm", c[0], c[1], c[2]);
  sink_t s = {0};
  const char *p = frame;
  while (*p) {
    const char *hit = strstr(p, k_lead);
    if (!hit) {
      sk_mem(&s, p, strlen(p));
      break;
    }
    sk_mem(&s, p, (size_t)(hit - p));
    const char *end = strchr(hit + 7, 'm');
    if (end) {
      sk_mem(&s, code, (size_t)code_len);
      p = end + 1;
    } else { /* no 'm' anywhere behind the lead-in (never in a rendered frame): the reference re-copies bytes it has
              * already copied and can run past its 2n buffer (color_filter.c:396-399); here the tail stays as it is */
      sk_byte(&s, *hit);
      p = hit + 1;
    }
  }
  return sk_finish(&s, out_len);
}
