/*
 * hostutil.c -- the small host-side utilities of the reference's render API surface: per-value colour
 * helpers, SGR string builders, the outbuf/RLE builders, string padding and the text-space grid.
 * These operate on single values or on finished (KB-sized) strings in host memory; they are part of the
 * drop-in boundary but not of the per-pixel hot loops, which run on the GPU.
 */
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <pthread.h>
#include <string.h>

#include "achip_host.h"
#include "asciichat_render.h"

char g_default_luminance_palette[256];

/* ---- colour quantisers (lib/video/terminal/ansi.c:360-379, 437-509) ------------------------- */
uint8_t rgb_to_256color(uint8_t r, uint8_t g, uint8_t b) {
  const int mean = (r + g + b) / 3;
  const int spread = abs(r - mean) + abs(g - mean) + abs(b - mean);
  if (spread < 30)
    return (uint8_t)(232 + mean * 23 / 255);
  return (uint8_t)(16 + (r * 5 / 255) * 36 + (g * 5 / 255) * 6 + (b * 5 / 255));
}

static const uint8_t ansi16_rgb[16][3] = {
    {0, 0, 0},       {128, 0, 0}, {0, 128, 0}, {128, 128, 0}, {0, 0, 128}, {128, 0, 128}, {0, 128, 128}, {192, 192, 192},
    {128, 128, 128}, {255, 0, 0}, {0, 255, 0}, {255, 255, 0}, {0, 0, 255}, {255, 0, 255}, {0, 255, 255}, {255, 255, 255}};

uint8_t rgb_to_16color(uint8_t r, uint8_t g, uint8_t b) {
  int pick = 0, pick_d = INT_MAX;
  for (int i = 0; i < 16; i++) {
    const int dr = r - ansi16_rgb[i][0], dg = g - ansi16_rgb[i][1], db = b - ansi16_rgb[i][2];
    const int d = dr * dr + dg * dg + db * db;
    if (d < pick_d) {
      pick_d = d;
      pick = i;
    }
  }
  return (uint8_t)pick;
}

void get_16color_rgb(uint8_t color_index, uint8_t *r, uint8_t *g, uint8_t *b) {
  if (color_index >= 16)
    color_index = 7;
  *r = ansi16_rgb[color_index][0];
  *g = ansi16_rgb[color_index][1];
  *b = ansi16_rgb[color_index][2];
}

/* rgb_to_16color_dithered (lib/video/terminal/ansi.c:511-583): ONE pixel of the Floyd-Steinberg pass against a
 * caller-held error buffer (width x height rgb_error_t) -- take the error parked for (x, y), quantise the clamped sum,
 * park 7/16, 3/16, 5/16, 1/16 of the UNclamped difference for the right / lower-left / lower / lower-right neighbours that
 * exist (C division: each share truncates toward zero).  error_buffer == NULL: plain rgb_to_16color of the clamped input.
 * The renderers do not go through this function (the device pass is dither16_rows); it is part of the API surface. */
static uint8_t clamp_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

uint8_t rgb_to_16color_dithered(int r, int g, int b, int x, int y, int width, int height, rgb_error_t *error_buffer) {
  rgb_error_t *here = error_buffer ? &error_buffer[(size_t)y * (size_t)width + (size_t)x] : NULL;
  if (here) {
    r += here->r;
    g += here->g;
    b += here->b;
    here->r = here->g = here->b = 0;
  }
  const uint8_t idx = rgb_to_16color(clamp_u8(r), clamp_u8(g), clamp_u8(b));
  if (!here)
    return idx;
  uint8_t pr, pg, pb;
  get_16color_rgb(idx, &pr, &pg, &pb);
  const int e[3] = {r - (int)pr, g - (int)pg, b - (int)pb};
  /* neighbour (dx, dy) and its sixteenths */
  static const int nb[4][3] = {{1, 0, 7}, {-1, 1, 3}, {0, 1, 5}, {1, 1, 1}};
  for (int k = 0; k < 4; k++) {
    const int nx = x + nb[k][0], ny = y + nb[k][1];
    if (nx < 0 || nx >= width || ny >= height)
      continue;
    rgb_error_t *t = &error_buffer[(size_t)ny * (size_t)width + (size_t)nx];
    t->r += (e[0] * nb[k][2]) / 16;
    t->g += (e[1] * nb[k][2]) / 16;
    t->b += (e[2] * nb[k][2]) / 16;
  }
  return idx;
}

/* ---- SGR builders (ansi.c:143-246, 326-435) --------------------------------------------------- */
static char *dec_u32(char *p, uint32_t v) {
  char rev[10];
  int n = 0;
  do {
    rev[n++] = (char)('0' + v % 10u);
    v /= 10u;
  } while (v);
  while (n)
    *p++ = rev[--n];
  return p;
}

static char *sgr_rgb(char *p, char layer, uint8_t r, uint8_t g, uint8_t b) {
  *p++ = '\033';
  *p++ = '[';
  *p++ = layer; /* '3' fg, '4' bg */
  *p++ = '8';
  *p++ = ';';
  *p++ = '2';
  *p++ = ';';
  p = dec_u32(p, r);
  *p++ = ';';
  p = dec_u32(p, g);
  *p++ = ';';
  p = dec_u32(p, b);
  *p++ = 'm';
  return p;
}

char *append_truecolor_fg(char *dst, uint8_t r, uint8_t g, uint8_t b) { return sgr_rgb(dst, '3', r, g, b); }
char *append_truecolor_bg(char *dst, uint8_t r, uint8_t g, uint8_t b) { return sgr_rgb(dst, '4', r, g, b); }

char *append_truecolor_fg_bg(char *dst, uint8_t fg_r, uint8_t fg_g, uint8_t fg_b, uint8_t bg_r, uint8_t bg_g,
                             uint8_t bg_b) {
  char *p = sgr_rgb(dst, '3', fg_r, fg_g, fg_b) - 1; /* drop the 'm' and continue the parameter list */
  memcpy(p, ";48;2;", 6);
  p += 6;
  p = dec_u32(p, bg_r);
  *p++ = ';';
  p = dec_u32(p, bg_g);
  *p++ = ';';
  p = dec_u32(p, bg_b);
  *p++ = 'm';
  return p;
}

static char *sgr_indexed(char *p, char layer, uint8_t idx) {
  *p++ = '\033';
  *p++ = '[';
  *p++ = layer;
  *p++ = '8';
  *p++ = ';';
  *p++ = '5';
  *p++ = ';';
  p = dec_u32(p, idx);
  *p++ = 'm';
  return p;
}
char *append_256color_fg(char *dst, uint8_t color_index) { return sgr_indexed(dst, '3', color_index); }
char *append_256color_bg(char *dst, uint8_t color_index) { return sgr_indexed(dst, '4', color_index); }

char *append_16color_fg(char *dst, uint8_t color_index) {
  if (color_index >= 16)
    color_index = 7;
  *dst++ = '\033';
  *dst++ = '[';
  dst = dec_u32(dst, color_index < 8 ? 30u + color_index : 82u + color_index);
  *dst++ = 'm';
  return dst;
}
char *append_16color_bg(char *dst, uint8_t color_index) {
  if (color_index >= 16)
    color_index = 0;
  *dst++ = '\033';
  *dst++ = '[';
  dst = dec_u32(dst, color_index < 8 ? 40u + color_index : 92u + color_index);
  *dst++ = 'm';
  return dst;
}

/* ---- colour-change RLE context (ansi.c:248-314) -------------------------------------------------- */
void ansi_rle_init(ansi_rle_context_t *ctx, char *buffer, size_t capacity, ansi_color_mode_t mode) {
  ctx->buffer = buffer;
  ctx->capacity = capacity;
  ctx->length = 0;
  ctx->mode = mode;
  ctx->first_pixel = true;
  ctx->last_r = ctx->last_g = ctx->last_b = 0xFF;
}

void ansi_rle_add_pixel(ansi_rle_context_t *ctx, uint8_t r, uint8_t g, uint8_t b, char ascii_char) {
  const bool changed = ctx->first_pixel || r != ctx->last_r || g != ctx->last_g || b != ctx->last_b;
  if (changed && ctx->length + 40 < ctx->capacity) {
    char *at = ctx->buffer + ctx->length;
    if (ctx->mode == ANSI_MODE_FOREGROUND)
      at = append_truecolor_fg(at, r, g, b);
    else if (ctx->mode == ANSI_MODE_BACKGROUND)
      at = append_truecolor_bg(at, r, g, b);
    else
      at = append_truecolor_fg_bg(at, r, g, b, 0, 0, 0);
    ctx->length = (size_t)(at - ctx->buffer);
    ctx->last_r = r;
    ctx->last_g = g;
    ctx->last_b = b;
    ctx->first_pixel = false;
  }
  if (ctx->length < ctx->capacity - 1)
    ctx->buffer[ctx->length++] = ascii_char;
}

void ansi_rle_finish(ansi_rle_context_t *ctx) {
  if (ctx->length + 5 <= ctx->capacity) {
    memcpy(ctx->buffer + ctx->length, "\033[0m", 4);
    ctx->length += 4;
  }
  if (ctx->length < ctx->capacity)
    ctx->buffer[ctx->length] = '\0';
}

/* ---- outbuf (lib/video/ascii/output_buffer.c) ------------------------------------------------------- */
void ob_reserve(outbuf_t *ob, size_t need) {
  if (!ob)
    return;
  if (ob->cap != 0 && ob->len + need <= ob->cap)
    return;
  size_t grown = ob->cap ? ob->cap : 4096;
  while (grown < ob->len + need)
    grown = grown * 3 / 2;
  ob->buf = (char *)realloc(ob->buf, grown);
  ob->cap = grown;
}
void ob_putc(outbuf_t *ob, char c) {
  if (!ob)
    return;
  ob_reserve(ob, 1);
  ob->buf[ob->len++] = c;
}
void ob_write(outbuf_t *ob, const char *s, size_t n) {
  if (!ob || !n)
    return;
  ob_reserve(ob, n);
  memcpy(ob->buf + ob->len, s, n);
  ob->len += n;
}
void ob_term(outbuf_t *ob) {
  if (!ob)
    return;
  if (ob->len >= ob->cap)
    ob_reserve(ob, 1);
  ob->buf[ob->len] = '\0';
}
void ob_u32(outbuf_t *ob, uint32_t v) {
  if (!ob)
    return;
  char t[12];
  char *e = dec_u32(t, v);
  ob_write(ob, t, (size_t)(e - t));
}
void ob_u8(outbuf_t *ob, uint8_t v) { ob_u32(ob, v); }
void emit_set_fg(outbuf_t *ob, uint8_t r, uint8_t g, uint8_t b) {
  char t[24];
  ob_write(ob, t, (size_t)(append_truecolor_fg(t, r, g, b) - t));
}
void emit_set_bg(outbuf_t *ob, uint8_t r, uint8_t g, uint8_t b) {
  char t[24];
  ob_write(ob, t, (size_t)(append_truecolor_bg(t, r, g, b) - t));
}
void emit_reset(outbuf_t *ob) { ob_write(ob, "\033[0m", 4); }

bool rep_is_profitable(uint32_t runlen) {
  if (runlen <= 2)
    return false;
  const uint32_t extra = runlen - 1;
  uint32_t digits = 1;
  for (uint32_t t = extra; t >= 10; t /= 10)
    digits++;
  return extra > digits + 3;
}
void emit_rep(outbuf_t *ob, uint32_t extra) {
  if (!ob)
    return;
  ob_write(ob, "\033[", 2);
  ob_u32(ob, extra);
  ob_putc(ob, 'b');
}

/* ---- glyph caches in the reference's host layout (common.c:380-490) ---------------------------------- */
static int split_palette(const char *s, const char *start[256], int blen[256]) {
  int n = 0;
  const char *end = s + strlen(s);
  while (s < end && n < 255) {
    const unsigned char c = (unsigned char)*s;
    const int l = (c & 0xE0) == 0xC0 ? 2 : (c & 0xF0) == 0xE0 ? 3 : (c & 0xF8) == 0xF0 ? 4 : 1;
    start[n] = s;
    blen[n] = l;
    n++;
    s = s + l <= end ? s + l : end;
  }
  return n;
}

static void fill_glyph(utf8_char_t *dst, const char *start, int len, const char *limit) {
  memset(dst, 0, sizeof(*dst));
  dst->byte_len = (uint8_t)len;
  for (int k = 0; k < len && start + k < limit; k++)
    dst->utf8_bytes[k] = (uint8_t)start[k];
}

void build_utf8_luminance_cache(const char *ascii_chars, utf8_char_t cache[256]) {
  if (!ascii_chars || !cache)
    return;
  const char *start[256];
  int blen[256];
  const int n = split_palette(ascii_chars, start, blen);
  if (n == 0)
    return;
  const char *limit = ascii_chars + strlen(ascii_chars);
  for (int i = 0; i < 256; i++) {
    int ci = n > 1 ? (i * (n - 1) + 127) / 255 : 0;
    if (ci >= n)
      ci = n - 1;
    fill_glyph(&cache[i], start[ci], blen[ci], limit);
  }
}

void build_utf8_ramp64_cache(const char *ascii_chars, utf8_char_t cache64[64], uint8_t char_index_ramp[256]) {
  if (!ascii_chars || !cache64 || !char_index_ramp)
    return;
  const char *start[256];
  int blen[256];
  const int n = split_palette(ascii_chars, start, blen);
  if (n == 0)
    return;
  const char *limit = ascii_chars + strlen(ascii_chars);
  for (int i = 0; i < 64; i++) {
    int ci = n > 1 ? (i * (n - 1) + 31) / 63 : 0;
    if (ci >= n)
      ci = n - 1;
    char_index_ramp[i] = (uint8_t)ci;
    fill_glyph(&cache64[i], start[ci], blen[ci], limit);
  }
}

/* get_utf8_palette_cache (common.c:270-377): palette string -> tables, built once, shared by all threads.
 * The reference evicts by a recency/frequency score once 2048 palettes are cached; entries here are 5 KB and
 * are kept for the life of the process up to the same 2048, after which the table stops growing (a palette
 * beyond that is rebuilt into a per-thread slot on every call). */
#define PALCACHE_BUCKETS 256
#define PALCACHE_MAX 2048
typedef struct palcache_node {
  utf8_palette_cache_t tables; /* first: the pointer handed out is the node */
  struct palcache_node *next;
  uint32_t key;
  char palette[256];
} palcache_node_t;
static palcache_node_t *g_palcache[PALCACHE_BUCKETS];
static int g_palcache_count;
static pthread_rwlock_t g_palcache_lock = PTHREAD_RWLOCK_INITIALIZER;

static palcache_node_t *palcache_find(uint32_t key, const char *chars) {
  for (palcache_node_t *n = g_palcache[key % PALCACHE_BUCKETS]; n; n = n->next)
    if (n->key == key && strcmp(n->palette, chars) == 0)
      return n;
  return NULL;
}

utf8_palette_cache_t *get_utf8_palette_cache(const char *ascii_chars) {
  if (!ascii_chars || ascii_chars[0] == '\0' || strlen(ascii_chars) >= sizeof(((palcache_node_t *)0)->palette))
    return NULL;
  uint32_t key = 2166136261u; /* FNV-1a, as the reference keys its table (common.c:275) */
  for (const unsigned char *p = (const unsigned char *)ascii_chars; *p; p++)
    key = (key ^ *p) * 16777619u;
  pthread_rwlock_rdlock(&g_palcache_lock);
  palcache_node_t *n = palcache_find(key, ascii_chars);
  pthread_rwlock_unlock(&g_palcache_lock);
  if (n)
    return &n->tables;
  pthread_rwlock_wrlock(&g_palcache_lock);
  n = palcache_find(key, ascii_chars); /* someone may have built it while the lock was dropped */
  if (!n) {
    static __thread palcache_node_t overflow_slot;
    const bool full = g_palcache_count >= PALCACHE_MAX;
    n = full ? &overflow_slot : (palcache_node_t *)calloc(1, sizeof(*n));
    if (n) {
      memset(n, 0, sizeof(*n));
      build_utf8_luminance_cache(ascii_chars, n->tables.cache);
      build_utf8_ramp64_cache(ascii_chars, n->tables.cache64, n->tables.char_index_ramp);
      n->key = key;
      memcpy(n->palette, ascii_chars, strlen(ascii_chars) + 1);
      if (!full) {
        n->next = g_palcache[key % PALCACHE_BUCKETS];
        g_palcache[key % PALCACHE_BUCKETS] = n;
        g_palcache_count++;
      }
    }
  }
  pthread_rwlock_unlock(&g_palcache_lock);
  return n ? &n->tables : NULL;
}

void ascii_simd_init(void) { /* common.c:576-604: default luminance palette over PALETTE_CHARS_STANDARD */
  static const char std_pal[] = PALETTE_CHARS_STANDARD;
  const size_t len = sizeof(std_pal) - 1;
  for (int i = 0; i < 256; i++) {
    size_t k = ((size_t)i * (len - 1) + 127) / 255;
    if (k >= len)
      k = len - 1;
    g_default_luminance_palette[i] = std_pal[k];
  }
}

/* ---- string padding (ascii.c:457-517, 902-941) --------------------------------------------------------- */
char *ascii_pad_frame_width(const char *frame, size_t pad_left) {
  if (!frame)
    return NULL;
  const size_t n = strlen(frame);
  size_t lines = 1;
  if (pad_left)
    for (size_t i = 0; i < n; i++)
      lines += frame[i] == '\n';
  char *out = (char *)malloc(n + (pad_left ? lines * pad_left : 0) + 1);
  if (!out)
    return NULL;
  if (!pad_left) {
    memcpy(out, frame, n + 1);
    return out;
  }
  char *w = out;
  bool at_bol = true;
  for (size_t i = 0; i < n; i++) {
    if (at_bol) {
      memset(w, ' ', pad_left);
      w += pad_left;
      at_bol = false;
    }
    *w++ = frame[i];
    at_bol = frame[i] == '\n';
  }
  *w = '\0';
  return out;
}

char *ascii_pad_frame_height(const char *frame, size_t pad_top) {
  if (!frame)
    return NULL;
  const size_t n = strlen(frame);
  char *out = (char *)malloc(pad_top + n + 1);
  if (!out)
    return NULL;
  memset(out, '\n', pad_top);
  memcpy(out + pad_top, frame, n + 1);
  return out;
}

/* ---- text-space grid (ascii.c:527-885) ------------------------------------------------------------------ */
/* CSI sequences (ESC [ ... final byte 0x40-0x7E) take no columns */
static int skip_csi(const char *d, int n, int i) {
  for (i += 2; i < n;) {
    const char c = d[i++];
    if (c >= '@' && c <= '~')
      break;
  }
  return i;
}
static int columns_of(const char *d, int n) {
  int cols = 0;
  for (int i = 0; i < n;) {
    if (d[i] == '\033' && i + 1 < n && d[i + 1] == '[') {
      i = skip_csi(d, n, i);
    } else {
      cols++;
      i++;
    }
  }
  return cols;
}
static int bytes_for_columns(const char *d, int n, int want) {
  int cols = 0, i = 0;
  while (i < n && cols < want) {
    if (d[i] == '\033' && i + 1 < n && d[i + 1] == '[') {
      i = skip_csi(d, n, i);
    } else {
      cols++;
      i++;
    }
  }
  return i;
}

static char *space_canvas(int width, int height, size_t *total) {
  const size_t sz = (size_t)width * (size_t)height + (size_t)height + 1;
  char *c = (char *)malloc(sz);
  if (!c)
    return NULL;
  memset(c, ' ', sz - 1);
  c[sz - 1] = '\0';
  for (int r = 0; r < height; r++)
    c[(size_t)r * (size_t)(width + 1) + (size_t)width] = '\n';
  *total = sz;
  return c;
}

char *ascii_create_grid(ascii_frame_source_t *sources, int source_count, int width, int height, size_t *out_size) {
  if (!sources || source_count <= 0 || width <= 0 || height <= 0 || !out_size)
    return NULL;
  size_t total = 0;

  if (source_count == 1) { /* centre the single frame (ascii.c:610-707) */
    char *canvas = space_canvas(width, height, &total);
    if (!canvas)
      return NULL;
    *out_size = total - 1;
    const char *src = sources[0].frame_data;
    const int n = (int)sources[0].frame_size;
    if (!src || n <= 0)
      return canvas;
    int newlines = 0;
    for (int i = 0; i < n; i++)
      newlines += src[i] == '\n';
    int row = (height - newlines) / 2;
    if (row < 0)
      row = 0;
    for (int pos = 0; pos < n && row < height; row++) {
      const int line = pos;
      while (pos < n && src[pos] != '\n')
        pos++;
      const int len = pos - line;
      int left = (width - columns_of(src + line, len)) / 2;
      if (left < 0)
        left = 0;
      const size_t at = (size_t)row * (size_t)(width + 1) + (size_t)left;
      const int take = bytes_for_columns(src + line, len, width - left);
      if (take > 0 && at + (size_t)take < total)
        memcpy(canvas + at, src + line, (size_t)take);
      if (pos < n && src[pos] == '\n')
        pos++;
    }
    return canvas;
  }

  /* choose the column count (ascii.c:712-769); float32 + logf as in the reference */
  float top = -1.0f;
  int cols = 1, rows = source_count;
  for (int c = 1; c <= source_count; c++) {
    const int r = (int)ceil((double)source_count / c);
    if (c * r - source_count > source_count / 2)
      continue;
    const int cw = (width - (c - 1)) / c, ch = (height - (r - 1)) / r;
    if (cw < 10 || ch < 3)
      continue;
    float squareness = 1.0f - fabsf(logf(((float)cw / (float)ch) / 2.0f));
    if (squareness < 0)
      squareness = 0;
    const float fill = (float)source_count / (float)(c * r);
    float score = source_count == 2 ? squareness * 0.9f + fill * 0.1f : squareness * 0.7f + fill * 0.3f;
    if (c == r)
      score += 0.05f;
    if (score > top) {
      top = score;
      cols = c;
      rows = r;
    }
  }
  const int cell_w = (width - (cols - 1)) / cols, cell_h = (height - (rows - 1)) / rows;
  if (cell_w < 10 || cell_h < 3) { /* too small: hand back a copy of the first frame (ascii.c:779-793) */
    char *copy = (char *)malloc(sources[0].frame_size + 1);
    if (!copy)
      return NULL;
    if (sources[0].frame_data && sources[0].frame_size > 0) {
      memcpy(copy, sources[0].frame_data, sources[0].frame_size);
      copy[sources[0].frame_size] = '\0';
      *out_size = sources[0].frame_size;
    } else {
      copy[0] = '\0';
      *out_size = 0;
    }
    return copy;
  }

  char *canvas = space_canvas(width, height, &total);
  if (!canvas)
    return NULL;
  for (int s = 0; s < source_count; s++) {
    const int gr = s / cols, gc = s % cols;
    const int row0 = gr * (cell_h + 1), col0 = gc * (cell_w + 1);
    const char *src = sources[s].frame_data;
    const int n = (int)sources[s].frame_size;
    int pos = 0;
    for (int line_no = 0; pos < n && line_no < cell_h && row0 + line_no < height; line_no++) {
      const int line = pos;
      while (pos < n && src[pos] != '\n')
        pos++;
      const int take = bytes_for_columns(src + line, pos - line, cell_w);
      /* raw bytes are pasted: escape-laden lines may overrun the cell in byte space, exactly as upstream -- but never
       * the canvas: upstream's SAFE_MEMCPY(dest, mixed_size - mixed_pos, ...) refuses such a copy as a whole */
      const size_t at = (size_t)(row0 + line_no) * (size_t)(width + 1) + (size_t)col0;
      if (take > 0 && col0 + columns_of(src + line, take) <= width && (size_t)take <= total - at)
        memcpy(canvas + at, src + line, (size_t)take);
      if (pos < n && src[pos] == '\n')
        pos++;
    }
    const bool right_edge = gc < cols - 1 && col0 + cell_w < width;
    if (right_edge)
      for (int r = row0; r < row0 + cell_h && r < height; r++) {
        const size_t at = (size_t)r * (size_t)(width + 1) + (size_t)(col0 + cell_w);
        if (at < total - 1)
          canvas[at] = '|';
      }
    if (gr < rows - 1 && row0 + cell_h < height) {
      for (int c = col0; c < col0 + cell_w && c < width; c++) {
        const size_t at = (size_t)(row0 + cell_h) * (size_t)(width + 1) + (size_t)c;
        if (at < total - 1)
          canvas[at] = '_';
      }
      if (right_edge) {
        const size_t at = (size_t)(row0 + cell_h) * (size_t)(width + 1) + (size_t)(col0 + cell_w);
        if (at < total - 1)
          canvas[at] = '+';
      }
    }
  }
  canvas[total - 1] = '\0'; /* a paste that ends exactly at the end of the canvas takes the terminator with it; upstream
                               then runs strlen() off the block (ascii.c:883) -- the only place where we differ */
  *out_size = strlen(canvas);
  return canvas;
}

/* ------------------------------------------------------------------------------------------- */
/* lib/video/ascii/rle.c:13-162 and lib/video/ascii/frame_validator.c:13-80                       */
/* ------------------------------------------------------------------------------------------- */
static size_t csi_params_end(const char *in, size_t n, size_t i, uint32_t *last_param) {
  uint32_t param = 0; /* rle.c:33-40: digits accumulate, ';' restarts the parameter */
  while (i < n && ((in[i] >= '0' && in[i] <= '9') || in[i] == ';')) {
    param = in[i] == ';' ? 0u : param * 10u + (uint32_t)(in[i] - '0');
    i++;
  }
  if (last_param)
    *last_param = param;
  return i;
}

char *ansi_expand_rle(const char *input, size_t input_len) {
  if (!input || input_len == 0)
    return NULL;
  outbuf_t ob = {0};
  ob_reserve(&ob, input_len * 2);
  char last_char[5] = " ";
  size_t last_len = 1;
  size_t i = 0;
  while (i < input_len) {
    if (input[i] == '\033' && i + 1 < input_len && input[i + 1] == '[') {
      const size_t start = i;
      uint32_t param = 0;
      i = csi_params_end(input, input_len, i + 2, &param);
      if (i < input_len) { /* a sequence cut off by the end of the input is dropped (rle.c:43) */
        const char final_byte = input[i++];
        if (final_byte == 'b' && param > 0) {
          for (uint32_t r = 0; r < param; r++)
            ob_write(&ob, last_char, last_len);
        } else {
          ob_write(&ob, input + start, i - start);
        }
      }
    } else {
      const unsigned char c = (unsigned char)input[i];
      size_t len = (c & 0xE0) == 0xC0 ? 2 : ((c & 0xF0) == 0xE0 ? 3 : ((c & 0xF8) == 0xF0 ? 4 : 1));
      if (i + len > input_len)
        len = input_len - i;
      ob_write(&ob, input + i, len);
      if (c >= 0x20 && c != 0x7F) {
        memcpy(last_char, input + i, len);
        last_char[len] = '\0';
        last_len = len;
      }
      i += len;
    }
  }
  ob_term(&ob);
  return ob.buf;
}

char *ansi_compress_rle(const char *input, size_t input_len) {
  if (!input || input_len == 0)
    return NULL;
  outbuf_t ob = {0};
  ob_reserve(&ob, input_len);
  size_t i = 0;
  while (i < input_len) {
    if (input[i] == '\033' && i + 1 < input_len && input[i + 1] == '[') {
      const size_t start = i;
      i = csi_params_end(input, input_len, i + 2, NULL);
      if (i < input_len)
        i++;
      ob_write(&ob, input + start, i - start);
    } else {
      const signed char c = (signed char)input[i]; /* the reference compares a plain (signed) char: bytes >= 0x80 never run */
      if (c >= 0x20 && c != 0x7F) {
        size_t run = 1;
        i++;
        while (i < input_len && input[i] == (char)c) {
          run++;
          i++;
        }
        ob_putc(&ob, (char)c);
        if (run > 1 && rep_is_profitable((uint32_t)run)) {
          emit_rep(&ob, (uint32_t)(run - 1));
        } else {
          for (size_t k = 1; k < run; k++)
            ob_putc(&ob, (char)c);
        }
      } else {
        ob_putc(&ob, (char)c);
        i++;
      }
    }
  }
  ob_term(&ob);
  return ob.buf;
}

static size_t final_reset_pos(const char *d, size_t n) { /* frame_validator.c:13-30: the LAST ESC[0m */
  if (!d || n < 4)
    return SIZE_MAX;
  for (size_t i = n - 4 + 1; i-- > 0;)
    if (memcmp(d + i, "\033[0m", 4) == 0)
      return i;
  return SIZE_MAX;
}

bool frame_validate_integrity(const char *frame_data, size_t frame_size) {
  if (!frame_data || frame_size == 0)
    return false;
  const size_t pos = final_reset_pos(frame_data, frame_size);
  return pos != SIZE_MAX && pos + 4 == frame_size; /* no reset at all, or bytes behind the last one: invalid */
}

size_t frame_get_valid_end(const char *frame_data, size_t frame_size) {
  if (!frame_data || frame_size < 4)
    return frame_size;
  const size_t pos = final_reset_pos(frame_data, frame_size);
  return pos == SIZE_MAX ? frame_size : pos + 4;
}

/* ---- COLOR_FILTER_RAINBOW on a finished frame (lib/video/rgba/color_filter.c:169-243, 348-408) ---------------- */
void color_filter_calculate_rainbow(float time, uint8_t *r, uint8_t *g, uint8_t *b) {
  if (r && g && b)
    achip_rainbow_color(time, r, g, b);
}

char *rainbow_replace_ansi_colors(const char *ansi_string, float time_seconds) {
  static const char lead[] = "\033[38;2;";
  if (!ansi_string)
    return NULL;
  const char *hit = strstr(ansi_string, lead);
  if (!hit)
    return NULL; /* nothing to recolour: the caller keeps its string (color_filter.c:363-365) */
  uint8_t r, g, b;
  achip_rainbow_color(time_seconds, &r, &g, &b);
  char code[24];
  const size_t code_len = (size_t)snprintf(code, sizeof code, "\033[38;2;%u;%u;%um", r, g, b);
  /* first pass: size.  Every SGR shrinks or grows by (code_len - its own length). */
  const size_t n = strlen(ansi_string);
  size_t out_n = n, count = 0;
  for (const char *p = hit; p;) {
    const char *end = strchr(p + 7, 'm');
    if (!end)
      break; /* a lead-in with no terminator behind it: the tail stays as it is */
    out_n = out_n - (size_t)(end + 1 - p) + code_len;
    count++;
    p = strstr(end + 1, lead);
  }
  char *out = (char *)malloc(out_n + 1);
  if (!out)
    return NULL;
  char *d = out;
  const char *src = ansi_string;
  for (const char *p = hit; p && count; count--) {
    const char *end = strchr(p + 7, 'm');
    memcpy(d, src, (size_t)(p - src));
    d += p - src;
    memcpy(d, code, code_len);
    d += code_len;
    src = end + 1;
    p = strstr(src, lead);
  }
  const size_t rest = n - (size_t)(src - ansi_string);
  memcpy(d, src, rest);
  d[rest] = '\0';
  return out;
}
