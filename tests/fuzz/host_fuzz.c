/*
 * host_fuzz.c -- sanitizer fuzz of the host-only C of libasciichat_hip.so (hostutil.c, achip_host.c): the string
 * utilities of the drop-in surface (grid, padding, REP expand/compress, frame validator, palette caches, SGR builders)
 * and the descriptor / geometry / layout helpers, fed with random and malformed input in exact-size heap blocks.
 * Built with -fsanitize=address,undefined by tests/test_host_fuzz.py; any report aborts the run.
 * (This is how the canvas overrun of ascii_create_grid's paste was found.)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "achip_host.h"
#include "asciichat_render.h"

static uint32_t rng = 12345;
static uint32_t rnd(void) {
  rng ^= rng << 13;
  rng ^= rng >> 17;
  rng ^= rng << 5;
  return rng;
}

static char *mkframe(size_t *len) {
  int rows = rnd() % 12 + 1; size_t cap = 4096, n = 0; char *s = malloc(cap);
  for (int r = 0; r < rows; r++) {
    int cols = rnd() % 40;
    for (int c = 0; c < cols && n + 32 < cap; c++) {
      uint32_t k = rnd() % 12;
      if (k == 0) n += (size_t)sprintf(s + n, "\033[38;2;%u;%u;%um", rnd() % 256, rnd() % 256, rnd() % 256);
      else if (k == 1) n += (size_t)sprintf(s + n, "\033[%ub", rnd() % 30);
      else if (k == 2) { memcpy(s + n, "\xe2\x96\x80", 3); n += 3; }
      else if (k == 3) { s[n++] = 27; }
      else s[n++] = (char)(' ' + rnd() % 90);
    }
    if (r + 1 < rows) s[n++] = '\n';
  }
  char *exact = malloc(n ? n : 1); memcpy(exact, s, n); free(s); *len = n; return exact;
}

static void fuzz_strings(int iters) {
  for (int it = 0; it < iters; it++) {
    size_t n = rnd() % 64 + 1;
    char *buf = malloc(n); /* exact-size heap block: overreads trip ASan */
    for (size_t i = 0; i < n; i++) {
      uint32_t r = rnd() % 10;
      buf[i] = r == 0 ? 27 : r == 1 ? '[' : r == 2 ? 'b' : r == 3 ? '0' + rnd() % 10 : r == 4 ? ';' : r == 5 ? (char)(0xC0 + rnd() % 0x38) : r == 6 ? 'a' : r == 7 ? 'm' : (char)(rnd() & 0xFF);
    }
    char *a = ansi_expand_rle(buf, n); free(a);
    char *c = ansi_compress_rle(buf, n); free(c);
    (void)frame_validate_integrity(buf, n); (void)frame_get_valid_end(buf, n);
    uint32_t w, h; const uint8_t *px;
    (void)achip_frame_blob_parse(buf, n, it & 1, &w, &h, &px);
    free(buf);
  }
  const char *pals[] = {"   ...',;:clodxkO0KXNWM", "   \xe2\x96\x91\xe2\x96\x91\xe2\x96\x92", "@", "ab"};
  for (int i = 0; i < iters / 50 + 10; i++) { char p[16]; snprintf(p, sizeof p, " .%d#", i); if (!get_utf8_palette_cache(i % 7 ? p : pals[i % 4])) abort(); }
}

static void fuzz_grid(int iters) {
  for (int it = 0; it < iters; it++) {
    int nsrc = rnd() % 10;
    ascii_frame_source_t src[10]; char *bufs[10];
    for (int i = 0; i < nsrc; i++) { size_t l; bufs[i] = mkframe(&l); src[i].frame_data = bufs[i]; src[i].frame_size = l; }
    size_t out = 0;
    char *g = ascii_create_grid(nsrc ? src : NULL, nsrc, (int)(rnd() % 200), (int)(rnd() % 70), &out);
    free(g);
    if (nsrc) {
      /* the pad helpers and the rainbow pass take NUL-terminated strings */
      size_t l = src[0].frame_size; char *z = malloc(l + 1); memcpy(z, bufs[0], l); z[l] = 0;
      for (size_t i = 0; i < l; i++) if (!z[i]) z[i] = 'x';
      { /* every ESC[38;2;..m becomes the colour of the moment; truncated lead-ins at the end stay */
        const float t = (float)(rnd() % 100000) / 997.0f - 3.0f;
        char *rb = rainbow_replace_ansi_colors(z, t);
        if (rb) {
          uint8_t cr, cg, cb; char code[32];
          color_filter_calculate_rainbow(t, &cr, &cg, &cb);
          snprintf(code, sizeof code, "\033[38;2;%u;%u;%um", cr, cg, cb);
          for (const char *q = rb; (q = strstr(q, "\033[38;2;")) != NULL; q += 7)
            if (strchr(q + 7, 'm') && strncmp(q, code, strlen(code)) != 0) abort();
          free(rb);
        } else if (strstr(z, "\033[38;2;")) abort();
      }
      char *p = ascii_pad_frame_width(z, rnd() % 9); free(p);
      p = ascii_pad_frame_height(z, rnd() % 5); free(p);
      free(z);
    }
    for (int i = 0; i < nsrc; i++) free(bufs[i]);
    ssize_t ow, oh; aspect_ratio(rnd() % 4000 + 1, rnd() % 2200 + 1, rnd() % 500 + 1, rnd() % 200 + 1, it & 1, &ow, &oh);
    achip_frame_t f; (void)achip_frame_setup(&f, (const uint8_t *)0x1000, rnd() % 4000, rnd() % 2300, (ssize_t)(rnd() % 600) - 5, (ssize_t)(rnd() % 300) - 5, rnd() % 3, it & 1, it & 2, it & 4);
  }
}

static void fuzz_geometry(int iters) {
  int caps[5] = {4096, 2048, 1024, 256, 2048};
  for (int it = 0; it < iters; it++) {
    /* palettes incl. malformed UTF-8 */
    char pal[40]; int pl = rnd() % 30 + 1;
    for (int i = 0; i < pl; i++) pal[i] = (char)((rnd() % 3) ? ' ' + rnd() % 90 : 0x80 + rnd() % 0x7F);
    pal[pl] = 0;
    for (int i = 0; i < pl; i++) if (!pal[i]) pal[i] = 'x';
    achip_lut_t lut; (void)achip_lut_build(pal, &lut); (void)achip_palette_ascii_only(pal);
    utf8_char_t c256[256], c64[64]; uint8_t ramp[256];
    build_utf8_luminance_cache(pal, c256); build_utf8_ramp64_cache(pal, c64, ramp);
    /* geometry policy on random batches */
    achip_frame_t fr[8]; int n = rnd() % 8 + 1;
    for (int i = 0; i < n; i++) {
      memset(&fr[i], 0, sizeof fr[i]);
      (void)achip_frame_setup(&fr[i], (const uint8_t *)0x1000, rnd() % 3900 + 1, rnd() % 2200 + 1, rnd() % 500 + 1, rnd() % 150 + 1, rnd() % 3, it & 1, it & 2, 0);
      if (fr[i].out_w <= 0) { fr[i].out_w = 1; fr[i].out_h = 1; }
    }
    achip_uniform_t uni; /* descriptors from independent setups are uniform only when they really are */
    if (achip_frames_uniform(fr, n, &uni)) {
      for (int i = 0; i < n; i++) {
        achip_frame_t a = uni.f; a.src = uni.f.src + (int64_t)i * uni.src_pitch;
        if (memcmp(&a, &fr[i], sizeof a) != 0) abort();
      }
    }
    (void)achip_frame_set_rainbow(&fr[0], (float)(rnd() % 9000) / 100.0f);
    if (!(fr[0].ops & ACHIP_OP_FG_OVERRIDE) || (fr[0].ops & ACHIP_OP_TINT)) abort();
    (void)achip_frame_set_display_ops(&fr[0], it & 1, it & 2, rnd() % 12);
    (void)achip_frame_set_dither_style(&fr[0], it & 4, !(it & 4) && (it & 8));
    int v, p, r;
    (void)achip_choose_geometry(rnd() % 10, fr, n, it & 1, caps, rnd() % 300 + 1, (int)(rnd() % 12) - 2, (int)(rnd() % 7) - 2 > 4 ? 4 : (int)(rnd() % 6) - 1, &v, &p, &r);
    for (int m = 0; m < 10; m++) (void)achip_out_bound(m, &fr[0]);
    /* layouts / composite geometry */
    int sw[9], sh[9]; const uint8_t *ptr[9]; int k = rnd() % 10;
    for (int i = 0; i < 9; i++) { sw[i] = rnd() % 2000 + 1; sh[i] = rnd() % 1200 + 1; ptr[i] = (const uint8_t *)0x2000; }
    int cols, rows; achip_grid_layout(sw, sh, k, rnd() % 300, rnd() % 120, &cols, &rows);
    achip_composite_t comp; if (k > 0) (void)achip_composite_setup(&comp, ptr, sw, sh, k > 9 ? 9 : k, rnd() % 300 + 1, rnd() % 100 + 1);
    /* SGR builders and the RLE context */
    char buf[128]; char *e = append_truecolor_fg_bg(buf, rnd(), rnd(), rnd(), rnd(), rnd(), rnd()); e = append_256color_bg(e, rnd()); e = append_16color_fg(e, rnd()); (void)e;
    char out[256]; ansi_rle_context_t ctx; ansi_rle_init(&ctx, out, rnd() % 200 + 1, rnd() % 3);
    for (int i = 0; i < 40; i++) ansi_rle_add_pixel(&ctx, rnd() % 4, rnd() % 4, rnd() % 4, 'a' + rnd() % 3);
    ansi_rle_finish(&ctx);
    outbuf_t ob = {0}; ob_u8(&ob, rnd()); ob_u32(&ob, rnd()); emit_rep(&ob, rnd() % 5000); emit_set_fg(&ob, 1, 2, 3); ob_term(&ob); free(ob.buf);
  }
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  fuzz_strings(iters);
  fuzz_grid(iters / 10 + 1);
  fuzz_geometry(iters / 2 + 1);
  puts("host fuzz ok");
  return 0;
}
