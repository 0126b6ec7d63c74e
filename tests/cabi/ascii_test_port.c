/*
 * ascii_test_port.c -- a plain-C caller of the drop-in layer, the way a relinked reference binary calls it.
 *
 * Includes ONLY include/asciichat_render.h and links ascii-chat_amd/libasciichat_hip.so.  It replays what the
 * reference's own unit tests assert at this boundary (tests/unit/video/ascii_test.c in the reference tree; Criterion
 * is not installed here, so the assertions are restated as plain checks):
 *   :102-115   ascii_convert of a 2x2 colour image -> non-NULL, non-empty
 *   :118-156   NULL image / NULL palette / NULL luminance palette -> NULL
 *   :467-476, :513-522   ascii_pad_frame_width / _height with pad 0 -> an equal copy; NULL frame -> NULL
 *   :584-637   ascii_create_grid: NULL sources, zero count, NULL out_size, zero dimensions -> NULL;
 *              two empty sources at 2x1 -> non-NULL with out_size == 0
 *   :741-765   every entry point with invalid parameters -> NULL
 *   :864-908   N x N grey gradient through ascii_convert(mono): line count == N for N in {2,4,8,16,32,64}
 * plus the error channel of SURVEY 8(b): a failing call reaches asciichat_set_errno_with_message when the host
 * process defines it (this program does, standing in for libasciichat's asciichat_errno.c:166-181).
 *
 * Exit status 0 = every check passed; prints one line per failed check.  Needs a GPU (the library has no CPU path).
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "asciichat_render.h"

static int g_failed = 0, g_checks = 0;
#define CHECK(cond, ...)                                                                                               \
  do {                                                                                                                 \
    g_checks++;                                                                                                        \
    if (!(cond)) {                                                                                                     \
      g_failed++;                                                                                                      \
      printf("FAIL %s:%d: ", __FILE__, __LINE__);                                                                      \
      printf(__VA_ARGS__);                                                                                             \
      printf("\n");                                                                                                    \
    }                                                                                                                  \
  } while (0)

/* the host binary's error channel (libasciichat defines this; the drop-in library binds to it weakly) */
static int g_errno_calls = 0, g_errno_last_code = 0;
static char g_errno_last_msg[256];
void asciichat_set_errno_with_message(int code, const char *file, int line, const char *function, const char *format, ...) {
  (void)file;
  (void)line;
  (void)function;
  va_list ap;
  va_start(ap, format);
  vsnprintf(g_errno_last_msg, sizeof(g_errno_last_msg), format, ap);
  va_end(ap);
  g_errno_calls++;
  g_errno_last_code = code;
}

static void make_luminance_palette(const char *palette, char out[257]) {
  const size_t n = strlen(palette);
  for (int i = 0; i < 256; i++)
    out[i] = palette[(size_t)i % n];
  out[256] = '\0';
}

static int count_lines(const char *s) {
  int lines = 0;
  const size_t len = strlen(s);
  for (const char *p = s; *p; p++)
    lines += *p == '\n';
  if (len > 0 && s[len - 1] != '\n')
    lines++;
  return lines;
}

int main(void) {
  const char *palette = "@#$%&*+=-:. ";
  char lum[257];
  make_luminance_palette(palette, lum);

  { /* basic conversion of a 2x2 colour image */
    image_t *img = image_new(2, 2);
    CHECK(img != NULL, "image_new(2,2)");
    if (img) {
      img->pixels[0] = (rgb_pixel_t){255, 0, 0};
      img->pixels[1] = (rgb_pixel_t){0, 255, 0};
      img->pixels[2] = (rgb_pixel_t){0, 0, 255};
      img->pixels[3] = (rgb_pixel_t){255, 255, 255};
      char *r = ascii_convert(img, 4, 4, true, false, false, palette, lum);
      CHECK(r != NULL && strlen(r) > 0, "ascii_convert(2x2 colour) must return a non-empty string");
      free(r);
      image_destroy(img);
    }
  }
  { /* NULL arguments -> NULL, and the error reaches the host's errno hook */
    const int before = g_errno_calls;
    CHECK(ascii_convert(NULL, 4, 4, false, false, false, palette, lum) == NULL, "NULL image");
    CHECK(g_errno_calls > before, "a failing call must reach asciichat_set_errno_with_message (SURVEY 8b)");
    CHECK(g_errno_last_code == 86 /* ERROR_INVALID_PARAM */, "errno code %d, want 86", g_errno_last_code);
    image_t *img = image_new(4, 4);
    CHECK(img != NULL, "image_new(4,4)");
    if (img) {
      CHECK(ascii_convert(img, 4, 4, false, false, false, NULL, lum) == NULL, "NULL palette");
      CHECK(ascii_convert(img, 4, 4, false, false, false, palette, NULL) == NULL, "NULL luminance palette");
      image_destroy(img);
    }
    CHECK(image_new(0, 0) == NULL, "image_new(0,0) must fail");
  }
  { /* padding helpers */
    const char *frame = "Hello\nWorld\nTest";
    char *w = ascii_pad_frame_width(frame, 0);
    CHECK(w && strcmp(w, frame) == 0, "pad_frame_width(0) must return an equal copy");
    free(w);
    char *h = ascii_pad_frame_height(frame, 0);
    CHECK(h && strcmp(h, frame) == 0, "pad_frame_height(0) must return an equal copy");
    free(h);
    CHECK(ascii_pad_frame_width(NULL, 5) == NULL && ascii_pad_frame_height(NULL, 2) == NULL, "NULL frame");
    char *e = ascii_pad_frame_width("", 5);
    CHECK(e && strlen(e) == 0, "pad_frame_width of an empty frame is empty");
    free(e);
    char *p3 = ascii_pad_frame_width("Hello", 3);
    CHECK(p3 && strlen(p3) > 5, "pad_frame_width(3) grows the line");
    free(p3);
    char *p2 = ascii_pad_frame_height(frame, 2);
    CHECK(p2 && strlen(p2) > strlen(frame), "pad_frame_height(2) grows the frame");
    free(p2);
  }
  { /* text grid */
    ascii_frame_source_t src[2];
    src[0].frame_data = "Hello\nWorld";
    src[0].frame_size = strlen(src[0].frame_data);
    src[1].frame_data = "Test\nGrid";
    src[1].frame_size = strlen(src[1].frame_data);
    size_t out_size = 123;
    CHECK(ascii_create_grid(NULL, 2, 2, 1, &out_size) == NULL, "grid: NULL sources");
    CHECK(ascii_create_grid(src, 0, 2, 1, &out_size) == NULL, "grid: zero count");
    CHECK(ascii_create_grid(src, 2, 2, 1, NULL) == NULL, "grid: NULL out_size");
    CHECK(ascii_create_grid(src, 2, 0, 0, &out_size) == NULL, "grid: zero dimensions");
    CHECK(ascii_create_grid(NULL, -1, -1, -1, &out_size) == NULL, "grid: invalid everything");
    ascii_frame_source_t empty[2] = {{"", 0}, {"", 0}};
    char *g = ascii_create_grid(empty, 2, 2, 1, &out_size);
    CHECK(g != NULL && out_size == 0, "grid of two empty sources at 2x1: non-NULL, out_size 0 (got %zu)", out_size);
    free(g);
    char *one = ascii_create_grid(src, 1, 1, 1, &out_size);
    CHECK(one != NULL && out_size > 0, "grid of one source at 1x1");
    free(one);
  }
  { /* invalid parameters everywhere */
    CHECK(ascii_convert(NULL, -1, -1, false, false, false, NULL, NULL) == NULL, "ascii_convert invalid");
    CHECK(ascii_convert_with_capabilities(NULL, -1, -1, NULL, false, false, NULL) == NULL, "with_capabilities invalid");
    CHECK(ascii_pad_frame_width(NULL, 0) == NULL && ascii_pad_frame_height(NULL, 0) == NULL, "pad invalid");
  }
  { /* N x N grey gradients: one text line per image row */
    const int sizes[] = {2, 4, 8, 16, 32, 64};
    for (size_t k = 0; k < sizeof(sizes) / sizeof(sizes[0]); k++) {
      const int n = sizes[k];
      image_t *img = image_new((size_t)n, (size_t)n);
      CHECK(img != NULL, "image_new(%d,%d)", n, n);
      if (!img)
        continue;
      for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
          const int den = n + n - 2;
          const int v = den > 0 ? (x + y) * 255 / den : 128;
          img->pixels[y * n + x] = (rgb_pixel_t){(uint8_t)v, (uint8_t)v, (uint8_t)v};
        }
      char *r = ascii_convert(img, n, n, false, false, false, palette, lum);
      CHECK(r != NULL, "ascii_convert(%dx%d mono)", n, n);
      if (r)
        CHECK(count_lines(r) == n, "%dx%d: %d lines", n, n, count_lines(r));
      free(r);
      image_destroy(img);
    }
  }
  { /* capabilities entry point on a real frame: every colour level x render mode returns a string ending as the
       reference's frames end (mono: no reset; the others: ESC[0m somewhere in the tail) */
    image_t *img = image_new(64, 48);
    if (img) {
      for (int i = 0; i < 64 * 48; i++)
        img->pixels[i] = (rgb_pixel_t){(uint8_t)(i * 7), (uint8_t)(i * 13), (uint8_t)(i * 29)};
      for (int cl = 0; cl <= 3; cl++)
        for (int rm = 0; rm <= 2; rm++) {
          terminal_capabilities_t caps;
          memset(&caps, 0, sizeof(caps));
          caps.color_level = (terminal_color_mode_t)cl;
          caps.render_mode = (render_mode_t)rm;
          char *r = ascii_convert_with_capabilities(img, 40, 12, &caps, true, false, "   ...',;:clodxkO0KXNWM");
          CHECK(r != NULL && strlen(r) > 0, "with_capabilities(color %d, mode %d)", cl, rm);
          if (r && cl > 0)
            CHECK(strstr(r, "\033[0m") != NULL, "colour frame (color %d, mode %d) carries a reset", cl, rm);
          free(r);
        }
      image_destroy(img);
    }
  }
  printf("%s: %d checks, %d failed\n", g_failed ? "FAILED" : "ok", g_checks, g_failed);
  return g_failed ? 1 : 0;
}
