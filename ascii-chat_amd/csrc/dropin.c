/*
 * dropin.c -- the reference's render entry points (asciichat_render.h) on top of the GPU path.
 *
 * Every call is a 1-frame batch: the calling thread owns a HIP stream and one pinned, device-mapped
 * block holding [frame descriptor | length | output slab].  The kernel reads the descriptor and writes
 * length + bytes straight into that block, so a call is: (stage source if it is not pool-pinned) ->
 * one kernel launch -> one stream sync -> malloc+memcpy of the result string the caller will free().
 *
 * Host logic here is only what the reference does outside its pixel loops: argument checks with the
 * same NULL conditions, aspect fit, padding sizes, mode dispatch.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <limits.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "achip_host.h"
#include "asciichat_hip.h"
#include "asciichat_render.h"
#include "hip_launch.h"
#include "internal.h"
#include "render_variants.h"

/* ------------------------------------------------------------------------------------------- */
/* per-thread GPU context                                                                        */
/* ------------------------------------------------------------------------------------------- */
#define PIN_DESC_OFF 0
#define PIN_LEN_OFF 64
#define PIN_OUT_OFF 128
#define DROPIN_INPLACE_MAX ((size_t)64 << 10) /* staged images up to this size are read in place (mapped pinned memory) */
#define DROPIN_POLLS 20000u /* hipStreamQuery polls (~0.3 us each) before the wait turns into a blocking synchronize */
#define PIN_KEEP_MAX ((size_t)32 << 20) /* pinned output blocks above 32 MiB are released after the call that needed them */

typedef struct {
  hipStream_t stream;
  uint8_t *pin;      /* host address of the pinned block */
  uint8_t *pin_dev;  /* its device alias                 */
  size_t pin_cap;
  uint8_t *stage;    /* device staging for sources that are not device-visible */
  size_t stage_cap;
  uint8_t *hstage;   /* pinned, device-mapped host staging: the sampled pixels of such a source */
  uint8_t *hstage_dev; /* its device alias: small staged images are read in place, larger ones sent with one DMA */
  size_t hstage_cap;
  uint8_t *scratch;  /* device scratch for image_resize() destinations */
  size_t scratch_cap;
  unsigned long long *part_sync; /* hand-off words of multi-workgroup frames (<= 2160 parts), zeroed once */
  uint32_t epoch;
} tls_ctx_t;

static pthread_key_t g_tls_key;
static pthread_once_t g_tls_once = PTHREAD_ONCE_INIT;

static void tls_destroy(void *p) {
  tls_ctx_t *c = (tls_ctx_t *)p;
  if (!c)
    return;
  if (c->stream)
    (void)hipStreamDestroy(c->stream);
  if (c->pin)
    (void)hipHostFree(c->pin);
  if (c->stage)
    (void)hipFree(c->stage);
  if (c->hstage)
    (void)hipHostFree(c->hstage);
  if (c->scratch)
    (void)hipFree(c->scratch);
  if (c->part_sync)
    (void)hipFree(c->part_sync);
  free(c);
}
static void tls_make_key(void) { pthread_key_create(&g_tls_key, tls_destroy); }

static tls_ctx_t *tls_get(void) {
  pthread_once(&g_tls_once, tls_make_key);
  tls_ctx_t *c = (tls_ctx_t *)pthread_getspecific(g_tls_key);
  if (c)
    return c;
  if (achip_require_device())
    return NULL;
  c = (tls_ctx_t *)calloc(1, sizeof(*c));
  if (!c)
    return NULL;
  if (achip_hip_check((int)hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate")) {
    free(c);
    return NULL;
  }
  pthread_setspecific(g_tls_key, c);
  return c;
}

static int ensure_pin(tls_ctx_t *c, size_t need) {
  if (c->pin_cap >= need)
    return 0;
  if (c->pin)
    (void)hipHostFree(c->pin);
  c->pin = NULL;
  c->pin_cap = 0;
  /* head room only while the block is small: one image_print_color of a 3840x2160 image bounds its output at
   * ~320 MB, and 1.5x of that pinned per calling thread is how a server runs out of pinnable memory */
  size_t cap = need <= PIN_KEEP_MAX ? need + need / 2 : need;
  void *host = NULL, *dev = NULL;
  if (achip_hip_check((int)hipHostMalloc(&host, cap, hipHostMallocMapped), "hipHostMalloc(output)"))
    return -1;
  if (hipHostGetDevicePointer(&dev, host, 0) != hipSuccess)
    dev = host;
  c->pin = (uint8_t *)host;
  c->pin_dev = (uint8_t *)dev;
  c->pin_cap = cap;
  return 0;
}

/* after a call: a block far above what a render thread's frames need does not stay pinned until thread exit */
static void trim_pin(tls_ctx_t *c) {
  if (c->pin_cap <= PIN_KEEP_MAX)
    return;
  (void)hipHostFree(c->pin);
  c->pin = NULL;
  c->pin_dev = NULL;
  c->pin_cap = 0;
}

static int ensure_dev(uint8_t **buf, size_t *cap, size_t need) {
  if (*cap >= need)
    return 0;
  if (*buf)
    (void)hipFree(*buf);
  *buf = NULL;
  *cap = 0;
  if (achip_hip_check((int)hipMalloc((void **)buf, need + need / 4), "hipMalloc(staging)"))
    return -1;
  *cap = need + need / 4;
  return 0;
}

/* device-visible address of `bytes` bytes of host pixels: pool-pinned frames are read in place,
 * anything else is copied to the thread's staging buffer first */
static const uint8_t *resolve_source(tls_ctx_t *c, const void *host_px, size_t bytes) {
  const void *alias = achip_pool_device_ptr(host_px);
  if (alias)
    return (const uint8_t *)alias;
  if (ensure_dev(&c->stage, &c->stage_cap, bytes))
    return NULL;
  if (achip_hip_check((int)hipMemcpyAsync(c->stage, host_px, bytes, hipMemcpyHostToDevice, c->stream),
                      "hipMemcpyAsync(source frame)"))
    return NULL;
  return c->stage;
}

/* A pageable source is not uploaded whole: the renderer point-samples out_h of its src_h rows (and out_w of the src_w
 * pixels of each), so only that part is gathered into pinned staging and sent with one DMA -- 138 KB of rows, or 5.6 KB of
 * samples, instead of 6.2 MB for 1080p -> 80x24 (achip_stage_gather: rows when the frame is more than half as wide as the
 * source, rows and columns below that).  The descriptor copy `d` is rewritten to address the compacted image (a flip is
 * folded into the gather).  Pool-pinned sources are read in place as before. */
static const uint8_t *stage_source(tls_ctx_t *c, achip_frame_t *d, size_t src_bytes) {
  const uint8_t *host_px = d->src;
  const void *alias = achip_pool_device_ptr(host_px);
  if (alias)
    return (const uint8_t *)alias;
  int sw, sh;
  const size_t need = d->comp ? 0 : achip_stage_extent(d, &sw, &sh);
  if (!need) /* every pixel is needed (or repeated): upload the image as it is */
    return resolve_source(c, host_px, src_bytes);
  if (c->hstage_cap < need) {
    if (c->hstage)
      (void)hipHostFree(c->hstage);
    c->hstage = NULL;
    c->hstage_dev = NULL;
    c->hstage_cap = 0;
    void *dev = NULL;
    if (achip_hip_check((int)hipHostMalloc((void **)&c->hstage, need + need / 4 + 16, hipHostMallocMapped),
                        "hipHostMalloc(pixel staging)"))
      return NULL;
    if (hipHostGetDevicePointer(&dev, c->hstage, 0) != hipSuccess)
      dev = c->hstage;
    c->hstage_dev = (uint8_t *)dev;
    c->hstage_cap = need + need / 4;
  }
  const achip_frame_t orig = *d;
  achip_stage_gather(&orig, host_px, c->hstage, d);
  /* a few KB of densely packed samples: the kernel reads them where they are (the DMA's set-up costs more than the
   * kernel's PCIe reads: 23 -> 6 us of issue time per launch in combine.c) */
  if (need <= DROPIN_INPLACE_MAX)
    return c->hstage_dev;
  if (ensure_dev(&c->stage, &c->stage_cap, need + 16))
    return NULL;
  if (achip_hip_check((int)hipMemcpyAsync(c->stage, c->hstage, need, hipMemcpyHostToDevice, c->stream),
                      "hipMemcpyAsync(sampled pixels)"))
    return NULL;
  return c->stage;
}

#define DROPIN_MAX_PARTS 2160 /* IMAGE_MAX_HEIGHT text rows at one row per part */

static int device_cu_count(void) {
  static int cached_cus = 0; /* idempotent: every thread that finds 0 stores the same value */
  int cached = __atomic_load_n(&cached_cus, __ATOMIC_RELAXED);
  if (!cached) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cached = n;
    __atomic_store_n(&cached_cus, cached, __ATOMIC_RELAXED);
  }
  return cached;
}

/* render one frame described by `f` (f->src = HOST pixels, `src_bytes` long) and return the malloc'd string */
static char *render_with_lut(tls_ctx_t *c, const achip_lut_t *lut, int mode, const char *palette, achip_frame_t *f,
                             size_t src_bytes) {
  /* what the kernel will see: the image as it is, or the pixels the target samples gathered into an image of their own
   * (stage_source).  The geometry is chosen for THAT descriptor -- a 1 x 1080 source sampled by a 200 x 1 target becomes a
   * 1 x 1 image, which needs the kernels' general sampler, and not every geometry carries it (round 5: rows geometry 26) */
  achip_frame_t staged = *f;
  const uint8_t *src_dev = stage_source(c, &staged, src_bytes);
  if (!src_dev)
    return NULL;
  /* a single frame on a 256-CU device: cut it into row bands so that many workgroups share it */
  int caps[ACHIP_VARIANT_COUNT], variant = -1, parts = 1, rows_per_part = 1;
  for (int v = 0; v < ACHIP_VARIANT_COUNT; v++)
    caps[v] = achip_variant_cap(v);
  /* row bands (the single frame cut over many workgroups that wait for each other) only while few calls are in flight:
   * with many launches from many threads on the GPU at once the bands of one frame can end up waiting behind other
   * launches' waiters (dispatch order across XCDs is undefined); whole-frame launches have no such dependency */
  const int split_request = achip_combine_callers() > 4 ? -1 : 0;
  if (achip_choose_geometry(mode, &staged, 1, achip_palette_ascii_only(palette), caps, device_cu_count(), split_request, -1, &variant,
                            &parts, &rows_per_part) != 0 ||
      variant < 0 || parts > DROPIN_MAX_PARTS) {
    achip_fail(ASCIICHAT_HIP_ERR_NOT_SUPPORTED, "row of %d cells exceeds the kernel chunk", f->pad_left + f->out_w);
    return NULL;
  }
  if (parts > 1 && !c->part_sync) {
    const size_t bytes = (size_t)DROPIN_MAX_PARTS * sizeof(unsigned long long);
    /* cleared on the thread's own (non-blocking) stream: a null-stream hipMemset is not ordered before its kernels */
    if (achip_hip_check((int)hipMalloc((void **)&c->part_sync, bytes), "hipMalloc(part_sync)") ||
        achip_hip_check((int)hipMemsetAsync(c->part_sync, 0, bytes, c->stream), "hipMemsetAsync(part_sync)"))
      return NULL;
  }
  c->epoch = c->epoch + 1u ? c->epoch + 1u : 1u;
  size_t stride = (achip_out_bound(mode, f) + 1 + 15) & ~(size_t)15;
  if (stride > 0xFFFFFFF0u) {
    achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame too large");
    return NULL;
  }
  if (ensure_pin(c, PIN_OUT_OFF + stride))
    return NULL;
  achip_frame_t *desc = (achip_frame_t *)(c->pin + PIN_DESC_OFF);
  *desc = staged;
  desc->src = src_dev;
  volatile uint32_t *len_host = (volatile uint32_t *)(c->pin + PIN_LEN_OFF);
  *len_host = ACHIP_LEN_BADDESC;
  achip_uniform_t uni; /* one frame: its descriptor rides in the kernel arguments (no PCIe read of the pinned copy) */
  (void)achip_frames_uniform(desc, 1, &uni);
  uni.flags = (achip_palette_ascii_only(palette) ? ACHIP_UNIFORM_PALETTE_ASCII : 0u) | ACHIP_UNIFORM_MAX_CELLS(achip_uniform_extent(mode, variant, desc, 1));
  const int generic = (long)desc->src_w * (long)desc->src_h == 1; /* 1x1 sources need the kernels' general sampler */
  if (achip_hip_check(achip_launch_render(mode, variant, generic, (const achip_frame_t *)(c->pin_dev + PIN_DESC_OFF), 1, lut,
                                          c->pin_dev + PIN_OUT_OFF, (uint64_t)stride,
                                          (uint32_t *)(c->pin_dev + PIN_LEN_OFF), NULL, parts, rows_per_part,
                                          parts > 1 ? c->part_sync : NULL, c->epoch, &uni, c->stream),
                      "render kernel launch"))
    return NULL;
  /* the caller is blocked on this frame: poll while it can have a CPU to itself (a blocking synchronize wakes tens of us
   * late), sleep in the runtime otherwise or when the frame takes long */
  hipError_t q = hipErrorNotReady;
  if (!achip_combine_crowded())
    for (unsigned polls = 0; polls < DROPIN_POLLS && (q = hipStreamQuery(c->stream)) == hipErrorNotReady; polls++) {
    }
  if (q == hipErrorNotReady)
    q = hipStreamSynchronize(c->stream);
  if (achip_hip_check((int)q, "hipStreamSynchronize"))
    return NULL;
  const uint32_t len = *len_host;
  if (len >= 0xFFFFFFF0u) {
    achip_fail(ASCIICHAT_HIP_ERR_BUFFER, "render kernel reported %s",
               len == ACHIP_LEN_OVERFLOW ? "output overflow" : "a bad descriptor");
    trim_pin(c);
    return NULL;
  }
  char *out = achip_out_take(c->pin + PIN_OUT_OFF, len);
  trim_pin(c);
  return out;
}

/* ---- where the finished string goes (internal.h) ------------------------------------------------------------------------ */
static __thread achip_out_target_t *tls_out_target;
achip_out_target_t *achip_out_target(void) { return tls_out_target; }
char *achip_out_take(const void *src, size_t len) {
  achip_out_target_t *t = tls_out_target;
  if (t) {
    t->needed = len;
    if (len + 1 > t->cap) {
      achip_fail(ASCIICHAT_HIP_ERR_BUFFER, "the frame needs %zu bytes, the caller's buffer holds %zu", len + 1, t->cap);
      return NULL;
    }
    memcpy(t->buf, src, len);
    t->buf[len] = '\0';
    return t->buf;
  }
  char *out = (char *)malloc(len + 1);
  if (!out) {
    achip_fail(ASCIICHAT_HIP_ERR_MEMORY, "out of memory");
    return NULL;
  }
  memcpy(out, src, len);
  out[len] = '\0';
  return out;
}

static char *render_unpadded_one(int mode, const char *palette, achip_frame_t *f, size_t src_bytes);

/* The kernel models padding as pseudo-cells of the text row, and a row holds at most 4096 cells.  A terminal so
 * wide that padding alone exceeds that (> 4096 columns) is served the way the reference does it: render the
 * unpadded frame, then ascii_pad_frame_width / ascii_pad_frame_height on the finished string (ascii.c:358-385). */
static char *render_one(int mode, const char *palette, achip_frame_t *f, size_t src_bytes) {
  if (f->pad_left + f->out_w <= achip_variant_cap(0))
    return render_unpadded_one(mode, palette, f, src_bytes);
  const size_t pad_left = (size_t)f->pad_left, pad_top = (size_t)f->pad_top;
  f->pad_left = 0;
  f->pad_top = 0;
  achip_out_target_t *target = tls_out_target; /* the intermediate strings are malloc blocks whatever the caller asked for */
  tls_out_target = NULL;
  char *plain = render_unpadded_one(mode, palette, f, src_bytes);
  char *wide = plain ? ascii_pad_frame_width(plain, pad_left) : NULL;
  free(plain);
  char *tall = wide ? ascii_pad_frame_height(wide, pad_top) : NULL;
  free(wide);
  tls_out_target = target;
  if (tall && target) {
    char *out = achip_out_take(tall, strlen(tall));
    free(tall);
    return out;
  }
  return tall;
}

static char *render_unpadded_one(int mode, const char *palette, achip_frame_t *f, size_t src_bytes) {
  if (achip_require_device())
    return NULL;
  const unsigned long long t_stats = achip_combine_stats_clock();
  if (!achip_frame_extent_ok(f)) {
    (void)achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "image spans 4 GiB or more (row stride %d)", f->src_stride);
    return NULL;
  }
  const achip_lut_t *lut = NULL;
  if (achip_lut_get(palette, &lut))
    return NULL;
  /* many concurrent callers -- the reference's one render thread per client -- share launches (combine.c) */
  achip_combine_enter();
  int handled = 0;
  char *out = achip_combine_render(mode, palette, lut, f, src_bytes, &handled);
  if (!handled) { /* few callers in flight, or too large for a shared generation: this thread's own stream and staging */
    tls_ctx_t *c = tls_get();
    out = c ? render_with_lut(c, lut, mode, palette, f, src_bytes) : NULL;
  }
  achip_combine_leave();
  achip_lut_put(lut); /* synchronous either way: the tables are free again */
  achip_combine_stats_call(t_stats);
  return out;
}

/* ------------------------------------------------------------------------------------------- */
/* images (lib/video/rgba/image.c:38-253)                                                        */
/* ------------------------------------------------------------------------------------------- */
static bool dims_valid(size_t w, size_t h) { return w && h && w <= IMAGE_MAX_WIDTH && h <= IMAGE_MAX_HEIGHT; }

image_t *image_new(size_t width, size_t height) {
  if (!dims_valid(width, height)) {
    achip_fail(ERROR_INVALID_PARAM, "image dimensions invalid or too large: %zu x %zu", width, height);
    return NULL;
  }
  image_t *im = (image_t *)malloc(sizeof(*im));
  void *px = NULL;
  if (!im || posix_memalign(&px, 64, width * height * sizeof(rgb_pixel_t)) != 0) {
    free(im);
    achip_fail(ERROR_MEMORY, "image_new: out of memory");
    return NULL;
  }
  im->w = (int)width;
  im->h = (int)height;
  im->pixels = (rgb_pixel_t *)px;
  im->alloc_method = IMAGE_ALLOC_SIMD;
  return im;
}

static size_t pooled_size(const image_t *im) {
  return sizeof(image_t) + (size_t)im->w * (size_t)im->h * sizeof(rgb_pixel_t);
}

void image_destroy(image_t *p) {
  if (!p)
    return;
  if (p->alloc_method == IMAGE_ALLOC_POOL) {
    if (p->w <= 0 || p->h <= 0)
      return;
    buffer_pool_free(NULL, p, pooled_size(p));
    return;
  }
  free(p->pixels);
  free(p);
}

image_t *image_new_from_pool(size_t width, size_t height) {
  if (!dims_valid(width, height)) {
    achip_fail(ERROR_INVALID_PARAM, "image_new_from_pool: invalid dimensions %zux%zu", width, height);
    return NULL;
  }
  const size_t total = sizeof(image_t) + width * height * sizeof(rgb_pixel_t);
  image_t *im = (image_t *)buffer_pool_alloc(NULL, total);
  if (!im) {
    achip_fail(ERROR_MEMORY, "image_new_from_pool: allocation of %zu bytes failed", total);
    return NULL;
  }
  im->w = (int)width;
  im->h = (int)height;
  im->pixels = (rgb_pixel_t *)((uint8_t *)im + sizeof(image_t));
  im->alloc_method = IMAGE_ALLOC_POOL;
  return im;
}

void image_destroy_to_pool(image_t *image) {
  if (!image || image->w <= 0 || image->h <= 0)
    return;
  buffer_pool_free(NULL, image, pooled_size(image));
}

void image_clear(image_t *p) {
  if (!p || !p->pixels || p->w <= 0 || p->h <= 0)
    return;
  memset(p->pixels, 0, (size_t)p->w * (size_t)p->h * sizeof(rgb_pixel_t));
}

image_t *image_new_copy(const image_t *source) {
  if (!source)
    return NULL;
  image_t *copy = image_new((size_t)source->w, (size_t)source->h);
  if (copy && source->pixels)
    memcpy(copy->pixels, source->pixels, (size_t)source->w * (size_t)source->h * sizeof(rgb_pixel_t));
  return copy;
}

void image_resize_interpolation(const image_t *source, image_t *dest) {
  if (!source || !dest || !source->pixels || !dest->pixels || source->w <= 0 || source->h <= 0 || dest->w <= 0 ||
      dest->h <= 0) {
    achip_fail(ERROR_INVALID_PARAM, "invalid parameters to image_resize_interpolation");
    return;
  }
  tls_ctx_t *c = tls_get();
  if (!c)
    return;
  const size_t src_bytes = (size_t)source->w * (size_t)source->h * 3u;
  const size_t dst_bytes = (size_t)dest->w * (size_t)dest->h * 3u;
  const uint8_t *src_dev = resolve_source(c, source->pixels, src_bytes);
  if (!src_dev)
    return;
  uint8_t *dst_alias = (uint8_t *)achip_pool_device_ptr(dest->pixels);
  uint8_t *dst_dev = dst_alias;
  if (!dst_dev) {
    if (ensure_dev(&c->scratch, &c->scratch_cap, dst_bytes))
      return;
    dst_dev = c->scratch;
  }
  if (achip_hip_check(achip_launch_resize(src_dev, source->w, source->h, dst_dev, dest->w, dest->h, c->stream),
                      "resize launch"))
    return;
  if (!dst_alias &&
      achip_hip_check((int)hipMemcpyAsync(dest->pixels, dst_dev, dst_bytes, hipMemcpyDeviceToHost, c->stream),
                      "hipMemcpyAsync(resized image)"))
    return;
  (void)achip_hip_check((int)hipStreamSynchronize(c->stream), "hipStreamSynchronize");
}

void image_resize(const image_t *s, image_t *d) {
  if (!s || !d) {
    achip_fail(ERROR_INVALID_PARAM, "image_resize: s or d is NULL");
    return;
  }
  image_resize_interpolation(s, d);
}

/* ------------------------------------------------------------------------------------------- */
/* renderers on an already-sized image (scalar/foreground.h, background.h, halfblock.h)          */
/* ------------------------------------------------------------------------------------------- */
static char *print_identity_ops(int mode, uint32_t ops, const uint8_t *rgb, int w, int h, int stride_bytes,
                                const char *palette) {
  achip_frame_t f;
  if (achip_frame_identity(&f, rgb, w, h) != 0) {
    achip_fail(ERROR_INVALID_PARAM, "invalid dimensions h=%d, w=%d", h, w);
    return NULL;
  }
  f.ops |= ops;
  f.src_stride = stride_bytes > 0 ? stride_bytes : w * 3;
  return render_one(mode, palette, &f, (size_t)f.src_stride * (size_t)(h - 1) + (size_t)w * 3u);
}
static char *print_identity(int mode, const uint8_t *rgb, int w, int h, int stride_bytes, const char *palette) {
  return print_identity_ops(mode, 0u, rgb, w, h, stride_bytes, palette);
}

static char *print_image(int mode, const image_t *p, const char *palette) {
  if (!p || !palette || !p->pixels) {
    achip_fail(ERROR_INVALID_PARAM, "image, pixels or palette is NULL");
    return NULL;
  }
  if (palette[0] == '\0') { /* get_utf8_palette_cache() rejects an empty palette (common.c:274-276) */
    achip_fail(ERROR_INVALID_STATE, "empty palette");
    return NULL;
  }
  return print_identity(mode, (const uint8_t *)p->pixels, p->w, p->h, 0, palette);
}

char *image_print(const image_t *p, const char *palette) { return print_image(ACHIP_MODE_MONO, p, palette); }
char *image_print_color(const image_t *p, const char *palette) { return print_image(ACHIP_MODE_TRUE_FG, p, palette); }
char *image_print_256color(const image_t *image, const char *palette) {
  return print_image(ACHIP_MODE_256_FG, image, palette);
}
char *image_print_16color(const image_t *image, const char *palette) {
  return print_image(ACHIP_MODE_16_FG, image, palette);
}
char *image_print_color_background(const image_t *p, const char *palette) {
  return print_image(ACHIP_MODE_TRUE_BG, p, palette);
}

/* sgr.c:413-436: despite its name this falls through to the scalar renderers */
char *image_print_color_simd(image_t *image, bool use_background_mode, bool use_256color, const char *ascii_chars) {
  if (!image || !ascii_chars) {
    achip_fail(ERROR_INVALID_PARAM, "image_print_color_simd: image or ascii_chars is NULL");
    return NULL;
  }
  if (use_background_mode) /* image_print_16color_dithered_with_background(image, true, ..) */
    return print_image(ACHIP_MODE_16_DITHER_BG, image, ascii_chars);
  return use_256color ? image_print_256color(image, ascii_chars) : image_print_color(image, ascii_chars);
}

/* foreground.c:752-846; with use_background this is what image_print_color_simd dispatches to */
char *image_print_16color_dithered_with_background(const image_t *image, bool use_background, const char *palette) {
  if (!image || !image->pixels || !palette) {
    achip_fail(ERROR_INVALID_PARAM, "image, pixels or palette is NULL");
    return NULL;
  }
  if (image->h <= 0 || image->w <= 0) {
    achip_fail(ERROR_INVALID_STATE, "invalid dimensions h=%d, w=%d", image->h, image->w);
    return NULL;
  }
  if (palette[0] == '\0') {
    achip_fail(ERROR_INVALID_STATE, "empty palette");
    return NULL;
  }
  return print_identity_ops(ACHIP_MODE_16_DITHER_BG, use_background ? 0u : ACHIP_OP_DITHER_FG,
                            (const uint8_t *)image->pixels, image->w, image->h, 0, palette);
}
/* foreground.c:650-750: foreground colour only, glyph through the 64-entry ramp */
char *image_print_16color_dithered(const image_t *image, const char *palette) {
  if (!image || !image->pixels || !palette) {
    achip_fail(ERROR_INVALID_PARAM, "image, pixels or palette is NULL");
    return NULL;
  }
  if (image->h <= 0 || image->w <= 0) {
    achip_fail(ERROR_INVALID_STATE, "invalid dimensions h=%d, w=%d", image->h, image->w);
    return NULL;
  }
  if (palette[0] == '\0') {
    achip_fail(ERROR_INVALID_STATE, "empty palette");
    return NULL;
  }
  return print_identity_ops(ACHIP_MODE_16_DITHER_BG, ACHIP_OP_DITHER_FG | ACHIP_OP_DITHER_RAMP,
                            (const uint8_t *)image->pixels, image->w, image->h, 0, palette);
}

static char *halfblock_entry(int mode, const uint8_t *rgb, int width, int height, int stride_bytes) {
  if (width <= 0 || height <= 0) { /* halfblock.c:50-51 */
    char *e = (char *)malloc(1);
    if (e)
      e[0] = '\0';
    return e;
  }
  if (!rgb)
    return NULL;
  return print_identity(mode, rgb, width, height, stride_bytes, PALETTE_CHARS_STANDARD /* unused by these modes */);
}

char *rgb_to_truecolor_halfblocks_scalar(const uint8_t *rgb, int width, int height, int stride_bytes) {
  return halfblock_entry(ACHIP_MODE_HB_TRUE, rgb, width, height, stride_bytes);
}
char *rgb_to_256color_halfblocks_scalar(const uint8_t *rgb, int width, int height, int stride_bytes,
                                        const char *palette) {
  (void)palette;
  return halfblock_entry(ACHIP_MODE_HB_256, rgb, width, height, stride_bytes);
}
char *rgb_to_16color_halfblocks_scalar(const uint8_t *rgb, int width, int height, int stride_bytes,
                                       const char *palette) {
  (void)palette;
  return halfblock_entry(ACHIP_MODE_HB_16, rgb, width, height, stride_bytes);
}
char *rgb_to_halfblocks_scalar(const uint8_t *rgb, int width, int height, int stride_bytes, const char *palette) {
  (void)palette;
  return halfblock_entry(ACHIP_MODE_HB_MONO, rgb, width, height, stride_bytes);
}

/* ascii.c:955-1002 */
char *image_print_with_capabilities(const image_t *image, const terminal_capabilities_t *caps, const char *palette) {
  if (!image || !caps || !palette)
    return NULL;
  if (caps->render_mode == RENDER_MODE_HALF_BLOCK) {
    const uint8_t *rgb = (const uint8_t *)image->pixels;
    switch (caps->color_level) {
    case TERM_COLOR_TRUECOLOR:
      return rgb_to_truecolor_halfblocks_scalar(rgb, image->w, image->h, 0);
    case TERM_COLOR_256:
      return rgb_to_256color_halfblocks_scalar(rgb, image->w, image->h, 0, palette);
    case TERM_COLOR_16:
      return rgb_to_16color_halfblocks_scalar(rgb, image->w, image->h, 0, palette);
    default:
      return rgb_to_halfblocks_scalar(rgb, image->w, image->h, 0, palette);
    }
  }
  switch (caps->color_level) {
  case TERM_COLOR_TRUECOLOR:
    return image_print_color_simd((image_t *)image, caps->render_mode == RENDER_MODE_BACKGROUND, false, palette);
  case TERM_COLOR_256:
    return image_print_256color(image, palette);
  case TERM_COLOR_16:
    return image_print_16color(image, palette);
  default:
    return image_print(image, palette);
  }
}

/* ------------------------------------------------------------------------------------------- */
/* ascii_convert / ascii_convert_with_capabilities (ascii.c:72-387): resize + render + pad fused  */
/* ------------------------------------------------------------------------------------------- */
static render_mode_t g_option_render_mode = RENDER_MODE_FOREGROUND;
void asciichat_hip_set_option_render_mode(render_mode_t mode) { g_option_render_mode = mode; }

char *ascii_convert_with_capabilities(image_t *original, const ssize_t width, const ssize_t height,
                                      const terminal_capabilities_t *caps, const bool use_aspect_ratio,
                                      const bool stretch, const char *palette_chars) {
  if (!original || !caps) {
    achip_fail(ERROR_INVALID_PARAM, "invalid parameters for ascii_convert_with_capabilities");
    return NULL;
  }
  if (original->w <= 0 || original->w > 10000 || original->h <= 0 || original->h > 10000 || !original->pixels) {
    achip_fail(ERROR_INVALID_PARAM, "invalid original image");
    return NULL;
  }
  if (!palette_chars || palette_chars[0] == '\0') { /* image_print_with_capabilities / palette cache reject these */
    achip_fail(ERROR_INVALID_PARAM, "NULL or empty palette");
    return NULL;
  }
  const int mode = achip_mode_from_caps((int)caps->color_level, (int)caps->render_mode);
  achip_frame_t f;
  if (achip_frame_setup(&f, (const uint8_t *)original->pixels, original->w, original->h, width, height,
                        (int)caps->render_mode, caps->wants_padding, use_aspect_ratio, stretch) != 0) {
    achip_fail(ERROR_INVALID_PARAM, "invalid dimensions for resize: width=%zd, height=%zd", width, height);
    return NULL;
  }
  return render_one(mode, palette_chars, &f, (size_t)original->w * (size_t)original->h * 3u);
}

/* Additive (not in the reference): the same call with the string left in the CALLER's buffer -- no malloc per frame, and a
 * render thread that reuses one buffer per client never touches the allocator (VERDICT r3 "next" 9).  Returns ASCIICHAT_OK
 * and the length in *out_len; ERROR_BUFFER when the frame (+ NUL) does not fit -- *out_len then says what it needs --;
 * the reference's codes for everything ascii_convert_with_capabilities returns NULL for. */
asciichat_error_t ascii_convert_with_capabilities_into(image_t *original, const ssize_t width, const ssize_t height,
                                                       const terminal_capabilities_t *caps, const bool use_aspect_ratio,
                                                       const bool stretch, const char *palette_chars, char *out,
                                                       size_t out_capacity, size_t *out_len) {
  if (out_len)
    *out_len = 0;
  if (!out || out_capacity == 0) {
    achip_fail(ERROR_INVALID_PARAM, "ascii_convert_with_capabilities_into: no output buffer");
    return ERROR_INVALID_PARAM;
  }
  achip_out_target_t target = {out, out_capacity, 0};
  achip_out_target_t *before = tls_out_target;
  tls_out_target = &target;
  char *r = ascii_convert_with_capabilities(original, width, height, caps, use_aspect_ratio, stretch, palette_chars);
  tls_out_target = before;
  if (out_len)
    *out_len = target.needed;
  if (r)
    return ASCIICHAT_OK;
  if (target.needed + 1 > out_capacity)
    return ERROR_BUFFER;
  return !original || !caps || !palette_chars || !palette_chars[0] ? ERROR_INVALID_PARAM : ERROR_INVALID_STATE;
}

char *ascii_convert(image_t *original, const ssize_t width, const ssize_t height, const bool color,
                    const bool _aspect_ratio, const bool stretch, const char *palette_chars,
                    const char luminance_palette[256]) {
  if (!original || !palette_chars || !luminance_palette || !original->pixels) {
    achip_fail(ERROR_INVALID_PARAM, "ascii_convert: invalid parameters");
    return NULL;
  }
  if (palette_chars[0] == '\0' || luminance_palette[0] == '\0') {
    achip_fail(ERROR_INVALID_PARAM, "ascii_convert: empty palette strings");
    return NULL;
  }
  if (original->w <= 0 || original->h <= 0)
    return NULL;
  int mode = ACHIP_MODE_MONO;
  if (color) { /* ascii.c:136-161: mode comes from the global render_mode option */
    if (g_option_render_mode == RENDER_MODE_HALF_BLOCK) {
      mode = ACHIP_MODE_HB_TRUE;
    } else if (g_option_render_mode == RENDER_MODE_BACKGROUND) {
      mode = ACHIP_MODE_16_DITHER_BG;
    } else {
      mode = ACHIP_MODE_TRUE_FG;
    }
  }
  /* same sizing as above except: padding whenever aspect correction is on, and the half-block renderer
   * receives the resized image without height doubling (ascii.c:97-131) */
  ssize_t rw = width, rh = height;
  if (_aspect_ratio)
    aspect_ratio(original->w, original->h, rw, rh, stretch, &rw, &rh);
  if (rw <= 0 || rh <= 0 || rw > IMAGE_MAX_WIDTH || rh > IMAGE_MAX_HEIGHT) {
    achip_fail(ERROR_INVALID_PARAM, "invalid dimensions for resize: width=%zd, height=%zd", rw, rh);
    return NULL;
  }
  achip_frame_t f;
  memset(&f, 0, sizeof(f));
  f.src = (const uint8_t *)original->pixels;
  f.src_w = original->w;
  f.src_h = original->h;
  f.out_w = (int32_t)rw;
  f.out_h = (int32_t)rh;
  if (_aspect_ratio) {
    f.pad_left = (int32_t)(width > rw ? (width - rw) / 2 : 0);
    f.pad_top = (int32_t)(height > rh ? (height - rh) / 2 : 0);
  }
  f.x_ratio = achip_nn_ratio(original->w, (int)rw);
  f.y_ratio = achip_nn_ratio(original->h, (int)rh);
  return render_one(mode, palette_chars, &f, (size_t)original->w * (size_t)original->h * 3u);
}
