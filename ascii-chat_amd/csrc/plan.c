/*
 * plan.c -- the batch layer of the C-ABI (asciichat_hip.h): plan objects, glyph-table cache, device
 * helpers.  Host code is plain C calling the HIP runtime C API; the kernels live in hip_launch.hip.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "achip_host.h"
#include "asciichat_hip.h"
#include "hip_launch.h"
#include "internal.h"
#include "render_variants.h"

/* ------------------------------------------------------------------------------------------- */
/* thread-local error text (stands for the reference's SET_ERRNO context, asciichat_errno.h:311)  */
/* ------------------------------------------------------------------------------------------- */
static _Thread_local char tl_err[256];

/* The reference's own error channel: SET_ERRNO(code, fmt, ...) expands to asciichat_set_errno_with_message(...)
 * (include/ascii-chat/asciichat_errno.h:311-319, lib/asciichat_errno.c:166-181), which fills the thread-local
 * asciichat_errno / asciichat_errno_context that callers read through HAS_ERRNO / GET_ERRNO.  A WEAK reference: in a
 * process that also carries libasciichat (the relinked server and client) the dynamic linker binds
 * it to the real function and every failure below lands in the caller's errno context, exactly as the failures of
 * the functions this library replaces did; in a process without it the reference is NULL and is skipped. */
extern void asciichat_set_errno_with_message(int code, const char *file, int line, const char *function,
                                             const char *format, ...) __attribute__((weak));

int achip_fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(tl_err, sizeof(tl_err), fmt, ap);
  va_end(ap);
  if (code == ASCIICHAT_HIP_ERR_NO_DEVICE || getenv("ASCIICHAT_HIP_VERBOSE"))
    fprintf(stderr, "asciichat_hip: %s\n", tl_err);
  if (asciichat_set_errno_with_message) /* codes are the reference's asciichat_error_t values; the one new code
                                           (no HIP device) is reported as ERROR_INVALID_STATE there */
    asciichat_set_errno_with_message(code == ASCIICHAT_HIP_ERR_NO_DEVICE ? ASCIICHAT_HIP_ERR_INVALID_STATE : code,
                                     "libasciichat_hip", 0, "achip_fail", "%s", tl_err);
  return code;
}

const char *asciichat_hip_last_error(void) { return tl_err; }

int asciichat_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int achip_require_device(void) {
  static int cached_count = -1; /* idempotent: whoever gets there first stores the same value */
  int cached = __atomic_load_n(&cached_count, __ATOMIC_RELAXED);
  if (cached < 0) {
    cached = asciichat_hip_device_count();
    __atomic_store_n(&cached_count, cached, __ATOMIC_RELAXED);
  }
  if (cached <= 0)
    return achip_fail(ASCIICHAT_HIP_ERR_NO_DEVICE,
                      "no HIP device visible: this library has no CPU fallback (built for gfx950)");
  return 0;
}

int achip_hip_check(int e, const char *what) {
  if (e == (int)hipSuccess)
    return 0;
  /* memory exhaustion, bad launches and a missing device are different failures */
  int code = ASCIICHAT_HIP_ERR_NO_DEVICE;
  if (e == (int)hipErrorOutOfMemory || e == (int)hipErrorMemoryAllocation)
    code = ASCIICHAT_HIP_ERR_MEMORY;
  else if (e == (int)hipErrorInvalidValue || e == (int)hipErrorInvalidConfiguration ||
           e == (int)hipErrorInvalidDeviceFunction || e == (int)hipErrorLaunchFailure ||
           e == (int)hipErrorLaunchOutOfResources || e == (int)hipErrorSharedObjectInitFailed)
    code = ASCIICHAT_HIP_ERR_INVALID_STATE;
  return achip_fail(code, "%s failed: %s", what, hipGetErrorString((hipError_t)e));
}

/* ------------------------------------------------------------------------------------------- */
/* glyph-table cache: palette string -> device achip_lut_t (per device)                           */
/* the counterpart of get_utf8_palette_cache (common.c:270-377): built once per palette, shared    */
/* ------------------------------------------------------------------------------------------- */
#define LUT_CACHE_MAX 2048 /* the reference's own cap (common.c: heap eviction at 2048 palettes) */
typedef struct {
  char *palette;
  int device;
  int pins;          /* users between achip_lut_get and achip_lut_put; only unpinned entries are evicted */
  unsigned long age; /* g_lut_clock at the last get */
  achip_lut_t *dev;
} lut_entry_t;
static lut_entry_t g_luts[LUT_CACHE_MAX];
static int g_lut_count;
static unsigned long g_lut_clock;
/* readers (a hit: every drop-in call, twice) share the lock and touch the entry with atomics; building, uploading and
 * recycling an entry take it exclusively.  It was a mutex: at 128 calling threads the two acquisitions per call cost
 * more than the render (profiles/r03_dropin_stats.txt). */
static pthread_rwlock_t g_lut_rw = PTHREAD_RWLOCK_INITIALIZER;

/* Pins the device tables of `palette` (building and uploading them on first use); release with achip_lut_put
 * once no queued work reads them any more. */
int achip_lut_get(const char *palette, const achip_lut_t **out_dev) {
  if (!palette || !palette[0])
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "empty palette");
  int device = 0;
  if (achip_hip_check((int)hipGetDevice(&device), "hipGetDevice"))
    return ASCIICHAT_HIP_ERR_NO_DEVICE;
  pthread_rwlock_rdlock(&g_lut_rw);
  for (int i = 0; i < g_lut_count; i++) {
    if (g_luts[i].device == device && strcmp(g_luts[i].palette, palette) == 0) {
      __atomic_add_fetch(&g_luts[i].pins, 1, __ATOMIC_ACQ_REL);
      __atomic_store_n(&g_luts[i].age, __atomic_add_fetch(&g_lut_clock, 1, __ATOMIC_RELAXED), __ATOMIC_RELAXED);
      *out_dev = g_luts[i].dev;
      pthread_rwlock_unlock(&g_lut_rw);
      return 0;
    }
  }
  pthread_rwlock_unlock(&g_lut_rw);
  pthread_rwlock_wrlock(&g_lut_rw);
  g_lut_clock++;
  for (int i = 0; i < g_lut_count; i++) { /* somebody else may have built it in between */
    if (g_luts[i].device == device && strcmp(g_luts[i].palette, palette) == 0) {
      g_luts[i].pins++;
      g_luts[i].age = g_lut_clock;
      *out_dev = g_luts[i].dev;
      pthread_rwlock_unlock(&g_lut_rw);
      return 0;
    }
  }
  achip_lut_t host;
  if (achip_lut_build(palette, &host) != 0) {
    pthread_rwlock_unlock(&g_lut_rw);
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "bad palette");
  }
  int slot = g_lut_count;
  if (slot == LUT_CACHE_MAX) { /* full: recycle the least recently used entry nobody holds */
    slot = -1;
    for (int i = 0; i < g_lut_count; i++)
      if (g_luts[i].pins == 0 && (slot < 0 || g_luts[i].age < g_luts[slot].age))
        slot = i;
    if (slot < 0) {
      pthread_rwlock_unlock(&g_lut_rw);
      return achip_fail(ASCIICHAT_HIP_ERR_MEMORY, "%d palettes are in use at once", LUT_CACHE_MAX);
    }
    (void)hipFree(g_luts[slot].dev); /* unpinned: no queued work reads it */
    free(g_luts[slot].palette);
    g_luts[slot].dev = NULL;
    g_luts[slot].palette = NULL;
  }
  achip_lut_t *dev = NULL;
  char *copy = strdup(palette);
  int rc = copy ? 0 : achip_fail(ASCIICHAT_HIP_ERR_MEMORY, "out of memory");
  if (!rc)
    rc = achip_hip_check((int)hipMalloc((void **)&dev, sizeof(host)), "hipMalloc(lut)");
  if (!rc)
    rc = achip_hip_check((int)hipMemcpy(dev, &host, sizeof(host), hipMemcpyHostToDevice), "hipMemcpy(lut)");
  if (rc) {
    if (dev)
      (void)hipFree(dev);
    free(copy);
    if (slot < g_lut_count) { /* the recycled slot stays empty: close the gap */
      g_luts[slot] = g_luts[g_lut_count - 1];
      g_lut_count--;
    }
    pthread_rwlock_unlock(&g_lut_rw);
    return rc;
  }
  g_luts[slot].palette = copy;
  g_luts[slot].device = device;
  g_luts[slot].pins = 1;
  g_luts[slot].age = g_lut_clock;
  g_luts[slot].dev = dev;
  if (slot == g_lut_count)
    g_lut_count++;
  *out_dev = dev;
  pthread_rwlock_unlock(&g_lut_rw);
  return 0;
}

void achip_lut_put(const achip_lut_t *dev) {
  if (!dev)
    return;
  pthread_rwlock_rdlock(&g_lut_rw);
  for (int i = 0; i < g_lut_count; i++)
    if (g_luts[i].dev == dev) {
      if (__atomic_sub_fetch(&g_luts[i].pins, 1, __ATOMIC_ACQ_REL) < 0) /* a put without a get: undo, stay at zero */
        __atomic_add_fetch(&g_luts[i].pins, 1, __ATOMIC_ACQ_REL);
      break;
    }
  pthread_rwlock_unlock(&g_lut_rw);
}

/* ------------------------------------------------------------------------------------------- */
/* plans                                                                                         */
/* ------------------------------------------------------------------------------------------- */
struct asciichat_hip_plan {
  int mode;
  int n;
  int variant;      /* resolved geometry */
  int variant_user; /* -1 = automatic */
  int max_wp;
  int has_comp; /* some frame samples a virtual composite */
  int all_dense; /* every source IS the image its target samples (ratio 1.0: the sampled-image ingest, frame_dense.c) */
  int big_src;   /* some source is wider than 1920 pixels (4K frames: a second gather costs more than a pack pass) */
  int parts, rows_per_part, split_request; /* multi-workgroup frames (achip_choose_geometry) */
  int whole_variant; /* the geometry of the wire-stage entry points (frame checksums, exact-length frames: a frame belongs to ONE
                        workgroup there): `variant`, unless that shares frames out over workgroups of the stream kernel */
  long max_cells;                          /* cells of the largest frame (ACHIP_UNIFORM_MAX_CELLS)  */
  int palette_ascii;
  unsigned long long *part_sync; /* n * parts_cap u64 hand-off words, zeroed once */
  int parts_cap;
  uint32_t epoch;
  achip_uniform_t uniform; /* the batch's common descriptor, when it has one (achip_frames_uniform) */
  int concurrency;         /* launches the caller keeps in flight on separate streams (>= 1) */
  int uniform_off;         /* asciichat_hip_plan_set_uniform(plan, 0): always read the device array */
  int fused_crc;           /* -1 = where it is the faster form (default), 0 = never, 1 = wherever the geometry carries it */
  size_t stride;
  achip_frame_t *frames_dev;
  achip_frame_t *frames_pinned; /* staging for async updates */
  int frames_dev_stale;         /* plan_update skipped the upload (a uniform launch carries its descriptor in the kernel
                                   arguments): frames_dev is brought up to date before the first launch that reads it */
  int frames_dma_queued;        /* a DMA out of frames_pinned may still be in flight on the stream of the last update */
  int exact_length;             /* -1 = the packed entry points are ONE launch wherever the plan qualifies (default), 0 = never */
  unsigned long long *pack_cursor; /* two device words of the PACK kernels, zero between launches (allocated on first use) */
  uint32_t *crc_scratch;           /* span registers of the stand-alone checksum pass behind this plan (frames above 128 KB):
                                      the plan's own block instead of a stream-ordered allocation per call -- hipMallocAsync +
                                      hipFreeAsync cost ~25 us of host time per call, four times a small launch's render */
  size_t crc_scratch_words;        /* the layout the block's counters were last zeroed for */
  size_t crc_scratch_cap;          /* words allocated */
  const void *pack_dst_seen;       /* the destination the automatic choice looked at last, and what it was */
  int pack_dst_host;
  const achip_lut_t *lut_dev;
};

static int device_cus(void) {
  static int cached_cus = 0; /* idempotent: every thread that finds 0 stores the same value */
  int cached = __atomic_load_n(&cached_cus, __ATOMIC_RELAXED);
  if (!cached) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cached = n;
    __atomic_store_n(&cached_cus, cached, __ATOMIC_RELAXED);
  }
  return cached;
}

/* a geometry whose multi-workgroup form shares a frame's BLOCKS out over workgroups of a wave-autonomous kernel (stream
 * geometry 18, rows geometries 31 / 32) -- its wire stage and its graph captures launch whole frames instead -- as opposed to the row
 * bands of the phase kernel */
static int variant_shares_out(int v) { return ACHIP_IS_STREAM_VARIANT(v) || ACHIP_IS_ROWS_VARIANT(v); }

static int choose_geometry(asciichat_hip_plan_t *p, const achip_frame_t *frames) {
  int caps[ACHIP_VARIANT_COUNT];
  for (int v = 0; v < ACHIP_VARIANT_COUNT; v++)
    caps[v] = achip_variant_cap(v);
  int variant = -1, parts = 1, rpp = 1;
  const char *env = getenv("ASCIICHAT_HIP_VARIANT");
  const int forced = p->variant_user >= 0 ? p->variant_user : (env && env[0] ? atoi(env) : -1);
  const int cus = device_cus() / (p->concurrency > 1 ? p->concurrency : 1);
  const int known = forced < ACHIP_VARIANT_COUNT || (forced >= ACHIP_STREAM_VARIANT_FIRST && achip_variant_block(forced) > 0);
  int rc = achip_choose_geometry(p->mode, frames, p->n, p->palette_ascii != 0, caps, cus > 0 ? cus : 1,
                                 forced >= 0 && p->split_request == 0 ? -1 : p->split_request, /* explicit geometry alone: whole frames */
                                 known ? forced : -1, &variant, &parts, &rpp);
  if (rc != 0 && forced >= 0 && p->variant_user < 0) /* ASCIICHAT_HIP_VARIANT names a geometry that does not apply to this
                                                        plan's mode or size: the override is process-wide, the plan is
                                                        not -- choose automatically */
    rc = achip_choose_geometry(p->mode, frames, p->n, p->palette_ascii != 0, caps, cus > 0 ? cus : 1, p->split_request, -1,
                               &variant, &parts, &rpp);
  if (rc != 0)
    return -1;
  p->variant = variant;
  p->parts = parts;
  p->rows_per_part = rpp;
  p->max_cells = achip_uniform_extent(p->mode, variant, frames, p->n);
  p->whole_variant = variant;
  if (parts > 1 && variant_shares_out(variant)) { /* what the same plan takes when it must not be shared out */
    int wv = -1, wp = 1, wr = 1;
    if (achip_choose_geometry(p->mode, frames, p->n, p->palette_ascii != 0, caps, cus > 0 ? cus : 1, -1, -1, &wv, &wp, &wr) != 0 ||
        wv < 0 || wp != 1)
      return -1;
    p->whole_variant = wv;
  }
  return 0;
}

/* Measures `frames` under the plan's current settings.  Transactional: everything is computed in a copy
 * and committed only when validation, geometry selection and the hand-off allocation have all succeeded -- after a
 * failed update or set_* the plan renders exactly what it rendered before. */
static int plan_measure(asciichat_hip_plan_t *p, const achip_frame_t *frames) {
  asciichat_hip_plan_t q = *p;
  size_t stride = 0;
  int max_wp = 0;
  q.has_comp = 0;
  q.all_dense = 1;
  q.big_src = 0;
  for (int i = 0; i < q.n; i++) {
    const achip_frame_t *f = &frames[i];
    if (f->comp || (long)f->src_w * (long)f->src_h == 1) /* the kernels' general sampler: composites, 1x1 sources */
      q.has_comp = 1;
    if (f->comp || f->src_w != f->out_w || f->src_h != f->out_h)
      q.all_dense = 0;
    if (!f->comp && f->src_w > 1920)
      q.big_src = 1;
    if (f->out_w <= 0 || f->out_h <= 0 || f->src_w <= 0 || f->src_h <= 0 || f->pad_left < 0 || f->pad_top < 0 ||
        (!f->src && !f->comp))
      return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame %d: bad descriptor", i);
    if (!achip_frame_extent_ok(f))
      return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame %d: source spans 4 GiB or more (row stride %d)", i,
                        f->src_stride);
    size_t b = achip_out_bound(q.mode, f) + 1;
    if (b > stride)
      stride = b;
    if (f->pad_left + f->out_w > max_wp)
      max_wp = f->pad_left + f->out_w;
  }
  /* a multiple of the 128-byte line: in a line-aligned slab every slot starts on a line, so the drains' whole-line
   * stores and the wire stage's 16-byte groups (thread t <-> group t) never straddle one (profiles/r04_rows_floor.txt:
   * 4.4 vs 5.7 TB/s on the write side).  Callers may pass any multiple of 16 >= this: the kernels follow the address. */
  stride = (stride + 127) & ~(size_t)127;
  if (stride > 0xFFFFFFF0u)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame output bound exceeds 4 GiB");
  q.stride = stride;
  q.max_wp = max_wp;
  (void)achip_frames_uniform(frames, q.n, &q.uniform);
  if (choose_geometry(&q, frames) != 0 || q.variant < 0 || achip_variant_cap(q.variant) < max_wp) {
    if (q.variant_user >= 0) /* a geometry forced with set_variant: say why it does not apply */
      return achip_fail(ASCIICHAT_HIP_ERR_NOT_SUPPORTED,
                        "geometry %d does not apply to this plan (mode %d, widest padded row %d cells, %ld cells in the largest "
                        "frame): stream geometries 16-19 take the per-cell modes, rows geometries 24-26 the run-structured "
                        "modes with rows of at most %d cells (27 / 29: rows of up to 4096 / 2560 cells cut into segments, 31 / 32: "
                        "rows of at most 128 / 512 cells; single sources only), 1-2 no half-block mode",
                        q.variant_user, q.mode, max_wp, achip_max_cells(frames, q.n), 64 * 7);
    return achip_fail(ASCIICHAT_HIP_ERR_NOT_SUPPORTED, "padded row of %d cells exceeds the kernel chunk (max %d)",
                      max_wp, achip_variant_cap(0));
  }
  unsigned long long *new_sync = NULL;
  if (q.parts > 1 && q.parts > q.parts_cap) { /* hand-off words for the multi-workgroup frames */
    const size_t bytes = (size_t)q.n * (size_t)q.parts * sizeof(unsigned long long);
    int rc = achip_hip_check((int)hipMalloc((void **)&new_sync, bytes), "hipMalloc(part_sync)");
    if (!rc)
      rc = achip_hip_check((int)hipMemset(new_sync, 0, bytes), "hipMemset(part_sync)");
    if (!rc) /* the launches that will poll these words run on the caller's streams, which do not wait for the null
                stream: make the clear complete before anything can be launched (a rare path: plan creation / growth) */
      rc = achip_hip_check((int)hipDeviceSynchronize(), "hipDeviceSynchronize(part_sync)");
    if (rc) {
      if (new_sync)
        (void)hipFree(new_sync);
      return rc;
    }
    q.part_sync = new_sync;
    q.parts_cap = q.parts;
  }
  if (new_sync && p->part_sync)
    (void)hipFree(p->part_sync); /* synchronises with launches that still poll the old words */
  *p = q;
  return 0;
}

int asciichat_hip_plan_create(asciichat_hip_plan_t **plan, int mode, const char *palette_chars,
                              const achip_frame_t *frames, int n_frames) {
  if (!plan || !frames || n_frames <= 0 || mode < 0 || mode >= ACHIP_MODE_COUNT)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "plan_create: bad arguments");
  *plan = NULL;
  int rc = achip_require_device();
  if (rc)
    return rc;
  /* (first plan of a device only; then a mutex and a look.  A failure here is not the plan's: callers that never checksum
   * must not lose their plan to it, and the wire entry points build the tables themselves and report then -- ADVICE r5) */
  (void)achip_launch_warm_crc_tables();
  asciichat_hip_plan_t *p = (asciichat_hip_plan_t *)calloc(1, sizeof(*p));
  if (!p)
    return achip_fail(ASCIICHAT_HIP_ERR_MEMORY, "out of memory");
  p->mode = mode;
  p->n = n_frames;
  p->variant_user = -1;
  p->fused_crc = -1;
  p->exact_length = -1;
  p->concurrency = 1;
  p->palette_ascii = achip_palette_ascii_only(palette_chars) ? 1 : 0;
  rc = plan_measure(p, frames);
  if (!rc)
    rc = achip_lut_get(palette_chars, &p->lut_dev);
  const size_t bytes = (size_t)n_frames * sizeof(achip_frame_t);
  if (!rc)
    rc = achip_hip_check((int)hipMalloc((void **)&p->frames_dev, bytes), "hipMalloc(frames)");
  if (!rc)
    rc = achip_hip_check((int)hipHostMalloc((void **)&p->frames_pinned, bytes, hipHostMallocDefault),
                         "hipHostMalloc(frames)");
  if (!rc) {
    memcpy(p->frames_pinned, frames, bytes);
    rc = achip_hip_check((int)hipMemcpy(p->frames_dev, p->frames_pinned, bytes, hipMemcpyHostToDevice),
                         "hipMemcpy(frames)");
  }
  if (rc) {
    asciichat_hip_plan_destroy(p);
    return rc;
  }
  *plan = p;
  return 0;
}

/* frames_pinned -> frames_dev on `stream` (launches of a plan are ordered on one stream, updates included) */
static int plan_upload_frames(asciichat_hip_plan_t *p, void *stream) {
  const int rc = achip_hip_check((int)hipMemcpyAsync(p->frames_dev, p->frames_pinned, (size_t)p->n * sizeof(achip_frame_t),
                                                     hipMemcpyHostToDevice, (hipStream_t)stream),
                                 "hipMemcpyAsync(frames)");
  if (!rc) {
    p->frames_dev_stale = 0;
    p->frames_dma_queued = 1;
  }
  return rc;
}
/* before a launch that reads the descriptor array: make it current */
static int plan_frames_current(asciichat_hip_plan_t *p, void *stream) {
  if (!p->frames_dev_stale || asciichat_hip_plan_get_uniform(p))
    return 0;
  return plan_upload_frames(p, stream);
}

int asciichat_hip_plan_update(asciichat_hip_plan_t *p, const achip_frame_t *frames, void *stream) {
  if (!p || !frames)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "plan_update: bad arguments");
  /* the pinned staging copy may still be in flight from the previous update on this stream: wait for it BEFORE the new
   * geometry is committed, so that a failure here leaves the plan as it was */
  int rc = 0;
  if (p->frames_dma_queued)
    rc = achip_hip_check((int)hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
  if (rc)
    return rc;
  p->frames_dma_queued = 0;
  rc = plan_measure(p, frames);
  if (rc)
    return rc;
  memcpy(p->frames_pinned, frames, (size_t)p->n * sizeof(achip_frame_t));
  p->frames_dev_stale = 1;
  /* descriptors that differ only by a constant source pitch (a tick's sampled images in one block, a batch of frames in
   * one slab) travel in the kernel arguments: nothing reads frames_dev, so a tick pays neither the wait nor the DMA */
  if (asciichat_hip_plan_get_uniform(p))
    return 0;
  return plan_upload_frames(p, stream);
}

size_t asciichat_hip_plan_out_stride(const asciichat_hip_plan_t *p) { return p ? p->stride : 0; }

int asciichat_hip_plan_set_variant(asciichat_hip_plan_t *p, int variant) {
  if (!p)
    return ASCIICHAT_HIP_ERR_INVALID_PARAM;
  if (variant >= 0 && achip_variant_block(variant) <= 0)
    return achip_fail(ASCIICHAT_HIP_ERR_NOT_SUPPORTED, "geometry %d is not in this build (render_variants.h: make EXTRA=-DACHIP_ALL_GEOMETRIES)", variant);
  if (variant >= 0 && !ACHIP_IS_STREAM_VARIANT(variant) && achip_variant_cap(variant) < p->max_wp) /* (a stream geometry's
                                                             "cap" is cells per frame: plan_measure judges those) */
    return achip_fail(ASCIICHAT_HIP_ERR_NOT_SUPPORTED, "variant %d cannot hold a %d-cell row", variant, p->max_wp);
  const int before = p->variant_user;
  p->variant_user = variant;
  const int rc = plan_measure(p, p->frames_pinned);
  if (rc)
    p->variant_user = before;
  return rc;
}

int asciichat_hip_plan_set_split(asciichat_hip_plan_t *p, int rows_per_part) {
  if (!p)
    return ASCIICHAT_HIP_ERR_INVALID_PARAM;
  const int before = p->split_request;
  p->split_request = rows_per_part;
  const int rc = plan_measure(p, p->frames_pinned);
  if (rc)
    p->split_request = before;
  return rc;
}

int asciichat_hip_plan_get_parts(const asciichat_hip_plan_t *p) { return p ? p->parts : 0; }

int asciichat_hip_plan_set_concurrency(asciichat_hip_plan_t *p, int launches_in_flight) {
  if (!p || launches_in_flight < 1)
    return ASCIICHAT_HIP_ERR_INVALID_PARAM;
  const int before = p->concurrency;
  p->concurrency = launches_in_flight;
  const int rc = plan_measure(p, p->frames_pinned);
  if (rc)
    p->concurrency = before;
  return rc;
}

int asciichat_hip_plan_set_uniform(asciichat_hip_plan_t *p, int allow) {
  if (!p)
    return ASCIICHAT_HIP_ERR_INVALID_PARAM;
  p->uniform_off = !allow;
  return 0;
}

int asciichat_hip_plan_get_uniform(const asciichat_hip_plan_t *p) {
  return p && p->uniform.enabled && !p->uniform_off;
}

int asciichat_hip_plan_get_variant(const asciichat_hip_plan_t *p) { return p ? p->variant : -1; }

/* whole != 0: every frame in ONE workgroup even where the plan's own choice shares frames out over workgroups of the stream
 * kernel (the hand-off words of that form carry a per-launch epoch: a captured launch cannot be replayed) */
static int render_range_as(asciichat_hip_plan_t *p, int first, int count, uint8_t *out_dev, size_t out_stride,
                           uint32_t *out_len_dev, unsigned long long *phase_cycles_dev, void *stream, int whole) {
  if (!p || !out_dev || !out_len_dev || first < 0 || count < 0 || first + count > p->n)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "plan_render: bad arguments");
  if (((uintptr_t)out_dev & 15u) || (out_stride & 15u) || out_stride < p->stride)
    return achip_fail(ASCIICHAT_HIP_ERR_BUFFER, "output slab must be 16-byte aligned with stride >= %zu (multiple of 16)",
                      p->stride);
  if (count == 0)
    return 0;
  /* a different epoch per launch makes last launch's hand-off words stale without clearing them; launches of
   * one plan must therefore be ordered (one stream), which updating its descriptors requires anyway */
  p->epoch = p->epoch + 1u ? p->epoch + 1u : 1u;
  const int fc = plan_frames_current(p, stream);
  if (fc)
    return fc;
  achip_uniform_t uni = p->uniform;
  if (p->uniform_off)
    uni.enabled = 0;
  uni.flags = (p->palette_ascii ? ACHIP_UNIFORM_PALETTE_ASCII : 0u) | ACHIP_UNIFORM_MAX_CELLS(p->max_cells);
  uni.f.src = uni.f.src ? uni.f.src + (int64_t)first * uni.src_pitch : NULL;
  const int shared_out = p->parts > 1 && variant_shares_out(p->variant);
  if (whole && p->parts > 1 && !shared_out)
    return achip_fail(ASCIICHAT_HIP_ERR_NOT_SUPPORTED, "this plan renders row bands");
  const int variant = whole && shared_out ? p->whole_variant : p->variant, parts = whole ? 1 : p->parts;
  return achip_hip_check(achip_launch_render(p->mode, variant, p->has_comp, p->frames_dev + first, count, p->lut_dev,
                                             out_dev, (uint64_t)out_stride, out_len_dev, phase_cycles_dev, parts,
                                             p->rows_per_part,
                                             parts > 1 && p->part_sync ? p->part_sync + (size_t)first * (size_t)p->parts : NULL,
                                             p->epoch, &uni, stream),
                         "render kernel launch");
}
static int render_range(asciichat_hip_plan_t *p, int first, int count, uint8_t *out_dev, size_t out_stride,
                        uint32_t *out_len_dev, unsigned long long *phase_cycles_dev, void *stream) {
  return render_range_as(p, first, count, out_dev, out_stride, out_len_dev, phase_cycles_dev, stream, 0);
}

int asciichat_hip_plan_render_range(asciichat_hip_plan_t *p, int first, int count, uint8_t *out_dev, size_t out_stride,
                                    uint32_t *out_len_dev, void *stream) {
  return render_range(p, first, count, out_dev, out_stride, out_len_dev, NULL, stream);
}

int asciichat_hip_plan_render_profiled(asciichat_hip_plan_t *p, uint8_t *out_dev, size_t out_stride,
                                       uint32_t *out_len_dev, unsigned long long *phase_cycles_dev, void *stream) {
  return render_range(p, 0, p ? p->n : 0, out_dev, out_stride, out_len_dev, phase_cycles_dev, stream);
}

int asciichat_hip_plan_render(asciichat_hip_plan_t *p, uint8_t *out_dev, size_t out_stride, uint32_t *out_len_dev,
                              void *stream) {
  return asciichat_hip_plan_render_range(p, 0, p ? p->n : 0, out_dev, out_stride, out_len_dev, stream);
}

/* Render + frame CRC-32C in one go ("a CRC over the output slab can ride the emit kernel").  Whole-frame
 * launches of the per-cell modes carry the checksum inside the stream kernel's drain; every other plan renders and
 * then runs the stand-alone CRC kernel on the slab -- same results either way. */
static int plan_wire_pass(asciichat_hip_plan_t *p, const uint8_t *slab_dev, size_t out_stride, const uint32_t *out_len_dev,
                          const uint32_t *dims_dev, uint32_t *crc_out_dev, uint8_t *hdr_out_dev, uint32_t *packet_crc_out_dev,
                          uint8_t *dst, size_t dst_capacity, uint64_t *off_out, uint32_t *len_out, void *stream);

static int plan_render_wire(asciichat_hip_plan_t *p, uint8_t *out_dev, size_t out_stride, uint32_t *out_len_dev,
                            const achip_wire_t *wire, unsigned long long *prof, void *stream) {
  if (!p || !out_dev || !out_len_dev || !wire->crc)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "plan_render_crc / plan_render_packets: bad arguments");
  if (((uintptr_t)out_dev & 15u) || (out_stride & 15u) || out_stride < p->stride)
    return achip_fail(ASCIICHAT_HIP_ERR_BUFFER, "output slab must be 16-byte aligned with stride >= %zu (multiple of 16)",
                      p->stride);
  if (((uintptr_t)wire->hdr & 7u))
    return achip_fail(ASCIICHAT_HIP_ERR_BUFFER, "packet headers must be 8-byte aligned");
  if (asciichat_hip_plan_has_fused_crc(p)) {
    const int fc = plan_frames_current(p, stream);
    if (fc)
      return fc;
    achip_uniform_t uni = p->uniform;
    if (p->uniform_off)
      uni.enabled = 0;
    uni.flags = (p->palette_ascii ? ACHIP_UNIFORM_PALETTE_ASCII : 0u) | ACHIP_UNIFORM_MAX_CELLS(p->max_cells);
    return achip_hip_check(achip_launch_render_crc(p->mode, p->whole_variant, p->has_comp, p->frames_dev, p->n, p->lut_dev, out_dev,
                                                   (uint64_t)out_stride, out_len_dev, wire, &uni, prof, stream),
                           "render + crc kernel launch");
  }
  int rc = render_range(p, 0, p->n, out_dev, out_stride, out_len_dev, prof, stream);
  if (!rc)
    rc = plan_wire_pass(p, out_dev, out_stride, out_len_dev, wire->dims, wire->crc, wire->hdr, wire->pkt_crc, NULL, 0, NULL, NULL,
                        stream);
  return rc;
}

/* Render + frame CRC-32C in one go ("a CRC over the output slab can ride the emit kernel").  Whole-frame
 * launches of the per-cell modes carry the checksum inside the stream kernel's drain; every other plan renders and
 * then runs the stand-alone CRC kernel on the slab -- same results either way. */
int asciichat_hip_plan_render_crc(asciichat_hip_plan_t *p, uint8_t *out_dev, size_t out_stride, uint32_t *out_len_dev,
                                  uint32_t *crc_out_dev, void *stream) {
  const achip_wire_t wire = {crc_out_dev, NULL, NULL, NULL};
  return plan_render_wire(p, out_dev, out_stride, out_len_dev, &wire, NULL, stream);
}

/* ... and the whole wire stage: frames, frame CRCs, 24-byte ascii_frame_packet_t headers, CRCs of header || frame --
 * ONE launch where the plan's geometry carries the fused CRC (the wave that finishes a frame writes its header and
 * packet CRC too), render + asciichat_hip_frame_packets otherwise. */
int asciichat_hip_plan_render_packets(asciichat_hip_plan_t *p, uint8_t *out_dev, size_t out_stride, uint32_t *out_len_dev,
                                      const uint32_t *dims_dev, uint32_t *crc_out_dev, uint8_t *hdr_out_dev,
                                      uint32_t *packet_crc_out_dev, void *stream) {
  if (!hdr_out_dev)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "plan_render_packets: no header buffer");
  const achip_wire_t wire = {crc_out_dev, dims_dev, hdr_out_dev, packet_crc_out_dev};
  return plan_render_wire(p, out_dev, out_stride, out_len_dev, &wire, NULL, stream);
}

/* ... and the frames at their exact lengths behind it (pack_frames' layout; dst may be mapped host memory): a plan whose
 * kernel carries the fused CRC renders (one launch) and packs; any other plan renders and then checksums AND packs in one
 * pass over the slab (asciichat_hip_frame_packets_packed) -- two launches either way. */
/* whether a wire-stage launch of this plan has every frame in ONE workgroup: whole-frame plans, and plans whose plain render
 * shares frames out over workgroups of the stream kernel (those launch whole_variant instead); not row bands */
static int plan_frames_whole(const asciichat_hip_plan_t *p) { return p->parts == 1 || variant_shares_out(p->variant); }

/* ---- frames at their exact lengths straight from the render kernel ------------------------- */
/* Whole-frame launches of the per-cell foreground modes whose frames fit the kernel's LDS image (48 KB: 1080p -> 80x24
 * truecolor is 36 KB) go through the PACK instantiations of the stream kernel: ONE launch leaves the frames back to back in
 * dst -- in the order in which they finish; off_out says where each one went -- together with the checksums and headers;
 * the slab is not written at all (ship exactly frame_size bytes, lib/network/acip/server.c:190-222). */
int asciichat_hip_plan_get_exact_length(const asciichat_hip_plan_t *p) {
  if (!p || p->exact_length == 0 || !plan_frames_whole(p) || p->has_comp || !ACHIP_IS_STREAM_VARIANT(p->whole_variant))
    return 0;
  if (!(p->mode == ACHIP_MODE_256_FG || p->mode == ACHIP_MODE_16_FG || (p->mode == ACHIP_MODE_TRUE_FG && p->palette_ascii)))
    return 0;
  return p->stride <= (size_t)achip_pack_frame_cap();
}
int asciichat_hip_plan_set_exact_length(asciichat_hip_plan_t *p, int mode) {
  if (!p || mode < -1 || mode > 1)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM,
                      "plan_set_exact_length: -1 (where it is the faster form), 0 (never) or 1 (wherever the plan qualifies)");
  p->exact_length = mode;
  return 0;
}
/* Whether THIS call takes the one-launch form.  Measured (profiles/r04_exact_length_timing.txt, 256 x 1080p -> 80x24
 * truecolor, us per step): into DEVICE memory the one launch wins -- frames only 9.4 against 11.5 with four launches in
 * flight -- but into mapped HOST memory it loses (206 against 176): a workgroup then holds its CU's LDS until its stores
 * have crossed PCIe, where the two-launch form's render is long gone and only the copy kernel's few registers wait.  So the
 * automatic choice looks at the destination (one hipPointerGetAttributes per new destination pointer; a tick reuses its
 * buffers). */
static int plan_packs_this_call(asciichat_hip_plan_t *p, const void *dst, const uint64_t *off_out) {
  if (!asciichat_hip_plan_get_exact_length(p))
    return 0;
  /* the one-launch form leaves the frames in COMPLETION order: only a caller that receives off_out can find them.  A
   * caller relying on pack_frames' documented layout (off_out == NULL: frame i behind the 16-byte rounded lengths of
   * frames 0..i-1) keeps the ordered two-pass form, whatever set_exact_length says */
  if (!off_out)
    return 0;
  if (p->exact_length > 0)
    return 1;
  if (p->pack_dst_seen != dst) {
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof(attr));
    const hipError_t e = hipPointerGetAttributes(&attr, dst);
    if (e != hipSuccess)
      (void)hipGetLastError(); /* an address the runtime knows nothing about: the caller's business, taken for device memory */
    p->pack_dst_seen = dst;
    p->pack_dst_host = e == hipSuccess && attr.type == hipMemoryTypeHost;
  }
  return !p->pack_dst_host;
}
/* ... and whether it takes the length-first form (frames beyond the PACK form's 48 KB; see plan_render_length_first below) */
static int plan_dst_is_device(asciichat_hip_plan_t *p, const void *dst) {
  if (p->pack_dst_seen != dst) {
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof(attr));
    const hipError_t e = hipPointerGetAttributes(&attr, dst);
    if (e != hipSuccess)
      (void)hipGetLastError();
    p->pack_dst_seen = dst;
    p->pack_dst_host = e == hipSuccess && attr.type == hipMemoryTypeHost;
  }
  return !p->pack_dst_host;
}
static int plan_length_first_ok(const asciichat_hip_plan_t *p);
static int plan_length_first_this_call(asciichat_hip_plan_t *p, const void *dst, const uint64_t *off_out) {
  if (p->exact_length == 0 || !off_out || !plan_length_first_ok(p) || asciichat_hip_plan_get_exact_length(p))
    return 0;
  if (p->exact_length > 0)
    return 1;
  /* (round 6's wire audit, scripts/gpu_wire_audit.py, profiles/r06_wire_audit.txt: sources up to 1080p as well -- 128 / 256 frames of
   * 1080p -> 200x60 49.2 / 62.3 us in one launch against 64.8 / 76.9 for render + pack pass, 320x90 94.6 / 123.4 against 144.4 / 175.2,
   * 160x45 34.4 / 46.2 against 39.4 / 49.1, 120x40 level; NOT a plan whose plain render shares its frames out over workgroups unless a
   * wave has one block: the one-launch form renders them whole, one workgroup a frame -- a lone 200x60 frame 34.9 us against 16.9) */
  if (p->parts > 1 && p->max_cells > 16 * 127)
    return 0;
  return (p->all_dense || !p->big_src) && plan_dst_is_device(p, dst);
}
static int plan_render_pack(asciichat_hip_plan_t *p, uint32_t *out_len_dev, const achip_wire_t *wire, uint8_t *dst,
                            size_t dst_capacity, uint64_t *off_out, uint32_t *len_out, void *stream) {
  if (!out_len_dev || !dst || !off_out || ((uintptr_t)dst & 15u) || ((uintptr_t)off_out & 7u) || ((uintptr_t)len_out & 3u) ||
      (wire && ((uintptr_t)wire->hdr & 7u)))
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "plan_render_*packed: bad arguments (16-byte aligned destination, 8-byte aligned offsets / headers)");
  if (!p->pack_cursor) {
    int rc = achip_hip_check((int)hipMalloc((void **)&p->pack_cursor, 2 * sizeof(unsigned long long)), "hipMalloc(pack cursor)");
    if (!rc)
      rc = achip_hip_check((int)hipMemset(p->pack_cursor, 0, 2 * sizeof(unsigned long long)), "hipMemset(pack cursor)");
    if (rc) {
      if (p->pack_cursor)
        (void)hipFree(p->pack_cursor);
      p->pack_cursor = NULL;
      return rc;
    }
  }
  const int fc = plan_frames_current(p, stream);
  if (fc)
    return fc;
  achip_uniform_t uni = p->uniform;
  if (p->uniform_off)
    uni.enabled = 0;
  uni.flags = (p->palette_ascii ? ACHIP_UNIFORM_PALETTE_ASCII : 0u) | ACHIP_UNIFORM_MAX_CELLS(achip_uniform_extent(p->mode, 16, p->frames_pinned, p->n));
  const achip_packdev_t pack = {dst, (uint64_t)dst_capacity, off_out, len_out, p->pack_cursor};
  return achip_hip_check(achip_launch_render_pack(p->mode, p->whole_variant, p->frames_dev, p->n, p->lut_dev, (uint64_t)p->stride, out_len_dev,
                                                  wire, &uni, &pack, stream),
                         "render + pack kernel launch");
}

/* ---- exact-length frames beyond the 48 KB of the one-launch PACK form: LENGTH-FIRST (round 6) -------------
 * The stream kernel's lean loop run twice -- lengths first, then the emission at the place the frame claimed -- for whole-frame
 * plans of truecolor foreground with an all-ASCII palette (render_stream.hpp LF).  Measured (profiles/r06_length_first_ab.txt,
 * 256 frames of 200x60): from sampled images 16.5 us against 34.5 for render + pack pass; from 4K sources the second gather
 * costs more than the pass (71.9 against 60.4) -- so the automatic choice takes it for dense sources and (the round's wire audit) for
 * sources up to 1080p, device destinations, callers that receive off_out (completion order, like the PACK form). */
static int plan_length_first_ok(const asciichat_hip_plan_t *p) {
  return p->mode == ACHIP_MODE_TRUE_FG && p->palette_ascii && !p->has_comp && plan_frames_whole(p) &&
         (p->whole_variant == 16 || p->whole_variant == 17);
}
/* whether plan_render_packed / _packets_packed may take the length-first form for this plan (frames beyond the one-launch
 * form's 48 KB; the automatic setting adds: dense sources, a device destination) */
int asciichat_hip_plan_get_length_first(const asciichat_hip_plan_t *p) {
  return p && p->exact_length != 0 && plan_length_first_ok(p) && !asciichat_hip_plan_get_exact_length(p);
}
int asciichat_hip_plan_render_length_first(asciichat_hip_plan_t *p, uint32_t *out_len_dev, uint8_t *dst, size_t dst_capacity,
                                           uint64_t *off_out, uint32_t *len_out, void *stream) {
  if (!p || !out_len_dev || !dst || !off_out || ((uintptr_t)dst & 15u) || ((uintptr_t)off_out & 7u) || ((uintptr_t)len_out & 3u))
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "plan_render_length_first: bad arguments");
  if (!plan_length_first_ok(p))
    return achip_fail(ASCIICHAT_HIP_ERR_NOT_SUPPORTED, "plan_render_length_first: truecolor foreground, all-ASCII palette, whole frames of single sources");
  if (!p->pack_cursor) {
    int rc = achip_hip_check((int)hipMalloc((void **)&p->pack_cursor, 2 * sizeof(unsigned long long)), "hipMalloc(pack cursor)");
    if (!rc)
      rc = achip_hip_check((int)hipMemset(p->pack_cursor, 0, 2 * sizeof(unsigned long long)), "hipMemset(pack cursor)");
    if (rc) {
      if (p->pack_cursor)
        (void)hipFree(p->pack_cursor);
      p->pack_cursor = NULL;
      return rc;
    }
  }
  const int fc = plan_frames_current(p, stream);
  if (fc)
    return fc;
  achip_uniform_t uni = p->uniform;
  if (p->uniform_off)
    uni.enabled = 0;
  uni.flags = ACHIP_UNIFORM_PALETTE_ASCII | ACHIP_UNIFORM_MAX_CELLS(achip_uniform_extent(p->mode, 16, p->frames_pinned, p->n));
  const achip_packdev_t pack = {dst, (uint64_t)dst_capacity, off_out, len_out, p->pack_cursor};
  const int e = achip_launch_render_length_first(p->whole_variant, p->frames_dev, p->n, p->lut_dev, (uint64_t)p->stride, out_len_dev, &uni,
                                                 &pack, stream);
  return achip_hip_check(e, "length-first render launch");
}

int asciichat_hip_plan_render_packets_packed(asciichat_hip_plan_t *p, uint8_t *slab_dev, size_t out_stride, uint32_t *out_len_dev,
                                             const uint32_t *dims_dev, uint32_t *crc_out_dev, uint8_t *hdr_out_dev,
                                             uint32_t *packet_crc_out_dev, uint8_t *dst, size_t dst_capacity, uint64_t *off_out,
                                             uint32_t *len_out, void *stream) {
  if (!p || !hdr_out_dev || !dst)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "plan_render_packets_packed: no header buffer or destination");
  if (crc_out_dev && plan_packs_this_call(p, dst, off_out)) {
    const achip_wire_t wire = {crc_out_dev, dims_dev, hdr_out_dev, packet_crc_out_dev};
    return plan_render_pack(p, out_len_dev, &wire, dst, dst_capacity, off_out, len_out, stream);
  }
  if (crc_out_dev && plan_length_first_this_call(p, dst, off_out)) { /* exact-length frames in one launch, checksummed where they lie */
    int rc = asciichat_hip_plan_render_length_first(p, out_len_dev, dst, dst_capacity, off_out, len_out, stream);
    if (!rc)
      rc = plan_wire_pass(p, dst, p->stride, out_len_dev, dims_dev, crc_out_dev, hdr_out_dev, packet_crc_out_dev, NULL, 0, off_out, NULL,
                          stream);
    return rc;
  }
  if (asciichat_hip_plan_has_fused_crc(p)) {
    int rc = asciichat_hip_plan_render_packets(p, slab_dev, out_stride, out_len_dev, dims_dev, crc_out_dev, hdr_out_dev,
                                               packet_crc_out_dev, stream);
    if (!rc)
      rc = asciichat_hip_pack_frames(slab_dev, out_stride, out_len_dev, p->n, dst, dst_capacity, off_out, len_out, stream);
    return rc;
  }
  if (((uintptr_t)dst & 15u) || ((uintptr_t)off_out & 7u) || ((uintptr_t)len_out & 3u))
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM,
                      "plan_render_packets_packed: a 16-byte aligned destination (8-byte aligned offsets, 4-byte aligned lengths) is required");
  int rc = asciichat_hip_plan_render(p, slab_dev, out_stride, out_len_dev, stream);
  if (!rc)
    rc = plan_wire_pass(p, slab_dev, out_stride, out_len_dev, dims_dev, crc_out_dev, hdr_out_dev, packet_crc_out_dev, dst,
                        dst_capacity, off_out, len_out, stream);
  return rc;
}

/* diagnostics: the same launch with the per-wave timestamps of plan_render_profiled */
int asciichat_hip_plan_render_crc_profiled(asciichat_hip_plan_t *p, uint8_t *out_dev, size_t out_stride,
                                           uint32_t *out_len_dev, uint32_t *crc_out_dev,
                                           unsigned long long *phase_cycles_dev, void *stream) {
  const achip_wire_t wire = {crc_out_dev, NULL, NULL, NULL};
  return plan_render_wire(p, out_dev, out_stride, out_len_dev, &wire, phase_cycles_dev, stream);
}

int asciichat_hip_plan_has_fused_crc(const asciichat_hip_plan_t *p) {
  if (!p || !plan_frames_whole(p) || p->fused_crc == 0 || !achip_variant_has_crc(p->whole_variant))
    return 0;
  if (p->mode == ACHIP_MODE_TRUE_FG && !p->palette_ascii) /* (the multi-byte-palette instantiation carries no checksum) */
    return 0;
  if (p->fused_crc > 0)
    return 1;
  /* a plan whose plain render shares its frames out over workgroups (a small launch) would have to launch them WHOLE for the
   * fused checksum -- one workgroup per frame: fine while a wave has one block (a lone 80x24 frame: 10.9 us fused against
   * 13.2 for shared-out render + stand-alone pass), a multiple of the render beyond (a lone 200x60 truecolor frame 49 us
   * fused, 320x90 116, against a 7 us render + a 19-26 us pass; scripts/gpu_wire_audit.py, profiles/r04_wire_audit.txt) */
  if (p->parts > 1 && p->max_cells > 16 * 127)
    return 0;
  /* (round 6's wire audit: the checksumming instantiations kept the general per-block loop, so beside the lean loop of truecolor
   * foreground the fused form pays only for small frames -- 256 frames of 80x24 13.0 us fused against 18.4 with the stand-alone pass,
   * 120x40 26.7 / 29.0, 160x45 34.4 / 35.4, but 200x60 55.9 / 49.6 and 320x90 129.0 / 102.9 (128 frames: 122.8 / 80.5); the other
   * per-cell modes up to 200x60: 256 colours 44.9 / 44.3, 320x90 101.3 / 89.3) */
  /* (... and from dense sources, where the lean render has no line fills to wait for, earlier still: truecolor foreground only while
   * a wave has one block -- 128 frames of 160x45 31.5 us fused against 27.2, 120x40 23.0 / 21.4, 80x24 12.5 / 14.5 the other way --,
   * 256 colours up to 160x45: 200x60 43.8 / 40.2) */
  const long fused_limit = p->mode == ACHIP_MODE_TRUE_FG ? (p->all_dense ? 16 * 127 : 8192) : (p->all_dense ? 8192 : 12288);
  if (ACHIP_IS_STREAM_VARIANT(p->whole_variant) && p->max_cells > fused_limit)
    return 0;
  return achip_variant_crc_pays(p->whole_variant);
}

int asciichat_hip_plan_set_fused_crc(asciichat_hip_plan_t *p, int mode) {
  if (!p || mode < -1 || mode > 1)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "plan_set_fused_crc: -1 (automatic), 0 (never) or 1 (always)");
  /* "always" on a plan whose geometry has no fused instantiation in this build of the library (the rows kernel's live in
   * -DACHIP_ALL_GEOMETRIES builds only; truecolor foreground with a multi-byte palette has none) would silently run the two
   * passes: say so (the setting is kept for a later geometry of the plan; plan_has_fused_crc tells what a call will do) */
  p->fused_crc = mode;
  if (mode == 1 && !asciichat_hip_plan_has_fused_crc(p))
    return achip_fail(ASCIICHAT_HIP_ERR_NOT_SUPPORTED, "plan_set_fused_crc(1): this plan's geometry carries no fused checksum in this build; "
                                                       "plan_render_crc / _packets will run render + stand-alone pass");
  return 0;
}

int asciichat_hip_packets_from_crc(const uint32_t *len_dev, const uint32_t *crc_dev, int n, const uint32_t *dims_dev,
                                   uint8_t *hdr_out_dev, uint32_t *packet_crc_out_dev, void *stream) {
  if (!len_dev || !crc_dev || !hdr_out_dev || n <= 0)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "packets_from_crc: bad arguments");
  int rc = achip_require_device();
  if (rc)
    return rc;
  return achip_hip_check(achip_launch_packets_from_crc(len_dev, crc_dev, dims_dev, n, hdr_out_dev, packet_crc_out_dev, stream),
                         "packet header launch");
}

/* K steps issued from C: step k renders plans[k % n_plans] on streams[k % n_streams] into that stream's slab.  A
 * server tick loop (and bench.py) calls this instead of paying an FFI round trip per launch: at ~8 us per 256-frame
 * step one interpreter thread cannot keep three streams fed. */
int asciichat_hip_render_many(asciichat_hip_plan_t *const *plans, int n_plans, uint8_t *const *out_dev,
                              uint32_t *const *out_len_dev, size_t out_stride, void *const *streams, int n_streams,
                              int first_step, int n_steps) {
  if (!plans || !out_dev || !out_len_dev || !streams || n_plans <= 0 || n_streams <= 0 || first_step < 0 || n_steps < 0 ||
      n_plans % n_streams != 0)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM,
                      "render_many: bad arguments (n_plans must be a multiple of n_streams: a plan stays on one stream)");
  for (int k = first_step; k < first_step + n_steps; k++) {
    const int s = k % n_streams;
    asciichat_hip_plan_t *p = plans[k % n_plans];
    const int rc = render_range(p, 0, p ? p->n : 0, out_dev[s], out_stride, out_len_dev[s], NULL, streams[s]);
    if (rc)
      return rc;
  }
  return 0;
}

/* diagnostics: the same loop with the kernels' per-wave timestamps, step k writing prof_dev + k * prof_stride_words
 * (scripts/gpu_burst_timeline.py: when does each launch of a short burst start and end on the device?) */
int asciichat_hip_render_many_profiled(asciichat_hip_plan_t *const *plans, int n_plans, uint8_t *const *out_dev,
                                       uint32_t *const *out_len_dev, size_t out_stride, void *const *streams,
                                       int n_streams, int first_step, int n_steps, unsigned long long *prof_dev,
                                       size_t prof_stride_words) {
  if (!plans || !out_dev || !out_len_dev || !streams || n_plans <= 0 || n_streams <= 0 || first_step < 0 || n_steps < 0 ||
      n_plans % n_streams != 0 || !prof_dev)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "render_many_profiled: bad arguments");
  for (int k = first_step; k < first_step + n_steps; k++) {
    const int s = k % n_streams;
    asciichat_hip_plan_t *p = plans[k % n_plans];
    const int rc = render_range(p, 0, p ? p->n : 0, out_dev[s], out_stride, out_len_dev[s],
                                prof_dev + (size_t)(k - first_step) * prof_stride_words, streams[s]);
    if (rc)
      return rc;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* A tick loop captured once and replayed: the n_steps launches of asciichat_hip_render_many as ONE hipGraph       */
/* (n_lanes parallel branches, one per independent batch stream).  A replay costs one graph launch instead of      */
/* n_steps kernel launches on n_lanes queues that each have to wake up after an idle period -- the fixed cost of a  */
/* short burst of steps (profiles/r02_region_overhead.txt).                                                         */
/* ------------------------------------------------------------------------------------------- */
struct asciichat_hip_schedule {
  hipGraph_t graph;
  hipGraphExec_t exec;
  int n_steps;
};

int asciichat_hip_schedule_create(asciichat_hip_schedule_t **sched, asciichat_hip_plan_t *const *plans, int n_plans,
                                  uint8_t *const *out_dev, uint32_t *const *out_len_dev, size_t out_stride, int n_lanes,
                                  int first_step, int n_steps) {
  if (!sched || !plans || !out_dev || !out_len_dev || n_plans <= 0 || n_lanes <= 0 || n_lanes > 16 || first_step < 0 ||
      n_steps <= 0 || n_plans % n_lanes != 0)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM,
                      "schedule_create: bad arguments (n_plans must be a multiple of n_lanes, n_lanes <= 16)");
  *sched = NULL;
  for (int i = 0; i < n_plans; i++)
    if (!plans[i] || (plans[i]->parts > 1 && !variant_shares_out(plans[i]->variant))) /* row-band launches carry a per-launch
                                                                       epoch: not replayable (shared-out frames of the
                                                                       stream kernel are captured whole instead) */
      return achip_fail(ASCIICHAT_HIP_ERR_NOT_SUPPORTED, "schedule_create: plan %d renders row bands (small batch)", i);
  int rc = achip_require_device();
  if (rc)
    return rc;
  asciichat_hip_schedule_t *s = (asciichat_hip_schedule_t *)calloc(1, sizeof(*s));
  if (!s)
    return achip_fail(ASCIICHAT_HIP_ERR_MEMORY, "out of memory");
  hipStream_t lane[16] = {0};
  hipEvent_t fork = NULL, join[16] = {0};
  int capturing = 0;
  for (int l = 0; l < n_lanes && !rc; l++)
    rc = achip_hip_check((int)hipStreamCreateWithFlags(&lane[l], hipStreamNonBlocking), "hipStreamCreate");
  if (!rc)
    rc = achip_hip_check((int)hipEventCreateWithFlags(&fork, hipEventDisableTiming), "hipEventCreate");
  for (int l = 1; l < n_lanes && !rc; l++)
    rc = achip_hip_check((int)hipEventCreateWithFlags(&join[l], hipEventDisableTiming), "hipEventCreate");
  if (!rc) {
    rc = achip_hip_check((int)hipStreamBeginCapture(lane[0], hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
    capturing = !rc;
  }
  if (!rc && n_lanes > 1) { /* the other lanes join the capture behind a fork event */
    rc = achip_hip_check((int)hipEventRecord(fork, lane[0]), "hipEventRecord");
    for (int l = 1; l < n_lanes && !rc; l++)
      rc = achip_hip_check((int)hipStreamWaitEvent(lane[l], fork, 0), "hipStreamWaitEvent");
  }
  for (int k = first_step; k < first_step + n_steps && !rc; k++) {
    const int l = k % n_lanes;
    asciichat_hip_plan_t *p = plans[k % n_plans];
    rc = render_range_as(p, 0, p->n, out_dev[l], out_stride, out_len_dev[l], NULL, lane[l], 1);
  }
  for (int l = 1; l < n_lanes && !rc; l++) {
    rc = achip_hip_check((int)hipEventRecord(join[l], lane[l]), "hipEventRecord");
    if (!rc)
      rc = achip_hip_check((int)hipStreamWaitEvent(lane[0], join[l], 0), "hipStreamWaitEvent");
  }
  if (capturing) {
    hipGraph_t g = NULL;
    const int e = (int)hipStreamEndCapture(lane[0], &g);
    if (!rc)
      rc = achip_hip_check(e, "hipStreamEndCapture");
    s->graph = g;
  }
  if (!rc)
    rc = achip_hip_check((int)hipGraphInstantiate(&s->exec, s->graph, NULL, NULL, 0), "hipGraphInstantiate");
  for (int l = 0; l < n_lanes; l++) {
    if (lane[l])
      (void)hipStreamDestroy(lane[l]);
    if (join[l])
      (void)hipEventDestroy(join[l]);
  }
  if (fork)
    (void)hipEventDestroy(fork);
  if (rc) {
    asciichat_hip_schedule_destroy(s);
    return rc;
  }
  s->n_steps = n_steps;
  *sched = s;
  return 0;
}

int asciichat_hip_schedule_launch(asciichat_hip_schedule_t *s, void *stream) {
  if (!s || !s->exec)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "schedule_launch: bad arguments");
  return achip_hip_check((int)hipGraphLaunch(s->exec, (hipStream_t)stream), "hipGraphLaunch");
}

void asciichat_hip_schedule_destroy(asciichat_hip_schedule_t *s) {
  if (!s)
    return;
  if (s->exec)
    (void)hipGraphExecDestroy(s->exec);
  if (s->graph)
    (void)hipGraphDestroy(s->graph);
  free(s);
}

/* Spin until every stream has drained.  Polling, because a blocking hipStreamSynchronize sleeps in the driver and wakes
 * tens of microseconds late (most of a 20-step timed region); and ROUND ROBIN over the streams, because the first query
 * of a stream with work outstanding costs a round trip of its own that proceeds asynchronously: queried one after the
 * other (each spun to completion before the next is touched) those round trips add up -- 4 streams, bursts of 4 / 20
 * steps: 83 / 205 us stream by stream, 58 / 189 us round robin (profiles/r02_wait_modes.txt). */
int asciichat_hip_streams_wait(void *const *streams, int n_streams) {
  if (!streams || n_streams <= 0 || n_streams > 64)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "streams_wait: bad arguments");
  unsigned long long pending = n_streams == 64 ? ~0ull : (1ull << n_streams) - 1ull;
  while (pending)
    for (int s = 0; s < n_streams; s++) {
      if (!(pending >> s & 1ull))
        continue;
      const hipError_t e = hipStreamQuery((hipStream_t)streams[s]);
      if (e == hipErrorNotReady)
        continue;
      if (e != hipSuccess)
        return achip_hip_check((int)e, "hipStreamQuery");
      pending &= ~(1ull << s);
    }
  return 0;
}

void asciichat_hip_plan_destroy(asciichat_hip_plan_t *p) {
  if (!p)
    return;
  if (p->frames_dev)
    (void)hipFree(p->frames_dev);
  if (p->frames_pinned)
    (void)hipHostFree(p->frames_pinned);
  if (p->part_sync)
    (void)hipFree(p->part_sync);
  if (p->pack_cursor)
    (void)hipFree(p->pack_cursor);
  if (p->crc_scratch)
    (void)hipFree(p->crc_scratch);
  achip_lut_put(p->lut_dev);
  free(p);
}

/* ------------------------------------------------------------------------------------------- */
int asciichat_hip_resize(const uint8_t *src_dev, int src_w, int src_h, uint8_t *dst_dev, int dst_w, int dst_h,
                         void *stream) {
  if (!src_dev || !dst_dev || src_w <= 0 || src_h <= 0 || dst_w <= 0 || dst_h <= 0)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "resize: bad arguments");
  int rc = achip_require_device();
  if (rc)
    return rc;
  return achip_hip_check(achip_launch_resize(src_dev, src_w, src_h, dst_dev, dst_w, dst_h, stream), "resize launch");
}

int asciichat_hip_composite_upload(const achip_composite_t *comp_host, achip_composite_t **comp_dev) {
  if (!comp_host || !comp_dev)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "composite_upload: bad arguments");
  int rc = achip_require_device();
  if (rc)
    return rc;
  achip_composite_t *d = NULL;
  achip_composite_t copy = *comp_host;
  copy._pad = 0; /* the kernels read this word as the pixel of a sample that hits no tile (sample_composite_lds) */
  rc = achip_hip_check((int)hipMalloc((void **)&d, sizeof(*d)), "hipMalloc(composite)");
  if (!rc)
    rc = achip_hip_check((int)hipMemcpy(d, &copy, sizeof(*d), hipMemcpyHostToDevice), "hipMemcpy(composite)");
  if (rc) {
    if (d)
      (void)hipFree(d);
    return rc;
  }
  *comp_dev = d;
  return 0;
}

int asciichat_hip_apply_color_filter(uint8_t *pixels_dev, int width, int height, int stride, int color_filter,
                                     void *stream) {
  if (!pixels_dev || width <= 0 || height <= 0 || stride < 3 * width)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "apply_color_filter: bad arguments");
  achip_frame_t probe;
  memset(&probe, 0, sizeof(probe));
  if (achip_frame_set_display_ops(&probe, false, false, color_filter) != 0)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "apply_color_filter: unknown or unsupported filter %d",
                      color_filter);
  if (!(probe.ops & ACHIP_OP_TINT))
    return 0; /* COLOR_FILTER_NONE */
  int rc = achip_require_device();
  if (rc)
    return rc;
  return achip_hip_check(achip_launch_tint(pixels_dev, width, height, stride, probe.ops, stream), "tint launch");
}

int asciichat_hip_image_flip(const uint8_t *src_dev, uint8_t *dst_dev, int width, int height, int flip_x, int flip_y,
                             void *stream) {
  if (!src_dev || !dst_dev || src_dev == dst_dev || width <= 0 || height <= 0)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "image_flip: bad arguments");
  int rc = achip_require_device();
  if (rc)
    return rc;
  uint32_t ops = 0;
  if (width > 1 && height > 1) /* display.c:549 */
    ops = (flip_x ? ACHIP_OP_FLIP_X : 0u) | (flip_y ? ACHIP_OP_FLIP_Y : 0u);
  return achip_hip_check(achip_launch_flip(src_dev, dst_dev, width, height, 3 * width, 3 * width, ops, stream),
                         "flip launch");
}

/* ---- wire stage -------------------------------------------------------------------------------- */
typedef struct {
  uint8_t *dst;
  size_t capacity;
  uint64_t *off_out;
  uint32_t *len_out;
} crc_pack_t;

/* at_dev != NULL: frame i lies at base_dev + at_dev[i] (16-byte aligned offsets: the exact-length forms of the render) */
static int crc_common_at(const uint8_t *base_dev, size_t stride, const uint64_t *at_dev, const uint32_t *len_dev, uint32_t fixed_len,
                         uint32_t max_len, int n, const uint32_t *dims_dev, uint32_t *crc_out_dev, uint8_t *hdr_out_dev,
                         uint32_t *packet_crc_out_dev, const crc_pack_t *pack, uint32_t *own_scratch, void *stream) {
  if (!base_dev || !crc_out_dev || n <= 0 || ((uintptr_t)base_dev & 15u) || (stride & 15u) ||
      (len_dev ? max_len == 0 : fixed_len > max_len) || max_len >= 0xFFFFFFF0u || (!at_dev && n > 1 && stride < max_len) ||
      (at_dev && (pack || !len_dev)))
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "crc32c: bad arguments");
  int rc = achip_require_device();
  if (rc)
    return rc;
  const int parts = achip_crc_parts(max_len, n);
  uint32_t *scratch = own_scratch; /* a plan's own block (n * parts words): launches of one plan are ordered on one stream */
  if (parts > 1 && !scratch) {
    /* span registers of large buffers: STREAM-ORDERED scratch, so that calls in flight on different streams (the
     * plan API's normal use) never share a block (one thread-local block was shared by every stream) */
    rc = achip_hip_check((int)hipMallocAsync((void **)&scratch, (size_t)n * (size_t)parts * sizeof(uint32_t),
                                             (hipStream_t)stream),
                         "hipMallocAsync(crc scratch)");
    if (rc)
      return rc;
  }
  /* a plan's own block carries n arrival counters (zero between launches) behind its n * parts span registers: ONE launch */
  uint32_t *counters = own_scratch && parts > 1 ? own_scratch + (size_t)n * (size_t)parts : NULL;
  rc = achip_hip_check(pack ? achip_launch_crc32c_pack(base_dev, stride, len_dev, max_len, n, scratch, counters, dims_dev, crc_out_dev,
                                                       hdr_out_dev, packet_crc_out_dev, pack->dst, (uint64_t)pack->capacity,
                                                       pack->off_out, pack->len_out, stream)
                       : at_dev ? achip_launch_crc32c_at(base_dev, at_dev, len_dev, max_len, n, scratch, counters, dims_dev, crc_out_dev,
                                                         hdr_out_dev, packet_crc_out_dev, stream)
                                : achip_launch_crc32c(base_dev, stride, len_dev, fixed_len, max_len, n, scratch, counters, dims_dev,
                                                      crc_out_dev, hdr_out_dev, packet_crc_out_dev, stream),
                       "crc32c launch");
  if (scratch && !own_scratch) {
    const int fr = achip_hip_check((int)hipFreeAsync(scratch, (hipStream_t)stream), "hipFreeAsync(crc scratch)");
    if (!rc)
      rc = fr;
  }
  return rc;
}

static int crc_common(const uint8_t *base_dev, size_t stride, const uint32_t *len_dev, uint32_t fixed_len, uint32_t max_len, int n,
                      const uint32_t *dims_dev, uint32_t *crc_out_dev, uint8_t *hdr_out_dev, uint32_t *packet_crc_out_dev,
                      const crc_pack_t *pack, uint32_t *own_scratch, void *stream) {
  return crc_common_at(base_dev, stride, NULL, len_dev, fixed_len, max_len, n, dims_dev, crc_out_dev, hdr_out_dev, packet_crc_out_dev, pack,
                       own_scratch, stream);
}

int asciichat_hip_crc32c(const uint8_t *base_dev, size_t stride, const uint32_t *len_dev, uint32_t fixed_len,
                         uint32_t max_len, int n, uint32_t *crc_out_dev, void *stream) {
  return crc_common(base_dev, stride, len_dev, fixed_len, max_len, n, NULL, crc_out_dev, NULL, NULL, NULL, NULL, stream);
}

int asciichat_hip_frame_packets(const uint8_t *base_dev, size_t stride, const uint32_t *len_dev, uint32_t max_len, int n,
                                const uint32_t *dims_dev, uint32_t *crc_out_dev, uint8_t *hdr_out_dev,
                                uint32_t *packet_crc_out_dev, void *stream) {
  if (!len_dev || !hdr_out_dev)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame_packets: lengths and a header buffer are required");
  return crc_common(base_dev, stride, len_dev, 0, max_len, n, dims_dev, crc_out_dev, hdr_out_dev, packet_crc_out_dev, NULL,
                    NULL, stream);
}

/* the wire stage AND the compaction in one pass over the slab (crc_kernels.hpp COPY instantiations) */
int asciichat_hip_frame_packets_packed(const uint8_t *base_dev, size_t stride, const uint32_t *len_dev, uint32_t max_len, int n,
                                       const uint32_t *dims_dev, uint32_t *crc_out_dev, uint8_t *hdr_out_dev,
                                       uint32_t *packet_crc_out_dev, uint8_t *dst, size_t dst_capacity, uint64_t *off_out,
                                       uint32_t *len_out, void *stream) {
  if (!len_dev || !hdr_out_dev || !dst || ((uintptr_t)dst & 15u) || ((uintptr_t)off_out & 7u) || ((uintptr_t)len_out & 3u))
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM,
                      "frame_packets_packed: lengths, a header buffer and a 16-byte aligned destination are required");
  const crc_pack_t pack = {dst, dst_capacity, off_out, len_out};
  return crc_common(base_dev, stride, len_dev, 0, max_len, n, dims_dev, crc_out_dev, hdr_out_dev, packet_crc_out_dev, &pack,
                    NULL, stream);
}

/* the stand-alone wire pass behind a plan's render (dst != NULL: + the compaction), with the plan's own span registers */
static int plan_wire_pass(asciichat_hip_plan_t *p, const uint8_t *slab_dev, size_t out_stride, const uint32_t *out_len_dev,
                          const uint32_t *dims_dev, uint32_t *crc_out_dev, uint8_t *hdr_out_dev, uint32_t *packet_crc_out_dev,
                          uint8_t *dst, size_t dst_capacity, uint64_t *off_out, uint32_t *len_out, void *stream) {
  if (packet_crc_out_dev && !hdr_out_dev)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame_packets: lengths and a header buffer are required");
  const int parts = out_stride < 0xFFFFFFF0u ? achip_crc_parts((uint32_t)out_stride, p->n) : 1;
  /* span registers + one arrival counter per frame (zero between launches: the last span to arrive re-arms it), laid out
   * for THIS parts count.  The block is kept at the largest layout seen: a call with another out_stride only re-zeroes the
   * counters at their new place, in stream order (a caller alternating strides paid a device-synchronising hipFree per call:
   * ADVICE r4).  Launches of one plan are ordered on ONE stream (asciichat_hip.h): the counters are the plan's. */
  const size_t words = parts > 1 ? (size_t)p->n * ((size_t)parts + 1u) : 0;
  if (words && words > p->crc_scratch_cap) {
    if (p->crc_scratch)
      (void)hipFree(p->crc_scratch); /* synchronises with a launch that still uses the old block */
    p->crc_scratch = NULL;
    p->crc_scratch_words = p->crc_scratch_cap = 0;
    int rc = achip_hip_check((int)hipMalloc((void **)&p->crc_scratch, words * sizeof(uint32_t)), "hipMalloc(crc scratch)");
    if (rc)
      return rc;
    p->crc_scratch_cap = words;
  }
  if (words && words != p->crc_scratch_words) {
    const int rc = achip_hip_check((int)hipMemsetAsync(p->crc_scratch + (size_t)p->n * (size_t)parts, 0, (size_t)p->n * sizeof(uint32_t),
                                                       (hipStream_t)stream),
                                   "hipMemsetAsync(crc arrival counters)");
    if (rc)
      return rc;
    p->crc_scratch_words = words;
  }
  const crc_pack_t pack = {dst, dst_capacity, off_out, len_out};
  /* dst == NULL with off_out: the frames already lie packed at slab_dev + off_out[i] (plan_render_length_first): in place */
  return crc_common_at(slab_dev, out_stride, !dst ? off_out : NULL, out_len_dev, 0, (uint32_t)out_stride, p->n,
                       hdr_out_dev || packet_crc_out_dev ? dims_dev : NULL, crc_out_dev, hdr_out_dev, packet_crc_out_dev, dst ? &pack : NULL,
                       words ? p->crc_scratch : NULL, stream);
}

/* ---- compacted output (lib/network/acip/server.c:190-222 ships frame_size bytes, not a stride) ------------- */
int asciichat_hip_pack_frames(const uint8_t *slab_dev, size_t stride, const uint32_t *len_dev, int n, uint8_t *dst,
                              size_t dst_capacity, uint64_t *off_out, uint32_t *len_out, void *stream) {
  if (!slab_dev || !len_dev || !dst || n <= 0 || ((uintptr_t)slab_dev & 15u) || (stride & 15u) || ((uintptr_t)dst & 15u) ||
      ((uintptr_t)off_out & 7u) || ((uintptr_t)len_out & 3u))
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM,
                      "pack_frames: bad arguments (slab, stride and dst 16-byte aligned, offsets 8-byte aligned)");
  int rc = achip_require_device();
  if (rc)
    return rc;
  return achip_hip_check(achip_launch_pack(slab_dev, (uint64_t)stride, len_dev, n, dst, (uint64_t)dst_capacity, off_out,
                                           len_out, stream),
                         "pack launch");
}

int asciichat_hip_plan_render_packed(asciichat_hip_plan_t *p, uint8_t *slab_dev, size_t out_stride, uint32_t *out_len_dev,
                                     uint8_t *dst, size_t dst_capacity, uint64_t *off_out, uint32_t *len_out, void *stream) {
  if (dst && plan_packs_this_call(p, dst, off_out))
    return plan_render_pack(p, out_len_dev, NULL, dst, dst_capacity, off_out, len_out, stream);
  if (dst && plan_length_first_this_call(p, dst, off_out))
    return asciichat_hip_plan_render_length_first(p, out_len_dev, dst, dst_capacity, off_out, len_out, stream);
  int rc = asciichat_hip_plan_render(p, slab_dev, out_stride, out_len_dev, stream);
  if (!rc)
    rc = asciichat_hip_pack_frames(slab_dev, out_stride, out_len_dev, p->n, dst, dst_capacity, off_out, len_out, stream);
  return rc;
}

/* pinned host memory that kernels can address (the packed output's destination on the host side of PCIe) */
int asciichat_hip_host_alloc(size_t bytes, void **host_ptr, void **device_alias) {
  if (!host_ptr || !device_alias || bytes == 0)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "host_alloc: bad arguments");
  *host_ptr = *device_alias = NULL;
  int rc = achip_require_device();
  if (rc)
    return rc;
  void *h = NULL, *d = NULL;
  rc = achip_hip_check((int)hipHostMalloc(&h, bytes, hipHostMallocMapped | hipHostMallocPortable), "hipHostMalloc(mapped)");
  if (!rc)
    rc = achip_hip_check((int)hipHostGetDevicePointer(&d, h, 0), "hipHostGetDevicePointer");
  if (rc) {
    if (h)
      (void)hipHostFree(h);
    return rc;
  }
  *host_ptr = h;
  *device_alias = d;
  return 0;
}

void asciichat_hip_host_free(void *host_ptr) {
  if (host_ptr)
    (void)hipHostFree(host_ptr);
}

void asciichat_hip_free(void *dev_ptr) {
  if (dev_ptr)
    (void)hipFree(dev_ptr);
}

int asciichat_hip_composite(const achip_composite_t *comp_host, uint8_t *dst_dev, void *stream) {
  if (!comp_host || !dst_dev)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "composite: bad arguments");
  achip_composite_t *d = NULL;
  int rc = asciichat_hip_composite_upload(comp_host, &d);
  if (rc)
    return rc;
  rc = achip_hip_check(achip_launch_composite(d, comp_host->canvas_w, comp_host->canvas_h, dst_dev, stream),
                       "composite launch");
  if (!rc)
    rc = achip_hip_check((int)hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
  (void)hipFree(d);
  return rc;
}
