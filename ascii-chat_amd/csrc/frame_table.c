/*
 * frame_table.c -- ingest side of the path: a device-resident table of every client's
 * latest camera frame.
 *
 * The reference keeps the latest frame of a client as a host blob [u32 BE width][u32 BE height][RGB24]
 * (video_frame_get_latest), and EVERY render thread copies EVERY client's blob twice per tick
 * (collect_video_sources, src/server/stream.c:221-463: SAFE_MALLOC + memcpy, then image_new_from_pool +
 * memcpy) -- 2*N^2 full-frame host copies per tick for N clients.  Here the receive path publishes a blob
 * once: it is validated exactly as collect_video_sources validates it, sent to HBM with one DMA, and every
 * render descriptor of the tick points at the same device frame.
 *
 * Slots are double-buffered and readers are tracked: latest() notes which stream was handed which buffer, and a
 * publish that is about to overwrite a buffer first records an event on every stream that was handed it and makes the
 * upload wait for those events -- the DMA is ordered behind every render ENQUEUED so far that may read the buffer
 * (the previous version only ordered uploads behind uploads).  The contract that remains with the caller:
 * a pointer from latest() is good for work enqueued before the publish after next on that slot.  Locks are per slot;
 * nothing blocking happens under a table-wide lock.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "achip_host.h"
#include "asciichat_hip.h"
#include "hip_launch.h"
#include "internal.h"

#define FT_MAX_READERS 32 /* consumer streams remembered per buffer; beyond that a publish synchronises the device */

typedef struct {
  pthread_mutex_t mu;   /* guards this slot only                                       */
  hipStream_t reader[2][FT_MAX_READERS]; /* streams that were handed buffer k by latest() since its last upload */
  int n_readers[2];
  int readers_overflow[2];
  hipEvent_t reader_done; /* scratch event: "everything enqueued on a reader stream so far" */
  uint8_t *dev[2];      /* frame buffers in HBM                                       */
  size_t cap[2];        /* bytes allocated                                            */
  hipEvent_t ready[2];  /* recorded after the upload of buffer k                      */
  uint8_t *stage[2];    /* pinned staging for blobs that are not in the pinned pool   */
  size_t stage_cap[2];
  uint8_t *rows_dev[2]; /* publish_rows: device side of the staged [index table][rows] block */
  size_t rows_cap[2];
  int cur;              /* buffer holding the latest complete frame, -1 = none yet    */
  int w, h;
  uint64_t generation;
} ft_slot_t;

struct asciichat_hip_frame_table {
  int n;
  ft_slot_t *slot;
};

int asciichat_hip_frame_table_create(asciichat_hip_frame_table_t **table, int n_slots) {
  if (!table || n_slots <= 0)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame_table_create: bad arguments");
  *table = NULL;
  int rc = achip_require_device();
  if (rc)
    return rc;
  asciichat_hip_frame_table_t *t = (asciichat_hip_frame_table_t *)calloc(1, sizeof(*t));
  if (t)
    t->slot = (ft_slot_t *)calloc((size_t)n_slots, sizeof(ft_slot_t));
  if (!t || !t->slot) {
    free(t);
    return achip_fail(ASCIICHAT_HIP_ERR_MEMORY, "out of memory");
  }
  t->n = n_slots;
  for (int i = 0; i < n_slots; i++) {
    t->slot[i].cur = -1;
    pthread_mutex_init(&t->slot[i].mu, NULL);
  }
  *table = t;
  return 0;
}

void asciichat_hip_frame_table_destroy(asciichat_hip_frame_table_t *t) {
  if (!t)
    return;
  for (int i = 0; i < t->n; i++) {
    pthread_mutex_destroy(&t->slot[i].mu);
    if (t->slot[i].reader_done)
      (void)hipEventDestroy(t->slot[i].reader_done);
  }
  for (int i = 0; i < t->n; i++)
    for (int k = 0; k < 2; k++) {
      ft_slot_t *s = &t->slot[i];
      if (s->ready[k]) {
        (void)hipEventSynchronize(s->ready[k]);
        (void)hipEventDestroy(s->ready[k]);
      }
      if (s->dev[k])
        (void)hipFree(s->dev[k]);
      if (s->stage[k])
        (void)hipHostFree(s->stage[k]);
      if (s->rows_dev[k])
        (void)hipFree(s->rows_dev[k]);
    }
  free(t->slot);
  free(t);
}

/* the source rows that the targets sample (the sampler's own rule: render_stream.hpp stream_request / render_kernels.hpp
 * sample_frame_raw -- sy = min((y * y_ratio) >> 16, src_h - 1), mirrored under ACHIP_OP_FLIP_Y), ascending, unique.
 * Returns their number, or -1 when a target does not describe an h-row source. */
static int sampled_rows(const achip_frame_t *targets, int n_targets, uint32_t h, uint32_t *rows_out, uint8_t *mark) {
  memset(mark, 0, h);
  for (int i = 0; i < n_targets; i++) {
    const achip_frame_t *f = &targets[i];
    if (f->comp || (uint32_t)f->src_h != h || f->out_h <= 0)
      return -1;
    for (uint32_t y = 0; y < (uint32_t)f->out_h; y++) {
      uint32_t sy = (uint32_t)(((uint64_t)y * f->y_ratio) >> 16);
      if (sy > h - 1u)
        sy = h - 1u;
      if (f->ops & ACHIP_OP_FLIP_Y)
        sy = h - 1u - sy;
      mark[sy] = 1;
    }
  }
  int n = 0;
  for (uint32_t r = 0; r < h; r++)
    if (mark[r])
      rows_out[n++] = r;
  return n;
}

static int publish_common(asciichat_hip_frame_table_t *t, int slot, const void *blob, size_t blob_size,
                          const achip_frame_t *targets, int n_targets, void *stream);

int asciichat_hip_frame_table_publish(asciichat_hip_frame_table_t *t, int slot, const void *blob, size_t blob_size,
                                      void *stream) {
  return publish_common(t, slot, blob, blob_size, NULL, 0, stream);
}

/* The same publish moving only the rows that the given targets will sample (138 KB instead of 6.2 MB for 1080p ->
 * 80x24; collect_video_sources copies the whole blob twice per render thread, src/server/stream.c:221-463).  The frame
 * buffer keeps the full frame's layout, so descriptors, plans and output bytes are those of a full publish -- for
 * renders whose (out_h, y_ratio, flip) are among `targets`; any other row of the buffer is stale. */
int asciichat_hip_frame_table_publish_rows(asciichat_hip_frame_table_t *t, int slot, const void *blob, size_t blob_size,
                                           const achip_frame_t *targets, int n_targets, void *stream) {
  if (!targets || n_targets <= 0)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame_table_publish_rows: no targets");
  return publish_common(t, slot, blob, blob_size, targets, n_targets, stream);
}

static int publish_common(asciichat_hip_frame_table_t *t, int slot, const void *blob, size_t blob_size,
                          const achip_frame_t *targets, int n_targets, void *stream) {
  if (!t || slot < 0 || slot >= t->n || !blob)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame_table_publish: bad arguments");
  uint32_t w = 0, h = 0;
  const uint8_t *pixels = NULL;
  const int pr = achip_frame_blob_parse(blob, blob_size, false, &w, &h, &pixels);
  if (pr != 0) /* the conditions under which collect_video_sources skips the client's frame */
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame blob rejected (%s)",
                      pr == ACHIP_BLOB_SHORT ? "shorter than a header and one pixel"
                                             : (pr == ACHIP_BLOB_DIMS ? "dimensions out of range" : "size below 8 + 3*w*h"));
  const size_t bytes = (size_t)w * (size_t)h * 3u;
  ft_slot_t *s = &t->slot[slot];
  pthread_mutex_lock(&s->mu);
  const int k = s->cur == 0 ? 1 : 0; /* the buffer that does not hold the latest frame */
  int rc = 0;
  if (s->ready[k]) /* the previous upload into this buffer reads the same pinned staging block: let it finish */
    rc = achip_hip_check((int)hipEventSynchronize(s->ready[k]), "hipEventSynchronize(frame buffer)");
  /* renders that were handed this buffer (one publish ago it was the latest frame) may still be queued or running:
   * the upload goes behind everything enqueued so far on their streams */
  if (!rc && s->n_readers[k] > 0 && !s->reader_done)
    rc = achip_hip_check((int)hipEventCreateWithFlags(&s->reader_done, hipEventDisableTiming), "hipEventCreate");
  for (int r = 0; r < s->n_readers[k] && !rc; r++) {
    if (hipEventRecord(s->reader_done, s->reader[k][r]) != hipSuccess) {
      (void)hipGetLastError(); /* the consumer destroyed its stream: nothing of it is left to wait for */
      continue;
    }
    rc = achip_hip_check((int)hipStreamWaitEvent((hipStream_t)stream, s->reader_done, 0), "hipStreamWaitEvent(readers)");
  }
  if (!rc && s->readers_overflow[k])
    rc = achip_hip_check((int)hipDeviceSynchronize(), "hipDeviceSynchronize(readers)");
  s->n_readers[k] = 0;
  s->readers_overflow[k] = 0;
  if (!rc && s->cap[k] < bytes) {
    if (s->dev[k])
      (void)hipFree(s->dev[k]);
    s->dev[k] = NULL;
    s->cap[k] = 0;
    rc = achip_hip_check((int)hipMalloc((void **)&s->dev[k], bytes), "hipMalloc(frame)");
    if (!rc)
      s->cap[k] = bytes;
  }
  if (!rc && !s->ready[k])
    rc = achip_hip_check((int)hipEventCreateWithFlags(&s->ready[k], hipEventDisableTiming), "hipEventCreate");
  if (!rc && targets) {
    /* sampled rows only: [index table][rows] packed into pinned staging by the host, ONE DMA, one scatter launch */
    const size_t row_bytes = (size_t)w * 3u;
    uint32_t *rows = (uint32_t *)malloc((size_t)h * sizeof(uint32_t));
    uint8_t *mark = (uint8_t *)malloc(h);
    const int n_rows = rows && mark ? sampled_rows(targets, n_targets, h, rows, mark) : -2;
    free(mark);
    if (n_rows < 0) {
      free(rows);
      pthread_mutex_unlock(&s->mu);
      return n_rows == -2 ? achip_fail(ASCIICHAT_HIP_ERR_MEMORY, "out of memory")
                          : achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM,
                                       "frame_table_publish_rows: a target does not describe this %ux%u frame", w, h);
    }
    const size_t table = ((size_t)n_rows * 4u + 15u) & ~(size_t)15;
    const size_t staged = table + (size_t)n_rows * row_bytes;
    if (s->stage_cap[k] < staged) {
      if (s->stage[k])
        (void)hipHostFree(s->stage[k]);
      s->stage[k] = NULL;
      s->stage_cap[k] = 0;
      rc = achip_hip_check((int)hipHostMalloc((void **)&s->stage[k], staged, hipHostMallocDefault), "hipHostMalloc(stage)");
      if (!rc)
        s->stage_cap[k] = staged;
    }
    if (!rc && s->rows_cap[k] < staged) {
      if (s->rows_dev[k])
        (void)hipFree(s->rows_dev[k]);
      s->rows_dev[k] = NULL;
      s->rows_cap[k] = 0;
      rc = achip_hip_check((int)hipMalloc((void **)&s->rows_dev[k], staged), "hipMalloc(row staging)");
      if (!rc)
        s->rows_cap[k] = staged;
    }
    if (!rc) {
      memcpy(s->stage[k], rows, (size_t)n_rows * 4u);
      for (int r = 0; r < n_rows; r++)
        memcpy(s->stage[k] + table + (size_t)r * row_bytes, pixels + (size_t)rows[r] * row_bytes, row_bytes);
      rc = achip_hip_check((int)hipMemcpyAsync(s->rows_dev[k], s->stage[k], staged, hipMemcpyHostToDevice, (hipStream_t)stream),
                           "hipMemcpyAsync(rows)");
    }
    if (!rc)
      rc = achip_hip_check(achip_launch_scatter_rows(s->rows_dev[k], (uint32_t)n_rows, (uint32_t)row_bytes, s->dev[k],
                                                     (uint64_t)row_bytes, stream),
                           "row scatter launch");
    free(rows);
  }
  const void *src = pixels;
  if (!rc && !targets && !achip_pool_device_ptr(pixels)) { /* pageable blob: one copy into pinned staging, then DMA */
    if (s->stage_cap[k] < bytes) {
      if (s->stage[k])
        (void)hipHostFree(s->stage[k]);
      s->stage[k] = NULL;
      s->stage_cap[k] = 0;
      rc = achip_hip_check((int)hipHostMalloc((void **)&s->stage[k], bytes, hipHostMallocDefault), "hipHostMalloc(stage)");
      if (!rc)
        s->stage_cap[k] = bytes;
    }
    if (!rc) {
      memcpy(s->stage[k], pixels, bytes);
      src = s->stage[k];
    }
  }
  if (!rc && !targets)
    rc = achip_hip_check((int)hipMemcpyAsync(s->dev[k], src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream),
                         "hipMemcpyAsync(frame)");
  if (!rc)
    rc = achip_hip_check((int)hipEventRecord(s->ready[k], (hipStream_t)stream), "hipEventRecord");
  if (!rc) {
    s->cur = k;
    s->w = (int)w;
    s->h = (int)h;
    s->generation++;
  }
  pthread_mutex_unlock(&s->mu);
  return rc;
}

int asciichat_hip_frame_table_latest(asciichat_hip_frame_table_t *t, int slot, void *consumer_stream,
                                     const uint8_t **pixels_dev, int *width, int *height, uint64_t *generation) {
  if (!t || slot < 0 || slot >= t->n || !pixels_dev)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame_table_latest: bad arguments");
  ft_slot_t *s = &t->slot[slot];
  pthread_mutex_lock(&s->mu);
  int rc = 0;
  if (s->cur < 0) {
    *pixels_dev = NULL; /* has_video = false (stream.c:272-274) */
    if (width)
      *width = 0;
    if (height)
      *height = 0;
    if (generation)
      *generation = 0;
  } else {
    /* work queued on consumer_stream after this call sees the complete upload */
    rc = achip_hip_check((int)hipStreamWaitEvent((hipStream_t)consumer_stream, s->ready[s->cur], 0), "hipStreamWaitEvent");
    /* remember the reader: the upload that will overwrite this buffer (the publish after next) waits for it */
    const int k = s->cur;
    int known = 0;
    for (int r = 0; r < s->n_readers[k]; r++)
      known |= s->reader[k][r] == (hipStream_t)consumer_stream;
    if (!known) {
      if (s->n_readers[k] < FT_MAX_READERS)
        s->reader[k][s->n_readers[k]++] = (hipStream_t)consumer_stream;
      else
        s->readers_overflow[k] = 1;
    }
    *pixels_dev = s->dev[s->cur];
    if (width)
      *width = s->w;
    if (height)
      *height = s->h;
    if (generation)
      *generation = s->generation;
  }
  pthread_mutex_unlock(&s->mu);
  return rc;
}

/* A consumer that destroys a stream it passed to latest() MUST call this first: publish() records events on the remembered
 * handles, and using a destroyed hipStream_t is undefined (ROCm happens to validate handles, which is what the skip in
 * publish() relies on as a last resort; a handle recycled by a later stream would make uploads wait on unrelated work). */
void asciichat_hip_frame_table_forget_stream(asciichat_hip_frame_table_t *t, void *consumer_stream) {
  if (!t)
    return;
  for (int i = 0; i < t->n; i++) {
    ft_slot_t *s = &t->slot[i];
    pthread_mutex_lock(&s->mu);
    for (int k = 0; k < 2; k++)
      for (int r = 0; r < s->n_readers[k];)
        if (s->reader[k][r] == (hipStream_t)consumer_stream)
          s->reader[k][r] = s->reader[k][--s->n_readers[k]];
        else
          r++;
    pthread_mutex_unlock(&s->mu);
  }
}
