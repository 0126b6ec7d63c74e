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

#include "frame_table_priv.h"

#define FT_SLOT_TURNED_DENSE (-1000) /* latest_one() to latest_frames(): look at the slot again */

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
  pthread_mutex_init(&t->batch_mu, NULL);
  pthread_mutex_init(&t->ev_mu, NULL);
  ft_dense_init(t);
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
  for (int k = 0; k < 2; k++) {
    if (t->batch_done[k]) {
      (void)hipEventSynchronize(t->batch_done[k]);
      (void)hipEventDestroy(t->batch_done[k]);
    }
    if (t->batch_host[k])
      (void)hipHostFree(t->batch_host[k]);
    if (t->batch_dev[k])
      (void)hipFree(t->batch_dev[k]);
  }
  for (int k = 0; k < FT_RING; k++)
    if (t->ring[k]) {
      (void)hipEventSynchronize(t->ring[k]);
      (void)hipEventDestroy(t->ring[k]);
    }
  ft_dense_destroy(t);
  pthread_mutex_destroy(&t->batch_mu);
  pthread_mutex_destroy(&t->ev_mu);
  free(t->slot);
  free(t);
}

/* the upload of buffer k is complete (host-side wait), or work queued on `consumer` later will see it complete */
static int upload_wait(asciichat_hip_frame_table_t *t, ft_slot_t *s, int k, int on_stream, hipStream_t consumer) {
  if (!s->batch_of[k]) {
    if (!s->ready[k])
      return 0;
    return on_stream ? achip_hip_check((int)hipStreamWaitEvent(consumer, s->ready[k], 0), "hipStreamWaitEvent")
                     : achip_hip_check((int)hipEventSynchronize(s->ready[k]), "hipEventSynchronize(frame buffer)");
  }
  const unsigned B = s->batch_of[k];
  int rc = 0;
  pthread_mutex_lock(&t->ev_mu);
  if (t->ring_seq[B % FT_RING] == B) /* else: out of the ring, so it was waited for when its ring slot was reused */
    rc = on_stream ? achip_hip_check((int)hipStreamWaitEvent(consumer, t->ring[B % FT_RING], 0), "hipStreamWaitEvent")
                   : achip_hip_check((int)hipEventSynchronize(t->ring[B % FT_RING]), "hipEventSynchronize(batch)");
  pthread_mutex_unlock(&t->ev_mu);
  return rc;
}

/* reader streams collected over a batch's slots: one "everything enqueued so far" event per distinct stream */
typedef struct {
  hipStream_t s[64];
  int n;
  unsigned done_batch; /* a batch publish known to be complete (the slots of a tick were mostly uploaded by one batch) */
} reader_set_t;

static int latest_one(asciichat_hip_frame_table_t *t, int slot, void *consumer_stream, const uint8_t **pixels_dev, int *width,
                      int *height, uint64_t *generation, unsigned *waited_batch);
static int publish_common(asciichat_hip_frame_table_t *t, int slot, const void *blob, size_t blob_size,
                          const achip_frame_t *targets, int n_targets, void *stream);
static int slot_prepare(asciichat_hip_frame_table_t *t, ft_slot_t *s, size_t bytes, void *stream, int *k_out, reader_set_t *defer);

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
  /* the fallible host-only work first: slot_prepare consumes the buffer's reader list, and a publish that
   * then failed would leave a later one on another stream unordered against renders still reading the buffer */
  uint32_t *rows = NULL;
  int n_rows = 0;
  if (targets) {
    rows = (uint32_t *)malloc((size_t)h * sizeof(uint32_t));
    uint8_t *mark = (uint8_t *)malloc(h);
    n_rows = rows && mark ? achip_sampled_rows(targets, n_targets, h, rows, mark) : -2;
    free(mark);
    if (n_rows < 0) {
      free(rows);
      return n_rows == -2 ? achip_fail(ASCIICHAT_HIP_ERR_MEMORY, "out of memory")
                          : achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM,
                                       "frame_table_publish_rows: a target does not describe this %ux%u frame", w, h);
    }
  }
  pthread_mutex_lock(&s->mu);
  int k = 0; /* the buffer that does not hold the latest frame */
  int rc = slot_prepare(t, s, bytes, stream, &k, NULL);
  if (!rc && targets) {
    /* sampled rows only: [index table][rows] packed into pinned staging by the host, ONE DMA, one scatter launch */
    const size_t row_bytes = (size_t)w * 3u;
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
  }
  free(rows);
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
    s->batch_of[k] = 0;
    s->cur = k;
    s->dense = 0;
    s->pend = 0; /* a staged sampled image that was not committed yet is older than this frame: it must not win at the commit */
    s->w = (int)w;
    s->h = (int)h;
    s->generation++;
  }
  pthread_mutex_unlock(&s->mu);
  return rc;
}

/* the part of a publish that happens under the slot's lock before the upload: pick the buffer that does not hold the
 * latest frame; the previous upload into it reads the same pinned staging block: let it finish; renders that were handed
 * this buffer (one publish ago it was the latest frame) may still be queued or running: the upload goes behind
 * everything enqueued so far on their streams; make room.  Returns the buffer index through *k_out; the caller enqueues the upload and calls slot_commit. */
static int slot_prepare(asciichat_hip_frame_table_t *t, ft_slot_t *s, size_t bytes, void *stream, int *k_out, reader_set_t *defer) {
  const int k = s->cur == 0 ? 1 : 0;
  int rc = 0;
  if (!(defer && s->batch_of[k] && s->batch_of[k] == defer->done_batch)) {
    rc = upload_wait(t, s, k, 0, NULL);
    if (!rc && defer)
      defer->done_batch = s->batch_of[k];
  }
  if (!rc && s->n_readers[k] > 0 && !s->reader_done && !defer)
    rc = achip_hip_check((int)hipEventCreateWithFlags(&s->reader_done, hipEventDisableTiming), "hipEventCreate");
  for (int r = 0; r < s->n_readers[k] && !rc; r++) {
    if (defer) { /* a batch: the caller makes `stream` wait for every distinct reader stream once */
      int known = 0;
      for (int q = 0; q < defer->n; q++)
        known |= defer->s[q] == s->reader[k][r];
      if (known)
        continue;
      if (defer->n < (int)(sizeof(defer->s) / sizeof(defer->s[0]))) {
        defer->s[defer->n++] = s->reader[k][r];
        continue;
      }
      if (!s->reader_done)
        rc = achip_hip_check((int)hipEventCreateWithFlags(&s->reader_done, hipEventDisableTiming), "hipEventCreate");
      if (rc)
        break;
    }
    if (hipEventRecord(s->reader_done, s->reader[k][r]) != hipSuccess) {
      (void)hipGetLastError();
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
  if (!rc && !s->ready[k] && !defer)
    rc = achip_hip_check((int)hipEventCreateWithFlags(&s->ready[k], hipEventDisableTiming), "hipEventCreate");
  *k_out = k;
  return rc;
}

/* A whole tick's clients at once: every client's sampled part -- the sampled rows, or only the
 * sampled pixels when the targets read at most half of the frame's columns (achip_sample_set_*) -- is packed into ONE
 * pinned block behind a table of 32-byte records, sent with ONE DMA and put in place by ONE launch, instead of a DMA and
 * a launch per client (256 clients: ~12 ms of per-call latencies; profiles/r03_bench.json tick_e2e).  The batch is guarded
 * by ONE event (the table's ring) and waits once per distinct reader stream.  Slots must be distinct; they are locked in
 * ascending order.  `targets` may be the tick's render descriptors: a blob is matched with those set up for its geometry,
 * a blob none of them describes is refused.  Semantics per slot are those of frame_table_publish_rows. */
int asciichat_hip_frame_table_publish_rows_batch(asciichat_hip_frame_table_t *t, const int *slots, const void *const *blobs,
                                                 const size_t *blob_sizes, int n, const achip_frame_t *targets, int n_targets,
                                                 void *stream) {
  if (!t || !slots || !blobs || !blob_sizes || n <= 0 || n > 65535 || !targets || n_targets <= 0)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame_table_publish_rows_batch: bad arguments");
  typedef struct {
    int slot, k, n_rows, set;
    uint32_t w, h;
    const uint8_t *pixels;
    size_t off;
  } item_t;
  item_t *it = (item_t *)calloc((size_t)n, sizeof(item_t));
  int *order = (int *)malloc((size_t)n * sizeof(int));
  enum { SETS_MAX = 8 };
  achip_sample_set_t sets[SETS_MAX];
  int n_sets = 0;
  if (!it || !order) {
    free(it);
    free(order);
    return achip_fail(ASCIICHAT_HIP_ERR_MEMORY, "out of memory");
  }
  int rc = 0;
  for (int i = 0; i < n && !rc; i++) {
    it[i].slot = slots[i];
    if (slots[i] < 0 || slots[i] >= t->n || !blobs[i])
      rc = achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame_table_publish_rows_batch: bad slot or blob at %d", i);
    else if (achip_frame_blob_parse(blobs[i], blob_sizes[i], false, &it[i].w, &it[i].h, &it[i].pixels) != 0)
      rc = achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame blob %d rejected", i);
    order[i] = i;
  }
  /* ascending slot order (insertion sort: n is a tick's client count), duplicates refused */
  for (int i = 1; i < n && !rc; i++)
    for (int j = i; j > 0 && it[order[j - 1]].slot > it[order[j]].slot; j--) {
      const int tmp = order[j];
      order[j] = order[j - 1];
      order[j - 1] = tmp;
    }
  for (int i = 1; i < n && !rc; i++)
    if (it[order[i]].slot == it[order[i - 1]].slot)
      rc = achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame_table_publish_rows_batch: slot %d named twice", it[order[i]].slot);
  /* sizes first: the block's layout (what is sampled depends on a frame's geometry only: one set per distinct w x h) */
  size_t total = ((size_t)n * 32u + 15u) & ~(size_t)15;
  uint32_t max_rows = 0, max_row_bytes = 0;
  for (int i = 0; i < n && !rc; i++) {
    int si = 0;
    while (si < n_sets && (sets[si].w != it[i].w || sets[si].h != it[i].h))
      si++;
    if (si == n_sets) {
      if (n_sets == SETS_MAX) { /* more geometries than this in one tick: rebuild the last set as needed */
        achip_sample_set_free(&sets[SETS_MAX - 1]);
        n_sets--;
        si = n_sets;
      }
      const int b = achip_sample_set_build(&sets[si], targets, n_targets, it[i].w, it[i].h);
      if (b) {
        rc = b == -2 ? achip_fail(ASCIICHAT_HIP_ERR_MEMORY, "out of memory")
                     : achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM,
                                  "frame_table_publish_rows_batch: a target does not describe blob %d (%ux%u)", i, it[i].w, it[i].h);
        break;
      }
      n_sets++;
    }
    it[i].set = si; /* (the last set slot may be rebuilt for another geometry later: re-checked when packing) */
    it[i].n_rows = sets[si].n_rows;
    it[i].off = total;
    total += achip_sample_set_block_bytes(&sets[si]);
    if ((uint32_t)sets[si].n_rows > max_rows)
      max_rows = (uint32_t)sets[si].n_rows;
    const uint32_t work = sets[si].n_cols ? (uint32_t)sets[si].n_cols * 3u : it[i].w * 3u;
    if (work > max_row_bytes)
      max_row_bytes = work;
  }
  if (!rc && total > 0xFFFFFFF0u)
    rc = achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame_table_publish_rows_batch: staged block exceeds 4 GiB");
  int locked = 0, have_batch = 0, par = 0;
  unsigned batch_seq = 0;
  reader_set_t readers;
  readers.n = 0;
  readers.done_batch = 0;
  if (!rc) {
    pthread_mutex_lock(&t->batch_mu);
    have_batch = 1;
    par = (int)(t->batch_no++ & 1u);
    batch_seq = t->batch_no ? t->batch_no : (t->batch_no = 1u); /* never 0: that means "own event" */
    if (t->batch_done[par]) /* the DMA of two batches ago read this block */
      rc = achip_hip_check((int)hipEventSynchronize(t->batch_done[par]), "hipEventSynchronize(batch staging)");
    if (!rc && t->batch_cap[par] < total) {
      if (t->batch_host[par])
        (void)hipHostFree(t->batch_host[par]);
      if (t->batch_dev[par])
        (void)hipFree(t->batch_dev[par]);
      t->batch_host[par] = t->batch_dev[par] = NULL;
      t->batch_cap[par] = 0;
      const size_t cap = total + total / 4;
      rc = achip_hip_check((int)hipHostMalloc((void **)&t->batch_host[par], cap, hipHostMallocDefault), "hipHostMalloc(batch staging)");
      if (!rc)
        rc = achip_hip_check((int)hipMalloc((void **)&t->batch_dev[par], cap), "hipMalloc(batch staging)");
      if (!rc)
        t->batch_cap[par] = cap;
    }
    if (!rc && !t->batch_done[par])
      rc = achip_hip_check((int)hipEventCreateWithFlags(&t->batch_done[par], hipEventDisableTiming), "hipEventCreate");
  }
  /* per slot, in ascending order: lock, prepare the target buffer, pack the record and the sampled part */
  for (int q = 0; q < n && !rc; q++) {
    item_t *I = &it[order[q]];
    ft_slot_t *s = &t->slot[I->slot];
    pthread_mutex_lock(&s->mu);
    locked = q + 1;
    rc = slot_prepare(t, s, (size_t)I->w * I->h * 3u, stream, &I->k, &readers);
    if (rc)
      break;
    achip_sample_set_t *S = &sets[I->set];
    achip_sample_set_t tmp;
    if (S->w != I->w || S->h != I->h) { /* its slot was recycled for another geometry (more than SETS_MAX in this tick) */
      if (achip_sample_set_build(&tmp, targets, n_targets, I->w, I->h)) {
        rc = achip_fail(ASCIICHAT_HIP_ERR_MEMORY, "out of memory");
        break;
      }
      S = &tmp;
    }
    achip_sample_set_pack(S, I->pixels, t->batch_host[par] + I->off);
    struct {
      uint64_t frame;
      uint32_t off, n_rows, row_bytes, n_cols, pad[2];
    } rec = {(uint64_t)(uintptr_t)s->dev[I->k], (uint32_t)I->off, (uint32_t)S->n_rows, I->w * 3u, (uint32_t)S->n_cols, {0, 0}};
    memcpy(t->batch_host[par] + (size_t)order[q] * 32u, &rec, 32);
    if (S == &tmp)
      achip_sample_set_free(&tmp);
  }
  /* renders that were handed the buffers about to be overwritten: the uploads go behind everything enqueued so far on
   * their streams -- once per distinct stream, not once per slot */
  if (!rc && readers.n > 0) {
    hipEvent_t ev = NULL;
    rc = achip_hip_check((int)hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
    for (int q = 0; q < readers.n && !rc; q++) {
      if (hipEventRecord(ev, readers.s[q]) != hipSuccess) { /* a stream destroyed without forget_stream: see there */
        (void)hipGetLastError();
        continue;
      }
      rc = achip_hip_check((int)hipStreamWaitEvent((hipStream_t)stream, ev, 0), "hipStreamWaitEvent(readers)");
    }
    if (ev)
      (void)hipEventDestroy(ev); /* (released once the recorded work has completed) */
  }
  if (!rc)
    rc = achip_hip_check((int)hipMemcpyAsync(t->batch_dev[par], t->batch_host[par], total, hipMemcpyHostToDevice, (hipStream_t)stream),
                         "hipMemcpyAsync(batch rows)");
  if (!rc)
    rc = achip_hip_check(achip_launch_scatter_rows_batch(t->batch_dev[par], (uint32_t)n, max_rows, max_row_bytes, stream),
                         "batched row scatter launch");
  if (!rc)
    rc = achip_hip_check((int)hipEventRecord(t->batch_done[par], (hipStream_t)stream), "hipEventRecord(batch)");
  if (!rc) { /* the batch's completion event, for every slot of it */
    pthread_mutex_lock(&t->ev_mu);
    const unsigned ri = batch_seq % FT_RING;
    if (t->ring[ri] && t->ring_seq[ri]) /* the batch this slot stood for leaves the ring: make "not in the ring" mean "done" */
      rc = achip_hip_check((int)hipEventSynchronize(t->ring[ri]), "hipEventSynchronize(batch ring)");
    if (!rc && !t->ring[ri])
      rc = achip_hip_check((int)hipEventCreateWithFlags(&t->ring[ri], hipEventDisableTiming), "hipEventCreate");
    if (!rc)
      rc = achip_hip_check((int)hipEventRecord(t->ring[ri], (hipStream_t)stream), "hipEventRecord(batch ring)");
    if (!rc)
      t->ring_seq[ri] = batch_seq;
    pthread_mutex_unlock(&t->ev_mu);
  }
  for (int q = 0; q < locked; q++) { /* commit (or leave untouched on failure) and unlock */
    item_t *I = &it[order[q]];
    ft_slot_t *s = &t->slot[I->slot];
    if (!rc) {
      s->batch_of[I->k] = batch_seq;
      s->cur = I->k;
      s->dense = 0;
      s->pend = 0;
      s->w = (int)I->w;
      s->h = (int)I->h;
      s->generation++;
    }
    pthread_mutex_unlock(&s->mu);
  }
  if (have_batch)
    pthread_mutex_unlock(&t->batch_mu);
  for (int i = 0; i < n_sets; i++)
    achip_sample_set_free(&sets[i]);
  free(order);
  free(it);
  return rc;
}

/* The latest frames of a tick's clients straight into their render descriptors: frames[i].src = the device frame of
 * slots[i] when it has one AND its geometry is what the descriptor was set up for (src_w x src_h), else NULL (no video yet,
 * or the client changed its resolution: set the descriptor up again).  The consumer stream is registered with every
 * frame handed out, as frame_table_latest does.  Returns the number of descriptors that got a source, < 0 on error. */
int asciichat_hip_frame_table_latest_frames(asciichat_hip_frame_table_t *t, const int *slots, int n, void *consumer_stream,
                                            achip_frame_t *frames) {
  if (!t || !slots || !frames || n < 0)
    return -achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame_table_latest_frames: bad arguments");
  int with_video = 0;
  unsigned waited_batch = 0; /* frames of one batch publish share its event: the consumer stream waits for it once */
  unsigned waited_dense = 0; /* ... and so do the sampled images of one commit (one bit per ring block) */
  for (int i = 0; i < n; i++) {
    const uint8_t *px = NULL;
    int w = 0, h = 0;
    if (slots[i] < 0 || slots[i] >= t->n)
      return -achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame_table_latest_frames: bad slot at %d", i);
    ft_slot_t *s = &t->slot[slots[i]];
  again:
    pthread_mutex_lock(&s->mu);
    if (s->dense) { /* the latest frame is the image one target samples of it (frame_dense.c) */
      const ft_dense_ref_t ref = {s->dense_blk, s->dense_seq, s->dense_off, s->dense_key, s->dense_geo};
      pthread_mutex_unlock(&s->mu);
      int drc = 0;
      const int got = ft_dense_latest(t, &ref, consumer_stream, &frames[i], &waited_dense, &drc);
      if (drc)
        return -drc;
      if (got < 0) /* its ring block was refilled between the two looks */
        goto again;
      if (!got)
        frames[i].src = NULL;
      with_video += got;
      continue;
    }
    pthread_mutex_unlock(&s->mu);
    const int rc = latest_one(t, slots[i], consumer_stream, &px, &w, &h, NULL, &waited_batch);
    if (rc == FT_SLOT_TURNED_DENSE) /* a commit slipped in between the two looks at the slot */
      goto again;
    if (rc)
      return -rc;
    const int fits = px && w == frames[i].src_w && h == frames[i].src_h;
    frames[i].src = fits ? px : NULL;
    with_video += fits;
  }
  return with_video;
}

/* latest() of one slot; *waited_batch (optional): a batch publish `consumer_stream` already waits for */
static int latest_one(asciichat_hip_frame_table_t *t, int slot, void *consumer_stream, const uint8_t **pixels_dev, int *width,
                      int *height, uint64_t *generation, unsigned *waited_batch) {
  ft_slot_t *s = &t->slot[slot];
  pthread_mutex_lock(&s->mu);
  int rc = 0;
  if (s->dense) { /* no full frame to point at: the slot holds what one render target samples of it */
    *pixels_dev = NULL;
    pthread_mutex_unlock(&s->mu);
    if (waited_batch) /* latest_frames(): it serves such slots itself */
      return FT_SLOT_TURNED_DENSE;
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM,
                      "frame_table_latest: slot %d holds a sampled image (frame_table_stage); use frame_table_latest_frames", slot);
  }
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
    const int k = s->cur;
    if (!(waited_batch && s->batch_of[k] && s->batch_of[k] == *waited_batch)) {
      rc = upload_wait(t, s, k, 1, (hipStream_t)consumer_stream);
      if (!rc && waited_batch)
        *waited_batch = s->batch_of[k];
    }
    /* remember the reader: the upload that will overwrite this buffer (the publish after next) waits for it */
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

int asciichat_hip_frame_table_latest(asciichat_hip_frame_table_t *t, int slot, void *consumer_stream,
                                     const uint8_t **pixels_dev, int *width, int *height, uint64_t *generation) {
  if (!t || slot < 0 || slot >= t->n || !pixels_dev)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "frame_table_latest: bad arguments");
  return latest_one(t, slot, consumer_stream, pixels_dev, width, height, generation, NULL);
}

/* A consumer that destroys a stream it passed to latest() MUST call this first: publish() records events on the remembered
 * handles, and using a destroyed hipStream_t is undefined (ROCm happens to validate handles, which is what the skip in
 * publish() relies on as a last resort; a handle recycled by a later stream would make uploads wait on unrelated work). */
void asciichat_hip_frame_table_forget_stream(asciichat_hip_frame_table_t *t, void *consumer_stream) {
  if (!t)
    return;
  ft_dense_forget_stream(t, (hipStream_t)consumer_stream);
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
