/*
 * frame_table_threads_mock.c -- the frame table's host-side locking under load, on the mock runtime (events are tokens, so
 * what is exercised is frame_table.c's own synchronisation: slot locks taken in ascending order by batches, the batch
 * staging lock, the ring of batch events, reader bookkeeping).  TESTS ONLY; run under ThreadSanitizer by
 * tests/test_mock_gpu.py.  Publishers (whole blobs, sampled rows, whole-tick batches over overlapping slot sets, sampled images staged
 * per blob and per tick -- frame_dense.c's block ring, carry-forward and ingest pool) race with
 * readers (latest, latest_frames + a "render" whose output depends on exactly the pixels the sampler reads).  A reader checks:
 * a frame handed out has the geometry the table reports, generations never go backwards, and the rendered line equals the
 * line of one of the images that can be in that slot.
 * usage: frame_table_threads_mock <seconds>
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "achip_host.h"
#include "asciichat_hip.h"

#define SLOTS 12
#define IMAGES 4
#define W 160
#define H 120

static asciichat_hip_frame_table_t *table;
static uint8_t *blob[IMAGES];
static size_t blob_bytes;
static char want[IMAGES][96];
static int stop_flag;
#define stop __atomic_load_n(&stop_flag, __ATOMIC_RELAXED)
static int failures;

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static void fail(const char *what) {
  fprintf(stderr, "%s [%s]\n", what, asciichat_hip_last_error());
  __atomic_add_fetch(&failures, 1, __ATOMIC_RELAXED);
}

static void target(achip_frame_t *f) { (void)achip_frame_setup(f, NULL, W, H, 40, 12, 0, false, false, true); }

/* one frame through a plan: the stand-in's line for it */
static int render_line(const achip_frame_t *f, char *out, size_t out_size) {
  asciichat_hip_plan_t *plan = NULL;
  if (asciichat_hip_plan_create(&plan, 1, "   ...',;:clodxkO0KXNWM", f, 1) != 0)
    return -1;
  const size_t stride = asciichat_hip_plan_out_stride(plan);
  uint8_t *slab = (uint8_t *)malloc(stride);
  uint32_t len = 0;
  int rc = asciichat_hip_plan_render(plan, slab, stride, &len, NULL);
  if (!rc && len < out_size) {
    memcpy(out, slab, len);
    out[len] = 0;
  } else
    rc = -1;
  free(slab);
  asciichat_hip_plan_destroy(plan);
  return rc;
}

static void *publisher(void *arg) {
  const int id = (int)(intptr_t)arg;
  unsigned x = 99u + (unsigned)id;
  achip_frame_t tgt;
  target(&tgt);
  while (!stop) {
    x ^= x << 13, x ^= x >> 17, x ^= x << 5;
    const int img = (int)(x % IMAGES), form = (int)((x >> 8) % 5);
    if (form == 0) {
      if (asciichat_hip_frame_table_publish(table, (int)((x >> 16) % SLOTS), blob[img], blob_bytes, NULL) != 0)
        fail("publish");
    } else if (form == 1) {
      if (asciichat_hip_frame_table_publish_rows(table, (int)((x >> 16) % SLOTS), blob[img], blob_bytes, &tgt, 1, NULL) != 0)
        fail("publish_rows");
    } else if (form == 3) { /* sampled images: a few receive-thread style stage() calls, then the tick's commit -- racing with
                               the other publishers' stages and commits (a commit waits for gathers in flight) */
      for (int k = 0; k < 3; k++) {
        x ^= x << 13, x ^= x >> 17, x ^= x << 5;
        if (asciichat_hip_frame_table_stage(table, (int)(x % SLOTS), blob[(img + k) % IMAGES], blob_bytes, &tgt) != 0)
          fail("stage");
      }
      if (asciichat_hip_frame_table_commit(table, NULL) != 0)
        fail("commit");
    } else if (form == 4) { /* ... and a whole tick on the ingest pool */
      int slots[SLOTS], n = 0;
      const void *blobs[SLOTS];
      size_t sizes[SLOTS];
      for (int s = (int)((x >> 16) % 2); s < SLOTS; s += 1 + (int)((x >> 20) % 2)) {
        slots[n] = s;
        blobs[n] = blob[(img + n) % IMAGES];
        sizes[n++] = blob_bytes;
      }
      if (asciichat_hip_frame_table_publish_sampled_batch(table, slots, blobs, sizes, n, &tgt, 1, NULL) != 0)
        fail("publish_sampled_batch");
    } else {
      int slots[SLOTS], n = 0;
      const void *blobs[SLOTS];
      size_t sizes[SLOTS];
      for (int s = (int)((x >> 16) % 3); s < SLOTS; s += 1 + (int)((x >> 20) % 3)) {
        slots[n] = SLOTS - 1 - s; /* descending on purpose: the table orders the locks itself */
        blobs[n] = blob[(img + n) % IMAGES];
        sizes[n++] = blob_bytes;
      }
      if (asciichat_hip_frame_table_publish_rows_batch(table, slots, blobs, sizes, n, &tgt, 1, NULL) != 0)
        fail("publish_rows_batch");
    }
  }
  return NULL;
}

static void *reader(void *arg) {
  const int id = (int)(intptr_t)arg;
  uint64_t last_gen[SLOTS] = {0};
  void *stream = (void *)(intptr_t)(0x1000 + id); /* a token: the mock never dereferences stream handles */
  int slots[SLOTS];
  for (int s = 0; s < SLOTS; s++)
    slots[s] = s;
  while (!stop) {
    achip_frame_t frames[SLOTS];
    for (int s = 0; s < SLOTS; s++)
      target(&frames[s]);
    if (id % 2) {
      if (asciichat_hip_frame_table_latest_frames(table, slots, SLOTS, stream, frames) < 0)
        fail("latest_frames");
    } else {
      for (int s = 0; s < SLOTS; s++) {
        const uint8_t *px = NULL;
        int w = 0, h = 0;
        uint64_t gen = 0;
        if (asciichat_hip_frame_table_latest(table, s, stream, &px, &w, &h, &gen) != 0) {
          if (px) /* a slot whose latest frame is a sampled image has no full frame to hand out: refused, px stays NULL */
            fail("latest");
          continue;
        }
        if (px && (w != W || h != H))
          fail("geometry of a frame handed out");
        if (gen < last_gen[s])
          fail("generation went backwards");
        last_gen[s] = gen;
        frames[s].src = px;
      }
    }
    /* (no pixel check here: a publisher may legitimately be overwriting the OTHER buffer of a slot, and on the mock a
     * "queued" upload is an immediate one -- ordering against readers is the GPU test's subject) */
  }
  asciichat_hip_frame_table_forget_stream(table, stream);
  return NULL;
}

int main(int argc, char **argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
  blob_bytes = 8 + (size_t)W * H * 3;
  unsigned x = 4242u;
  achip_frame_t tgt;
  for (int i = 0; i < IMAGES; i++) {
    blob[i] = (uint8_t *)malloc(blob_bytes);
    const uint8_t hdr[8] = {0, 0, 0, W, 0, 0, 0, H};
    memcpy(blob[i], hdr, 8);
    for (size_t k = 8; k < blob_bytes; k++) {
      x ^= x << 13, x ^= x >> 17, x ^= x << 5;
      blob[i][k] = (uint8_t)x;
    }
    target(&tgt);
    tgt.src = blob[i] + 8;
    if (render_line(&tgt, want[i], sizeof want[i]) != 0) {
      fail("reference line");
      return 1;
    }
  }
  if (asciichat_hip_frame_table_create(&table, SLOTS) != 0) {
    fail("frame_table_create");
    return 1;
  }
  pthread_t pub[4], rd[4];
  for (int i = 0; i < 4; i++)
    pthread_create(&pub[i], NULL, publisher, (void *)(intptr_t)i);
  for (int i = 0; i < 4; i++)
    pthread_create(&rd[i], NULL, reader, (void *)(intptr_t)i);
  const double t0 = now();
  while (now() - t0 < seconds) {
    struct timespec ts = {0, 20000000};
    nanosleep(&ts, NULL);
  }
  __atomic_store_n(&stop_flag, 1, __ATOMIC_RELAXED);
  for (int i = 0; i < 4; i++)
    pthread_join(pub[i], NULL);
  for (int i = 0; i < 4; i++)
    pthread_join(rd[i], NULL);
  /* quiescent: every slot that has a frame renders to the line of one of the images */
  int slots[SLOTS], with_video, matched = 0;
  achip_frame_t frames[SLOTS];
  for (int s = 0; s < SLOTS; s++) {
    slots[s] = s;
    target(&frames[s]);
  }
  with_video = asciichat_hip_frame_table_latest_frames(table, slots, SLOTS, NULL, frames);
  for (int s = 0; s < SLOTS && with_video >= 0; s++) {
    if (!frames[s].src)
      continue;
    char line[96];
    if (render_line(&frames[s], line, sizeof line) != 0) {
      fail("render of a published frame");
      continue;
    }
    int ok = 0;
    for (int i = 0; i < IMAGES; i++)
      ok |= strcmp(line, want[i]) == 0;
    if (!ok)
      fail("a slot holds pixels of no published image");
    matched += ok;
  }
  asciichat_hip_frame_table_destroy(table);
  for (int i = 0; i < IMAGES; i++)
    free(blob[i]);
  if (failures)
    return 1;
  printf("ok: 4 publishers (blobs, sampled rows, batches, staged sampled images) x 4 readers for %.1f s, %d slots with a frame, all of a published image\n",
         seconds, matched);
  return 0;
}
