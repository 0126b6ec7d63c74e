/*
 * oracle_bench.c -- times the CPU oracle for bench.py's cpu_baseline leg.
 * TEST/BENCH INFRASTRUCTURE ONLY (see asciichat_oracle.h).
 */
#include "asciichat_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <time.h>

typedef struct {
  const uint8_t *rgb;
  int sw, sh, w, h, color, mode, iters;
  const char *palette;
  uint64_t bytes;
} job_t;

static void *worker(void *p) {
  job_t *j = (job_t *)p;
  for (int i = 0; i < j->iters; i++) {
    size_t n = 0;
    char *s = orc_convert_with_caps(j->rgb, j->sw, j->sh, j->w, j->h, j->color, j->mode, false, false, false,
                                    j->palette, &n);
    j->bytes += n;
    free(s);
  }
  return NULL;
}

/* Runs `iters` conversions on each of `threads` threads (shared read-only source frame);
 * returns elapsed seconds, *out_bytes = total output bytes produced. */
double orc_bench_convert(const uint8_t *rgb, int sw, int sh, int w, int h, int color_level, int render_mode,
                         const char *palette, int iters, int threads, uint64_t *out_bytes) {
  if (threads < 1)
    threads = 1;
  if (threads > 256)
    threads = 256;
  pthread_t tid[256];
  job_t jobs[256];
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int t = 0; t < threads; t++) {
    jobs[t] = (job_t){rgb, sw, sh, w, h, color_level, render_mode, iters, palette, 0};
    pthread_create(&tid[t], NULL, worker, &jobs[t]);
  }
  uint64_t total = 0;
  for (int t = 0; t < threads; t++) {
    pthread_join(tid[t], NULL);
    total += jobs[t].bytes;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (out_bytes)
    *out_bytes = total;
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
