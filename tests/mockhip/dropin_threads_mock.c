/*
 * dropin_threads_mock.c -- T threads through the drop-in entry point with coalescing forced, on the mock runtime with the
 * arithmetic stand-in for the kernels (mock_launch_simple.c).  TESTS ONLY; built by tests/mockgpu.py, run plain and under
 * ThreadSanitizer / AddressSanitizer by tests/test_mock_gpu.py.  Every thread has its own image, size and mode; every result
 * must equal the thread's first one and what the stand-in computes for that image (a frame handed to the wrong caller, a
 * descriptor torn between generations or a half-staged image changes the line).
 * usage: dropin_threads_mock <threads> <calls per thread>
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "asciichat_render.h"

extern int asciichat_hip_set_coalesce_min_callers(int n);
extern const char *asciichat_hip_last_error(void);

typedef struct {
  image_t *img;
  int id, calls, w, h, failed;
  terminal_capabilities_t caps;
  char first[128];
} job_t;

static pthread_barrier_t gate;

static void *worker(void *arg) {
  job_t *j = (job_t *)arg;
  pthread_barrier_wait(&gate);
  for (int k = 0; k < j->calls; k++) {
    char *s = ascii_convert_with_capabilities(j->img, j->w, j->h, &j->caps, false, false, k % 3 ? PALETTE_CHARS_STANDARD : "ab");
    if (!s) {
      fprintf(stderr, "thread %d call %d failed: %s\n", j->id, k, asciichat_hip_last_error());
      j->failed = 1;
      return NULL;
    }
    if (k == 0) {
      snprintf(j->first, sizeof j->first, "%s", s);
    } else if (strcmp(s, j->first) != 0) {
      fprintf(stderr, "thread %d call %d: got \"%s\", expected \"%s\"\n", j->id, k, s, j->first);
      j->failed = 1;
      free(s);
      return NULL;
    }
    free(s);
  }
  return NULL;
}

int main(int argc, char **argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 8, calls = argc > 2 ? atoi(argv[2]) : 200;
  asciichat_hip_set_coalesce_min_callers(1);
  job_t *jobs = (job_t *)calloc((size_t)T, sizeof(job_t));
  pthread_t *tid = (pthread_t *)calloc((size_t)T, sizeof(pthread_t));
  pthread_barrier_init(&gate, NULL, (unsigned)T);
  static const int dims[4][2] = {{160, 120}, {97, 61}, {320, 200}, {40, 30}};
  static const int terms[3][2] = {{40, 12}, {33, 17}, {80, 24}};
  for (int t = 0; t < T; t++) {
    const int w = dims[t % 4][0], h = dims[t % 4][1];
    jobs[t].img = t % 5 == 4 ? image_new_from_pool((size_t)w, (size_t)h) : image_new((size_t)w, (size_t)h);
    unsigned x = 777u + (unsigned)t;
    unsigned char *px = (unsigned char *)jobs[t].img->pixels;
    for (size_t i = 0; i < (size_t)w * (size_t)h * 3; i++) {
      x ^= x << 13, x ^= x >> 17, x ^= x << 5;
      px[i] = (unsigned char)x;
    }
    jobs[t].id = t;
    jobs[t].calls = calls;
    jobs[t].w = terms[t % 3][0];
    jobs[t].h = terms[t % 3][1];
    jobs[t].caps.color_level = (terminal_color_mode_t)(t % 4);
    jobs[t].caps.render_mode = t % 7 == 3 ? RENDER_MODE_HALF_BLOCK : RENDER_MODE_FOREGROUND;
    jobs[t].caps.utf8_support = true;
    pthread_create(&tid[t], NULL, worker, &jobs[t]);
  }
  int failed = 0;
  for (int t = 0; t < T; t++) {
    pthread_join(tid[t], NULL);
    failed |= jobs[t].failed;
  }
  /* different images give different lines (the stand-in really looks at the staged pixels) */
  for (int t = 1; t < T && !failed; t++)
    if (strcmp(jobs[t].first, jobs[0].first) == 0) {
      fprintf(stderr, "threads 0 and %d got the same line\n", t);
      failed = 1;
    }
  for (int t = 0; t < T; t++) {
    if (t % 5 == 4)
      image_destroy_to_pool(jobs[t].img);
    else
      image_destroy(jobs[t].img);
  }
  if (!failed)
    printf("ok: %d threads x %d calls through the combiner, every line as expected (\"%.40s...\")\n", T, calls, jobs[0].first);
  return failed;
}
