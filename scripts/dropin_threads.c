/* The unmodified-server scenario: T render threads (the reference runs one per client, src/server/render.c:1233), each
 * calling the reference's own entry point ascii_convert_with_capabilities() on its client's 1080p frame and freeing
 * the string -- through libasciichat_hip.so.  Reports calls/s for 1..T threads, pageable and pool (pinned) images.
 * Build: gcc -O2 -I include scripts/dropin_threads.c -o scripts/dropin_threads -L ascii-chat_amd -lasciichat_hip
 *        -Wl,-rpath,'$ORIGIN/../ascii-chat_amd' -lpthread
 * usage: dropin_threads [max_threads [src_w src_h term_w term_h color_level render_mode]] */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <time.h>

#include "asciichat_render.h"

typedef struct {
  image_t *img;
  int calls, w, h;
  terminal_capabilities_t caps;
  size_t bytes;
} job_t;

static pthread_barrier_t gate;

static void *worker(void *arg) {
  job_t *j = (job_t *)arg;
  /* the first result is the thread's reference: every later call on the same image must return the same bytes (the GPU
   * tests compare against the oracle; this catches a race that hands a caller somebody else's or a torn frame) */
  char *first = ascii_convert_with_capabilities(j->img, j->w, j->h, &j->caps, false, false, PALETTE_CHARS_STANDARD);
  const size_t first_len = first ? strlen(first) : 0;
  for (int k = 0; k < 20; k++)
    free(ascii_convert_with_capabilities(j->img, j->w, j->h, &j->caps, false, false, PALETTE_CHARS_STANDARD));
  /* DT_INTO=1: the additive entry point that leaves the frame in the caller's buffer (one buffer per render thread) */
  const char *into_env = getenv("DT_INTO");
  const int into = into_env && into_env[0] == '1';
  const size_t own_cap = first_len + 64;
  char *own = into ? (char *)malloc(own_cap) : NULL;
  pthread_barrier_wait(&gate);
  for (int k = 0; k < j->calls; k++) {
    char *s;
    if (into) {
      size_t n = 0;
      s = ascii_convert_with_capabilities_into(j->img, j->w, j->h, &j->caps, false, false, PALETTE_CHARS_STANDARD, own, own_cap, &n) == ASCIICHAT_OK
              ? own : NULL;
    } else
      s = ascii_convert_with_capabilities(j->img, j->w, j->h, &j->caps, false, false, PALETTE_CHARS_STANDARD);
    if (!s) {
      extern const char *asciichat_hip_last_error(void);
      fprintf(stderr, "render failed: %s\n", asciichat_hip_last_error());
      exit(1);
    }
    const size_t len = strlen(s);
    if (len != first_len || memcmp(s, first, len) != 0) {
      fprintf(stderr, "call %d returned a different frame (%zu bytes, expected %zu)\n", k, len, first_len);
      exit(1);
    }
    j->bytes += len;
    if (!into)
      free(s);
  }
  free(own);
  free(first);
  pthread_barrier_wait(&gate);
  return NULL;
}

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int main(int argc, char **argv) {
  setvbuf(stdout, NULL, _IOLBF, 0); /* a run that is cut off keeps the lines it produced */
  const int max_threads = argc > 1 ? atoi(argv[1]) : 32;
  const int sw = argc > 7 ? atoi(argv[2]) : 1920, sh = argc > 7 ? atoi(argv[3]) : 1080;
  const int tw = argc > 7 ? atoi(argv[4]) : 80, th = argc > 7 ? atoi(argv[5]) : 24;
  const int cl = argc > 7 ? atoi(argv[6]) : TERM_COLOR_TRUECOLOR, rm = argc > 7 ? atoi(argv[7]) : RENDER_MODE_FOREGROUND;
  const int min_threads = getenv("DT_MIN_T") ? atoi(getenv("DT_MIN_T")) : 1; /* start the sweep here */
  const int only = getenv("DT_POOLED") ? atoi(getenv("DT_POOLED")) : -1;       /* 0 pageable, 1 pooled, unset both */
  for (int pooled = 0; pooled <= 1; pooled++) {
    if (only >= 0 && pooled != only)
      continue;
    for (int T = min_threads; T <= max_threads; T *= 2) {
      job_t *jobs = (job_t *)calloc((size_t)T, sizeof(job_t));
      pthread_t *tid = (pthread_t *)calloc((size_t)T, sizeof(pthread_t));
      pthread_barrier_init(&gate, NULL, (unsigned)T + 1);
      for (int t = 0; t < T; t++) {
        jobs[t].img = pooled ? image_new_from_pool((size_t)sw, (size_t)sh) : image_new((size_t)sw, (size_t)sh);
        unsigned x = 12345u + (unsigned)t;
        unsigned char *px = (unsigned char *)jobs[t].img->pixels;
        for (size_t i = 0; i < (size_t)sw * (size_t)sh * 3; i++) {
          x ^= x << 13, x ^= x >> 17, x ^= x << 5;
          px[i] = (unsigned char)x;
        }
        jobs[t].calls = 2000;
        jobs[t].w = tw;
        jobs[t].h = th;
        memset(&jobs[t].caps, 0, sizeof jobs[t].caps);
        jobs[t].caps.color_level = (terminal_color_mode_t)cl;
        jobs[t].caps.render_mode = (render_mode_t)rm;
        jobs[t].caps.utf8_support = true;
        pthread_create(&tid[t], NULL, worker, &jobs[t]);
      }
      pthread_barrier_wait(&gate);
      struct rusage ru0, ru1;
      getrusage(RUSAGE_SELF, &ru0);
      const double t0 = now();
      pthread_barrier_wait(&gate);
      const double dt = now() - t0;
      getrusage(RUSAGE_SELF, &ru1);
      size_t bytes = 0;
      for (int t = 0; t < T; t++) {
        pthread_join(tid[t], NULL);
        bytes += jobs[t].bytes;
        if (pooled)
          image_destroy_to_pool(jobs[t].img);
        else
          image_destroy(jobs[t].img);
      }
      printf("%dx%d -> %dx%d colour %d mode %d, %s images, %2d render threads: %9.0f calls/s (%6.1f us per call per thread, "
             "%.1f MB/s of frames)\n",
             sw, sh, tw, th, cl, rm, pooled ? "pool (pinned)" : "pageable     ", T, T * 2000.0 / dt, dt / 2000.0 * 1e6,
             (double)bytes / dt / 1e6);
      if (getenv("DT_RUSAGE")) { /* CPU time the whole process spent per call (every thread, the library's own included) */
        const double us = (double)(ru1.ru_utime.tv_sec - ru0.ru_utime.tv_sec) * 1e6 + (double)(ru1.ru_utime.tv_usec - ru0.ru_utime.tv_usec);
        const double ss = (double)(ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) * 1e6 + (double)(ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec);
        printf("   CPU per call: user %.1f us, system %.1f us; %.1f CPUs busy; context switches per call: %.2f voluntary, %.2f forced\n",
               us / (T * 2000.0), ss / (T * 2000.0), (us + ss) / (dt * 1e6), (double)(ru1.ru_nvcsw - ru0.ru_nvcsw) / (T * 2000.0),
               (double)(ru1.ru_nivcsw - ru0.ru_nivcsw) / (T * 2000.0));
      }
      pthread_barrier_destroy(&gate);
      free(jobs);
      free(tid);
    }
  }
  return 0;
}
