/*
 * buffer_pool_test_port.c -- a plain-C caller of the buffer_pool_* exports of libasciichat_hip.so.  Includes only the
 * public drop-in header and restates, as one sequential program, the invariants the reference's own unit tests hold
 * for the pool (tests/unit/util/buffer_pool_test.c: creation 13-24, repeated creation 26-33, NULL destroy 35-42,
 * global pool 44-69, allocation round trip 76-109, zero size 111-124, NULL pool 126-134, reuse 140-165, mixed sizes
 * 167-203, statistics 205-225, global helpers 247-294, many allocations 296-324, very large 326-344, stress 350-385,
 * free(NULL) 387-395, shrink 397-412), then the properties this library adds: frames above BUFFER_POOL_MAX_SINGLE_SIZE
 * come back as pinned blocks when a GPU is present (plain host blocks otherwise), interior pointers resolve through
 * buffer_pool_is_pinned, and the pool stays consistent under concurrent alloc / free from many threads.
 * Exit code 0 and a last line "ok: <n> checks" on success.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "asciichat_render.h"

static int g_checks;
#define CHECK(cond, ...)                                                                                               \
  do {                                                                                                                 \
    g_checks++;                                                                                                        \
    if (!(cond)) {                                                                                                     \
      fprintf(stderr, "FAILED %s:%d: %s -- ", __FILE__, __LINE__, #cond);                                              \
      fprintf(stderr, __VA_ARGS__);                                                                                    \
      fprintf(stderr, "\n");                                                                                           \
      exit(1);                                                                                                         \
    }                                                                                                                  \
  } while (0)

static buffer_pool_t *fresh(void) {
  buffer_pool_t *p = buffer_pool_create(BUFFER_POOL_MAX_BYTES, BUFFER_POOL_SHRINK_DELAY_NS);
  CHECK(p != NULL, "pool creation");
  return p;
}

static void stats(buffer_pool_t *p, size_t *cur, size_t *used, size_t *fre) {
  *cur = *used = *fre = (size_t)-1;
  buffer_pool_get_stats(p, cur, used, fre);
}

static void lifecycle(void) {
  buffer_pool_t *p = fresh();
  size_t cur, used, fre;
  stats(p, &cur, &used, &fre);
  CHECK(used == 0, "a new pool has nothing in use (%zu)", used);
  buffer_pool_destroy(p);
  for (int i = 0; i < 5; i++)
    buffer_pool_destroy(fresh());
  buffer_pool_destroy(NULL); /* must be harmless */
  for (int i = 0; i < 3; i++) {
    buffer_pool_init_global();
    CHECK(buffer_pool_get_global() != NULL, "global pool, cycle %d", i);
    buffer_pool_cleanup_global();
  }
  buffer_pool_cleanup_global(); /* twice in a row */
}

static void round_trips(void) {
  static const size_t sizes[] = {512, 1024, 32768, 65536, 131072, 262144, 655360, 1048576};
  for (size_t k = 0; k < sizeof sizes / sizeof sizes[0]; k++) {
    const size_t n = sizes[k];
    buffer_pool_t *p = fresh();
    unsigned char *b = (unsigned char *)buffer_pool_alloc(p, n);
    CHECK(b != NULL, "allocation of %zu bytes", n);
    const unsigned char pat = (unsigned char)((n ^ 0xAB) & 0xFF);
    memset(b, pat, n);
    CHECK(b[0] == pat && b[n / 2] == pat && b[n - 1] == pat, "%zu bytes readable end to end", n);
    CHECK(((size_t)b & 63u) == 0, "payloads are 64-byte aligned (%p)", (void *)b);
    buffer_pool_free(p, b, n);
    buffer_pool_destroy(p);
  }
  buffer_pool_t *p = fresh();
  void *z = buffer_pool_alloc(p, 0); /* NULL or a block: both fine, neither may crash */
  if (z)
    buffer_pool_free(p, z, 0);
  void *g = buffer_pool_alloc(NULL, 1024); /* NULL pool = the global pool (created on demand) or a plain block */
  if (g)
    buffer_pool_free(NULL, g, 1024);
  buffer_pool_free(p, NULL, 1024); /* free(NULL) is a no-op */
  buffer_pool_destroy(p);
}

static void reuse_and_mixed(void) {
  static const size_t sizes[] = {1024, 65536, 131072};
  for (size_t k = 0; k < 3; k++) {
    buffer_pool_t *p = fresh();
    void *b[5];
    for (int cycle = 0; cycle < 3; cycle++) {
      for (int i = 0; i < 5; i++) {
        b[i] = buffer_pool_alloc(p, sizes[k]);
        CHECK(b[i] != NULL, "allocation %d of cycle %d (%zu bytes)", i, cycle, sizes[k]);
        memset(b[i], cycle * 16 + i, sizes[k]);
      }
      for (int i = 0; i < 5; i++)
        for (int j = i + 1; j < 5; j++)
          CHECK(b[i] != b[j], "live blocks are distinct");
      for (int i = 0; i < 5; i++) {
        CHECK(((unsigned char *)b[i])[sizes[k] - 1] == (unsigned char)(cycle * 16 + i), "block %d kept its bytes", i);
        buffer_pool_free(p, b[i], sizes[k]);
      }
    }
    size_t cur, used, fre;
    stats(p, &cur, &used, &fre);
    CHECK(used == 0, "everything returned: used = %zu", used);
    CHECK(fre > 0 && cur >= fre, "returned blocks are kept for reuse (current %zu, free %zu)", cur, fre);
    buffer_pool_destroy(p);
  }
  buffer_pool_t *p = fresh();
  unsigned char *s = buffer_pool_alloc(p, 512), *m = buffer_pool_alloc(p, 32768), *l = buffer_pool_alloc(p, 131072),
                *x = buffer_pool_alloc(p, 655360);
  CHECK(s && m && l && x, "four size classes at once");
  memset(s, 0xAA, 512);
  memset(m, 0xBB, 32768);
  memset(l, 0xCC, 131072);
  memset(x, 0xDD, 655360);
  CHECK(s[0] == 0xAA && m[0] == 0xBB && l[0] == 0xCC && x[0] == 0xDD, "blocks do not overlap");
  CHECK(s[511] == 0xAA && m[32767] == 0xBB && l[131071] == 0xCC && x[655359] == 0xDD, "blocks do not overlap (ends)");
  buffer_pool_free(p, s, 512);
  buffer_pool_free(p, m, 32768);
  buffer_pool_free(p, l, 131072);
  buffer_pool_free(p, x, 655360);
  buffer_pool_destroy(p);
}

static void statistics_and_shrink(void) {
  buffer_pool_t *p = fresh();
  size_t cur, used0, fre, used1;
  stats(p, &cur, &used0, &fre);
  void *a = buffer_pool_alloc(p, 1024), *b = buffer_pool_alloc(p, 32768);
  CHECK(a && b, "two allocations");
  stats(p, &cur, &used1, &fre);
  CHECK(used1 > used0 && used1 >= 1024 + 32768, "used bytes grow with allocations (%zu -> %zu)", used0, used1);
  buffer_pool_free(p, a, 1024);
  buffer_pool_free(p, b, 32768);
  stats(p, &cur, &used1, &fre);
  CHECK(used1 == used0, "and shrink back on free (%zu)", used1);
  buffer_pool_shrink(p); /* nothing is old enough to be released: must not crash, must not lose live blocks */
  void *c = buffer_pool_alloc(p, 2048);
  CHECK(c != NULL, "allocation after shrink");
  buffer_pool_free(p, c, 2048);
  buffer_pool_destroy(p);
  /* a pool with no shrink delay gives its idle blocks back */
  p = buffer_pool_create(BUFFER_POOL_MAX_BYTES, 0);
  CHECK(p != NULL, "pool with zero shrink delay");
  a = buffer_pool_alloc(p, 4096);
  buffer_pool_free(p, a, 4096);
  buffer_pool_shrink(p);
  stats(p, &cur, &used1, &fre);
  CHECK(used1 == 0, "nothing in use after shrink");
  buffer_pool_destroy(p);
}

static void global_helpers(void) {
  buffer_pool_init_global();
  unsigned char *b = (unsigned char *)buffer_pool_alloc(NULL, 1024);
  CHECK(b != NULL, "allocation from the global pool");
  memset(b, 0x99, 1024);
  CHECK(b[0] == 0x99 && b[1023] == 0x99, "writable");
  buffer_pool_free(NULL, b, 1024);
  unsigned char *v[20];
  for (int i = 0; i < 20; i++) {
    v[i] = (unsigned char *)buffer_pool_alloc(NULL, 2048);
    CHECK(v[i] != NULL, "global allocation %d", i);
    memset(v[i], i + 0x10, 2048);
  }
  for (int i = 0; i < 20; i++) {
    CHECK(v[i][0] == (unsigned char)(i + 0x10) && v[i][2047] == (unsigned char)(i + 0x10), "global block %d intact", i);
    buffer_pool_free(NULL, v[i], 2048);
  }
  buffer_pool_cleanup_global();
}

static void many_and_large(int have_gpu) {
  buffer_pool_t *p = fresh();
  void *v[100];
  for (int i = 0; i < 100; i++) {
    v[i] = buffer_pool_alloc(p, 1024);
    CHECK(v[i] != NULL, "allocation %d of 100", i);
    memset(v[i], i, 1024);
  }
  for (int i = 0; i < 100; i++) {
    CHECK(((unsigned char *)v[i])[1023] == (unsigned char)i, "block %d intact", i);
    buffer_pool_free(p, v[i], 1024);
  }
  /* above the largest pooled size: the reference falls back to malloc; here it is the pinned frame class */
  const size_t huge = (size_t)BUFFER_POOL_MAX_SINGLE_SIZE + 1024;
  unsigned char *h = (unsigned char *)buffer_pool_alloc(p, huge);
  CHECK(h != NULL, "allocation of %zu bytes", huge);
  memset(h, 0x77, 4096);
  h[huge - 1] = 0x78;
  CHECK(h[0] == 0x77 && h[huge - 1] == 0x78, "huge block writable");
  CHECK(buffer_pool_is_pinned(h) == (have_gpu != 0), "pinned exactly when a GPU is present");
  CHECK(buffer_pool_is_pinned(h + huge / 2) == (have_gpu != 0), "interior pointers resolve as well");
  CHECK(!buffer_pool_is_pinned(v), "a stack address is not in the pool");
  if (have_gpu)
    CHECK(buffer_pool_pinned_blocks(p) >= 1, "the pinned block is accounted for");
  buffer_pool_free(p, h, huge);
  unsigned char *h2 = (unsigned char *)buffer_pool_alloc(p, huge);
  CHECK(h2 != NULL, "second huge allocation");
  if (have_gpu)
    CHECK(h2 == h, "an idle pinned block of the same size is recycled");
  buffer_pool_free(NULL, h2, huge); /* header magic: the pool argument is optional */
  buffer_pool_destroy(p);
  CHECK(!buffer_pool_is_pinned(h), "destroying the pool unregisters its pinned blocks");
}

static void stress(void) {
  static const size_t sizes[] = {256, 1024, 4096, 16384};
  for (size_t k = 0; k < 4; k++) {
    buffer_pool_t *p = fresh();
    void *b[10];
    for (int cycle = 0; cycle < 10; cycle++) {
      for (int i = 0; i < 10; i++) {
        b[i] = buffer_pool_alloc(p, sizes[k]);
        CHECK(b[i] != NULL, "stress allocation");
        memset(b[i], cycle, sizes[k]);
      }
      for (int i = 0; i < 10; i++)
        buffer_pool_free(p, b[i], sizes[k]);
    }
    buffer_pool_destroy(p);
  }
}

/* ---- concurrency: T threads, each cycling blocks of its own pattern through ONE pool ------------ */
typedef struct {
  buffer_pool_t *pool;
  int id, big, failures;
} worker_t;

static void *worker(void *arg) {
  worker_t *w = (worker_t *)arg;
  unsigned seed = 12345u + 977u * (unsigned)w->id;
  for (int it = 0; it < 400; it++) {
    unsigned char *b[6];
    size_t n[6];
    for (int i = 0; i < 6; i++) {
      seed = seed * 1664525u + 1013904223u;
      n[i] = 64u + (seed >> 8) % (w->big && i == 0 ? (size_t)BUFFER_POOL_MAX_SINGLE_SIZE + 65536u : 70000u);
      b[i] = (unsigned char *)buffer_pool_alloc(w->pool, n[i]);
      if (!b[i]) {
        w->failures++;
        continue;
      }
      b[i][0] = (unsigned char)w->id;
      b[i][n[i] - 1] = (unsigned char)(w->id ^ it);
    }
    for (int i = 0; i < 6; i++) {
      if (!b[i])
        continue;
      if (b[i][0] != (unsigned char)w->id || b[i][n[i] - 1] != (unsigned char)(w->id ^ it))
        w->failures += 1000; /* somebody else wrote into a live block */
      if (n[i] > BUFFER_POOL_MAX_SINGLE_SIZE)
        (void)buffer_pool_is_pinned(b[i] + n[i] / 2); /* registry lookups race with other threads' registrations */
      buffer_pool_free((i & 1) ? NULL : w->pool, b[i], n[i]);
    }
  }
  return NULL;
}

static void threads(int have_gpu) {
  enum { T = 8 };
  buffer_pool_t *p = fresh();
  pthread_t th[T];
  worker_t w[T];
  for (int i = 0; i < T; i++) {
    w[i].pool = p;
    w[i].id = i + 1;
    w[i].big = have_gpu && (i % 4 == 0); /* two threads also cycle pinned frame blocks */
    w[i].failures = 0;
    CHECK(pthread_create(&th[i], NULL, worker, &w[i]) == 0, "thread %d", i);
  }
  int failures = 0;
  for (int i = 0; i < T; i++) {
    pthread_join(th[i], NULL);
    failures += w[i].failures;
  }
  CHECK(failures == 0, "%d failures under %d threads", failures, T);
  size_t cur, used, fre;
  stats(p, &cur, &used, &fre);
  CHECK(used == 0, "all blocks returned: used = %zu", used);
  buffer_pool_destroy(p);
}

int main(int argc, char **argv) {
  const int have_gpu = argc > 1 && strcmp(argv[1], "gpu") == 0;
  lifecycle();
  round_trips();
  reuse_and_mixed();
  statistics_and_shrink();
  global_helpers();
  many_and_large(have_gpu);
  stress();
  threads(have_gpu);
  printf("ok: %d checks (%s)\n", g_checks, have_gpu ? "with the pinned frame class" : "host blocks only");
  return 0;
}
