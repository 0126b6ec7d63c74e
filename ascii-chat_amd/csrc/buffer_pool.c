/*
 * buffer_pool.c -- buffer_pool_* of the reference (lib/buffer_pool.c:122-277, include/ascii-chat/buffer_pool.h)
 * re-designed for a GPU renderer:
 *
 *  - small objects (64 B .. 4 MiB) keep the reference's scheme: one header+payload malloc block, recycled
 *    through a LOCK-FREE LIFO (a tagged-pointer Treiber stack, like the reference's lock-free pop/push,
 *    lib/buffer_pool.c:122-200), freed for real by buffer_pool_shrink() after shrink_delay_ns of idleness;
 *  - objects above 4 MiB -- every 1080p (6.2 MB) / 4K (24.9 MB) frame, which the reference hands to the
 *    malloc fallback -- come from a new PINNED size class: hipHostMalloc(mapped) blocks kept in
 *    per-size free lists, so the GPU gathers its ~2 K samples per frame straight out of the producer's
 *    buffer over PCIe and nothing is staged or copied;
 *  - the header-magic contract is kept: buffer_pool_free(NULL, p, size) works on any block from here.
 *
 * Host-side memory management only; no pixel arithmetic happens in this file.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "asciichat_render.h"
#include "internal.h"

#define MAGIC_POOLED 0xBF00B001u   /* MAGIC_BUFFER_POOL_VALID,    include/ascii-chat/util/magic.h:23 */
#define MAGIC_FALLBACK 0xBF00FA11u /* MAGIC_BUFFER_POOL_FALLBACK, magic.h:26 */
#define MAGIC_PINNED 0xBF00D1A1u   /* new: pinned + device-mapped frame block */

typedef struct pool_node {
  uint32_t magic;
  uint32_t _pad;
  size_t size; /* payload capacity */
  struct pool_node *next;
  uint64_t returned_at_ns;
  struct buffer_pool *pool;
  void *device_alias; /* pinned blocks: device address of the payload */
  uint64_t _reserved[2];
} pool_node_t; /* 64 bytes: payloads stay 64-byte aligned */
_Static_assert(sizeof(pool_node_t) == 64, "the header keeps payloads 64-byte aligned");

struct buffer_pool {
  pthread_mutex_t mu;       /* the pinned class (best fit over a short list) and nothing else */
  uint64_t free_small;      /* lock-free LIFO head: {node pointer : 48, ABA tag : 16} */
  int small_pops;           /* threads inside a pop: shrink waits for 0 before it frees detached nodes */
  int shrinking;            /* a shrink is in progress (try-lock) */
  pool_node_t *free_pinned;
  size_t max_bytes;
  uint64_t shrink_delay_ns;
  size_t current_bytes, used_bytes, peak_bytes;
  size_t pinned_bytes, pinned_live;
  uint64_t hits, allocs, returns, shrink_freed, fallbacks;
};

/* ---- the small class: a Treiber stack with a tagged head.  Nodes are only ever FREED by shrink / destroy, and shrink
 * detaches the whole list and waits until no pop is in flight before it frees anything, so a pop may always read
 * n->next of the node it saw at the head; the tag makes a stale compare-and-swap fail. */
#define HEAD_PTR(v) ((pool_node_t *)(uintptr_t)((v) & 0x0000FFFFFFFFFFFFull))
#define HEAD_PACK(n, old) (((uint64_t)(uintptr_t)(n) & 0x0000FFFFFFFFFFFFull) | ((((old) >> 48) + 1ull) << 48))

static void small_push(buffer_pool_t *pool, pool_node_t *n) {
  uint64_t old = __atomic_load_n(&pool->free_small, __ATOMIC_RELAXED);
  do {
    __atomic_store_n(&n->next, HEAD_PTR(old), __ATOMIC_RELAXED);
  } while (!__atomic_compare_exchange_n(&pool->free_small, &old, HEAD_PACK(n, old), 1, __ATOMIC_RELEASE, __ATOMIC_RELAXED));
}

/* the head node if it is large enough (LIFO head only, like the reference), else NULL */
static inline void cpu_relax(void) {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  __asm__ __volatile__("yield");
#endif
}

static pool_node_t *small_pop(buffer_pool_t *pool, size_t size) {
  /* Dekker-style handshake with buffer_pool_shrink (announce, THEN look at the head; shrink detaches the head, THEN
   * looks at the announcements): both sides need their store ordered before their load, i.e. sequential consistency --
   * acquire / release alone only worked because x86's locked instructions are full barriers */
  __atomic_add_fetch(&pool->small_pops, 1, __ATOMIC_SEQ_CST);
  pool_node_t *n;
  uint64_t old = __atomic_load_n(&pool->free_small, __ATOMIC_SEQ_CST);
  for (;;) {
    n = HEAD_PTR(old);
    /* relaxed atomics: another thread may win this node and clear its link while we look (the tag then fails our swap) */
    if (!n || __atomic_load_n(&n->size, __ATOMIC_RELAXED) < size) {
      n = NULL;
      break;
    }
    pool_node_t *next = __atomic_load_n(&n->next, __ATOMIC_RELAXED);
    if (__atomic_compare_exchange_n(&pool->free_small, &old, HEAD_PACK(next, old), 1, __ATOMIC_ACQUIRE, __ATOMIC_ACQUIRE))
      break;
  }
  __atomic_sub_fetch(&pool->small_pops, 1, __ATOMIC_RELEASE);
  return n;
}

static void note_peak(buffer_pool_t *pool, size_t used) {
  size_t peak = __atomic_load_n(&pool->peak_bytes, __ATOMIC_RELAXED);
  while (used > peak && !__atomic_compare_exchange_n(&pool->peak_bytes, &peak, used, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED))
    ;
}

static buffer_pool_t *g_pool;
static pthread_mutex_t g_pool_mu = PTHREAD_MUTEX_INITIALIZER;

/* Registry of live pinned blocks, so that any interior pointer can be resolved to its device alias.  Every drop-in
 * render call asks (is this image in the pinned pool?), from every render thread at once, while blocks come and go
 * rarely: a SORTED array read under a sequence lock -- readers take no lock and write no shared word (a binary search
 * between two reads of the sequence counter, retried if a writer was active), writers serialise on a mutex.
 * (Round 1 took a global mutex and scanned up to 1024 entries linearly on every call.) */
#define PIN_REG_MAX 1024
typedef struct {
  const uint8_t *lo, *hi;
  const uint8_t *dev;
} pin_entry_t;
static pin_entry_t g_pins[PIN_REG_MAX]; /* sorted by lo; blocks never overlap */
static int g_pin_count;
static unsigned g_pin_seq; /* odd while a writer is inside */
static pthread_mutex_t g_pin_mu = PTHREAD_MUTEX_INITIALIZER;

static uint64_t now_ns(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

static void *payload_of(pool_node_t *n) { return (uint8_t *)n + sizeof(pool_node_t); }
static pool_node_t *node_of(const void *p) { return (pool_node_t *)((uint8_t *)p - sizeof(pool_node_t)); }

static void pin_write_begin(void) {
  pthread_mutex_lock(&g_pin_mu);
  __atomic_store_n(&g_pin_seq, g_pin_seq + 1u, __ATOMIC_RELAXED);
  __atomic_thread_fence(__ATOMIC_RELEASE); /* the odd counter is visible before any entry changes */
}
static void pin_write_end(void) {
  __atomic_store_n(&g_pin_seq, g_pin_seq + 1u, __ATOMIC_RELEASE); /* entries are visible before the even counter */
  pthread_mutex_unlock(&g_pin_mu);
}

static void pin_register(pool_node_t *n) {
  const uint8_t *lo = (const uint8_t *)payload_of(n);
  pin_write_begin();
  if (g_pin_count < PIN_REG_MAX) {
    int at = g_pin_count;
    while (at > 0 && g_pins[at - 1].lo > lo) { /* keep the array sorted: shift the tail up */
      g_pins[at] = g_pins[at - 1];
      at--;
    }
    g_pins[at].lo = lo;
    g_pins[at].hi = lo + n->size;
    g_pins[at].dev = (const uint8_t *)n->device_alias;
    __atomic_store_n(&g_pin_count, g_pin_count + 1, __ATOMIC_RELAXED);
  }
  pin_write_end();
}

static void pin_unregister(pool_node_t *n) {
  const uint8_t *lo = (const uint8_t *)payload_of(n);
  pin_write_begin();
  for (int i = 0; i < g_pin_count; i++) {
    if (g_pins[i].lo == lo) {
      for (int k = i; k + 1 < g_pin_count; k++)
        g_pins[k] = g_pins[k + 1];
      __atomic_store_n(&g_pin_count, g_pin_count - 1, __ATOMIC_RELAXED);
      break;
    }
  }
  pin_write_end();
}

const void *achip_pool_device_ptr(const void *host_ptr) {
  const uint8_t *p = (const uint8_t *)host_ptr;
  for (;;) {
    const unsigned s0 = __atomic_load_n(&g_pin_seq, __ATOMIC_ACQUIRE);
    if (s0 & 1u) { /* a writer is inside: it holds the mutex only for a few dozen stores */
      cpu_relax();
      continue;
    }
    /* the entry with the greatest lo <= p; indices stay inside the static array whatever a concurrent writer does,
     * and a torn read is discarded by the sequence check below */
    int lo_i = 0, hi_i = __atomic_load_n(&g_pin_count, __ATOMIC_RELAXED);
    if (hi_i > PIN_REG_MAX)
      hi_i = PIN_REG_MAX;
    while (lo_i < hi_i) {
      const int mid = (lo_i + hi_i) >> 1;
      if (__atomic_load_n(&g_pins[mid].lo, __ATOMIC_RELAXED) <= p)
        lo_i = mid + 1;
      else
        hi_i = mid;
    }
    const void *out = NULL;
    if (lo_i > 0) {
      const uint8_t *elo = __atomic_load_n(&g_pins[lo_i - 1].lo, __ATOMIC_RELAXED);
      const uint8_t *ehi = __atomic_load_n(&g_pins[lo_i - 1].hi, __ATOMIC_RELAXED);
      const uint8_t *edev = __atomic_load_n(&g_pins[lo_i - 1].dev, __ATOMIC_RELAXED);
      if (p >= elo && p < ehi)
        out = edev + (p - elo);
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (__atomic_load_n(&g_pin_seq, __ATOMIC_RELAXED) == s0)
      return out;
  }
}

bool buffer_pool_is_pinned(const void *data) { return achip_pool_device_ptr(data) != NULL; }

buffer_pool_t *buffer_pool_create(size_t max_bytes, uint64_t shrink_delay_ns) {
  buffer_pool_t *p = (buffer_pool_t *)calloc(1, sizeof(*p));
  if (!p)
    return NULL;
  pthread_mutex_init(&p->mu, NULL);
  p->max_bytes = max_bytes ? max_bytes : BUFFER_POOL_MAX_BYTES;
  p->shrink_delay_ns = shrink_delay_ns ? shrink_delay_ns : BUFFER_POOL_SHRINK_DELAY_NS;
  return p;
}

static void release_node(pool_node_t *n) {
  if (n->magic == MAGIC_PINNED) {
    pin_unregister(n);
    n->magic = 0;
    (void)hipHostFree(n);
  } else {
    n->magic = 0;
    free(n);
  }
}

void buffer_pool_destroy(buffer_pool_t *pool) {
  if (!pool)
    return;
  for (pool_node_t *n = HEAD_PTR(pool->free_small); n;) {
    pool_node_t *nx = n->next;
    release_node(n);
    n = nx;
  }
  for (pool_node_t *n = pool->free_pinned; n;) {
    pool_node_t *nx = n->next;
    release_node(n);
    n = nx;
  }
  pthread_mutex_destroy(&pool->mu);
  free(pool);
}

static pool_node_t *fallback_node(buffer_pool_t *pool, size_t size) {
  pool_node_t *n = (pool_node_t *)malloc(sizeof(pool_node_t) + size);
  if (!n)
    return NULL;
  memset(n, 0, sizeof(*n));
  n->magic = MAGIC_FALLBACK;
  n->size = size;
  n->pool = pool;
  return n;
}

static void *alloc_pinned(buffer_pool_t *pool, size_t size) {
  /* best fit among idle pinned blocks (frames come in very few distinct sizes) */
  pthread_mutex_lock(&pool->mu);
  pool_node_t **best = NULL;
  for (pool_node_t **pp = &pool->free_pinned; *pp; pp = &(*pp)->next)
    if ((*pp)->size >= size && (!best || (*pp)->size < (*best)->size))
      best = pp;
  if (best && (*best)->size <= size + size / 4) {
    pool_node_t *n = *best;
    *best = n->next;
    n->next = NULL;
    pool->pinned_live++;
    pthread_mutex_unlock(&pool->mu);
    note_peak(pool, __atomic_add_fetch(&pool->used_bytes, n->size, __ATOMIC_RELAXED)); /* shared with the lock-free class */
    __atomic_add_fetch(&pool->hits, 1, __ATOMIC_RELAXED);
    return payload_of(n);
  }
  const int room = pool->pinned_bytes + size + sizeof(pool_node_t) <= BUFFER_POOL_PINNED_MAX_BYTES;
  pthread_mutex_unlock(&pool->mu);

  pool_node_t *n = NULL;
  if (room) {
    void *raw = NULL;
    if (hipHostMalloc(&raw, sizeof(pool_node_t) + size, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess) {
      n = (pool_node_t *)raw;
      memset(n, 0, sizeof(*n));
      void *dev = NULL;
      if (hipHostGetDevicePointer(&dev, payload_of(n), 0) != hipSuccess)
        dev = payload_of(n); /* unified addressing: same value */
      n->magic = MAGIC_PINNED;
      n->size = size;
      n->pool = pool;
      n->device_alias = dev;
      pin_register(n);
    } else {
      (void)hipGetLastError(); /* no device / out of pinned memory: plain host block, as the reference does */
    }
  }
  pthread_mutex_lock(&pool->mu);
  if (n) {
    pool->pinned_bytes += sizeof(pool_node_t) + size;
    pool->pinned_live++;
  }
  pthread_mutex_unlock(&pool->mu);
  if (n) {
    note_peak(pool, __atomic_add_fetch(&pool->used_bytes, size, __ATOMIC_RELAXED));
    __atomic_add_fetch(&pool->allocs, 1, __ATOMIC_RELAXED);
  } else {
    __atomic_add_fetch(&pool->fallbacks, 1, __ATOMIC_RELAXED);
  }
  if (!n) {
    n = fallback_node(pool, size);
    if (!n)
      return NULL;
  }
  return payload_of(n);
}

void *buffer_pool_alloc(buffer_pool_t *pool, size_t size) {
  if (!pool)
    pool = buffer_pool_get_global();
  if (!pool || size < BUFFER_POOL_MIN_SIZE) {
    pool_node_t *n = fallback_node(pool, size);
    return n ? payload_of(n) : NULL;
  }
  if (size > BUFFER_POOL_MAX_SINGLE_SIZE)
    return alloc_pinned(pool, size);

  pool_node_t *n = small_pop(pool, size);
  if (n) {
    __atomic_store_n(&n->next, (pool_node_t *)NULL, __ATOMIC_RELAXED);
    note_peak(pool, __atomic_add_fetch(&pool->used_bytes, n->size, __ATOMIC_RELAXED));
    __atomic_add_fetch(&pool->hits, 1, __ATOMIC_RELAXED);
    return payload_of(n);
  }
  const size_t total = sizeof(pool_node_t) + size;
  const int room = __atomic_add_fetch(&pool->current_bytes, total, __ATOMIC_RELAXED) <= pool->max_bytes;
  if (!room)
    __atomic_sub_fetch(&pool->current_bytes, total, __ATOMIC_RELAXED);
  if (room) {
    void *raw = NULL;
    if (posix_memalign(&raw, 64, total) == 0) {
      n = (pool_node_t *)raw;
      memset(n, 0, sizeof(*n));
      n->magic = MAGIC_POOLED;
      n->size = size;
      n->pool = pool;
      note_peak(pool, __atomic_add_fetch(&pool->used_bytes, size, __ATOMIC_RELAXED));
      __atomic_add_fetch(&pool->allocs, 1, __ATOMIC_RELAXED);
      return payload_of(n);
    }
    __atomic_sub_fetch(&pool->current_bytes, total, __ATOMIC_RELAXED);
  }
  __atomic_add_fetch(&pool->fallbacks, 1, __ATOMIC_RELAXED);
  n = fallback_node(pool, size);
  return n ? payload_of(n) : NULL;
}

void buffer_pool_free(buffer_pool_t *pool, const void *data, size_t size) {
  (void)size;
  if (!data)
    return;
  pool_node_t *n = node_of(data);
  if (n->magic == MAGIC_FALLBACK) {
    n->magic = 0;
    free(n);
    return;
  }
  if (n->magic != MAGIC_POOLED && n->magic != MAGIC_PINNED) {
    free((void *)data); /* unknown allocation: the reference frees the payload pointer itself */
    return;
  }
  if (!pool)
    pool = n->pool;
  if (!pool)
    return;
  __atomic_sub_fetch(&pool->used_bytes, n->size, __ATOMIC_RELAXED);
  const int do_shrink = __atomic_add_fetch(&pool->returns, 1, __ATOMIC_RELAXED) % 100 == 0;
  n->returned_at_ns = now_ns();
  if (n->magic == MAGIC_PINNED) {
    pthread_mutex_lock(&pool->mu);
    n->next = pool->free_pinned;
    pool->free_pinned = n;
    pool->pinned_live--;
    pthread_mutex_unlock(&pool->mu);
  } else {
    small_push(pool, n);
  }
  if (do_shrink)
    buffer_pool_shrink(pool);
}

void buffer_pool_shrink(buffer_pool_t *pool) {
  if (!pool || pool->shrink_delay_ns == 0)
    return;
  const uint64_t now = now_ns();
  const uint64_t cutoff = now > pool->shrink_delay_ns ? now - pool->shrink_delay_ns : 0;
  pool_node_t *doomed = NULL;
  /* one shrinker at a time; a caller that finds one at work skips its turn (it runs on every 100th free) */
  int expected = 0;
  if (!__atomic_compare_exchange_n(&pool->shrinking, &expected, 1, 0, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED))
    return;
  /* small class: take the whole list, wait -- a bounded while -- for the pops that may still look at its nodes, keep the
   * young ones.  While the list is detached every alloc misses it, so the wait must not depend on the callers going
   * quiet: pops that started AFTER the detach cannot see a detached node, but they are counted too, and under a steady
   * stream of them the count may never read zero.  After the bound nothing is freed this time. */
  uint64_t old = __atomic_load_n(&pool->free_small, __ATOMIC_ACQUIRE);
  while (!__atomic_compare_exchange_n(&pool->free_small, &old, HEAD_PACK(NULL, old), 1, __ATOMIC_SEQ_CST, __ATOMIC_ACQUIRE))
    ;
  bool quiet = false;
  for (int spin = 0; spin < 20000 && !quiet; spin++) {
    quiet = __atomic_load_n(&pool->small_pops, __ATOMIC_SEQ_CST) == 0;
    if (!quiet)
      cpu_relax();
  }
  if (!quiet && HEAD_PTR(old)) {
    /* poppers are still about: they may be looking at any detached node, so no node is freed and no link inside the
     * chain is written -- the chain goes back as it is (only its tail is linked to the current head, atomically) */
    pool_node_t *first = HEAD_PTR(old), *tail = first;
    while (tail->next)
      tail = tail->next;
    uint64_t cur = __atomic_load_n(&pool->free_small, __ATOMIC_RELAXED);
    do {
      __atomic_store_n(&tail->next, HEAD_PTR(cur), __ATOMIC_RELAXED);
    } while (!__atomic_compare_exchange_n(&pool->free_small, &cur, HEAD_PACK(first, cur), 1, __ATOMIC_RELEASE, __ATOMIC_RELAXED));
    old = HEAD_PACK(NULL, old); /* nothing left to sort below */
  }
  pool_node_t *keep = NULL;
  for (pool_node_t *n = HEAD_PTR(old); n;) {
    pool_node_t *nx = n->next;
    if (n->returned_at_ns < cutoff) {
      __atomic_sub_fetch(&pool->current_bytes, sizeof(pool_node_t) + n->size, __ATOMIC_RELAXED);
      __atomic_add_fetch(&pool->shrink_freed, 1, __ATOMIC_RELAXED);
      n->next = doomed;
      doomed = n;
    } else {
      n->next = keep; /* reversed here, reversed again by the pushes below: the LIFO order survives */
      keep = n;
    }
    n = nx;
  }
  while (keep) {
    pool_node_t *nx = keep->next;
    small_push(pool, keep);
    keep = nx;
  }
  pthread_mutex_lock(&pool->mu);
  for (pool_node_t **pp = &pool->free_pinned; *pp;) {
    pool_node_t *n = *pp;
    if (n->returned_at_ns < cutoff) {
      *pp = n->next;
      pool->pinned_bytes -= sizeof(pool_node_t) + n->size;
      __atomic_add_fetch(&pool->shrink_freed, 1, __ATOMIC_RELAXED);
      n->next = doomed;
      doomed = n;
    } else {
      pp = &n->next;
    }
  }
  pthread_mutex_unlock(&pool->mu);
  while (doomed) {
    pool_node_t *nx = doomed->next;
    release_node(doomed);
    doomed = nx;
  }
  __atomic_store_n(&pool->shrinking, 0, __ATOMIC_RELEASE);
}

void buffer_pool_get_stats(buffer_pool_t *pool, size_t *current_bytes, size_t *used_bytes, size_t *free_bytes) {
  size_t cur = 0, used = 0;
  if (pool) {
    pthread_mutex_lock(&pool->mu);
    cur = __atomic_load_n(&pool->current_bytes, __ATOMIC_RELAXED) + pool->pinned_bytes;
    pthread_mutex_unlock(&pool->mu);
    used = __atomic_load_n(&pool->used_bytes, __ATOMIC_RELAXED);
  }
  if (current_bytes)
    *current_bytes = cur;
  if (used_bytes)
    *used_bytes = used;
  if (free_bytes)
    *free_bytes = cur > used ? cur - used : 0;
}

size_t buffer_pool_pinned_blocks(buffer_pool_t *pool) {
  if (!pool)
    pool = buffer_pool_get_global();
  if (!pool)
    return 0;
  pthread_mutex_lock(&pool->mu);
  size_t n = pool->pinned_live;
  pthread_mutex_unlock(&pool->mu);
  return n;
}

void buffer_pool_init_global(void) {
  pthread_mutex_lock(&g_pool_mu);
  if (!g_pool)
    g_pool = buffer_pool_create(0, 0);
  pthread_mutex_unlock(&g_pool_mu);
}

void buffer_pool_cleanup_global(void) {
  pthread_mutex_lock(&g_pool_mu);
  buffer_pool_t *p = g_pool;
  g_pool = NULL;
  pthread_mutex_unlock(&g_pool_mu);
  buffer_pool_destroy(p);
}

buffer_pool_t *buffer_pool_get_global(void) {
  if (!g_pool)
    buffer_pool_init_global();
  return g_pool;
}
