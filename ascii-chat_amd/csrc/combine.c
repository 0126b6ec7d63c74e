/*
 * combine.c -- transparent coalescing of concurrent drop-in calls (flat combining).
 *
 * The reference's server renders once per client per tick from one thread per client
 * (src/server/render.c:340-600 -> create_mixed_ascii_frame_for_client -> ascii_convert_with_capabilities,
 * src/server/stream.c:841).  Relinked against this library, each of those calls used to be its own upload + launch +
 * synchronise: ~25-33 us for a 1080p -> 80x24 frame, 5.5 us of which is serialised HIP-runtime work, so throughput
 * stopped at ~175 k calls/s however many threads called (profiles/r01_dropin_threads.txt).  Here concurrent callers
 * share launches without any change on their side:
 *
 *   * a caller takes a slot in the OPEN generation and copies the pixels its frame samples (the sampled rows, and only
 *     the sampled columns of them when the frame is at most half as wide as its source: achip_stage_gather -- 5.6 KB
 *     instead of 6.2 MB for 1080p -> 80x24) into that generation's pinned, device-mapped staging arena, in parallel with
 *     the other callers; then it sleeps until its result is ready -- except the generation's FIRST member, its launcher;
 *   * the launcher waits until the generation is ripe (it holds a third of the calls in flight, or is full, or has been open
 *     for 30 us) and one of the launch slots is free, closes it, launches ONE kernel per (mode, palette) group of the
 *     generation -- the same kernels, geometry policy and descriptors as the batch API; small staged images are read in
 *     place over PCIe, larger ones uploaded with ONE DMA first -- and waits once;
 *   * every caller then copies its own string out of the generation's output slab (mapped host memory the kernels wrote
 *     directly) into a malloc block (the ownership contract of the reference's API), in parallel; the last one out frees
 *     the generation.
 *
 * Generations are set up on demand (4 to 16), so that joining one does not have to wait.  Requests that do not fit a
 * generation (a 4K identity render bounds its output at hundreds of MB) are not combined.
 *
 * How callers wait matters more than anything else here (profiles/r03_dropin_stats.txt is the log of finding that out):
 * they poll only while every call in flight can have a CPU of its own -- by the affinity mask AND the cgroup's CPU quota --
 * and sleep on a futex otherwise, woken as a tree (each woken thread wakes two more); the launcher sleeps on a
 * blocking-sync event in that case.  On a box that shows 256 hardware threads and grants 16 CPUs' worth of time, callers
 * that polled froze the whole process (CFS throttling) from 64 threads on.
 *
 * Measured (scripts/dropin_threads.c, 1080p -> 80x24 truecolor, one MI355X box with that 16-CPU quota;
 * profiles/r03_dropin_threads.txt), calls/s pageable / pooled images: 37 k / 39 k from one thread, 127 k / 118 k from 4
 * (each call its own launch on the thread's stream: below ASCIICHAT_HIP_COALESCE = 6 calls in flight that is faster),
 * 378 k / 444 k from 16, 505 k / 476 k from 32, 483 k / 509 k from 64, 559 k / 517 k from 128 with the process confined
 * to 16 CPUs (taskset); free to roam over all 256 the kernel's per-CPU quota slices run dry and throughput falls again
 * past 32-64 threads (407 k, 461 k, 169 k) -- ASCIICHAT_HIP_CONFINE=1 applies that confinement to the calling threads.
 * Round 2: 105 k / 161 k at the peak, 72 k / 64 k at 64 threads.
 */
#define _GNU_SOURCE /* CPU_COUNT, sched_getaffinity */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <errno.h>
#include <limits.h>
#include <linux/futex.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include "achip_host.h"
#include "asciichat_hip.h"
#include "hip_launch.h"
#include "internal.h"
#include "render_variants.h"

#define CB_MAX 64                      /* requests per generation                                  */
#define CB_ARENA ((size_t)16 << 20)    /* pinned staging bytes per generation (sampled pixels)        */
#define CB_SLAB ((size_t)8 << 20)     /* pinned output bytes per generation                         */
#define CB_DEVICES 16
#define CB_GENS 16     /* generations per device, at most: CB_GENS_START are set up when the layer first engages, more only
                          when a caller finds every one of them busy (a generation stays busy until its last member has
                          copied its string out, and a crowd of sleepers takes a while to wake) -- joining a generation
                          should never have to wait, so that a call sleeps once (for its frame), not twice */
#define CB_GENS_START 4
#define CB_INFLIGHT 6  /* generations that may be in flight at once (each on its own stream); a ripe generation whose
                          launcher finds them all taken waits for a launch slot and keeps filling meanwhile          */
#define CB_SHARE 3     /* a generation is launched once it holds 1/CB_SHARE of the calls in flight (at least one, at most
                          CB_MAX): round 3's first form launched an OPEN generation as soon as one member had filled its slot
                          and a launch slot was free, so generations held 3-7 members whatever the load, a generation is
                          busy from its first member until its last one has copied out, and at 128 callers 8 x 5 = 40 were
                          inside generations while 88 queued for a slot (1.4 ms of a 1.9 ms call: profiles/
                          r03_dropin_stats.txt).  A third of the callers per generation keeps three generations cycling:
                          one filling, one on the GPU, one being copied out */
#define CB_LINGER_NS 30000ull /* ... or once it has been open this long: the callers it waited for went elsewhere */
#define CB_INPLACE_MAX ((size_t)64 << 10) /* staged images up to this size are read by the kernel in place (mapped pinned
                          memory, dense after the gather) instead of through a DMA into HBM */
#define CB_RUN_DEADLINE_S 10 /* a generation whose stream has not drained by then is reported as failed (a wedged GPU
                                must not hang every caller of the library for ever) */

enum { GEN_FREE = 0, GEN_OPEN, GEN_CLOSED, GEN_DONE };

typedef struct {
  achip_frame_t desc; /* src: device-visible (pool alias, the mapped arena or its HBM twin at stage_off) */
  int mode, ascii;
  const achip_lut_t *lut;
  size_t bound;     /* worst-case bytes incl. NUL, multiple of 16 */
  size_t stage_off; /* (size_t)-1: read in place */
  size_t out_off;
  uint32_t len;
  int failed; /* the launch of this request's (mode, palette) group failed: only its members get the error */
} cb_req_t;

typedef struct {
  int state __attribute__((aligned(64))); /* polled by every waiting member: a line of its own */
  int n __attribute__((aligned(64)));
  int filled, copied, failed;
  int full; /* no more members fit */
  int n_pub; /* n, published for the members' lock-free look at how full the generation is */
  size_t arena_used, max_bound;
  cb_req_t req[CB_MAX];
  uint8_t *arena_host, *arena_dev; /* pinned staging and its HBM twin */
  uint8_t *arena_map;              /* device alias of arena_host (mapped): small staged images are read in place */
  unsigned long long t_open;       /* when the generation was opened (CB_LINGER_NS) */
  int need_dma;                    /* a member staged something for the HBM twin */
  uint8_t *slab_host, *slab_dev;   /* pinned, device-mapped output slab: the kernels write it over PCIe */
  achip_frame_t *descs_host, *descs_dev;
  uint32_t *lens_host, *lens_dev;
  hipStream_t stream; /* a generation launches on its own stream: generations in flight overlap on the GPU */
  hipEvent_t done_ev; /* blocking-sync event: how a launcher waits when there are more callers than CPUs */
  unsigned long long *part_sync;
  size_t part_sync_n;
  uint32_t epoch;
  char err[160];
} cb_gen_t;

typedef struct {
  pthread_mutex_t mu;
  cb_gen_t gen[CB_GENS];
  int n_gens, grow_failed; /* generations set up so far; a set-up failed: stay with these */
  int open; /* index of the OPEN generation, -1 = none */
  int inflight; /* generations between CLOSED and DONE */
  int turnover; /* bumped whenever a generation changes state in a way that may let a waiting caller in (futex word) */
  int launch_seq; /* bumped when an in-flight slot frees up: members of OPEN generations waiting to launch (futex word) */
  int ready; /* 0 = untried, 1 = usable, -1 = initialisation failed (callers use the direct path) */
  int cus;
} cb_t;

/* Waiting.  A generation is in flight for ~50 us and a futex sleep + wake costs about as much, so nobody sleeps at first:
 * a waiter polls ONE word (its generation's state, or the table's turnover / launch counters; atomics, no lock) for
 * CB_SPIN_NS and only then parks on a FUTEX on that very word.  Wake-ups are a TREE, never a broadcast: the thread that
 * changes the word wakes two sleepers and every thread that returns from a park wakes two more -- with 40-100 sleepers on
 * one word (128 callers) a wake-all costs its caller, the thread that just finished a generation, hundreds of
 * microseconds of serial wake-ups, and the crowd then arrives at the mutex at once (profiles/r03_dropin_stats.txt).
 * Round 2 used 100 us timed waits on one condition variable under the table's mutex (72 k calls/s at 64 threads). */
#define CB_SPIN_NS 100000ull
#define CB_POLL_PAUSE 0 /* pauses between two hipStreamQuery calls of a generation's launcher */

static cb_t g_cb[CB_DEVICES];
static pthread_once_t g_cb_once = PTHREAD_ONCE_INIT;
static int g_cb_enabled = 1;
static int g_cb_min_callers = 6; /* calls in flight from which coalescing pays (profiles/r03_dropin_threads.txt: 8
                                    threads 216 k / 246 k calls/s combined against 163 k / 141 k direct; 4 threads launch
                                    on their own streams at 120 k) */
/* The CPUs this process may really keep busy: the affinity mask, capped by the cgroup's CPU quota (cpu.max of cgroup v2,
 * cfs_quota_us / cfs_period_us of v1).  Waiters poll only while every call in flight can have a CPU of its own; beyond
 * that they sleep at once.  Polling past the quota is worse than useless: the MI355X boxes these figures come from report
 * 256 hardware threads and grant 16 CPUs' worth of time (cpu.max 1600000 100000; scripts/cpu_scaling.c), so 128 polling
 * callers used up a 100 ms period's quota in 12 ms and the kernel then froze ALL of them, the launchers included, for the
 * other 88 ms -- the "collapse" of profiles/r03_dropin_threads.txt at 64 and 128 threads. */
static int g_cb_callers; /* drop-in render calls currently inside achip_combine_render / the direct path */
static int g_cpu_budget = 1;
static int g_cb_confine;
static void cpu_budget_init(void) {
  cpu_set_t set;
  int n = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : (int)sysconf(_SC_NPROCESSORS_ONLN);
  if (n < 1)
    n = 1;
  long long quota = -1, period = 0;
  FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
  if (f) {
    char q[32] = "";
    if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm')
      quota = atoll(q);
    fclose(f);
  } else {
    FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
    if (fq && fp && (fscanf(fq, "%lld", &quota) != 1 || fscanf(fp, "%lld", &period) != 1))
      quota = -1;
    if (fq)
      fclose(fq);
    if (fp)
      fclose(fp);
  }
  int quota_limited = 0;
  if (quota > 0 && period > 0) {
    const long long cpus = (quota + period - 1) / period;
    if (cpus < n) {
      n = (int)(cpus < 1 ? 1 : cpus);
      quota_limited = 1;
    }
  }
  const char *k = getenv("ASCIICHAT_HIP_CPU_BUDGET");
  if (k && atoi(k) >= 1)
    n = atoi(k);
  g_cpu_budget = n;
  /* default: confine exactly when a cgroup quota is worth fewer CPUs than the affinity mask shows -- the one situation in
   * which roaming costs throughput (below); ASCIICHAT_HIP_CONFINE=0 / =1 overrides either way */
  k = getenv("ASCIICHAT_HIP_CONFINE");
  g_cb_confine = k && k[0] ? k[0] != '0' : quota_limited;
}
/* A calling thread is confined (once, on its first call) to the first g_cpu_budget CPUs of its affinity mask when the
 * cgroup quota is smaller than the mask.  CFS hands the quota out in per-CPU slices; a hundred threads that sleep and
 * wake all over a 256-CPU box strand it on CPUs that have nothing to run, and the process is throttled at a fraction of
 * its quota (128 callers: 169 k calls/s roaming, 559 k confined to 16 CPUs).  On by default in exactly that situation
 * (round 4: the figure should not depend on the operator having read INTEGRATION.md); ASCIICHAT_HIP_CONFINE=0 leaves
 * the threads alone, =1 forces it; `taskset` on the server does the same from outside. */
static void cb_confine_thread(void) {
  static __thread int done;
  if (done)
    return;
  done = 1;
  cpu_set_t set, keep;
  if (sched_getaffinity(0, sizeof(set), &set) != 0 || CPU_COUNT(&set) <= g_cpu_budget)
    return;
  CPU_ZERO(&keep);
  int kept = 0;
  for (int c = 0; c < CPU_SETSIZE && kept < g_cpu_budget; c++)
    if (CPU_ISSET(c, &set)) {
      CPU_SET(c, &keep);
      kept++;
    }
  const int rc = sched_setaffinity(0, sizeof(keep), &keep);
  /* the host application's threads are touched without having asked: say so, once per process */
  static int said;
  if (!__atomic_exchange_n(&said, 1, __ATOMIC_RELAXED)) {
    const char *q = getenv("ASCIICHAT_HIP_QUIET");
    if (!(q && q[0] && q[0] != '0'))
      fprintf(stderr,
              "libasciichat_hip: threads that call the render entry points are confined to %d of their %d CPUs (the cgroup CPU quota; "
              "threads they start later inherit the mask)%s -- ASCIICHAT_HIP_CONFINE=0 leaves them alone, ASCIICHAT_HIP_QUIET=1 drops this line\n",
              g_cpu_budget, CPU_COUNT(&set), rc ? " [sched_setaffinity failed: not applied]" : "");
  }
}
static void cb_global_init(void);
int achip_cpu_budget(void) {
  pthread_once(&g_cb_once, cb_global_init);
  return g_cpu_budget;
}
static inline int cb_crowded(void) { return __atomic_load_n(&g_cb_callers, __ATOMIC_RELAXED) > g_cpu_budget; }

/* tuning knobs (environment, read once): ASCIICHAT_HIP_CB_{SPIN_US,INFLIGHT,SHARE,LINGER_US,POLL_PAUSE} */
static unsigned long long g_cb_spin_ns = CB_SPIN_NS, g_cb_linger_ns = CB_LINGER_NS;
static int g_cb_inflight = CB_INFLIGHT, g_cb_share = CB_SHARE, g_cb_poll_pause = CB_POLL_PAUSE, g_cb_fanout = 2, g_cb_block = 1;
static int g_cb_inplace = 1;      /* ASCIICHAT_HIP_COMBINE_INPLACE=0: always DMA staged pixels into HBM */

/* ASCIICHAT_HIP_COMBINE_STATS=1: where a combined call spends its time, printed at exit (diagnostics) */
static int g_cb_stats;
static struct {
  unsigned long long calls, gens, ns_slot, ns_fill, ns_wait, ns_out, ns_issue, ns_poll, all_calls, ns_all;
} g_st;
static inline unsigned long long now_ns(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (unsigned long long)t.tv_sec * 1000000000ull + (unsigned long long)t.tv_nsec;
}
#define ST_ADD(f, v) __atomic_add_fetch(&g_st.f, (v), __ATOMIC_RELAXED)
static void cb_stats_print(void) {
  const double c = g_st.calls ? (double)g_st.calls : 1.0, g = g_st.gens ? (double)g_st.gens : 1.0;
  fprintf(stderr,
          "[asciichat_hip combine] calls %llu generations %llu (%.1f members): per call us: slot %.1f fill %.1f wait %.1f out "
          "%.1f; per generation us: issue %.1f poll %.1f; all %llu render calls: %.1f us each\n",
          g_st.calls, g_st.gens, c / g, g_st.ns_slot / c / 1e3, g_st.ns_fill / c / 1e3, g_st.ns_wait / c / 1e3,
          g_st.ns_out / c / 1e3, g_st.ns_issue / g / 1e3, g_st.ns_poll / g / 1e3, g_st.all_calls,
          g_st.ns_all / (g_st.all_calls ? (double)g_st.all_calls : 1.0) / 1e3);
}

static void cb_global_init(void) {
  const char *st = getenv("ASCIICHAT_HIP_COMBINE_STATS");
  if (st && st[0] && st[0] != '0') {
    g_cb_stats = 1;
    atexit(cb_stats_print);
  }
  cpu_budget_init();
  const char *k;
  if ((k = getenv("ASCIICHAT_HIP_CB_SPIN_US")) && k[0])
    g_cb_spin_ns = (unsigned long long)atoll(k) * 1000ull;
  if ((k = getenv("ASCIICHAT_HIP_CB_LINGER_US")) && k[0])
    g_cb_linger_ns = (unsigned long long)atoll(k) * 1000ull;
  if ((k = getenv("ASCIICHAT_HIP_CB_INFLIGHT")) && atoi(k) >= 1 && atoi(k) <= CB_GENS - 1)
    g_cb_inflight = atoi(k);
  if ((k = getenv("ASCIICHAT_HIP_CB_SHARE")) && atoi(k) >= 1)
    g_cb_share = atoi(k);
  if ((k = getenv("ASCIICHAT_HIP_CB_POLL_PAUSE")) && k[0])
    g_cb_poll_pause = atoi(k);
  if ((k = getenv("ASCIICHAT_HIP_CB_FANOUT")) && atoi(k) >= 1)
    g_cb_fanout = atoi(k);
  if ((k = getenv("ASCIICHAT_HIP_CB_BLOCK")) && k[0])
    g_cb_block = atoi(k);
  const char *ip = getenv("ASCIICHAT_HIP_COMBINE_INPLACE");
  if (ip && ip[0] == '0')
    g_cb_inplace = 0;
  for (int d = 0; d < CB_DEVICES; d++) {
    pthread_mutex_init(&g_cb[d].mu, NULL);
    g_cb[d].open = -1;
  }
  /* ASCIICHAT_HIP_COALESCE: 0 = never, 1 = always, N >= 2 = from N concurrent callers on (default 6: below that
   * every call launching on its own thread's stream is as fast or faster, above it the HIP runtime's serialised
   * per-launch work makes throughput collapse and shared launches hold it) */
  const char *e = getenv("ASCIICHAT_HIP_COALESCE");
  if (e && e[0]) {
    const int v = atoi(e);
    if (v <= 0)
      g_cb_enabled = 0;
    else
      g_cb_min_callers = v;
  }
}

/* 0 = never coalesce, 1 = always, N >= 2 = from N concurrent callers on; returns the previous setting */
int asciichat_hip_set_coalesce_min_callers(int n) {
  pthread_once(&g_cb_once, cb_global_init);
  const int before = g_cb_enabled ? g_cb_min_callers : 0;
  g_cb_enabled = n > 0;
  if (n > 0)
    g_cb_min_callers = n;
  return before;
}

/* diagnostics: dropin.c reports the whole duration of a render call (palette tables, either path, string) */
unsigned long long achip_combine_stats_clock(void) {
  pthread_once(&g_cb_once, cb_global_init);
  return g_cb_stats ? now_ns() : 0;
}
void achip_combine_stats_call(unsigned long long t0) {
  if (g_cb_stats && t0) {
    ST_ADD(all_calls, 1);
    ST_ADD(ns_all, now_ns() - t0);
  }
}

/* dropin.c brackets every render call with these: the number of callers in flight decides between the two paths */
void achip_combine_enter(void) {
  pthread_once(&g_cb_once, cb_global_init);
  if (g_cb_confine)
    cb_confine_thread();
  __atomic_add_fetch(&g_cb_callers, 1, __ATOMIC_RELAXED);
}
void achip_combine_leave(void) { __atomic_sub_fetch(&g_cb_callers, 1, __ATOMIC_RELAXED); }
int achip_combine_callers(void) { return __atomic_load_n(&g_cb_callers, __ATOMIC_RELAXED); }
int achip_combine_crowded(void) { return cb_crowded(); }

static int pinned_mapped(void **host, void **dev, size_t bytes) {
  if (hipHostMalloc(host, bytes, hipHostMallocMapped) != hipSuccess)
    return -1;
  if (hipHostGetDevicePointer(dev, *host, 0) != hipSuccess)
    *dev = *host;
  return 0;
}

/* streams and buffers of one generation; 0 on success (what was allocated before a failure stays with the slot, unused) */
static int cb_gen_setup(cb_gen_t *G) {
  void *h = NULL, *d = NULL;
  if (hipStreamCreateWithFlags(&G->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&G->done_ev, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess)
    return -1;
  if (pinned_mapped(&h, &d, CB_ARENA) || hipMalloc((void **)&G->arena_dev, CB_ARENA) != hipSuccess)
    return -1;
  G->arena_host = (uint8_t *)h;
  G->arena_map = (uint8_t *)d;
  if (pinned_mapped(&h, &d, CB_SLAB))
    return -1;
  G->slab_host = (uint8_t *)h;
  G->slab_dev = (uint8_t *)d;
  if (pinned_mapped(&h, &d, CB_MAX * sizeof(achip_frame_t)))
    return -1;
  G->descs_host = (achip_frame_t *)h;
  G->descs_dev = (achip_frame_t *)d;
  if (pinned_mapped(&h, &d, CB_MAX * sizeof(uint32_t)))
    return -1;
  G->lens_host = (uint32_t *)h;
  G->lens_dev = (uint32_t *)d;
  G->state = GEN_FREE;
  return 0;
}

/* called with cb->mu held */
static void cb_device_init(cb_t *cb) {
  cb->ready = -1;
  for (int g = 0; g < CB_GENS_START; g++) {
    if (cb_gen_setup(&cb->gen[g]))
      return;
    cb->n_gens = g + 1;
  }
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
    n = 256;
  cb->cus = n;
  cb->ready = 1;
}

#define LOAD(x) __atomic_load_n(&(x), __ATOMIC_ACQUIRE)
#define STORE(x, v) __atomic_store_n(&(x), (v), __ATOMIC_RELEASE)

static inline void cpu_relax(void) {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  __asm__ __volatile__("yield");
#endif
}

/* park until *word != seen (or a spurious wake-up; the callers re-check).  A bounded wait: a lost wake-up costs 2 ms,
 * not a hang. */
static void futex_park(int *word, int seen) {
  const struct timespec ts = {0, 2000000};
  (void)syscall(SYS_futex, word, FUTEX_WAIT_PRIVATE, seen, &ts, NULL, 0);
}
static void futex_wake_all(int *word) { (void)syscall(SYS_futex, word, FUTEX_WAKE_PRIVATE, INT_MAX, NULL, NULL, 0); }

static void futex_wake_n(int *word, int n) { (void)syscall(SYS_futex, word, FUTEX_WAKE_PRIVATE, n, NULL, NULL, 0); }

/* wait until *word != seen: polling for CB_SPIN_NS (measured from *t0, 0 = not started), parked afterwards; passes the
 * wake-up on when it slept (mu NOT held) */
static void cb_wait_change(int *word, int seen, unsigned long long *t0) {
  int parked = 0;
  const int crowded = cb_crowded(); /* more calls in flight than CPUs to poll on: sleep at once */
  for (unsigned polls = 0; LOAD(*word) == seen; polls++) {
    if (crowded) {
      futex_park(word, seen);
      parked = 1;
      continue;
    }
    if ((polls & 63u) == 63u || parked) {
      const unsigned long long t = now_ns();
      if (!*t0)
        *t0 = t;
      if (t - *t0 >= g_cb_spin_ns) {
        futex_park(word, seen);
        parked = 1;
        continue;
      }
    }
    cpu_relax();
  }
  if (parked)
    futex_wake_n(word, g_cb_fanout);
}

/* a generation became FREE, or left the OPEN state: callers waiting for a slot may try again */
static void cb_turnover(cb_t *cb) {
  __atomic_add_fetch(&cb->turnover, 1, __ATOMIC_RELEASE);
  futex_wake_n(&cb->turnover, g_cb_fanout);
}
/* a launch slot is free: the launchers of OPEN generations (at most two exist) may take it */
static void cb_launch_slot(cb_t *cb) {
  __atomic_add_fetch(&cb->launch_seq, 1, __ATOMIC_RELEASE);
  futex_wake_all(&cb->launch_seq);
}

/* the generation's launches: one DMA for the staged rows, one kernel per (mode, palette) group, one wait */
static void cb_run(cb_t *cb, cb_gen_t *G) {
  cb_gen_t *const S = G; /* stream, hand-off words and epoch belong to the generation */
  int caps[ACHIP_VARIANT_COUNT];
  for (int v = 0; v < ACHIP_VARIANT_COUNT; v++)
    caps[v] = achip_variant_cap(v);
  hipError_t e = hipSuccess;
  const char *what = "";
  const unsigned long long t_issue = g_cb_stats ? now_ns() : 0;
  if (G->arena_used && G->need_dma)
    e = hipMemcpyAsync(G->arena_dev, G->arena_host, G->arena_used, hipMemcpyHostToDevice, S->stream), what = "hipMemcpyAsync";
  int order[CB_MAX], done[CB_MAX] = {0}, placed = 0;
  size_t cursor = 0;
  int base_of_group[CB_MAX], n_of_group[CB_MAX], groups = 0;
  for (int i = 0; i < G->n && e == hipSuccess; i++) {
    if (done[i])
      continue;
    const int base = placed;
    size_t stride = 0;
    int generic = 0;
    for (int j = i; j < G->n; j++)
      if (!done[j] && G->req[j].mode == G->req[i].mode && G->req[j].lut == G->req[i].lut) {
        done[j] = 1;
        order[placed] = j;
        G->descs_host[placed] = G->req[j].desc;
        generic |= (long)G->req[j].desc.src_w * (long)G->req[j].desc.src_h == 1;
        if (G->req[j].bound > stride)
          stride = G->req[j].bound;
        placed++;
      }
    const int n = placed - base;
    for (int k = 0; k < n; k++) {
      G->req[order[base + k]].out_off = cursor + (size_t)k * stride;
      G->lens_host[base + k] = ACHIP_LEN_BADDESC;
    }
    int variant = -1, parts = 1, rpp = 1;
    /* whole frames, never row bands: the bands of a frame wait for each other across workgroups, which is safe for ONE
     * launch (a launch dispatches in order) but not for six generations and a crowd of direct callers in flight at once --
     * workgroups of different launches then fill the CUs of one XCD while the bands they wait for queue on another, and
     * the bounded wait turns the stall into ACHIP_LEN_OVERFLOW (seen at 128 calling threads; the guide: dispatch order
     * across XCDs is undefined).  A generation's launch is latency-bound either way (~10 us). */
    if (achip_choose_geometry(G->req[i].mode, G->descs_host + base, n, G->req[i].ascii != 0, caps, cb->cus, -1, -1, &variant,
                              &parts, &rpp) != 0 ||
        variant < 0) {
      e = hipErrorInvalidValue, what = "geometry selection";
      break;
    }
    if (parts > 1 && (size_t)n * (size_t)parts > S->part_sync_n) {
      if (S->part_sync)
        (void)hipFree(S->part_sync);
      S->part_sync = NULL;
      S->part_sync_n = 0;
      const size_t words = (size_t)n * (size_t)parts * 2;
      e = hipMalloc((void **)&S->part_sync, words * sizeof(unsigned long long));
      if (e == hipSuccess) /* STREAM-ORDERED: a plain hipMemset runs on the null stream, which this non-blocking stream
                              does not wait for -- the band kernels would read recycled words whose epochs may match */
        e = hipMemsetAsync(S->part_sync, 0, words * sizeof(unsigned long long), S->stream);
      what = "hipMalloc(part_sync)";
      if (e != hipSuccess)
        break;
      S->part_sync_n = words;
    }
    S->epoch = S->epoch + 1u ? S->epoch + 1u : 1u;
    achip_uniform_t uni;
    (void)achip_frames_uniform(G->descs_host + base, n, &uni);
    uni.flags = (G->req[i].ascii ? ACHIP_UNIFORM_PALETTE_ASCII : 0u) | ACHIP_UNIFORM_MAX_CELLS(achip_uniform_extent(G->req[i].mode, variant, G->descs_host + base, n));
    e = (hipError_t)achip_launch_render(G->req[i].mode, variant, generic, G->descs_dev + base, n, G->req[i].lut,
                                        G->slab_dev + cursor, (uint64_t)stride, G->lens_dev + base, NULL, parts, rpp,
                                        parts > 1 ? S->part_sync : NULL, S->epoch, &uni, S->stream);
    what = "render kernel launch";
    if (e != hipSuccess)
      break; /* this group is not counted: `groups` is exactly the number of launches that were enqueued */
    base_of_group[groups] = base;
    n_of_group[groups++] = n;
    cursor += (size_t)n * stride;
  }
  /* every path that leaves the loop early (geometry selection, part_sync allocation, the launch itself) does so BEFORE
   * counting its group, so the groups counted are the groups whose kernels are in the queue -- and whenever anything
   * at all was enqueued (a kernel, the arena DMA) the stream is drained before the generation may be recycled */
  const int groups_launched = groups;
  const int enqueued = groups > 0 || (G->arena_used && G->need_dma);
  if (enqueued || e == hipSuccess) {
    /* the callers of this generation are parked on it: poll, do not add a driver wake-up to their latency -- but not for
     * ever: past the deadline the stream is synchronised (which reports a wedged queue) and the generation fails */
    hipError_t q;
    struct timespec t0, t;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const unsigned long long t_poll = g_cb_stats ? now_ns() : 0;
    unsigned polls = 0;
    if (g_cb_block && cb_crowded() && hipEventRecord(S->done_ev, S->stream) == hipSuccess) {
      /* more callers than CPUs: sleep until the GPU's interrupt.  A polling launcher holds a CPU for the whole run of its
       * generation, six of them a third of a 16-CPU quota, and when the quota runs out it is the launchers that stand
       * still with everybody waiting for them (poll 20 us -> 430 us at 128 callers) */
      q = hipEventSynchronize(S->done_ev);
    } else
      while ((q = hipStreamQuery(S->stream)) == hipErrorNotReady) {
        for (int k = 0; k < g_cb_poll_pause; k++)
          cpu_relax();
        if ((++polls & 0xFFFu) == 0u) {
          clock_gettime(CLOCK_MONOTONIC, &t);
          if (t.tv_sec - t0.tv_sec >= CB_RUN_DEADLINE_S) {
            q = hipStreamSynchronize(S->stream);
            if (q == hipSuccess)
              q = hipErrorNotReady; /* it did drain, but far too late for anyone to trust this queue */
            break;
          }
        }
      }
    if (g_cb_stats) {
      const unsigned long long t_end = now_ns();
      ST_ADD(gens, 1);
      ST_ADD(ns_issue, t_poll - t_issue);
      ST_ADD(ns_poll, t_end - t_poll);
    }
    if (q != hipSuccess && e == hipSuccess)
      e = q, what = "hipStreamQuery", groups = 0; /* nothing of this generation can be trusted */
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    /* groups that were launched before the failure completed above: their members get their frames; the members of the
     * failing group and of the groups never launched get the error (one bad group used to fail them all) */
    for (int g = 0; g < (groups_launched > 0 && groups > 0 ? groups_launched : 0); g++)
      for (int k = 0; k < n_of_group[g]; k++)
        G->req[order[base_of_group[g] + k]].len = G->lens_host[base_of_group[g] + k];
    for (int i = 0; i < G->n; i++)
      G->req[i].failed = G->req[i].len == ACHIP_LEN_BADDESC;
    G->failed = 1;
    const char *msg = hipGetErrorString(e);
    size_t k = 0;
    for (const char *p = what; *p && k + 1 < sizeof(G->err); p++)
      G->err[k++] = *p;
    for (const char *p = " failed: "; *p && k + 1 < sizeof(G->err); p++)
      G->err[k++] = *p;
    for (const char *p = msg; *p && k + 1 < sizeof(G->err); p++)
      G->err[k++] = *p;
    G->err[k] = 0;
    return;
  }
  for (int g = 0; g < groups; g++)
    for (int k = 0; k < n_of_group[g]; k++)
      G->req[order[base_of_group[g] + k]].len = G->lens_host[base_of_group[g] + k];
}

/* an OPEN generation is launched when it holds its share of the calls in flight, or has lingered long enough */
static int cb_ripe(cb_gen_t *G) {
  int want = __atomic_load_n(&g_cb_callers, __ATOMIC_RELAXED) / g_cb_share;
  want = want < 1 ? 1 : want > CB_MAX ? CB_MAX : want;
  return __atomic_load_n(&G->n_pub, __ATOMIC_ACQUIRE) >= want || LOAD(G->full) || now_ns() - G->t_open >= g_cb_linger_ns;
}

/* Render one frame through the combiner.  f->src is HOST pixels (src_bytes long).  Returns the malloc'd string, or
 * NULL with *handled = 1 on failure (achip_fail has the reason), or NULL with *handled = 0 when the request is not
 * combinable and the caller should take the direct path. */
char *achip_combine_render(int mode, const char *palette, const achip_lut_t *lut, const achip_frame_t *f, size_t src_bytes,
                           int *handled) {
  *handled = 0;
  pthread_once(&g_cb_once, cb_global_init);
  if (!g_cb_enabled || f->comp)
    return NULL;
  { /* with hysteresis: coalescing starts at min_callers calls in flight and stops below half of that.  The count of T
     * steadily calling threads hovers a little below T (they also free strings and loop), and a threshold without
     * memory made T = min_callers threads flip between the two paths call by call -- slower than either */
    static int engaged;
    const int callers = __atomic_load_n(&g_cb_callers, __ATOMIC_RELAXED);
    int on = __atomic_load_n(&engaged, __ATOMIC_RELAXED);
    if (!on && callers >= g_cb_min_callers)
      __atomic_store_n(&engaged, on = 1, __ATOMIC_RELAXED);
    else if (on && 2 * callers < g_cb_min_callers)
      __atomic_store_n(&engaged, on = 0, __ATOMIC_RELAXED);
    if (!on)
      return NULL;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CB_DEVICES)
    return NULL;
  cb_t *cb = &g_cb[dev];

  /* what has to be staged: of an image that is sampled sparsely only the pixels the sampler asks for (achip_stage_gather:
   * the sampled rows; their sampled columns too when the frame is at most half as wide as the source; image.c:293-312) --
   * of a pool-pinned image as well: 1920 three-byte samples picked up by the CPU and read densely by the kernel beat 1920
   * sparse reads across PCIe (pooled 199 k calls/s in place against 250 k staged at 32 callers); an image every pixel of
   * which is needed is read in place when it is pool-pinned and copied whole otherwise */
  achip_frame_t d = *f;
  const uint8_t *host_px = f->src;
  const void *alias = achip_pool_device_ptr(host_px);
  int sw, sh;
  const size_t part = achip_stage_extent(&d, &sw, &sh);
  if (part)
    alias = NULL;
  const size_t need = alias ? 0 : (part ? part : src_bytes);
  const int inplace = g_cb_inplace && need && need <= CB_INPLACE_MAX;
  const size_t need_al = (need + 255u) & ~(size_t)255;
  const size_t bound = (achip_out_bound(mode, f) + 1 + 15) & ~(size_t)15;
  if (need_al > CB_ARENA / 4 || bound > CB_SLAB / 4)
    return NULL; /* a giant: not worth holding a generation for */
  { /* a frame no kernel geometry can render must fail alone, on the direct path, not take a generation down with it */
    int caps[ACHIP_VARIANT_COUNT], variant = -1, parts = 1, rpp = 1;
    for (int v = 0; v < ACHIP_VARIANT_COUNT; v++)
      caps[v] = achip_variant_cap(v);
    if (achip_choose_geometry(mode, f, 1, achip_palette_ascii_only(palette), caps, 256, -1, -1, &variant, &parts, &rpp) != 0 ||
        variant < 0)
      return NULL;
  }

  const unsigned long long t_a = g_cb_stats ? now_ns() : 0;
  pthread_mutex_lock(&cb->mu);
  if (cb->ready == 0)
    cb_device_init(cb);
  if (cb->ready < 0) {
    pthread_mutex_unlock(&cb->mu);
    return NULL;
  }
  *handled = 1;
  cb_gen_t *G = NULL;
  unsigned long long t_wait0 = 0;
  for (;;) { /* a slot in the open generation (mu held at the top of every iteration) */
    if (cb->open < 0)
      for (int g = 0; g < cb->n_gens && cb->open < 0; g++)
        if (LOAD(cb->gen[g].state) == GEN_FREE) {
          cb_gen_t *N = &cb->gen[g];
          N->n = N->filled = N->copied = N->failed = 0;
          N->n_pub = N->full = 0;
          N->arena_used = N->max_bound = 0;
          N->need_dma = 0;
          N->t_open = now_ns();
          STORE(N->state, GEN_OPEN);
          cb->open = g;
        }
    if (cb->open >= 0) {
      G = &cb->gen[cb->open];
      const size_t mb = bound > G->max_bound ? bound : G->max_bound;
      if (G->n < CB_MAX && G->arena_used + need_al <= CB_ARENA && (size_t)(G->n + 1) * mb <= CB_SLAB)
        break;
      /* full: its members launch it; the next generation opens now if one is free */
      STORE(G->full, 1);
      cb->open = -1;
      int have_free = 0;
      for (int g = 0; g < cb->n_gens; g++)
        have_free |= LOAD(cb->gen[g].state) == GEN_FREE;
      if (have_free)
        continue;
    }
    if (cb->n_gens < CB_GENS && !cb->grow_failed) { /* every generation is busy: one more (a few ms, once) */
      if (cb_gen_setup(&cb->gen[cb->n_gens]) == 0) {
        cb->n_gens++;
        continue;
      }
      (void)hipGetLastError();
      cb->grow_failed = 1;
    }
    /* every generation is busy, or the open one is full (its members are about to launch it): wait for a turnover */
    /* (on the word, NOT through the mutex: waiters that re-took the mutex after every pause kept it so contended that the
     * members trying to launch their generation could not get it -- at 128 callers a generation that runs for 30 us was
     * launched every 200 us and a call spent 1.4 ms here) */
    const int seen = LOAD(cb->turnover);
    pthread_mutex_unlock(&cb->mu);
    cb_wait_change(&cb->turnover, seen, &t_wait0);
    pthread_mutex_lock(&cb->mu);
  }
  cb_req_t *r = &G->req[G->n++];
  uint8_t *const arena_dev = inplace ? G->arena_map : G->arena_dev;
  r->mode = mode;
  r->lut = lut;
  r->ascii = achip_palette_ascii_only(palette) ? 1 : 0;
  r->bound = bound;
  r->len = ACHIP_LEN_BADDESC;
  r->failed = 0;
  r->stage_off = alias ? (size_t)-1 : G->arena_used;
  G->arena_used += need_al;
  if (need && !inplace)
    G->need_dma = 1;
  __atomic_store_n(&G->n_pub, G->n, __ATOMIC_RELEASE);
  if (bound > G->max_bound)
    G->max_bound = bound;
  pthread_mutex_unlock(&cb->mu);
  const unsigned long long t_b = g_cb_stats ? now_ns() : 0;

  /* ---- fill the slot (every caller in parallel) */
  if (alias) {
    d.src = (const uint8_t *)alias;
  } else if (part) {
    achip_stage_gather(f, host_px, G->arena_host + r->stage_off, &d);
    d.src = arena_dev + r->stage_off;
  } else {
    memcpy(G->arena_host + r->stage_off, host_px, src_bytes);
    d.src = arena_dev + r->stage_off;
  }
  r->desc = d;
  const unsigned long long t_c = g_cb_stats ? now_ns() : 0;

  __atomic_add_fetch(&G->filled, 1, __ATOMIC_RELEASE);
  if (r == &G->req[0]) {
    /* the generation's first member launches it: once it is ripe (its share of the calls in flight has joined, it is full,
     * or it has lingered long enough -- a bounded poll), and once one of the CB_INFLIGHT launch slots is free */
    while (!cb_ripe(G))
      cpu_relax();
    unsigned long long t0 = 0;
    for (;;) {
      const int seen = LOAD(cb->launch_seq);
      if (LOAD(cb->inflight) < g_cb_inflight) {
        pthread_mutex_lock(&cb->mu);
        if (LOAD(cb->inflight) < g_cb_inflight) /* (decremented outside the mutex: an atomic read) */
          break;
        pthread_mutex_unlock(&cb->mu);
      }
      cb_wait_change(&cb->launch_seq, seen, &t0);
    }
    __atomic_add_fetch(&cb->inflight, 1, __ATOMIC_ACQ_REL); /* (an atomic RMW: the decrement below runs outside the mutex,
                                                               and a plain read-add-store here lost decrements -- the count
                                                               crept up until no generation could launch) */
    STORE(G->state, GEN_CLOSED);
    if (cb->open >= 0 && &cb->gen[cb->open] == G)
      cb->open = -1;
    const int members = G->n; /* final from here on */
    pthread_mutex_unlock(&cb->mu);
    cb_turnover(cb); /* the next caller opens a fresh generation */
    while (LOAD(G->filled) < members) /* members still copying their pixels: a memcpy away */
      cpu_relax();
    cb_run(cb, G);
    STORE(G->state, GEN_DONE);
    futex_wake_n(&G->state, g_cb_fanout); /* the members that went to sleep; each passes it on */
    __atomic_sub_fetch(&cb->inflight, 1, __ATOMIC_RELEASE);
    cb_launch_slot(cb);
  } else {
    /* everybody else waits for DONE on the generation's state word (OPEN -> CLOSED is none of their business: a park on
     * a stale value returns at once) */
    unsigned long long t0 = 0;
    for (int st; (st = LOAD(G->state)) != GEN_DONE;)
      cb_wait_change(&G->state, st, &t0);
  }
  const unsigned long long t_d = g_cb_stats ? now_ns() : 0;
  const int failed = G->failed && r->failed;
  const uint32_t len = r->len;
  const size_t out_off = r->out_off;
  char errbuf[160];
  memcpy(errbuf, G->err, sizeof(errbuf));

  /* ---- take the result out (every caller in parallel) */
  char *out = NULL;
  if (failed) {
    achip_fail(ASCIICHAT_HIP_ERR_INVALID_STATE, "%s", errbuf);
  } else if (len >= 0xFFFFFFF0u) {
    achip_fail(ASCIICHAT_HIP_ERR_BUFFER, "render kernel reported %s",
               len == ACHIP_LEN_OVERFLOW ? "output overflow" : "a bad descriptor");
  } else {
    out = achip_out_take(G->slab_host + out_off, len); /* a malloc block, or the caller's buffer (..._into) */
  }
  /* (the member count is read BEFORE this member counts itself out: once it has, the others may finish, the generation
   * may be recycled and G->n may belong to its next life -- a member that then compared its count with the new n could
   * "free" a generation in use; with sixteen-member generations cycling fast that happened within seconds) */
  const int members_out = G->n;
  if (__atomic_add_fetch(&G->copied, 1, __ATOMIC_ACQ_REL) == members_out) { /* last one out */
    STORE(G->state, GEN_FREE);
    cb_turnover(cb);
  }
  if (g_cb_stats) {
    const unsigned long long t_e = now_ns();
    ST_ADD(calls, 1);
    ST_ADD(ns_slot, t_b - t_a);
    ST_ADD(ns_fill, t_c - t_b);
    ST_ADD(ns_wait, t_d - t_c);
    ST_ADD(ns_out, t_e - t_d);
  }
  return out;
}
