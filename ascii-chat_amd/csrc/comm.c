/*
 * comm.c -- the multi-GPU side of the C-ABI (asciichat_hip.h): one process per GPU, RCCL over xGMI.
 *
 * Frames are independent, so the render path itself needs no collective: a batch is partitioned over the
 * ranks and every rank renders its block.  Collectives exist where a consumer needs remote data:
 *   * asciichat_hip_comm_all_gather_slab -- every rank's block of a fixed-stride output slab + its uint32 lengths, in
 *     place, as ONE group of two ncclAllGather calls per batch (messages are KB..MB: latency-bound on xGMI, never
 *     issue one per frame);
 *   * asciichat_hip_grid_* -- BASELINE config 4, the server's pixel-space grid (create_multi_source_composite,
 *     src/server/stream.c:664-779; consumer convert_composite_to_ascii :790-854): every rank nearest-neighbour-resizes
 *     the sources it owns into their composite tiles (<= 53x30x3 B each for nine 1080p sources at 160x48), one
 *     ncclAllGather moves all tiles to every rank, and each rank then renders the composite for ITS target clients
 *     straight from the gathered tiles (the fused composite sampler; the W x 2H canvas stays virtual).
 *
 * librccl is loaded lazily with dlopen: the drop-in library must stay loadable on hosts that never go multi-GPU.
 * The unique id of a communicator travels out of band (the server's control plane; torch.distributed in bench.py).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "achip_host.h"
#include "asciichat_hip.h"
#include "hip_launch.h"
#include "internal.h"

/* ------------------------------------------------------------------------------------------- */
/* librccl, resolved on first use.  The handful of types and prototypes this file needs are declared here instead of  */
/* including <rccl/rccl.h>: the library is dlopen'ed, so hosts without the RCCL development headers can still build    */
/* the drop-in library.  Values follow nccl.h (ncclUniqueId = 128 opaque bytes, ncclUint8 = 1,         */
/* ncclSuccess = 0), which RCCL keeps ABI-stable.                                                                      */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
  char internal[ASCIICHAT_HIP_COMM_ID_BYTES];
} rccl_unique_id_t;
typedef struct ncclComm *rccl_comm_t;
typedef int rccl_result_t; /* ncclResult_t: 0 = ncclSuccess */
enum { RCCL_SUCCESS = 0, RCCL_UINT8 = 1 /* RCCL_UINT8 */ };

static struct {
  void *handle;
  rccl_result_t (*GetUniqueId)(rccl_unique_id_t *);
  rccl_result_t (*CommInitRank)(rccl_comm_t *, int, rccl_unique_id_t, int);
  rccl_result_t (*CommDestroy)(rccl_comm_t);
  rccl_result_t (*CommCount)(rccl_comm_t, int *);
  rccl_result_t (*AllGather)(const void *, void *, size_t, int, rccl_comm_t, hipStream_t);
  rccl_result_t (*GroupStart)(void);
  rccl_result_t (*GroupEnd)(void);
  const char *(*GetErrorString)(rccl_result_t);
  int state; /* 0 = not tried, 1 = ready, -1 = unavailable */
  char why[160]; /* the loader's message, kept from the one attempt (dlerror() is NULL on later calls) */
} g_rccl;
static pthread_mutex_t g_rccl_mu = PTHREAD_MUTEX_INITIALIZER;

static int rccl_load(void) {
  pthread_mutex_lock(&g_rccl_mu);
  if (g_rccl.state == 0) {
    /* ASCIICHAT_HIP_RCCL_LIB names another library with the same seven entry points: the tests use it to run this
     * file with a world of two on a one-GPU box over a stand-in transport (tests/cabi/loopback_rccl.c) */
    const char *env = getenv("ASCIICHAT_HIP_RCCL_LIB");
    const char *names[] = {env && env[0] ? env : "librccl.so.1", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    const size_t n_names = env && env[0] ? 1 : sizeof(names) / sizeof(names[0]);
    for (size_t i = 0; i < n_names && !g_rccl.handle; i++)
      g_rccl.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    g_rccl.state = -1;
    if (g_rccl.handle) {
      *(void **)&g_rccl.GetUniqueId = dlsym(g_rccl.handle, "ncclGetUniqueId");
      *(void **)&g_rccl.CommInitRank = dlsym(g_rccl.handle, "ncclCommInitRank");
      *(void **)&g_rccl.CommDestroy = dlsym(g_rccl.handle, "ncclCommDestroy");
      *(void **)&g_rccl.CommCount = dlsym(g_rccl.handle, "ncclCommCount");
      *(void **)&g_rccl.AllGather = dlsym(g_rccl.handle, "ncclAllGather");
      *(void **)&g_rccl.GroupStart = dlsym(g_rccl.handle, "ncclGroupStart");
      *(void **)&g_rccl.GroupEnd = dlsym(g_rccl.handle, "ncclGroupEnd");
      *(void **)&g_rccl.GetErrorString = dlsym(g_rccl.handle, "ncclGetErrorString");
      if (g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.CommCount && g_rccl.AllGather &&
          g_rccl.GroupStart && g_rccl.GroupEnd && g_rccl.GetErrorString)
        g_rccl.state = 1;
      else
        snprintf(g_rccl.why, sizeof(g_rccl.why), "%s lacks an nccl* entry point", names[0]);
    } else {
      const char *e = dlerror();
      snprintf(g_rccl.why, sizeof(g_rccl.why), "%s", e ? e : "dlopen failed");
    }
  }
  const int ok = g_rccl.state == 1;
  pthread_mutex_unlock(&g_rccl_mu);
  return ok ? 0 : achip_fail(ASCIICHAT_HIP_ERR_NOT_SUPPORTED, "librccl.so.1 is not loadable: %s", g_rccl.why);
}

/* a failed collective is a broken communicator, not a missing device (ERR_NO_DEVICE always prints to stderr) */
static int rccl_check(rccl_result_t r, const char *what) {
  if (r == RCCL_SUCCESS)
    return 0;
  return achip_fail(ASCIICHAT_HIP_ERR_INVALID_STATE, "%s failed: %s", what, g_rccl.GetErrorString(r));
}

/* ------------------------------------------------------------------------------------------- */
/* partition of n independent items over `world` ranks: contiguous, balanced (the first n % world ranks hold one    */
/* more) -- nine sources over eight GPUs is (2,1,1,1,1,1,1,1), not (2,2,2,2,1,0,0,0)                                */
/* ------------------------------------------------------------------------------------------- */
void achip_shard_bounds(int n_items, int world, int rank, int *first, int *count) {
  if (world < 1)
    world = 1;
  if (n_items < 0)
    n_items = 0;
  const int base = n_items / world, extra = n_items % world;
  const int f = rank * base + (rank < extra ? rank : extra);
  if (first)
    *first = f;
  if (count)
    *count = base + (rank < extra ? 1 : 0);
}

int achip_shard_owner(int n_items, int world, int item) {
  if (world < 1 || item < 0 || item >= n_items)
    return -1;
  const int base = n_items / world, extra = n_items % world;
  const int big = extra * (base + 1); /* items held by the ranks that hold base + 1 */
  if (item < big)
    return item / (base + 1);
  return base ? extra + (item - big) / base : -1;
}

/* slots every rank reserves in a gathered buffer: the largest block (blocks are padded to it so that ONE
 * fixed-size ncclAllGather moves everything) */
int achip_shard_slots(int n_items, int world) {
  if (world < 1)
    world = 1;
  return n_items <= 0 ? 0 : (n_items + world - 1) / world;
}

/* ------------------------------------------------------------------------------------------- */
/* communicator                                                                                   */
/* ------------------------------------------------------------------------------------------- */
struct asciichat_hip_comm {
  rccl_comm_t comm;
  int world, rank;
};

int asciichat_hip_comm_unique_id(void *id_out, size_t id_bytes) {
  if (!id_out || id_bytes < ASCIICHAT_HIP_COMM_ID_BYTES)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "comm_unique_id: need a %d-byte buffer", ASCIICHAT_HIP_COMM_ID_BYTES);
  int rc = rccl_load();
  if (rc)
    return rc;
  rccl_unique_id_t id;
  rc = rccl_check(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  if (!rc)
    memcpy(id_out, &id, sizeof(id));
  return rc;
}

int asciichat_hip_comm_init(asciichat_hip_comm_t **comm, int world, int rank, const void *id, size_t id_bytes) {
  if (!comm || world < 1 || rank < 0 || rank >= world || !id || id_bytes < ASCIICHAT_HIP_COMM_ID_BYTES)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "comm_init: bad arguments");
  *comm = NULL;
  int rc = achip_require_device();
  if (!rc)
    rc = rccl_load();
  if (rc)
    return rc;
  asciichat_hip_comm_t *c = (asciichat_hip_comm_t *)calloc(1, sizeof(*c));
  if (!c)
    return achip_fail(ASCIICHAT_HIP_ERR_MEMORY, "out of memory");
  rccl_unique_id_t uid;
  memcpy(&uid, id, sizeof(uid));
  rc = rccl_check(g_rccl.CommInitRank(&c->comm, world, uid, rank), "ncclCommInitRank");
  if (rc) {
    free(c);
    return rc;
  }
  c->world = world;
  c->rank = rank;
  *comm = c;
  return 0;
}

int asciichat_hip_comm_world(const asciichat_hip_comm_t *c) { return c ? c->world : 1; }
/* ranks the communicator itself reports (ncclCommCount) -- what bench.py prints next to n_gpus */
int asciichat_hip_comm_count(const asciichat_hip_comm_t *c) {
  int n = 0;
  if (!c || !c->comm || g_rccl.state != 1 || rccl_check(g_rccl.CommCount(c->comm, &n), "ncclCommCount"))
    return -1;
  return n;
}
int asciichat_hip_comm_rank(const asciichat_hip_comm_t *c) { return c ? c->rank : 0; }

void asciichat_hip_comm_destroy(asciichat_hip_comm_t *c) {
  if (!c)
    return;
  if (c->comm && g_rccl.state == 1)
    (void)g_rccl.CommDestroy(c->comm);
  free(c);
}

int asciichat_hip_comm_all_gather(asciichat_hip_comm_t *c, const void *send_dev, void *recv_dev, size_t bytes_per_rank,
                                  void *stream) {
  if (!c || !send_dev || !recv_dev)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "comm_all_gather: bad arguments");
  if (bytes_per_rank == 0)
    return 0;
  return rccl_check(g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, RCCL_UINT8, c->comm, (hipStream_t)stream),
                    "ncclAllGather");
}

/* In-place all-gather of a sharded output slab: rank r rendered its frames into slots [r*slots, (r+1)*slots) of
 * slab_dev (slot i at slab_dev + i*stride) and their lengths into len_dev[r*slots ..]; afterwards every rank holds
 * all world*slots slots.  One group = one launch on the wire for bytes and lengths together. */
int asciichat_hip_comm_all_gather_slab(asciichat_hip_comm_t *c, uint8_t *slab_dev, size_t stride, uint32_t *len_dev,
                                       int slots_per_rank, void *stream) {
  if (!c || !slab_dev || !len_dev || slots_per_rank < 0)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "comm_all_gather_slab: bad arguments");
  if (slots_per_rank == 0)
    return 0;
  const size_t block = (size_t)slots_per_rank * stride;
  int rc = rccl_check(g_rccl.GroupStart(), "ncclGroupStart");
  if (rc)
    return rc;
  rccl_result_t a = g_rccl.AllGather(slab_dev + (size_t)c->rank * block, slab_dev, block, RCCL_UINT8, c->comm,
                                    (hipStream_t)stream);
  rccl_result_t b = g_rccl.AllGather(len_dev + (size_t)c->rank * (size_t)slots_per_rank, len_dev,
                                    (size_t)slots_per_rank * sizeof(uint32_t), RCCL_UINT8, c->comm, (hipStream_t)stream);
  rccl_result_t e = g_rccl.GroupEnd();
  if ((rc = rccl_check(a, "ncclAllGather(slab)")) || (rc = rccl_check(b, "ncclAllGather(lengths)")))
    return rc;
  return rccl_check(e, "ncclGroupEnd");
}

/* The same exchange with compacted blocks: what crosses xGMI is the bytes in use, not the worst-case stride (SURVEY 8e:
 * "prefer gathering compacted per-rank buffers ... lengths first").  Three steps on `stream`: (1) in-place all-gather
 * of the lengths, read back by the host (the one synchronisation: collectives need equal sizes on every rank, and only
 * the host can size them); (2) pack_frames of this rank's block to its place in packed_dev; (3) in-place all-gather of
 * max-over-ranks packed bytes. */
int asciichat_hip_comm_all_gather_packed(asciichat_hip_comm_t *c, const uint8_t *slab_dev, size_t stride, uint32_t *len_dev,
                                         int slots_per_rank, uint8_t *packed_dev, size_t packed_capacity_per_rank,
                                         uint64_t *off_host, uint32_t *len_host, size_t *block_bytes, void *stream) {
  if (!c || !slab_dev || !len_dev || !packed_dev || !off_host || slots_per_rank < 0 || (packed_capacity_per_rank & 15u) ||
      ((uintptr_t)packed_dev & 15u))
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "comm_all_gather_packed: bad arguments");
  if (block_bytes)
    *block_bytes = 0;
  if (slots_per_rank == 0)
    return 0;
  const size_t slots = (size_t)slots_per_rank, total = slots * (size_t)c->world;
  /* everything that can fail on THIS rank alone happens before the first collective: a rank that returned
   * between the two all-gathers would leave the others blocked inside RCCL */
  uint32_t *lens = len_host ? len_host : (uint32_t *)malloc(total * sizeof(uint32_t));
  if (!lens)
    return achip_fail(ASCIICHAT_HIP_ERR_MEMORY, "out of memory");
  int rc = rccl_check(g_rccl.AllGather(len_dev + (size_t)c->rank * slots, len_dev, slots * sizeof(uint32_t), RCCL_UINT8,
                                       c->comm, (hipStream_t)stream),
                      "ncclAllGather(lengths)");
  if (rc) {
    if (!len_host)
      free(lens);
    return rc;
  }
  rc = achip_hip_check((int)hipMemcpyAsync(lens, len_dev, total * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream),
                       "hipMemcpyAsync(lengths)");
  if (!rc)
    rc = achip_hip_check((int)hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
  size_t block = 0;
  if (!rc) {
    for (int r = 0; r < c->world; r++) { /* every rank computes the same table */
      size_t b = 0;
      for (size_t i = 0; i < slots; i++) {
        const uint32_t l = lens[(size_t)r * slots + i];
        b += l >= 0xFFFFFFF0u ? 0u : ((size_t)l + 15u) & ~(size_t)15;
      }
      if (b > block)
        block = b;
    }
    if (block == 0)
      block = 16; /* an all-gather of nothing is still a well-formed exchange */
    if (block > packed_capacity_per_rank)
      rc = achip_fail(ASCIICHAT_HIP_ERR_BUFFER, "comm_all_gather_packed: a rank's block needs %zu bytes, capacity is %zu", block,
                      packed_capacity_per_rank);
  }
  if (!rc) {
    for (int r = 0; r < c->world; r++) {
      size_t o = (size_t)r * block;
      for (size_t i = 0; i < slots; i++) {
        const uint32_t l = lens[(size_t)r * slots + i];
        off_host[(size_t)r * slots + i] = o;
        o += l >= 0xFFFFFFF0u ? 0u : ((size_t)l + 15u) & ~(size_t)15;
      }
    }
    /* from here on every rank holds the same table and the same `block` (the capacity must be the same on every rank: a
     * block that does not fit fails on all of them alike, above, and nobody enters the second collective).  A pack launch
     * that fails on this rank alone still joins the exchange -- the others are already on their way into it -- and
     * reports its error afterwards */
    const int prc = asciichat_hip_pack_frames(slab_dev + (size_t)c->rank * slots * stride, stride, len_dev + (size_t)c->rank * slots,
                                              slots_per_rank, packed_dev + (size_t)c->rank * block, block, NULL, NULL, stream);
    rc = rccl_check(g_rccl.AllGather(packed_dev + (size_t)c->rank * block, packed_dev, block, RCCL_UINT8, c->comm,
                                     (hipStream_t)stream),
                    "ncclAllGather(packed)");
    if (prc)
      rc = prc;
  }
  if (!len_host)
    free(lens);
  if (!rc && block_bytes)
    *block_bytes = block;
  return rc;
}

/* "Every rank needs every rank's frames" behind ONE entry, the form picked per call or by the environment -- so that the first
 * visit to a real multi-GPU node is a one-shot A/B (src/server/render.c:1233: one render thread per client
 * there, one process per GPU here):
 *   form 0 / ASCIICHAT_HIP_GATHER=packed   comm_all_gather_packed: lengths first, the host sizes the second collective -- the
 *                                          bytes in use cross the links, at the price of one stream synchronisation;
 *   form 1 / ASCIICHAT_HIP_GATHER=slab     comm_all_gather_slab: lengths + the worst-case-stride slab in ONE group -- no host
 *                                          synchronisation, stride bytes per frame on the links;
 *   form -1                                what the environment says (default: packed).
 * Where the frames are afterwards: *base_out + off_host[i] (the slab itself in form 1, packed_dev in form 0); len_host is
 * filled in form 0 only (form 1 never visits the host: the lengths are in len_dev).  *form_out = the form taken. */
int asciichat_hip_comm_all_gather_frames(asciichat_hip_comm_t *c, int form, uint8_t *slab_dev, size_t stride, uint32_t *len_dev,
                                         int slots_per_rank, uint8_t *packed_dev, size_t packed_capacity_per_rank,
                                         const uint8_t **base_out, uint64_t *off_host, uint32_t *len_host, size_t *block_bytes,
                                         int *form_out, void *stream) {
  if (!c || form < -1 || form > 1 || !off_host || slots_per_rank < 0)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "comm_all_gather_frames: bad arguments");
  if (form < 0) {
    const char *e = getenv("ASCIICHAT_HIP_GATHER");
    form = e && (strcmp(e, "slab") == 0 || strcmp(e, "1") == 0) ? 1 : 0;
  }
  if (form_out)
    *form_out = form;
  if (form == 0) {
    if (base_out)
      *base_out = packed_dev;
    return asciichat_hip_comm_all_gather_packed(c, slab_dev, stride, len_dev, slots_per_rank, packed_dev, packed_capacity_per_rank,
                                                off_host, len_host, block_bytes, stream);
  }
  const int rc = asciichat_hip_comm_all_gather_slab(c, slab_dev, stride, len_dev, slots_per_rank, stream);
  if (rc)
    return rc;
  const size_t total = (size_t)slots_per_rank * (size_t)c->world;
  for (size_t i = 0; i < total; i++)
    off_host[i] = (uint64_t)i * stride;
  if (base_out)
    *base_out = slab_dev;
  if (block_bytes)
    *block_bytes = (size_t)slots_per_rank * stride;
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* the pixel-space grid across GPUs                                                               */
/* ------------------------------------------------------------------------------------------- */
struct asciichat_hip_grid {
  asciichat_hip_comm_t *comm; /* NULL: single GPU */
  int world, rank, n_src, slots; /* slots = tile slots every rank contributes (padded) */
  size_t tile_stride;
  achip_composite_t geom; /* the reference's geometry: tile sizes and origins of the (<= 9) PLACED sources */
  int placed_of[ASCIICHAT_HIP_GRID_MAX_SOURCES]; /* source -> index into geom.s, -1 = no video / beyond the ninth */
  uint8_t *tiles_dev;     /* world * slots * tile_stride bytes: slot of source k = owner(k)*slots + (k - first(owner)) */
  achip_composite_t *comp_dev; /* samples the gathered tiles: identity ratios, tile-sized "sources" */
  /* direct mode (one GPU): no tiles, no resize launch, no collective -- the render samples the clients' frames themselves
   * through comp_direct_dev (the reference's geometry with real ratios); a tick only refreshes its nine source pointers */
  int direct;
  achip_composite_t *comp_direct_dev;
  achip_comp_poke_t poked; /* the pointers comp_direct_dev currently holds */
  int poked_valid;
};

static size_t grid_slot_of(const asciichat_hip_grid_t *g, int k) {
  const int owner = achip_shard_owner(g->n_src, g->world, k);
  int first = 0;
  achip_shard_bounds(g->n_src, g->world, owner, &first, NULL);
  return (size_t)owner * (size_t)g->slots + (size_t)(k - first);
}

int asciichat_hip_grid_create(asciichat_hip_grid_t **grid, asciichat_hip_comm_t *comm, const int *src_w, const int *src_h,
                              const unsigned char *has_video, int n_src, int term_w, int term_h) {
  if (!grid || !src_w || !src_h || n_src < 1 || n_src > ASCIICHAT_HIP_GRID_MAX_SOURCES || term_w < 1 || term_h < 1)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "grid_create: bad arguments (1..%d sources)",
                      ASCIICHAT_HIP_GRID_MAX_SOURCES);
  *grid = NULL;
  int rc = achip_require_device();
  if (rc)
    return rc;
  asciichat_hip_grid_t *g = (asciichat_hip_grid_t *)calloc(1, sizeof(*g));
  if (!g)
    return achip_fail(ASCIICHAT_HIP_ERR_MEMORY, "out of memory");
  g->comm = comm;
  g->world = asciichat_hip_comm_world(comm);
  g->rank = asciichat_hip_comm_rank(comm);
  g->n_src = n_src;
  g->slots = achip_shard_slots(n_src, g->world);
  /* geometry only: every rank knows every source's size; a non-NULL dummy marks "has video" */
  const uint8_t *ptrs[ASCIICHAT_HIP_GRID_MAX_SOURCES];
  int placed = 0;
  for (int k = 0; k < n_src; k++) { /* the placement rule of achip_composite_setup: the first nine with video */
    const int video = (!has_video || has_video[k]) && src_w[k] > 0 && src_h[k] > 0;
    ptrs[k] = video ? (const uint8_t *)(uintptr_t)16 : NULL;
    g->placed_of[k] = video && placed < 9 ? placed++ : -1;
  }
  achip_composite_setup(&g->geom, ptrs, src_w, src_h, n_src, term_w, term_h);
  g->tile_stride = 16;
  for (int k = 0; k < g->geom.n_src; k++) {
    const size_t b = ((size_t)g->geom.s[k].tile_w * (size_t)g->geom.s[k].tile_h * 3u + 15u) & ~(size_t)15;
    if (g->geom.s[k].src && b > g->tile_stride)
      g->tile_stride = b;
  }
  const size_t bytes = (size_t)g->world * (size_t)g->slots * g->tile_stride;
  rc = achip_hip_check((int)hipMalloc((void **)&g->tiles_dev, bytes), "hipMalloc(grid tiles)");
  if (!rc)
    rc = achip_hip_check((int)hipMemset(g->tiles_dev, 0, bytes), "hipMemset(grid tiles)");
  if (!rc) /* the memset runs on the null stream, which the callers' (non-blocking) streams do not wait for */
    rc = achip_hip_check((int)hipDeviceSynchronize(), "hipDeviceSynchronize(grid tiles)");
  if (!rc) { /* the composite that samples the gathered tiles */
    achip_composite_t c2 = g->geom;
    for (int k = 0; k < n_src; k++) {
      if (g->placed_of[k] < 0)
        continue;
      achip_comp_src_t *s = &c2.s[g->placed_of[k]];
      if (!s->src)
        continue; /* a degenerate tile: stays an empty cell */
      s->src = g->tiles_dev + grid_slot_of(g, k) * g->tile_stride;
      s->src_w = s->tile_w;
      s->src_h = s->tile_h;
      s->src_stride = 3 * s->tile_w;
      s->x_ratio = s->y_ratio = 65537u; /* ((n << 16) / n) + 1: the tile is sampled 1:1 */
    }
    rc = asciichat_hip_composite_upload(&c2, &g->comp_dev);
  }
  if (!rc && g->world == 1) { /* the direct form: sources NULL until the first exchange names them */
    achip_composite_t c3 = g->geom;
    for (int k = 0; k < 9; k++)
      c3.s[k].src = NULL;
    rc = asciichat_hip_composite_upload(&c3, &g->comp_direct_dev);
  }
  if (rc) {
    asciichat_hip_grid_destroy(g);
    return rc;
  }
  *grid = g;
  return 0;
}

int asciichat_hip_grid_owner(const asciichat_hip_grid_t *g, int source) {
  return g ? achip_shard_owner(g->n_src, g->world, source) : -1;
}

const achip_composite_t *asciichat_hip_grid_composite_dev(const asciichat_hip_grid_t *g) {
  return !g ? NULL : g->direct ? g->comp_direct_dev : g->comp_dev;
}

/* One GPU only: on = render straight from the sources (a tick is a pointer refresh; pays while the targets are few:
 * every target's workgroups gather from the full-size frames), off = through the resized tiles (one resize per source
 * and tick, dense samples for every target).  Changes what grid_composite_dev() returns: fetch it again. */
int asciichat_hip_grid_set_direct(asciichat_hip_grid_t *g, int on) {
  if (!g)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "grid_set_direct: bad arguments");
  if (on && (g->world != 1 || !g->comp_direct_dev))
    return achip_fail(ASCIICHAT_HIP_ERR_NOT_SUPPORTED, "grid_set_direct: the sources of a %d-rank grid live on other GPUs",
                      g->world);
  g->direct = on ? 1 : 0;
  return 0;
}

const achip_composite_t *asciichat_hip_grid_geometry(const asciichat_hip_grid_t *g) { return g ? &g->geom : NULL; }

/* One tick: resize the sources this rank owns into their tile slots, then one all-gather of the tile slots.  After
 * `stream` has passed this point every rank's tiles are current and plans whose frames point at
 * asciichat_hip_grid_composite_dev() render the grid. */
int asciichat_hip_grid_exchange(asciichat_hip_grid_t *g, const uint8_t *const *local_src_dev, void *stream) {
  if (!g || !local_src_dev)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "grid_exchange: bad arguments");
  if (g->direct) { /* one GPU: name the sources; nothing to launch when they are the ones already named */
    achip_comp_poke_t poke;
    memset(&poke, 0, sizeof(poke));
    for (int k = 0; k < g->n_src; k++) {
      if (g->placed_of[k] < 0 || !g->geom.s[g->placed_of[k]].src)
        continue;
      if (!local_src_dev[k])
        return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "grid_exchange: source %d is owned by this rank but NULL", k);
      poke.src[g->placed_of[k]] = local_src_dev[k];
    }
    if (g->poked_valid && memcmp(&poke, &g->poked, sizeof(poke)) == 0)
      return 0;
    const int rc = achip_hip_check(achip_launch_comp_poke(g->comp_direct_dev, &poke, stream), "composite pointer launch");
    if (!rc) {
      g->poked = poke;
      g->poked_valid = 1;
    }
    return rc;
  }
  int first = 0, count = 0;
  achip_shard_bounds(g->n_src, g->world, g->rank, &first, &count);
  achip_resize_batch_t batch; /* every tile this rank owns in ONE launch: the tiles are KBs, a launch each costs more */
  batch.n = 0;
  for (int k = first; k <= first + count; k++) {
    if (batch.n == ACHIP_RESIZE_BATCH_MAX || (k == first + count && batch.n > 0)) {
      const int rc = achip_hip_check(achip_launch_resize_batch(&batch, stream), "resize launch");
      if (rc)
        return rc;
      batch.n = 0;
    }
    if (k == first + count || g->placed_of[k] < 0)
      continue; /* no video in this slot (or beyond the ninth placed source) */
    const achip_comp_src_t *s = &g->geom.s[g->placed_of[k]];
    if (!s->src)
      continue;
    if (!local_src_dev[k])
      return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "grid_exchange: source %d is owned by this rank but NULL", k);
    achip_resize_item_t *it = &batch.item[batch.n++];
    it->src = local_src_dev[k];
    it->dst = g->tiles_dev + grid_slot_of(g, k) * g->tile_stride;
    it->sw = s->src_w;
    it->sh = s->src_h;
    it->dw = s->tile_w;
    it->dh = s->tile_h;
    it->x_ratio = it->y_ratio = 0;
  }
  if (!g->comm || g->world == 1)
    return g->comm ? asciichat_hip_comm_all_gather(g->comm, g->tiles_dev, g->tiles_dev,
                                                   (size_t)g->slots * g->tile_stride, stream)
                   : 0;
  const size_t block = (size_t)g->slots * g->tile_stride;
  return asciichat_hip_comm_all_gather(g->comm, g->tiles_dev + (size_t)g->rank * block, g->tiles_dev, block, stream);
}

void asciichat_hip_grid_destroy(asciichat_hip_grid_t *g) {
  if (!g)
    return;
  if (g->tiles_dev)
    (void)hipFree(g->tiles_dev);
  if (g->comp_dev)
    (void)hipFree(g->comp_dev);
  if (g->comp_direct_dev)
    (void)hipFree(g->comp_direct_dev);
  free(g);
}
