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
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "achip_host.h"
#include "asciichat_hip.h"
#include "hip_launch.h"
#include "internal.h"

/* ------------------------------------------------------------------------------------------- */
/* librccl, resolved on first use                                                                  */
/* ------------------------------------------------------------------------------------------- */
static struct {
  void *handle;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *);
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  ncclResult_t (*GroupStart)(void);
  ncclResult_t (*GroupEnd)(void);
  const char *(*GetErrorString)(ncclResult_t);
  int state; /* 0 = not tried, 1 = ready, -1 = unavailable */
} g_rccl;
static pthread_mutex_t g_rccl_mu = PTHREAD_MUTEX_INITIALIZER;

static int rccl_load(void) {
  pthread_mutex_lock(&g_rccl_mu);
  if (g_rccl.state == 0) {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (size_t i = 0; i < sizeof(names) / sizeof(names[0]) && !g_rccl.handle; i++)
      g_rccl.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    g_rccl.state = -1;
    if (g_rccl.handle) {
      *(void **)&g_rccl.GetUniqueId = dlsym(g_rccl.handle, "ncclGetUniqueId");
      *(void **)&g_rccl.CommInitRank = dlsym(g_rccl.handle, "ncclCommInitRank");
      *(void **)&g_rccl.CommDestroy = dlsym(g_rccl.handle, "ncclCommDestroy");
      *(void **)&g_rccl.AllGather = dlsym(g_rccl.handle, "ncclAllGather");
      *(void **)&g_rccl.GroupStart = dlsym(g_rccl.handle, "ncclGroupStart");
      *(void **)&g_rccl.GroupEnd = dlsym(g_rccl.handle, "ncclGroupEnd");
      *(void **)&g_rccl.GetErrorString = dlsym(g_rccl.handle, "ncclGetErrorString");
      if (g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllGather && g_rccl.GroupStart &&
          g_rccl.GroupEnd && g_rccl.GetErrorString)
        g_rccl.state = 1;
    }
  }
  const int ok = g_rccl.state == 1;
  pthread_mutex_unlock(&g_rccl_mu);
  return ok ? 0 : achip_fail(ASCIICHAT_HIP_ERR_NOT_SUPPORTED, "librccl.so.1 is not loadable: %s", dlerror());
}

static int rccl_check(ncclResult_t r, const char *what) {
  if (r == ncclSuccess)
    return 0;
  return achip_fail(ASCIICHAT_HIP_ERR_NO_DEVICE, "%s failed: %s", what, g_rccl.GetErrorString(r));
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
  ncclComm_t comm;
  int world, rank;
};

int asciichat_hip_comm_unique_id(void *id_out, size_t id_bytes) {
  if (!id_out || id_bytes < ASCIICHAT_HIP_COMM_ID_BYTES)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "comm_unique_id: need a %d-byte buffer", ASCIICHAT_HIP_COMM_ID_BYTES);
  int rc = rccl_load();
  if (rc)
    return rc;
  ncclUniqueId id;
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
  ncclUniqueId uid;
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
  return rccl_check(g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, ncclUint8, c->comm, (hipStream_t)stream),
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
  ncclResult_t a = g_rccl.AllGather(slab_dev + (size_t)c->rank * block, slab_dev, block, ncclUint8, c->comm,
                                    (hipStream_t)stream);
  ncclResult_t b = g_rccl.AllGather(len_dev + (size_t)c->rank * (size_t)slots_per_rank, len_dev,
                                    (size_t)slots_per_rank * sizeof(uint32_t), ncclUint8, c->comm, (hipStream_t)stream);
  ncclResult_t e = g_rccl.GroupEnd();
  if ((rc = rccl_check(a, "ncclAllGather(slab)")) || (rc = rccl_check(b, "ncclAllGather(lengths)")))
    return rc;
  return rccl_check(e, "ncclGroupEnd");
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

const achip_composite_t *asciichat_hip_grid_composite_dev(const asciichat_hip_grid_t *g) { return g ? g->comp_dev : NULL; }

const achip_composite_t *asciichat_hip_grid_geometry(const asciichat_hip_grid_t *g) { return g ? &g->geom : NULL; }

/* One tick: resize the sources this rank owns into their tile slots, then one all-gather of the tile slots.  After
 * `stream` has passed this point every rank's tiles are current and plans whose frames point at
 * asciichat_hip_grid_composite_dev() render the grid. */
int asciichat_hip_grid_exchange(asciichat_hip_grid_t *g, const uint8_t *const *local_src_dev, void *stream) {
  if (!g || !local_src_dev)
    return achip_fail(ASCIICHAT_HIP_ERR_INVALID_PARAM, "grid_exchange: bad arguments");
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
  free(g);
}
