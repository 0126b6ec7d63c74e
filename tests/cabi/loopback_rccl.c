/*
 * loopback_rccl.c -- TEST INFRASTRUCTURE, never part of the product.
 *
 * A stand-in TRANSPORT with librccl's entry points (the eight that ascii-chat_amd/csrc/comm.c resolves with dlsym), so
 * that comm.c can be driven with a world of TWO ranks on a box that has ONE GPU -- RCCL itself refuses two ranks on the
 * same device ("duplicate GPU").  The ranks are processes on this host; a collective moves the bytes through a POSIX
 * shared-memory segment named by the unique id: device -> segment, barrier, segment -> device, barrier.  What this
 * exercises is everything comm.c decides by itself -- the in-place offsets of the slab / packed all-gathers, the tile
 * slots of uneven source shards (grid_slot_of), the lengths-first protocol -- not RCCL and not xGMI, which stay
 * unmeasured until a multi-GPU node runs tests/test_comm_two_ranks.py's real-RCCL twin.
 *
 * Selected with ASCIICHAT_HIP_RCCL_LIB=<this .so> (comm.c: rccl_load).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <fcntl.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define LB_ID_BYTES 128
#define LB_MAX_WORLD 16
#define LB_CHUNK ((size_t)8 << 20) /* bytes of one rank's slot in the segment */

typedef struct {
  char internal[LB_ID_BYTES];
} lb_unique_id_t;

typedef struct {
  _Atomic uint32_t arrived;    /* ranks inside the current barrier */
  _Atomic uint32_t generation; /* bumped by the last rank to arrive */
  _Atomic uint32_t attached;
  uint32_t world;
} lb_header_t;

struct ncclComm {
  lb_header_t *hdr;
  unsigned char *slots; /* world x LB_CHUNK */
  size_t map_bytes;
  int world, rank;
  char name[LB_ID_BYTES];
};
typedef struct ncclComm *lb_comm_t;

static int lb_barrier(lb_comm_t c) {
  const uint32_t gen = atomic_load(&c->hdr->generation);
  if (atomic_fetch_add(&c->hdr->arrived, 1u) + 1u == (uint32_t)c->world) {
    atomic_store(&c->hdr->arrived, 0u);
    atomic_fetch_add(&c->hdr->generation, 1u);
    return 0;
  }
  struct timespec t0, t;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  while (atomic_load(&c->hdr->generation) == gen) {
    clock_gettime(CLOCK_MONOTONIC, &t);
    if (t.tv_sec - t0.tv_sec > 60)
      return 5; /* ncclInvalidUsage: a peer never arrived (bounded: a test must not hang) */
    usleep(50);
  }
  return 0;
}

int ncclGetUniqueId(lb_unique_id_t *id) {
  static unsigned counter;
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/achip_loopback_%d_%u_%lx", (int)getpid(), counter++, (unsigned long)time(NULL));
  return 0;
}

int ncclCommInitRank(lb_comm_t *out, int world, lb_unique_id_t id, int rank) {
  if (!out || world < 1 || world > LB_MAX_WORLD || rank < 0 || rank >= world || id.internal[0] != '/')
    return 4; /* ncclInvalidArgument */
  lb_comm_t c = (lb_comm_t)calloc(1, sizeof(*c));
  if (!c)
    return 1;
  c->world = world;
  c->rank = rank;
  memcpy(c->name, id.internal, LB_ID_BYTES);
  c->name[LB_ID_BYTES - 1] = 0;
  c->map_bytes = 4096 + (size_t)world * LB_CHUNK;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
    free(c);
    return 2; /* ncclSystemError */
  }
  void *m = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) {
    free(c);
    return 2;
  }
  c->hdr = (lb_header_t *)m; /* a fresh segment reads as zeros */
  c->slots = (unsigned char *)m + 4096;
  atomic_fetch_add(&c->hdr->attached, 1u);
  *out = c;
  return lb_barrier(c); /* like ncclCommInitRank: returns once every rank has joined */
}

int ncclCommCount(lb_comm_t c, int *count) {
  if (!c || !count)
    return 4;
  *count = c->world;
  return 0;
}

int ncclCommDestroy(lb_comm_t c) {
  if (!c)
    return 4;
  if (atomic_fetch_sub(&c->hdr->attached, 1u) == 1u)
    shm_unlink(c->name);
  munmap(c->hdr, c->map_bytes);
  free(c);
  return 0;
}

int ncclGroupStart(void) { return 0; }
int ncclGroupEnd(void) { return 0; }
const char *ncclGetErrorString(int r) { return r == 0 ? "no error" : r == 5 ? "loopback transport: peer missing" : "loopback transport error"; }

/* every rank contributes `count` elements; rank r's land at recv + r*bytes on every rank (send may alias that place) */
int ncclAllGather(const void *send, void *recv, size_t count, int dtype, lb_comm_t c, hipStream_t stream) {
  if (!c || !send || !recv || dtype != 1 /* ncclUint8: all comm.c uses */)
    return 4;
  const size_t bytes = count;
  if (hipStreamSynchronize(stream) != hipSuccess) /* stream order: the producers of `send` have finished */
    return 1;
  for (size_t done = 0; done < bytes || (bytes == 0 && done == 0); done += LB_CHUNK) {
    const size_t n = bytes - done < LB_CHUNK ? bytes - done : LB_CHUNK;
    if (n && hipMemcpy(c->slots + (size_t)c->rank * LB_CHUNK, (const unsigned char *)send + done, n, hipMemcpyDeviceToHost) != hipSuccess)
      return 1;
    int rc = lb_barrier(c);
    if (rc)
      return rc;
    for (int r = 0; r < c->world && n; r++)
      if (hipMemcpy((unsigned char *)recv + (size_t)r * bytes + done, c->slots + (size_t)r * LB_CHUNK, n, hipMemcpyHostToDevice) !=
          hipSuccess)
        return 1;
    rc = lb_barrier(c); /* nobody refills a slot that a peer still reads */
    if (rc)
      return rc;
    if (bytes == 0)
      break;
  }
  return 0;
}
