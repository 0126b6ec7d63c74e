/* frame_table_priv.h -- the frame table's state, shared by frame_table.c (full-frame publishes, the getters) and
 * frame_dense.c (the sampled-image ingest: stage / commit).  Not installed. */
#ifndef ACHIP_FRAME_TABLE_PRIV_H
#define ACHIP_FRAME_TABLE_PRIV_H

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>

#include "achip_types.h"

#define FT_RING 8
#define FT_MAX_READERS 32 /* consumer streams remembered per buffer; beyond that a publish synchronises the device */

typedef struct {
  pthread_mutex_t mu;   /* guards this slot only                                       */
  hipStream_t reader[2][FT_MAX_READERS]; /* streams that were handed buffer k by latest() since its last upload */
  int n_readers[2];
  int readers_overflow[2];
  hipEvent_t reader_done; /* scratch event: "everything enqueued on a reader stream so far" */
  uint8_t *dev[2];      /* frame buffers in HBM                                       */
  size_t cap[2];        /* bytes allocated                                            */
  hipEvent_t ready[2];  /* recorded after the upload of buffer k (single-slot publishes)  */
  unsigned batch_of[2]; /* != 0: buffer k was uploaded by batch publish number batch_of[k]; that batch's event (the
                           table's ring) stands for `ready[k]`, which was not recorded                        */
  uint8_t *stage[2];    /* pinned staging for blobs that are not in the pinned pool   */
  size_t stage_cap[2];
  uint8_t *rows_dev[2]; /* publish_rows: device side of the staged [index table][rows] block */
  size_t rows_cap[2];
  int cur;              /* buffer holding the latest complete frame, -1 = none yet    */
  int w, h;
  uint64_t generation;
  /* sampled-image ingest (frame_dense.c): the latest frame may instead be the W x Hs image one render target samples of
   * it, somewhere in a ring block -- `dense` says which kind is the latest */
  int dense;               /* 1: the latest frame is the sampled image below; the full-frame buffers are stale */
  int dense_blk;           /* ring block it lives in                                                     */
  unsigned dense_seq;      /* number of the commit that put it there (the block's seq at that time)       */
  uint32_t dense_off, dense_bytes;
  achip_frame_t dense_key; /* the target it was gathered for (src unused)                                 */
  achip_frame_t dense_geo; /* that target rewritten onto the sampled image (achip_stage_gather; src unused) */
  int pend;                /* staged in the open block, not committed yet (same four fields)               */
  uint32_t pend_off, pend_bytes;
  achip_frame_t pend_key, pend_geo;
  int pend_w, pend_h;
} ft_slot_t;

#define FT_DENSE_RING 4 /* a sampled image handed out by latest_frames() stays where it is for this many commits - 1 */
typedef struct {
  uint8_t *host, *dev; /* pinned block the host gathers into; its twin in HBM (ONE DMA per commit) */
  size_t cap, dev_cap;
  size_t used;         /* bytes handed out (under dense_mu)                                        */
  hipEvent_t done;     /* the DMA of the commit that filled it                                     */
  unsigned seq;        /* that commit's number, 0 = never filled                                   */
  hipStream_t reader[FT_MAX_READERS]; /* streams handed pointers into the twin since that DMA      */
  int n_readers, readers_overflow;
} ft_dense_blk_t;

struct asciichat_hip_frame_table {
  int n;
  ft_slot_t *slot;
  /* publish_rows_batch: one pinned staging block and its device twin per parity (a tick's block is still being DMA'd
   * while the next tick's is filled), guarded by batch_mu */
  pthread_mutex_t batch_mu;
  uint8_t *batch_host[2], *batch_dev[2];
  size_t batch_cap[2];
  hipEvent_t batch_done[2];
  unsigned batch_no;
  /* completion of batch publishes: ONE event per batch (not one per slot: 256 hipEventRecord calls and then 256
   * hipStreamWaitEvent calls per tick were most of a tick's host time).  A ring: slot B % FT_RING holds batch B's event
   * while ring_seq says so; before the slot is re-recorded for batch B + FT_RING the host waits for batch B, so a
   * buffer whose batch is no longer in the ring is known to be complete. */
  pthread_mutex_t ev_mu;
  hipEvent_t ring[FT_RING];
  unsigned ring_seq[FT_RING];
  /* sampled-image ingest */
  pthread_mutex_t dense_mu; /* opening / committing a block, its reader list */
  ft_dense_blk_t dense[FT_DENSE_RING];
  int dense_open;           /* ring index of the block being filled, -1 = none */
  int dense_next;           /* the block the next tick opens                   */
  unsigned dense_seq;
  size_t dense_want;        /* capacity the next block starts out with (a block that had to grow says so) */
  int dense_inflight;       /* stage() calls gathering right now (atomic)      */
  int dense_zero_copy;      /* ASCIICHAT_HIP_INGEST_ZERO_COPY=1: renders read the pinned block itself, no DMA, no twin */
};


/* frame_dense.c */
void ft_dense_init(struct asciichat_hip_frame_table *t);
void ft_dense_destroy(struct asciichat_hip_frame_table *t);
void ft_dense_forget_stream(struct asciichat_hip_frame_table *t, hipStream_t s);
/* what latest_frames() copies out of a slot (under its lock) whose latest frame is a sampled image */
typedef struct {
  int dense_blk;
  unsigned dense_seq; /* the block's commit number when the snapshot was taken: a block refilled since then is stale */
  uint32_t dense_off;
  achip_frame_t dense_key, dense_geo;
} ft_dense_ref_t;
/* rewrites *f onto that sampled image when f asks for what was staged (or is last tick's rewritten descriptor); *waited:
 * bit r set = the consumer stream already waits for ring block r.  1 = source handed out, 0 = no match or *rc != 0,
 * -1 = the block was refilled after the snapshot was taken (two commits slipped in): look at the slot again */
int ft_dense_latest(struct asciichat_hip_frame_table *t, const ft_dense_ref_t *s, void *consumer_stream, achip_frame_t *f,
                    unsigned *waited, int *rc);

#endif
