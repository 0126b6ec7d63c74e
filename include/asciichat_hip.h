/*
 * asciichat_hip.h -- C-ABI of libasciichat_hip.so, the MI355X-native image->ASCII render path.
 *
 * Two layers are exported by the same shared object:
 *
 *  (1) the DROP-IN layer, declared in asciichat_render.h: the reference's own libasciichat render
 *      entry points (ascii_convert, ascii_convert_with_capabilities, image_print_*, rgb_to_*_halfblocks_*,
 *      ascii_create_grid, image_*, buffer_pool_*) with identical C signatures, ownership and NULL/error
 *      behaviour.  Each call runs a 1-frame batch on the GPU.
 *
 *  (2) the BATCH layer below (additive, not in the reference): a plan object renders N independent
 *      frames per launch from device-resident RGB24 into an HBM slab.  This is what a server with many
 *      clients binds instead of calling ascii_convert_with_capabilities once per client per tick
 *      (reference call site: src/server/stream.c:841, one call per client render thread).
 *
 * Plain C, plain pointers and sizes; `stream` arguments are hipStream_t passed as void* (NULL = the
 * null stream).  All functions return an asciichat_error_t-compatible int: 0 = ASCIICHAT_OK
 * (include/ascii-chat/common/error_codes.h:51-109 in the reference).
 *
 * The library has no CPU fallback: without a visible gfx950 device every compute entry point fails
 * with ASCIICHAT_HIP_ERR_NO_DEVICE and prints the reason to stderr.
 */
#ifndef ASCIICHAT_HIP_H
#define ASCIICHAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "achip_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* error codes shared with the drop-in layer (values = the reference's asciichat_error_t) */
enum {
  ASCIICHAT_HIP_OK = 0,
  ASCIICHAT_HIP_ERR_MEMORY = 3,         /* ERROR_MEMORY        */
  ASCIICHAT_HIP_ERR_NOT_SUPPORTED = 30, /* ERROR_NOT_SUPPORTED */
  ASCIICHAT_HIP_ERR_BUFFER = 81,        /* ERROR_BUFFER        */
  ASCIICHAT_HIP_ERR_INVALID_STATE = 85, /* ERROR_INVALID_STATE */
  ASCIICHAT_HIP_ERR_INVALID_PARAM = 86, /* ERROR_INVALID_PARAM */
  ASCIICHAT_HIP_ERR_NO_DEVICE = 200,    /* no HIP device / HIP runtime failure (new) */
};

typedef struct asciichat_hip_plan asciichat_hip_plan_t;

/* Number of visible HIP devices (0 when there is none); never fails. */
int asciichat_hip_device_count(void);

/* Last error message of the calling thread ("" when none). */
const char *asciichat_hip_last_error(void);

/*
 * Create a plan for `n_frames` frames rendered in `mode` (ACHIP_MODE_*, achip_types.h) with the glyph
 * palette `palette_chars` (UTF-8; what the reference passes as palette_chars / client_palette_chars).
 * `frames` is a HOST array; the src/comp pointers inside must be device-visible.  Use
 * achip_frame_setup()/achip_frame_identity() (achip_host.h, also exported) to fill descriptors exactly
 * as ascii_convert_with_capabilities sizes them.
 */
int asciichat_hip_plan_create(asciichat_hip_plan_t **plan, int mode, const char *palette_chars,
                              const achip_frame_t *frames, int n_frames);

/* Replace the frame descriptors (e.g. new source pointers for the next tick); same n_frames. */
int asciichat_hip_plan_update(asciichat_hip_plan_t *plan, const achip_frame_t *frames, void *stream);

/* Bytes each frame needs in the output slab (worst case incl. NUL, a multiple of 128: slots of a line-aligned slab start on lines; any multiple of 16 >= this may be passed to the render calls). */
size_t asciichat_hip_plan_out_stride(const asciichat_hip_plan_t *plan);

/* Kernel geometry: -1 = automatic (achip_choose_geometry: by mode, row width, frames per CU of the plan's share and the kind
 * of source), else a variant id from render_variants.h; NOT_SUPPORTED when the geometry cannot carry the plan (rows geometries
 * 26 / 27 / 29 / 31, for one, have no general sampler: no composites, no 1 x 1 sources; 24-26 hold rows of at most 448 cells,
 * 27 / 29 cut rows of up to 4096 / 2560 cells into segments) or is not in this build. */
int asciichat_hip_plan_set_variant(asciichat_hip_plan_t *plan, int variant);
int asciichat_hip_plan_get_variant(const asciichat_hip_plan_t *plan);

/* Multi-workgroup frames: 0 = automatic (small batches are shared out over several workgroups so that more of the GPU works
 * on them: a frame's blocks over four-wave workgroups of the stream kernel for the per-cell modes, variant 18, and of the rows
 * kernel for the run-structured modes while a row is at most 128 cells, variant 31; row bands of the phase kernel otherwise),
 * < 0 = never, > 0 = row bands of this many text rows per workgroup.  get_parts() reports
 * workgroups per frame.  The wire-stage entry points (render_crc, render_packets*, *_packed) of a shared-out plan launch its
 * whole-frame geometry instead: a frame's checksum and its exact-length image belong to one workgroup. */
int asciichat_hip_plan_set_split(asciichat_hip_plan_t *plan, int rows_per_part);
/* Pipelining hint.  One launch is a gather burst (HBM-bound) followed by token work (latency-bound, HBM idle); a caller
 * that keeps `launches_in_flight` independent batches in flight on separate streams -- the reference's model: one render
 * thread per client (src/server/render.c:1233) -- lets those phases overlap, and the plan then picks the geometry
 * for its share of the GPU (1080p->80x24, 256 frames per launch: 13.7 us per launch alone, 8.8 us with three in
 * flight; profiles/r01_overlap.txt).  Default 1.  Launches of ONE plan must still be ordered (one stream). */
int asciichat_hip_plan_set_concurrency(asciichat_hip_plan_t *plan, int launches_in_flight);
/* A plan whose descriptors differ only by a constant source pitch (equally sized client frames in one slab; or one
 * frame) passes the common descriptor in the kernel arguments instead of having every workgroup fetch its own --
 * one memory round trip less in front of the first gather.  Detected at create / update; on by default.
 * set_uniform(plan, 0) forces the descriptor array (A/B measurements); get_uniform: 1 when the fast path is used. */
int asciichat_hip_plan_set_uniform(asciichat_hip_plan_t *plan, int allow);
int asciichat_hip_plan_get_uniform(const asciichat_hip_plan_t *plan);
int asciichat_hip_plan_get_parts(const asciichat_hip_plan_t *plan);

/*
 * Render all frames: frame i's bytes go to out_dev + i*out_stride (16-byte aligned base, stride a
 * multiple of 16, >= plan_out_stride), its length to out_len_dev[i] (ACHIP_LEN_OVERFLOW /
 * ACHIP_LEN_BADDESC on error).  Asynchronous on `stream`; inputs must already be resident.
 */
int asciichat_hip_plan_render(asciichat_hip_plan_t *plan, uint8_t *out_dev, size_t out_stride, uint32_t *out_len_dev,
                              void *stream);

/* Render only frames [first, first+count) of the plan (multi-GPU sharding of one logical batch). */
int asciichat_hip_plan_render_range(asciichat_hip_plan_t *plan, int first, int count, uint8_t *out_dev,
                                    size_t out_stride, uint32_t *out_len_dev, void *stream);

/* Diagnostics: as plan_render, and additionally fills phase_cycles_dev.  Phase-kernel geometries (variants 0-4):
 * phase_cycles_dev[frame*8 + k] = shader-clock cycles spent per kernel phase (0 setup, 1 gather, 2 heads, 3 lengths,
 * 4 scan, 5 emit tokens, 6 drain to HBM, 7 total).  Stream geometries (16-19): eight 100 MHz wall-clock stamps per WAVE of
 * every workgroup, phase_cycles_dev[((frame * parts + part) * waves + wave) * 8 + k] (render_stream.hpp) -- the buffer must
 * hold n_frames * plan_get_parts() * (workgroup size / 64) * 8 words.  Rows geometries (24, 25, 26) write nothing. */
int asciichat_hip_plan_render_profiled(asciichat_hip_plan_t *plan, uint8_t *out_dev, size_t out_stride,
                                       uint32_t *out_len_dev, unsigned long long *phase_cycles_dev, void *stream);

/*
 * Issue n_steps launches from C: step k (k = first_step .. first_step + n_steps - 1) renders plans[k % n_plans] on
 * streams[k % n_streams] into out_dev[k % n_streams] / out_len_dev[k % n_streams].  n_plans must be a multiple of
 * n_streams so that a plan always runs on the same stream (launches of one plan must be ordered).  This is the
 * tick loop of a server with several independent client batches -- one FFI call instead of one per launch.
 * Asynchronous; asciichat_hip_streams_wait() spins (no driver sleep) until the given streams have drained.
 */
int asciichat_hip_render_many(asciichat_hip_plan_t *const *plans, int n_plans, uint8_t *const *out_dev,
                              uint32_t *const *out_len_dev, size_t out_stride, void *const *streams, int n_streams,
                              int first_step, int n_steps);
/* diagnostics: the same loop, step k writing the kernels' per-wave timestamps to prof_dev + k * prof_stride_words */
int asciichat_hip_render_many_profiled(asciichat_hip_plan_t *const *plans, int n_plans, uint8_t *const *out_dev,
                                       uint32_t *const *out_len_dev, size_t out_stride, void *const *streams,
                                       int n_streams, int first_step, int n_steps, unsigned long long *prof_dev,
                                       size_t prof_stride_words);
int asciichat_hip_streams_wait(void *const *streams, int n_streams);

/*
 * The same tick loop captured ONCE into a HIP graph with n_lanes parallel branches (lane l renders into out_dev[l] /
 * out_len_dev[l], step k runs on lane k % n_lanes) and replayed with one graph launch on `stream`.  For a server whose
 * set of client batches is stable from tick to tick this removes the per-launch host cost and the wake-up of n_lanes
 * idle queues that a short burst of launches pays.  Row-band plans cannot be captured (their hand-off words carry a
 * per-launch epoch): batches of >= 3/4 frame per CU, or set_split(plan, -1); small plans of the per-cell modes, which
 * plan_render shares out over several workgroups, are captured in their whole-frame geometry.  Plans must not be updated
 * or rendered elsewhere while a replay is in flight.
 */
typedef struct asciichat_hip_schedule asciichat_hip_schedule_t;
int asciichat_hip_schedule_create(asciichat_hip_schedule_t **sched, asciichat_hip_plan_t *const *plans, int n_plans,
                                  uint8_t *const *out_dev, uint32_t *const *out_len_dev, size_t out_stride, int n_lanes,
                                  int first_step, int n_steps);
int asciichat_hip_schedule_launch(asciichat_hip_schedule_t *sched, void *stream);
void asciichat_hip_schedule_destroy(asciichat_hip_schedule_t *sched);

void asciichat_hip_plan_destroy(asciichat_hip_plan_t *plan);

/*
 * Drop-in layer: coalescing of concurrent calls (combine.c).  The reference's server calls
 * ascii_convert_with_capabilities from one render thread per client; from `n` calls in flight on, those calls share
 * launches (flat combining: one upload, one kernel per (mode, palette) group, one wait per generation of callers).
 * n = 0 never, 1 always, default 6 (with hysteresis: off again below 3) -- below that every call launching on its own thread's stream is as fast or faster
 * (profiles/r03_dropin_threads.txt).  Also settable with the environment variable ASCIICHAT_HIP_COALESCE.  Returns the
 * previous setting.
 */
int asciichat_hip_set_coalesce_min_callers(int n);

/* image_resize on device memory: nearest-neighbour, lib/video/rgba/image.c:267-328 */
int asciichat_hip_resize(const uint8_t *src_dev, int src_w, int src_h, uint8_t *dst_dev, int dst_w, int dst_h,
                         void *stream);

/* Materialise the W x 2H pixel-space grid composite (src/server/stream.c:664-779) on device.
 * comp_host is filled by achip_composite_setup(); dst_dev holds canvas_w*canvas_h*3 bytes. */
int asciichat_hip_composite(const achip_composite_t *comp_host, uint8_t *dst_dev, void *stream);

/* The client display path's full-frame passes on DEVICE images (RGB24): apply_color_filter
 * (lib/video/rgba/color_filter.c:274-345; color_filter = the reference's color_filter_t 0..11, in place) and the
 * x/y flips of src/common/session/display.c:546-600 (out of place).  The render path does not need them --
 * achip_frame_set_display_ops() folds both into its sampler -- they exist for callers that want the image. */
int asciichat_hip_apply_color_filter(uint8_t *pixels_dev, int width, int height, int stride, int color_filter,
                                     void *stream);
int asciichat_hip_image_flip(const uint8_t *src_dev, uint8_t *dst_dev, int width, int height, int flip_x, int flip_y,
                             void *stream);

/*
 * Ingest: a device-resident table of every client's latest camera frame, replacing the two
 * full-frame host copies that each render thread makes of each client's frame per tick
 * (collect_video_sources, src/server/stream.c:221-463).
 *   publish   validates the blob [u32 BE width][u32 BE height][RGB24] like collect_video_sources does
 *             (achip_frame_blob_parse, exact = false) and uploads it on `stream`.  A blob inside the pinned
 *             buffer pool is DMA'd in place and must stay valid until `stream` has passed the copy; any other
 *             blob is copied to pinned staging first and may be released when publish returns.
 *   latest    device pointer + size of the newest published frame of a slot (NULL / 0 while the client has
 *             sent none: has_video = false) and makes consumer_stream wait for its upload.  The table remembers that
 *             consumer_stream was handed this buffer: the upload that will overwrite it (the publish after next on
 *             the slot) is ordered behind everything enqueued on consumer_stream up to that publish.  The pointer is
 *             therefore good for work enqueued before the publish after next; call latest() again every tick.
 *   forget_stream  REQUIRED before destroying a stream that was passed to latest(): the table records events on the streams
 *             it remembers, and a destroyed handle (or one recycled by a later stream) must not be among them
 */
typedef struct asciichat_hip_frame_table asciichat_hip_frame_table_t;
int asciichat_hip_frame_table_create(asciichat_hip_frame_table_t **table, int n_slots);
void asciichat_hip_frame_table_destroy(asciichat_hip_frame_table_t *table);
int asciichat_hip_frame_table_publish(asciichat_hip_frame_table_t *table, int slot, const void *blob, size_t blob_size,
                                      void *stream);
/* publish moving only the source rows that renders described by `targets` (their out_h, y_ratio, src_h and the FLIP_Y
 * bit of ops; src is ignored; targets for sources of another height are passed over, at least one must fit) will sample: 24 of 1080 rows for an 80x24 target, 138 KB instead of 6.2 MB over PCIe.  The
 * buffer keeps the full frame's layout, so descriptors, plans and output bytes are those of a full publish for such
 * renders; rows nobody named are stale. */
int asciichat_hip_frame_table_publish_rows(asciichat_hip_frame_table_t *table, int slot, const void *blob, size_t blob_size,
                                           const achip_frame_t *targets, int n_targets, void *stream);
/* ... for a whole tick's clients at once: ONE packed block, ONE DMA, ONE launch instead of one of each per client (slots
 * distinct).  `targets` may simply be the tick's render descriptors: a blob is matched with the targets set up for sources
 * of its height (of its size, for the columns); a blob none of them describes is refused.  Per slot the semantics of publish_rows; when the targets sample
 * at most half of a frame's columns (x_ratio, src_w, out_w and the FLIP_X bit) only the sampled PIXELS are staged -- 5.6 KB
 * of a 1080p frame for an 80x24 target -- and pixels nobody named are stale like rows nobody named. */
int asciichat_hip_frame_table_publish_rows_batch(asciichat_hip_frame_table_t *table, const int *slots, const void *const *blobs,
                                                 const size_t *blob_sizes, int n, const achip_frame_t *targets, int n_targets,
                                                 void *stream);
/*
 * Ingest of SAMPLED IMAGES (frame_dense.c): the table keeps, per slot, the W x Hs pixels ONE render target samples of the
 * client's frame (image.c:293-325: 1 920 of a 1080p frame's 2 073 600 for an 80x24 target), in target raster order with
 * the flips folded in, and latest_frames() points the descriptor at that image (src_w x src_h = sampled size, ratios 1.0:
 * (x * 65536) >> 16 == x) -- no scatter back to full-frame positions, and the render's gather is a dense read.
 *   stage    any thread, e.g. each receive thread for the blob it holds (the replacement of collect_video_sources' two
 *            copies, src/server/stream.c:221-463): validates the blob like publish and gathers what `target` (src_w x src_h
 *            = the blob's size, no composite; src ignored) samples of it into the tick's pinned block.  ERR_INVALID_PARAM
 *            when the target takes the frame as it is (nothing to compact: publish the blob).  The block grows as needed.
 *            A slot staged twice in a tick keeps the later frame.
 *   commit   once per tick, after every stage() of the tick has returned: ONE DMA of the block into HBM on `stream`, no
 *            kernel.  The staged slots' latest frame is now the sampled image; a tick without stage() is a no-op.
 *   publish_sampled_batch = stage for n blobs on the library's own pool of ingest threads (the caller's included;
 *            ASCIICHAT_HIP_INGEST_THREADS, default half of the CPUs the process may keep busy, at most 8) + commit.
 *            n_targets == 1 (every blob is rendered to the same target) or == n (targets[i] belongs to blobs[i]).
 * Getter: frame_table_latest_frames() -- a descriptor that asks for the staged target (the sampling fields: src_w, src_h,
 * out_w, out_h, ratios, flips), or that latest_frames() itself rewrote a tick ago, is rewritten onto the image; padding,
 * tints and the other ops stay the caller's; any other descriptor gets src = NULL (stage that target).  A pointer handed
 * out stays good for work enqueued before the THIRD commit after the one that uploaded it; frame_table_latest() has no
 * full frame to return for such a slot and fails.  ASCIICHAT_HIP_INGEST_ZERO_COPY=1: no DMA and no twin, renders read
 * the mapped pinned block itself.
 */
int asciichat_hip_frame_table_stage(asciichat_hip_frame_table_t *table, int slot, const void *blob, size_t blob_size,
                                    const achip_frame_t *target);
int asciichat_hip_frame_table_commit(asciichat_hip_frame_table_t *table, void *stream);
int asciichat_hip_frame_table_publish_sampled_batch(asciichat_hip_frame_table_t *table, const int *slots,
                                                    const void *const *blobs, const size_t *blob_sizes, int n,
                                                    const achip_frame_t *targets, int n_targets, void *stream);
int asciichat_hip_ingest_threads(void); /* threads publish_sampled_batch gathers on (starts the pool) */
int asciichat_hip_frame_table_latest(asciichat_hip_frame_table_t *table, int slot, void *consumer_stream,
                                     const uint8_t **pixels_dev, int *width, int *height, uint64_t *generation);
/* a tick's latest frames straight into the render descriptors: frames[i].src = the device frame of slots[i] when it has
 * one of the geometry the descriptor was set up for (src_w x src_h), else NULL (has_video = false, or a new resolution:
 * set the descriptor up again).  Returns the number of descriptors that got a source, or -(error code). */
int asciichat_hip_frame_table_latest_frames(asciichat_hip_frame_table_t *table, const int *slots, int n,
                                            void *consumer_stream, achip_frame_t *frames);
void asciichat_hip_frame_table_forget_stream(asciichat_hip_frame_table_t *table, void *consumer_stream);

/*
 * Wire stage after render, on DEVICE buffers, for the frames of a slab (frame i at
 * base_dev + i*stride, len_dev[i] bytes; or fixed_len bytes each when len_dev == NULL; max_len bounds every
 * length -- pass the slab stride for a plan's output):
 *   crc_out_dev[i]   asciichat_crc32(frame i) -- CRC-32C, lib/network/crc32.c:95-190
 *   hdr_out_dev      n x 24 bytes: the ascii_frame_packet_t that acip_send_ascii_frame builds
 *                    (lib/network/acip/server.c:186-214; include/ascii-chat/network/packet/packet.h:847-862):
 *                    {width, height, original_size = len, compressed_size = 0, checksum, flags = 0}, network
 *                    byte order; dims_dev = n x {width, height} (uint32, host byte order)
 *   packet_crc_out_dev[i]  CRC-32C of header || frame, the packet_header_t.crc32 that
 *                    packet_send_via_transport computes over the payload (lib/network/acip/send.c:59-69)
 * hdr_out_dev / dims_dev / packet_crc_out_dev may be NULL.  A frame whose length is a render error code
 * (>= 0xFFFFFFF0) gets CRC 0.  base_dev and stride must be 16-byte aligned.
 */
int asciichat_hip_crc32c(const uint8_t *base_dev, size_t stride, const uint32_t *len_dev, uint32_t fixed_len,
                         uint32_t max_len, int n, uint32_t *crc_out_dev, void *stream);
int asciichat_hip_frame_packets(const uint8_t *base_dev, size_t stride, const uint32_t *len_dev, uint32_t max_len, int n,
                                const uint32_t *dims_dev, uint32_t *crc_out_dev, uint8_t *hdr_out_dev,
                                uint32_t *packet_crc_out_dev, void *stream);
/* ... and, in the same pass over the slab, the frames at their exact lengths: frame i also goes to dst + off_out[i]
 * (asciichat_hip_pack_frames' layout below: 16-byte aligned starts, off_out[n] = bytes used, dst may be the device alias of
 * mapped pinned host memory).  What a send thread needs of a tick -- checksums, headers, frame bytes -- for one read of
 * the slab instead of two. */
int asciichat_hip_frame_packets_packed(const uint8_t *base_dev, size_t stride, const uint32_t *len_dev, uint32_t max_len, int n,
                                       const uint32_t *dims_dev, uint32_t *crc_out_dev, uint8_t *hdr_out_dev,
                                       uint32_t *packet_crc_out_dev, uint8_t *dst, size_t dst_capacity, uint64_t *off_out,
                                       uint32_t *len_out, void *stream);

/*
 * Compacted output (SURVEY.md 8e "prefer gathering compacted per-rank buffers ... lengths first").  A render leaves frame
 * i at slab + i*stride, stride = the worst case (44.5 KB for 80x24 truecolor; real video is 2-4 KB per frame).  What
 * crosses PCIe or xGMI should be the bytes in use -- the reference ships exactly frame_size bytes per client
 * (lib/network/acip/server.c:190-222).  pack_frames copies frame i to dst + off[i], off[i] = sum over j < i of
 * round16(len[j]) (frame starts stay 16-byte aligned: <= 15 bytes of padding per frame); frames whose length is a render
 * error code take no room.  off_out (n + 1 entries; [n] = total bytes) and len_out (n, copy of the lengths) may be NULL.
 * dst, off_out and len_out may be device memory or the device alias of mapped pinned host memory (host_alloc below): the
 * kernel's stores are then the transfer itself -- exact length, no second DMA, no host round trip to learn a size.  Frames
 * travel in whole 16-byte groups: a frame whose last group (round16(len) bytes from its start) would end beyond
 * dst_capacity is not copied, and nothing is ever stored at or behind dst + dst_capacity, whatever the capacity's
 * alignment (off_out[n] > dst_capacity tells).  plan_render_packed = plan_render + pack_frames on the same stream.
 */
int asciichat_hip_pack_frames(const uint8_t *slab_dev, size_t stride, const uint32_t *len_dev, int n, uint8_t *dst,
                              size_t dst_capacity, uint64_t *off_out, uint32_t *len_out, void *stream);
int asciichat_hip_plan_render_packed(asciichat_hip_plan_t *plan, uint8_t *slab_dev, size_t out_stride,
                                     uint32_t *out_len_dev, uint8_t *dst, size_t dst_capacity, uint64_t *off_out,
                                     uint32_t *len_out, void *stream);
int asciichat_hip_host_alloc(size_t bytes, void **host_ptr, void **device_alias);
void asciichat_hip_host_free(void *host_ptr);

/*
 * Multi-GPU (comm.c): one process per GPU, RCCL over xGMI (librccl is dlopen'ed on first use).
 * Frames are independent, so the render path needs no collective -- a batch is partitioned over the ranks
 * (achip_shard_bounds, balanced and contiguous) and every rank renders its block.  Collectives exist where a consumer
 * needs remote data:
 *   comm_all_gather_slab   in-place all-gather of a sharded output slab + its lengths (one group per batch):
 *                          rank r rendered its frames into slots [r*slots, (r+1)*slots), slots = achip_shard_slots();
 *   grid_*                 BASELINE config 4, the server's pixel-space grid (create_multi_source_composite +
 *                          convert_composite_to_ascii, src/server/stream.c:664-854): grid_exchange resizes the sources
 *                          this rank owns into their composite tiles and all-gathers the tiles (one ncclAllGather of
 *                          <= 2 x 4.8 KB per rank for nine 1080p sources at 160x48); every rank then renders the grid
 *                          for its own target clients from plans whose frames point at grid_composite_dev().
 * The 128-byte unique id goes from rank 0 to the others out of band (the server's control plane).  comm == NULL
 * means a single GPU everywhere below.
 */
#define ASCIICHAT_HIP_COMM_ID_BYTES 128
#define ASCIICHAT_HIP_GRID_MAX_SOURCES 64 /* clients the layout counts; the first nine with video are placed (stream.c:687) */
typedef struct asciichat_hip_comm asciichat_hip_comm_t;
typedef struct asciichat_hip_grid asciichat_hip_grid_t;
int asciichat_hip_comm_unique_id(void *id_out, size_t id_bytes);
int asciichat_hip_comm_init(asciichat_hip_comm_t **comm, int world, int rank, const void *id, size_t id_bytes);
int asciichat_hip_comm_world(const asciichat_hip_comm_t *comm);
int asciichat_hip_comm_rank(const asciichat_hip_comm_t *comm);
int asciichat_hip_comm_count(const asciichat_hip_comm_t *comm); /* ncclCommCount of the communicator; -1 on error */
void asciichat_hip_comm_destroy(asciichat_hip_comm_t *comm);
int asciichat_hip_comm_all_gather(asciichat_hip_comm_t *comm, const void *send_dev, void *recv_dev, size_t bytes_per_rank,
                                  void *stream);
int asciichat_hip_comm_all_gather_slab(asciichat_hip_comm_t *comm, uint8_t *slab_dev, size_t stride, uint32_t *len_dev,
                                       int slots_per_rank, void *stream);
/* The same exchange moving only the bytes in use: lengths first (one small in-place all-gather + a host read of them),
 * then every rank packs its block (pack_frames) and ONE all-gather of max-over-ranks packed bytes moves the blocks.
 *   slab_dev / len_dev   as for all_gather_slab (this rank's block rendered at slots [rank*slots, ..)); len_dev holds
 *                        all world*slots lengths afterwards, the slab is NOT gathered
 *   packed_dev           world * packed_capacity_per_rank bytes; rank r's packed block arrives at r * (*block_bytes)
 *   off_host             world*slots entries (host): byte offset in packed_dev of every frame after the gather
 *   len_host             world*slots entries (host, may be NULL): the gathered lengths
 *   block_bytes          bytes every rank contributed = what crossed the links per rank (max over ranks, multiple of 16)
 * Synchronises `stream` once (the host must know the sizes).  ERR_BUFFER when a block exceeds the capacity -- which must
 * be the SAME on every rank: all of them then fail alike before the second collective; a failure local to one rank
 * (its pack launch) is reported after that rank has joined the exchange, so no rank is left waiting inside RCCL. */
int asciichat_hip_comm_all_gather_packed(asciichat_hip_comm_t *comm, const uint8_t *slab_dev, size_t stride,
                                         uint32_t *len_dev, int slots_per_rank, uint8_t *packed_dev,
                                         size_t packed_capacity_per_rank, uint64_t *off_host, uint32_t *len_host,
                                         size_t *block_bytes, void *stream);
/* Both forms behind one entry, picked per call (form 0 = packed, 1 = slab) or by the environment (form -1:
 * ASCIICHAT_HIP_GATHER=packed|slab, default packed): the frames are at *base_out + off_host[i] afterwards (packed_dev or the
 * slab itself), len_host is filled by the packed form only (the slab form never visits the host: lengths stay in len_dev),
 * *form_out says which form ran.  For the first A/B on a real multi-GPU node (bench.py --gpus N prints both). */
int asciichat_hip_comm_all_gather_frames(asciichat_hip_comm_t *comm, int form, uint8_t *slab_dev, size_t stride, uint32_t *len_dev,
                                         int slots_per_rank, uint8_t *packed_dev, size_t packed_capacity_per_rank,
                                         const uint8_t **base_out, uint64_t *off_host, uint32_t *len_host, size_t *block_bytes,
                                         int *form_out, void *stream);
/* partition of n independent items: rank's [first, first+count); owner of an item; slots every rank reserves */
void achip_shard_bounds(int n_items, int world, int rank, int *first, int *count);
int achip_shard_owner(int n_items, int world, int item);
int achip_shard_slots(int n_items, int world);
/* has_video: NULL = every source has video, else n_src flags (a client without video takes no cell) */
int asciichat_hip_grid_create(asciichat_hip_grid_t **grid, asciichat_hip_comm_t *comm, const int *src_w, const int *src_h,
                              const unsigned char *has_video, int n_src, int term_w, int term_h);
int asciichat_hip_grid_owner(const asciichat_hip_grid_t *grid, int source); /* rank that must pass this source's pixels */
/* local_src_dev[k]: device pixels of source k (RGB24, tightly packed) for the sources this rank owns; others ignored */
int asciichat_hip_grid_exchange(asciichat_hip_grid_t *grid, const uint8_t *const *local_src_dev, void *stream);
const achip_composite_t *asciichat_hip_grid_composite_dev(const asciichat_hip_grid_t *grid); /* for achip_frame_t.comp */
const achip_composite_t *asciichat_hip_grid_geometry(const asciichat_hip_grid_t *grid);      /* host copy: canvas size .. */
/* One GPU (world == 1) only.  on: no tiles, no resize launch, no collective -- plans render straight from the clients'
 * frames and grid_exchange only refreshes the (<= 9) source pointers of the descriptor (a one-wave launch, or nothing
 * when they did not change): the whole tick is the render launch.  Pays while the target clients are few (every target's
 * workgroups then gather from the full-size frames; with hundreds of targets the resized tiles are the better source).
 * Changes what grid_composite_dev() returns -- fetch it again after switching.  ERR_NOT_SUPPORTED when world > 1. */
int asciichat_hip_grid_set_direct(asciichat_hip_grid_t *grid, int on);
void asciichat_hip_grid_destroy(asciichat_hip_grid_t *grid);

/*
 * Render + checksum in one launch: crc_out_dev[i] = asciichat_crc32(frame i) (0 for a frame that did not fit its slot).
 * For whole-frame launches of the per-cell modes (plan_has_fused_crc() == 1) the CRC rides the render kernel's drain --
 * the frame's bytes are checksummed while they sit in LDS on their way out -- instead of a second pass over the slab;
 * any other plan renders and then runs asciichat_hip_crc32c, with the same results.  packets_from_crc() then builds the
 * 24-byte ascii_frame_packet_t headers and the header || frame CRCs from lengths + frame CRCs (one thread per frame),
 * the rest of what asciichat_hip_frame_packets does.
 */
int asciichat_hip_plan_render_crc(asciichat_hip_plan_t *plan, uint8_t *out_dev, size_t out_stride, uint32_t *out_len_dev,
                                  uint32_t *crc_out_dev, void *stream);
/* ... and the whole wire stage with the render: one launch where plan_has_fused_crc(), render + frame_packets otherwise.
 * dims_dev ({width, height} per frame) may be NULL (zeros in the headers); packet_crc_out_dev may be NULL.
 * The wire entry points (plan_render_crc / _packets / _packets_packed / _packed) of ONE plan must be called on ONE stream:
 * the plan owns the span registers, arrival counters and the pack cursor they use between launches. */
int asciichat_hip_plan_render_packets(asciichat_hip_plan_t *plan, uint8_t *out_dev, size_t out_stride,
                                      uint32_t *out_len_dev, const uint32_t *dims_dev, uint32_t *crc_out_dev,
                                      uint8_t *hdr_out_dev, uint32_t *packet_crc_out_dev, void *stream);
/* ... plus the compaction.  Three forms, chosen by the plan:
 *   (1) ONE launch that writes the frames at their exact lengths itself -- whole-frame plans of the per-cell foreground
 *       modes (truecolor with an all-ASCII palette, 256, 16 colours; single sources) whose slab stride is at most 48 KB
 *       (asciichat_hip_plan_get_exact_length() != 0; 1080p -> 80x24 truecolor is 36 KB per frame) AND, by default, only
 *       for a destination in device memory (into mapped host memory form (2) is the faster one: a workgroup of (1) holds
 *       its CU until its stores have crossed PCIe).  The slab is NOT written
 *       (slab_dev may be NULL); frames lie in dst back to back in the order in which they FINISH, every start 16-byte
 *       aligned: off_out[i] says where frame i went, off_out[n] the total.  off_out is the only way to find a frame of
 *       this form, so a call WITHOUT off_out never takes it: it keeps forms (2) / (3), whose layout pack_frames documents
 *       (frame i behind the 16-byte rounded lengths of frames 0..i-1), whatever set_exact_length says.  The
 *       padding behind a frame is zeros.  asciichat_hip_plan_set_exact_length(plan, 0) keeps such a plan on (2).
 *   (2) render (+ fused wire stage) and pack_frames where the render kernel carries the CRC,
 *   (3) render and ONE pass that checksums and packs otherwise (plans of the run-structured modes, row bands);
 *       in (2) and (3) frame i lies at off_out[i] = sum over j < i of round16(len[j]) and the slab holds the frames too.
 * asciichat_hip_plan_render_packed (no wire stage) follows the same rule. */
int asciichat_hip_plan_get_exact_length(const asciichat_hip_plan_t *plan);
int asciichat_hip_plan_set_exact_length(asciichat_hip_plan_t *plan, int mode); /* -1 where it is the faster form (default:
                                                                                   destinations in device memory), 0 never,
                                                                                   1 wherever the plan qualifies */
int asciichat_hip_plan_render_packets_packed(asciichat_hip_plan_t *plan, uint8_t *slab_dev, size_t out_stride,
                                             uint32_t *out_len_dev, const uint32_t *dims_dev, uint32_t *crc_out_dev,
                                             uint8_t *hdr_out_dev, uint32_t *packet_crc_out_dev, uint8_t *dst,
                                             size_t dst_capacity, uint64_t *off_out, uint32_t *len_out, void *stream);
/* Exact-length frames beyond the 48 KB of the one-launch form above: LENGTH-FIRST -- the stream kernel's loop run twice,
 * lengths first, then the emission at the place the frame claimed (whole-frame plans of truecolor foreground with an all-ASCII
 * palette; ASCIICHAT_HIP_ERR_NOT_SUPPORTED otherwise).  Frames land in completion order (off_out[i], 16-byte aligned;
 * off_out[n] = total), lengths in out_len_dev / len_out.  plan_render_packed / plan_render_packets_packed take this form by
 * themselves for plans whose sources are the sampled images (ratio 1.0: 16.5 us against 34.5 for render + pack pass per 256
 * frames of 200x60) or at most 1920 pixels wide (128 frames of 1080p -> 320x90: 94.6 us against 144.4; from 4K sources the second
 * gather costs more than the pass -- profiles/r06_length_first_ab.txt, r06_wire_audit.txt), device destinations, whole-frame
 * launches; plan_set_exact_length(1) forces it wherever it applies, 0 turns it off. */
int asciichat_hip_plan_render_length_first(asciichat_hip_plan_t *plan, uint32_t *out_len_dev, uint8_t *dst, size_t dst_capacity,
                                           uint64_t *off_out, uint32_t *len_out, void *stream);
int asciichat_hip_plan_get_length_first(const asciichat_hip_plan_t *plan); /* 1: the packed entry points may take that form for this plan */
/* Which form plan_render_crc / plan_render_packets take: -1 (default) the fused one where it is the faster form (the
 * per-cell modes' stream kernel, frames of at most 8192 cells in truecolor foreground / 12288 in the other modes (2032 / 8192 from sampled-image sources): beside the
 * lean render of larger frames the stand-alone pass is faster, profiles/r06_wire_audit.txt -- for a small launch whose render
 * is shared out over workgroups only while a wave has one block: fusing means one workgroup per frame, and a lone 320x90 frame
 * then takes 116 us where render + stand-alone pass take 17), 1 wherever the plan's geometry carries it (also the rows kernel of the run-structured modes, where the
 * stand-alone pass is measured faster), 0 never.  plan_has_fused_crc() tells what a call will do.  Mode 1 on a plan whose
 * geometry has no fused instantiation in this build (the rows kernel's exist in -DACHIP_ALL_GEOMETRIES builds only) returns
 * ASCIICHAT_HIP_ERR_NOT_SUPPORTED -- the setting is kept, the calls run render + stand-alone pass. */
int asciichat_hip_plan_set_fused_crc(asciichat_hip_plan_t *plan, int mode);
int asciichat_hip_plan_render_crc_profiled(asciichat_hip_plan_t *plan, uint8_t *out_dev, size_t out_stride,
                                           uint32_t *out_len_dev, uint32_t *crc_out_dev,
                                           unsigned long long *phase_cycles_dev, void *stream); /* diagnostics */
int asciichat_hip_plan_has_fused_crc(const asciichat_hip_plan_t *plan);
int asciichat_hip_packets_from_crc(const uint32_t *len_dev, const uint32_t *crc_dev, int n, const uint32_t *dims_dev,
                                   uint8_t *hdr_out_dev, uint32_t *packet_crc_out_dev, void *stream);

/* Upload a composite descriptor for use as achip_frame_t.comp; free with asciichat_hip_free. */
int asciichat_hip_composite_upload(const achip_composite_t *comp_host, achip_composite_t **comp_dev);
void asciichat_hip_free(void *dev_ptr);

#ifdef __cplusplus
}
#endif
#endif
