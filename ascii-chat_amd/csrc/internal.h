/* internal.h -- declarations shared by the host C sources of libasciichat_hip.so (not installed). */
#ifndef ACHIP_INTERNAL_H
#define ACHIP_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

#include "achip_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* records a thread-local message, returns `code` */
int achip_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
/* 0 when a HIP device exists, else ASCIICHAT_HIP_ERR_NO_DEVICE (and a message on stderr) */
int achip_require_device(void);
/* maps a hipError_t to 0 / ASCIICHAT_HIP_ERR_NO_DEVICE with a message */
int achip_hip_check(int hip_error, const char *what);
/* device glyph tables for a palette string, cached per device; get pins the entry, put releases it */
int achip_lut_get(const char *palette, const achip_lut_t **out_dev);
void achip_lut_put(const achip_lut_t *dev);

/* combine.c: one frame through the flat-combining layer (f->src = host pixels).  *handled = 0: not combinable, take the
 * direct path; else the malloc'd string or NULL (achip_fail has the reason). */
void achip_combine_enter(void);
void achip_combine_leave(void);
unsigned long long achip_combine_stats_clock(void); /* 0 unless ASCIICHAT_HIP_COMBINE_STATS */
void achip_combine_stats_call(unsigned long long t0);
int achip_combine_callers(void); /* drop-in render calls in flight right now */
int achip_cpu_budget(void);      /* CPUs this process may keep busy: affinity mask capped by the cgroup quota */
int achip_combine_crowded(void); /* ... more of them than CPUs this process may keep busy: sleep, do not poll */
char *achip_combine_render(int mode, const char *palette, const achip_lut_t *lut, const achip_frame_t *f, size_t src_bytes,
                           int *handled);

/* dropin.c: where a drop-in render leaves its string.  NULL target (the default): a malloc block, the reference's ownership
 * contract.  ascii_convert_with_capabilities_into() sets a target for the duration of its call: the string goes into the
 * caller's buffer instead (no malloc; `needed` is set either way) and the buffer's address is returned; a string that does
 * not fit fails with ASCIICHAT_HIP_ERR_BUFFER.  Thread-local: concurrent callers do not see each other's targets. */
typedef struct {
  char *buf;
  size_t cap;
  size_t needed; /* strlen of the frame (set even when it did not fit) */
} achip_out_target_t;
achip_out_target_t *achip_out_target(void); /* this thread's target, or NULL */
/* the hand-over itself: `len` bytes at `src` -> the thread's target or a fresh malloc block, NUL-terminated */
char *achip_out_take(const void *src, size_t len);

/* achip_host.c: the part of a HOST image a frame's point sampler reads.  stage_extent() gives the size of the compacted
 * image (sampled rows; sampled columns too when the frame is at most half as wide as its source) and returns its bytes, or 0
 * when the whole image is needed; stage_gather() copies that part to dst (row-major, tight) and rewrites d (a copy of the
 * descriptor; d->src is left for the caller) so that sampling the compacted image gives the pixels the original would. */
size_t achip_stage_extent(const achip_frame_t *f, int *w, int *h);
void achip_stage_gather(const achip_frame_t *f, const uint8_t *host_px, uint8_t *dst, achip_frame_t *d);

/* achip_host.c: what a set of render targets reads of a w x h source frame (frame_table_publish_rows*): the sampled rows
 * (ascending, unique), and the sampled columns when they are at most half of the frame's (n_cols = 0: whole rows).
 * build: 0, -1 (a target does not describe this frame), -2 (memory).  pack writes [row table][column table, if
 * any][rows, or rows x columns pixels] (tables padded to 16 bytes) -- the block scatter_rows_batch_kernel puts in place. */
typedef struct {
  uint32_t w, h;
  int n_rows, n_cols;
  uint32_t *rows, *cols;
} achip_sample_set_t;
int achip_sampled_rows(const achip_frame_t *targets, int n_targets, uint32_t h, uint32_t *rows_out, uint8_t *mark);
int achip_sample_set_build(achip_sample_set_t *S, const achip_frame_t *targets, int n_targets, uint32_t w, uint32_t h);
void achip_sample_set_free(achip_sample_set_t *S);
size_t achip_sample_set_block_bytes(const achip_sample_set_t *S);
void achip_sample_set_pack(const achip_sample_set_t *S, const uint8_t *pixels, uint8_t *blk);

/* buffer_pool.c: device alias of a pointer inside a pinned pool block, or NULL */
const void *achip_pool_device_ptr(const void *host_ptr);

#ifdef __cplusplus
}
#endif
#endif
