/* tick_latency.c -- what ONE server tick costs in C, at the client counts and terminal sizes a real server runs
 * (bench.py's tick_e2e issues the same calls from Python, whose call overhead is a third of a nine-client tick):
 *   n clients' 1080p blobs in the pinned pool  ->  asciichat_hip_frame_table_publish_sampled_batch (the images the targets
 *   sample: one pinned block, one DMA)  ->  frame_table_latest_frames  ->  plan_update  ->  plan_render_packets_packed
 *   (frames at their exact lengths + checksums + headers into mapped HOST memory)  ->  the tick's synchronisation.
 * Per (n, W x H): median / p90 / p99 of 2000 ticks, the publish call's share, and the last tick's first frame against the
 * drop-in entry point of the same library (ascii_convert_with_capabilities on the same image).
 * Build: gcc -O2 -I include -I /opt/rocm/include scripts/tick_latency.c -o scripts/tick_latency -L ascii-chat_amd -lasciichat_hip
 *        -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/ascii-chat_amd -lm
 * usage: tick_latency [ticks]  */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "achip_host.h"
#include "asciichat_hip.h"
#include "asciichat_render.h"

#define DIE(...)                                                                                                       \
  do {                                                                                                                 \
    fprintf(stderr, __VA_ARGS__);                                                                                      \
    fprintf(stderr, " [%s]\n", asciichat_hip_last_error());                                                            \
    exit(1);                                                                                                           \
  } while (0)

static double now_us(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return 1e6 * (double)t.tv_sec + 1e-3 * (double)t.tv_nsec;
}
static int cmp_d(const void *a, const void *b) { return *(const double *)a < *(const double *)b ? -1 : *(const double *)a > *(const double *)b; }

#define MAXC 256
#define DISTINCT 16

int main(int argc, char **argv) {
  const int ticks = argc > 1 ? atoi(argv[1]) : 2000;
  const int sw = 1920, sh = 1080;
  const size_t px = (size_t)sw * sh * 3, blob_bytes = 8 + px;
  uint8_t *blob[DISTINCT];
  uint32_t x = 12345u;
  for (int k = 0; k < DISTINCT; k++) { /* > 4 MiB: the pool's pinned, device-mapped class */
    blob[k] = (uint8_t *)buffer_pool_alloc(NULL, blob_bytes);
    if (!blob[k])
      DIE("buffer_pool_alloc");
    const uint8_t hdr[8] = {0, 0, (uint8_t)(sw >> 8), (uint8_t)sw, 0, 0, (uint8_t)(sh >> 8), (uint8_t)sh};
    memcpy(blob[k], hdr, 8);
    for (size_t i = 0; i < px; i++) {
      x ^= x << 13, x ^= x >> 17, x ^= x << 5;
      blob[k][8 + i] = (uint8_t)x;
    }
  }
  hipStream_t st;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
    DIE("hipStreamCreate");
  void *streams[1] = {st};
  terminal_capabilities_t caps;
  memset(&caps, 0, sizeof caps);
  caps.color_level = TERM_COLOR_TRUECOLOR;
  caps.render_mode = RENDER_MODE_FOREGROUND;
  printf("# ticks of n clients (1080p blobs in the pinned pool -> W x H truecolor), %d ticks each: us per tick, the publish call's share, ingest threads %d\n", ticks,
         asciichat_hip_ingest_threads());
  const int sizes[4][2] = {{80, 24}, {120, 40}, {160, 45}, {200, 60}};
  const int counts[5] = {1, 4, 9, 32, 64};
  double *t_all = (double *)malloc(sizeof(double) * (size_t)ticks), *t_pub = (double *)malloc(sizeof(double) * (size_t)ticks);
  for (int si = 0; si < 4; si++)
    for (int ci = 0; ci < 5; ci++) {
      const int W = sizes[si][0], H = sizes[si][1], n = counts[ci];
      asciichat_hip_frame_table_t *table = NULL;
      if (asciichat_hip_frame_table_create(&table, n) != 0)
        DIE("frame table");
      achip_frame_t target, targets[MAXC], frames[MAXC];
      if (achip_frame_setup(&target, NULL, sw, sh, W, H, caps.render_mode, false, false, false) != 0)
        DIE("frame_setup");
      int slots[MAXC];
      const void *blobs[MAXC];
      size_t sizes_b[MAXC];
      uint32_t dims_h[2 * MAXC];
      for (int c = 0; c < n; c++)
        slots[c] = c, sizes_b[c] = blob_bytes, targets[c] = target, dims_h[2 * c] = (uint32_t)W, dims_h[2 * c + 1] = (uint32_t)H;
      asciichat_hip_plan_t *plan = NULL;
      uint8_t *slab = NULL, *hdr = NULL;
      uint32_t *len = NULL, *crc = NULL, *pkt = NULL, *dims = NULL;
      size_t stride = 0, cap = 0, tab = 0;
      void *host = NULL, *alias = NULL;
      for (int tick = -8; tick < ticks; tick++) { /* the first ticks allocate: untimed */
        for (int c = 0; c < n; c++)
          blobs[c] = blob[(c + tick + 8) % DISTINCT];
        const double t0 = now_us();
        if (asciichat_hip_frame_table_publish_sampled_batch(table, slots, blobs, sizes_b, n, targets, n, st) != 0)
          DIE("publish_sampled_batch");
        const double t1 = now_us();
        memcpy(frames, targets, sizeof(achip_frame_t) * (size_t)n);
        if (asciichat_hip_frame_table_latest_frames(table, slots, n, st, frames) != n)
          DIE("latest_frames");
        if (!plan) {
          if (asciichat_hip_plan_create(&plan, achip_mode_from_caps(caps.color_level, caps.render_mode), PALETTE_CHARS_STANDARD, frames, n) != 0)
            DIE("plan_create");
          if (getenv("TL_EXACT")) /* A/B: the one-launch exact-length form also into host memory */
            (void)asciichat_hip_plan_set_exact_length(plan, atoi(getenv("TL_EXACT")));
          stride = asciichat_hip_plan_out_stride(plan);
          cap = stride * (size_t)n;
          tab = (8 * ((size_t)n + 1) + 4 * (size_t)n + 15) / 16 * 16;
          if (hipMalloc((void **)&slab, cap) || hipMalloc((void **)&hdr, 24 * (size_t)n) || hipMalloc((void **)&len, 4 * (size_t)n) ||
              hipMalloc((void **)&crc, 4 * (size_t)n) || hipMalloc((void **)&pkt, 4 * (size_t)n) || hipMalloc((void **)&dims, 8 * (size_t)n) ||
              hipMemcpy(dims, dims_h, 8 * (size_t)n, hipMemcpyHostToDevice))
            DIE("hipMalloc");
          if (asciichat_hip_host_alloc(tab + cap, &host, &alias) != 0)
            DIE("host_alloc");
        } else if (asciichat_hip_plan_update(plan, frames, st) != 0)
          DIE("plan_update");
        if (asciichat_hip_plan_render_packets_packed(plan, slab, stride, len, dims, crc, hdr, pkt, (uint8_t *)alias + tab, cap, (uint64_t *)alias,
                                                     (uint32_t *)((uint8_t *)alias + 8 * ((size_t)n + 1)), st) != 0)
          DIE("render_packets_packed");
        if (asciichat_hip_streams_wait(streams, 1) != 0)
          DIE("streams_wait");
        const double t2 = now_us();
        if (tick >= 0)
          t_all[tick] = t2 - t0, t_pub[tick] = t1 - t0;
      }
      /* what arrived on the host for client 0 is what the drop-in entry point makes of the same image */
      const uint64_t *off = (const uint64_t *)host;
      const uint32_t *plen = (const uint32_t *)((const uint8_t *)host + 8 * ((size_t)n + 1));
      image_t img;
      memset(&img, 0, sizeof img);
      img.w = sw, img.h = sh, img.pixels = (rgb_pixel_t *)(blob[(0 + ticks - 1 + 8) % DISTINCT] + 8);
      char *want = ascii_convert_with_capabilities(&img, W, H, &caps, false, false, PALETTE_CHARS_STANDARD);
      const int same = want && strlen(want) == plen[0] && memcmp(want, (const uint8_t *)host + tab + off[0], plen[0]) == 0;
      free(want);
      qsort(t_all, (size_t)ticks, sizeof(double), cmp_d);
      qsort(t_pub, (size_t)ticks, sizeof(double), cmp_d);
      printf("%3d x %3dx%-3d: median %7.1f  p90 %7.1f  p99 %7.1f us   publish median %6.1f us   %8.0f frames/s   %7.1f KB to the host   frame 0 %s\n", n, W, H,
             t_all[ticks / 2], t_all[ticks * 9 / 10], t_all[ticks * 99 / 100], t_pub[ticks / 2], 1e6 * n / t_all[ticks / 2],
             (double)off[n] / 1e3, same ? "= the drop-in call's" : "DIFFERS from the drop-in call's");
      fflush(stdout);
      if (!same)
        return 2;
      asciichat_hip_host_free(host);
      asciichat_hip_plan_destroy(plan);
      asciichat_hip_frame_table_destroy(table);
      (void)hipFree(slab), (void)hipFree(hdr), (void)hipFree(len), (void)hipFree(crc), (void)hipFree(pkt), (void)hipFree(dims);
    }
  return 0;
}
