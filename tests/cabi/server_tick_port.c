/*
 * server_tick_port.c -- one tick of a many-client server in plain C against the BATCH layer of the C-ABI
 * (include/asciichat_hip.h + achip_host.h) and the HIP runtime's C API, the way INTEGRATION.md section 2 describes it:
 *
 *   receive side   every client's latest camera frame arrives as the reference's blob [u32 BE width][u32 BE height][RGB24]
 *                  (src/server/stream.c:330-372) and is PUBLISHED once into the device-resident frame table
 *   render tick    one plan renders every client's frame for its terminal (sizes as ascii_convert_with_capabilities,
 *                  lib/video/ascii/ascii.c:194-387), and the same launch leaves the wire stage's results: frame CRC-32C,
 *                  the 24-byte ascii_frame_packet_t headers (lib/network/acip/server.c:186-214) and the CRC of
 *                  header || frame (lib/network/acip/send.c:59-69)
 *   send side      one copy back; every packet is checked HERE, on the host, with this file's own bit-serial CRC-32C:
 *                  the frame against what the drop-in entry point returns for the same image, the header field by field,
 *                  both checksums
 *
 * A second tick renders the server's 3x3 grid with the tiles exchanged through the library's RCCL layer (world of one).
 *
 * No Python, no PyTorch: gcc, libasciichat_hip.so, libamdhip64.  Exit status 0 and a last line "ok: <n> checks".
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "achip_host.h"
#include "asciichat_hip.h"
#include "asciichat_render.h"

static int g_checks;
#define CHECK(cond, ...)                                                                                               \
  do {                                                                                                                 \
    g_checks++;                                                                                                        \
    if (!(cond)) {                                                                                                     \
      fprintf(stderr, "FAILED %s:%d: %s -- ", __FILE__, __LINE__, #cond);                                              \
      fprintf(stderr, __VA_ARGS__);                                                                                    \
      fprintf(stderr, " [%s]\n", asciichat_hip_last_error());                                                          \
      exit(1);                                                                                                         \
    }                                                                                                                  \
  } while (0)
#define HIP(call) CHECK((call) == hipSuccess, "%s", #call)

/* CRC-32C (Castagnoli), reflected, init and final xor 0xFFFFFFFF: asciichat_crc32 (lib/network/crc32.c:95-190), bit by bit */
static uint32_t crc32c(const uint8_t *p, size_t n, uint32_t crc) {
  crc = ~crc;
  for (size_t i = 0; i < n; i++) {
    crc ^= p[i];
    for (int k = 0; k < 8; k++)
      crc = (crc & 1u) ? (crc >> 1) ^ 0x82F63B78u : crc >> 1;
  }
  return ~crc;
}

static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

enum { CLIENTS = 12 };

static void run_tick(int force_variant) {
  static const int src_w[CLIENTS] = {320, 640, 160, 320, 1280, 64, 320, 333, 640, 2, 320, 800};
  static const int src_h[CLIENTS] = {240, 480, 120, 200, 720, 48, 240, 111, 360, 2, 240, 600};
  static const int term_w[CLIENTS] = {80, 120, 40, 80, 200, 20, 80, 97, 132, 10, 80, 100};
  static const int term_h[CLIENTS] = {24, 40, 12, 24, 60, 10, 50, 31, 43, 5, 24, 30};
  const char *palette = "   ...',;:clodxkO0KXNWM";
  terminal_capabilities_t caps;
  memset(&caps, 0, sizeof caps);
  caps.color_level = 3; /* truecolor */
  caps.render_mode = RENDER_MODE_FOREGROUND;
  caps.utf8_support = true;
  caps.wants_padding = true;

  hipStream_t upload, render;
  HIP(hipStreamCreateWithFlags(&upload, hipStreamNonBlocking));
  HIP(hipStreamCreateWithFlags(&render, hipStreamNonBlocking));

  /* ---- receive side */
  asciichat_hip_frame_table_t *table = NULL;
  CHECK(asciichat_hip_frame_table_create(&table, CLIENTS) == 0, "frame table");
  uint8_t *blob[CLIENTS];
  uint32_t seed = 2463534242u;
  for (int c = 0; c < CLIENTS; c++) {
    const size_t px = (size_t)src_w[c] * (size_t)src_h[c] * 3u;
    blob[c] = (uint8_t *)malloc(8 + px);
    CHECK(blob[c] != NULL, "malloc");
    const uint32_t w = (uint32_t)src_w[c], h = (uint32_t)src_h[c];
    blob[c][0] = (uint8_t)(w >> 24), blob[c][1] = (uint8_t)(w >> 16), blob[c][2] = (uint8_t)(w >> 8), blob[c][3] = (uint8_t)w;
    blob[c][4] = (uint8_t)(h >> 24), blob[c][5] = (uint8_t)(h >> 16), blob[c][6] = (uint8_t)(h >> 8), blob[c][7] = (uint8_t)h;
    for (size_t i = 0; i < px; i++) { /* blocks of flat colour with noise on top: runs, repeats and changes */
      seed ^= seed << 13, seed ^= seed >> 17, seed ^= seed << 5;
      const size_t pixel = i / 3u, x = pixel % (size_t)src_w[c], y = pixel / (size_t)src_w[c];
      const uint8_t flat = (uint8_t)(((x / 23u) * 37u + (y / 17u) * 91u + (i % 3u) * 60u + (size_t)c * 11u) & 0xFFu);
      blob[c][8 + i] = (seed & 7u) ? flat : (uint8_t)(seed >> 11);
    }
    CHECK(asciichat_hip_frame_table_publish(table, c, blob[c], 8 + px, upload) == 0, "publish client %d", c);
  }
  /* a blob the reference would skip is refused and leaves the slot untouched */
  CHECK(asciichat_hip_frame_table_publish(table, 0, blob[0], 11, upload) != 0, "short blob must be refused");

  /* ---- render tick */
  achip_frame_t frames[CLIENTS];
  uint32_t dims_host[2 * CLIENTS];
  for (int c = 0; c < CLIENTS; c++) {
    const uint8_t *px = NULL;
    int w = 0, h = 0;
    uint64_t gen = 0;
    CHECK(asciichat_hip_frame_table_latest(table, c, render, &px, &w, &h, &gen) == 0 && px && w == src_w[c] && h == src_h[c] && gen == 1,
          "latest frame of client %d", c);
    CHECK(achip_frame_setup(&frames[c], px, w, h, term_w[c], term_h[c], caps.render_mode, caps.wants_padding, true, false) == 0,
          "frame_setup client %d", c);
    dims_host[2 * c] = (uint32_t)term_w[c];
    dims_host[2 * c + 1] = (uint32_t)term_h[c];
  }
  asciichat_hip_plan_t *plan = NULL;
  CHECK(asciichat_hip_plan_create(&plan, achip_mode_from_caps(caps.color_level, caps.render_mode), palette, frames, CLIENTS) == 0,
        "plan_create");
  if (force_variant >= 0) {
    CHECK(asciichat_hip_plan_set_variant(plan, force_variant) == 0, "set_variant %d", force_variant);
    /* (the whole-frame tick is here for the fused checksum: asked for -- by itself the plan takes it for small frames only, and
     * one of these terminals is 200 columns wide; round 6's wire audit) */
    CHECK(asciichat_hip_plan_set_fused_crc(plan, 1) == 0, "set_fused_crc(1) on geometry %d", force_variant);
  }
  const size_t stride = asciichat_hip_plan_out_stride(plan);
  uint8_t *slab = NULL, *hdr = NULL;
  uint32_t *len = NULL, *crc = NULL, *pkt = NULL, *dims = NULL;
  HIP(hipMalloc((void **)&slab, stride * CLIENTS));
  HIP(hipMalloc((void **)&hdr, 24 * CLIENTS));
  HIP(hipMalloc((void **)&len, 4 * CLIENTS));
  HIP(hipMalloc((void **)&crc, 4 * CLIENTS));
  HIP(hipMalloc((void **)&pkt, 4 * CLIENTS));
  HIP(hipMalloc((void **)&dims, 8 * CLIENTS));
  HIP(hipMemcpy(dims, dims_host, 8 * CLIENTS, hipMemcpyHostToDevice));
  CHECK(asciichat_hip_plan_render_packets(plan, slab, stride, len, dims, crc, hdr, pkt, render) == 0, "render_packets");

  /* ---- send side */
  uint8_t *slab_h = (uint8_t *)malloc(stride * CLIENTS), hdr_h[24 * CLIENTS];
  uint32_t len_h[CLIENTS], crc_h[CLIENTS], pkt_h[CLIENTS];
  CHECK(slab_h != NULL, "malloc");
  void *streams[1] = {render};
  CHECK(asciichat_hip_streams_wait(streams, 1) == 0, "streams_wait");
  HIP(hipMemcpy(slab_h, slab, stride * CLIENTS, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(hdr_h, hdr, sizeof hdr_h, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(len_h, len, sizeof len_h, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(crc_h, crc, sizeof crc_h, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(pkt_h, pkt, sizeof pkt_h, hipMemcpyDeviceToHost));
  for (int c = 0; c < CLIENTS; c++) {
    const uint8_t *frame = slab_h + (size_t)c * stride, *h = hdr_h + 24 * c;
    CHECK(len_h[c] < 0xFFFFFFF0u && len_h[c] < stride && frame[len_h[c]] == 0, "client %d: length %u, NUL behind it", c, len_h[c]);
    /* the same image through the reference's own entry point (the drop-in layer) */
    image_t img = {src_w[c], src_h[c], (rgb_pixel_t *)(blob[c] + 8), IMAGE_ALLOC_SIMD};
    char *ref = ascii_convert_with_capabilities(&img, term_w[c], term_h[c], &caps, true, false, palette);
    CHECK(ref != NULL, "drop-in render of client %d", c);
    CHECK(strlen(ref) == len_h[c] && memcmp(ref, frame, len_h[c]) == 0, "client %d: batch frame == drop-in frame (%zu vs %u bytes)", c,
          strlen(ref), len_h[c]);
    free(ref);
    const uint32_t want = crc32c(frame, len_h[c], 0);
    CHECK(crc_h[c] == want, "client %d: frame CRC %08x, expected %08x", c, crc_h[c], want);
    CHECK(be32(h) == (uint32_t)term_w[c] && be32(h + 4) == (uint32_t)term_h[c] && be32(h + 8) == len_h[c] && be32(h + 12) == 0 &&
              be32(h + 16) == want && be32(h + 20) == 0,
          "client %d: packet header fields", c);
    uint32_t p = crc32c(h, 24, 0);            /* CRC over header || frame, continued across the two pieces */
    p = crc32c(frame, len_h[c], p);
    CHECK(pkt_h[c] == p, "client %d: packet CRC %08x, expected %08x", c, pkt_h[c], p);
  }
  printf("tick (geometry %d, fused CRC %s): %d clients, frames == drop-in renders, headers and both checksums verified on the host\n",
         asciichat_hip_plan_get_variant(plan), asciichat_hip_plan_has_fused_crc(plan) ? "yes" : "no", CLIENTS);

  /* ---- the next tick through the whole-tick forms: every client's SAMPLED pixels in one packed
   * block / one DMA / one launch (the targets are simply the tick's render descriptors: every blob is matched with the
   * descriptors of its own geometry), every descriptor's source pointer in one call, and the frames delivered at their
   * exact lengths into mapped host memory.  Same blobs, so the frames must equal tick one's. */
  int slots[CLIENTS];
  const void *blobs[CLIENTS];
  size_t blob_sizes[CLIENTS];
  for (int c = 0; c < CLIENTS; c++) {
    slots[c] = c;
    blobs[c] = blob[c];
    blob_sizes[c] = 8 + (size_t)src_w[c] * (size_t)src_h[c] * 3u;
  }
  CHECK(asciichat_hip_frame_table_publish_rows_batch(table, slots, blobs, blob_sizes, CLIENTS, frames, CLIENTS, upload) == 0,
        "publish_rows_batch");
  CHECK(asciichat_hip_frame_table_latest_frames(table, slots, CLIENTS, render, frames) == CLIENTS, "latest_frames");
  CHECK(asciichat_hip_plan_update(plan, frames, render) == 0, "plan_update");
  const size_t cap = stride * CLIENTS, tab = ((size_t)(CLIENTS + 1) * 8u + (size_t)CLIENTS * 4u + 15u) & ~(size_t)15;
  void *host = NULL, *alias = NULL;
  CHECK(asciichat_hip_host_alloc(tab + cap, &host, &alias) == 0, "host_alloc");
  uint64_t *off_h = (uint64_t *)host;
  uint32_t *plen_h = (uint32_t *)((uint8_t *)host + 8u * (CLIENTS + 1));
  CHECK(asciichat_hip_plan_render_packed(plan, slab, stride, len, (uint8_t *)alias + tab, cap, (uint64_t *)alias,
                                         (uint32_t *)((uint8_t *)alias + 8u * (CLIENTS + 1)), render) == 0,
        "render_packed");
  CHECK(asciichat_hip_streams_wait(streams, 1) == 0, "streams_wait");
  size_t packed_bytes = 0;
  for (int c = 0; c < CLIENTS; c++) {
    CHECK(plen_h[c] == len_h[c] && (off_h[c] & 15u) == 0 && off_h[c] + plen_h[c] <= cap, "client %d: packed length %u at %llu", c,
          plen_h[c], (unsigned long long)off_h[c]);
    CHECK(memcmp((uint8_t *)host + tab + off_h[c], slab_h + (size_t)c * stride, len_h[c]) == 0,
          "client %d: frame of the sampled-pixel tick == frame of the whole-blob tick", c);
    packed_bytes = (size_t)off_h[c] + plen_h[c];
  }
  CHECK(off_h[CLIENTS] >= packed_bytes && off_h[CLIENTS] <= cap, "packed total");
  printf("tick two (sampled pixels in one batch, latest_frames, packed output in mapped host memory): %d frames identical, %zu of %zu "
         "slab bytes crossed PCIe\n", CLIENTS, (size_t)off_h[CLIENTS], cap);
  asciichat_hip_host_free(host);

  /* ---- tick three: the table keeps the IMAGES the targets sample, the render reads them where
   * the tick's one DMA put them, and the send side -- frames at their exact lengths, checksums, headers -- is one call.
   * Clients whose target takes their frame as it is (the 2x2 source: nothing to compact) publish the whole blob; the others
   * are staged, first one by one as receive threads would, then a fourth tick through the batch form. */
  achip_frame_t targets[CLIENTS], frames3[CLIENTS];
  for (int c = 0; c < CLIENTS; c++)
    CHECK(achip_frame_setup(&targets[c], NULL, src_w[c], src_h[c], term_w[c], term_h[c], caps.render_mode, caps.wants_padding, true,
                            false) == 0,
          "target of client %d", c);
  uint8_t *dst3 = NULL;
  uint64_t *off3 = NULL;
  uint32_t *plen3 = NULL;
  HIP(hipMalloc((void **)&dst3, cap));
  HIP(hipMalloc((void **)&off3, 8 * (CLIENTS + 1)));
  HIP(hipMalloc((void **)&plen3, 4 * CLIENTS));
  uint8_t *dst3_h = (uint8_t *)malloc(cap);
  CHECK(dst3_h != NULL, "malloc");
  for (int tick = 3; tick <= 4; tick++) {
    int staged = 0, whole = 0, sslots[CLIENTS];
    const void *sblobs[CLIENTS];
    size_t ssizes[CLIENTS];
    achip_frame_t stargets[CLIENTS];
    for (int c = 0; c < CLIENTS; c++) {
      const int compacts = targets[c].out_h < src_h[c] || 2 * targets[c].out_w <= src_w[c];
      if (!compacts) {
        CHECK(asciichat_hip_frame_table_stage(table, c, blob[c], blob_sizes[c], &targets[c]) != 0, "client %d: nothing to compact", c);
        CHECK(asciichat_hip_frame_table_publish(table, c, blob[c], blob_sizes[c], upload) == 0, "publish client %d", c);
        whole++;
      } else if (tick == 3) {
        CHECK(asciichat_hip_frame_table_stage(table, c, blob[c], blob_sizes[c], &targets[c]) == 0, "stage client %d", c);
        staged++;
      } else {
        sslots[staged] = c, sblobs[staged] = blob[c], ssizes[staged] = blob_sizes[c], stargets[staged] = targets[c];
        staged++;
      }
    }
    if (tick == 3)
      CHECK(asciichat_hip_frame_table_commit(table, upload) == 0, "commit");
    else
      CHECK(asciichat_hip_frame_table_publish_sampled_batch(table, sslots, sblobs, ssizes, staged, stargets, staged, upload) == 0,
            "publish_sampled_batch");
    memcpy(frames3, targets, sizeof frames3);
    CHECK(asciichat_hip_frame_table_latest_frames(table, slots, CLIENTS, render, frames3) == CLIENTS, "latest_frames (tick %d)", tick);
    int on_images = 0;
    for (int c = 0; c < CLIENTS; c++)
      on_images += frames3[c].src_h == frames3[c].out_h && frames3[c].y_ratio == 65536u;
    CHECK(on_images == staged, "%d of %d staged clients render from their sampled image", on_images, staged);
    CHECK(asciichat_hip_plan_update(plan, frames3, render) == 0, "plan_update (tick %d)", tick);
    HIP(hipMemsetAsync(crc, 0, 4 * CLIENTS, render));
    CHECK(asciichat_hip_plan_render_packets_packed(plan, slab, stride, len, dims, crc, hdr, pkt, dst3, cap, off3, plen3, render) == 0,
          "render_packets_packed (tick %d)", tick);
    CHECK(asciichat_hip_streams_wait(streams, 1) == 0, "streams_wait");
    uint64_t off3_h[CLIENTS + 1];
    uint32_t plen3_h[CLIENTS], crc3_h[CLIENTS], pkt3_h[CLIENTS];
    uint8_t hdr3_h[24 * CLIENTS];
    HIP(hipMemcpy(dst3_h, dst3, cap, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(off3_h, off3, sizeof off3_h, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(plen3_h, plen3, sizeof plen3_h, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(crc3_h, crc, sizeof crc3_h, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(pkt3_h, pkt, sizeof pkt3_h, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(hdr3_h, hdr, sizeof hdr3_h, hipMemcpyDeviceToHost));
    size_t sum = 0;
    for (int c = 0; c < CLIENTS; c++) { /* every frame at its own offset (any order), its bytes, checksums and header as in tick one */
      CHECK(plen3_h[c] == len_h[c] && (off3_h[c] & 15u) == 0 && off3_h[c] + plen3_h[c] <= cap, "client %d: exact length %u at %llu", c,
            plen3_h[c], (unsigned long long)off3_h[c]);
      CHECK(memcmp(dst3_h + off3_h[c], slab_h + (size_t)c * stride, len_h[c]) == 0, "client %d: frame of tick %d == frame of tick one", c, tick);
      CHECK(crc3_h[c] == crc_h[c] && pkt3_h[c] == pkt_h[c] && memcmp(hdr3_h + 24 * c, hdr_h + 24 * c, 24) == 0,
            "client %d: wire stage of tick %d == tick one", c, tick);
      sum += ((size_t)plen3_h[c] + 15u) & ~(size_t)15;
    }
    CHECK(off3_h[CLIENTS] == sum, "frames tile the destination: total %llu, expected %zu", (unsigned long long)off3_h[CLIENTS], sum);
    printf("tick %d (%s, %d sampled images + %d whole blobs, frames + wire stage %s): %d frames identical\n", tick,
           tick == 3 ? "stage x N + commit" : "publish_sampled_batch", staged, whole,
           asciichat_hip_plan_get_exact_length(plan) ? "in ONE launch at their exact lengths" : "packed behind the render", CLIENTS);
  }
  (void)hipFree(dst3), (void)hipFree(off3), (void)hipFree(plen3);
  free(dst3_h);

  asciichat_hip_plan_destroy(plan);
  asciichat_hip_frame_table_destroy(table);
  (void)hipFree(slab), (void)hipFree(hdr), (void)hipFree(len), (void)hipFree(crc), (void)hipFree(pkt), (void)hipFree(dims);
  free(slab_h);
  for (int c = 0; c < CLIENTS; c++)
    free(blob[c]);
  HIP(hipStreamDestroy(upload));
  HIP(hipStreamDestroy(render));
}

/* The server's pixel-space grid (create_multi_source_composite + convert_composite_to_ascii, src/server/stream.c:664-854)
 * with the sources spread over GPUs -- here a world of ONE rank, so that the real RCCL collective runs on one GPU: the
 * tiles go through asciichat_hip_grid_exchange (one batched resize launch + one ncclAllGather), every target client is
 * rendered from the gathered tiles, and the result must equal the single-GPU composite path (achip_composite_setup:
 * the same geometry, sampled straight from the sources) byte for byte. */
static void run_grid_tick(void) {
  enum { SOURCES = 9, TARGETS = 4, TW = 160, TH = 48 };
  int sw[SOURCES], sh[SOURCES];
  uint8_t *src_dev[SOURCES];
  const uint8_t *src_c[SOURCES];
  uint32_t seed = 88172645u;
  for (int k = 0; k < SOURCES; k++) {
    sw[k] = 320 + 64 * (k % 3);
    sh[k] = 180 + 36 * (k / 3);
    const size_t px = (size_t)sw[k] * (size_t)sh[k] * 3u;
    uint8_t *h = (uint8_t *)malloc(px);
    CHECK(h != NULL, "malloc");
    for (size_t i = 0; i < px; i++) {
      seed ^= seed << 13, seed ^= seed >> 17, seed ^= seed << 5;
      h[i] = (uint8_t)(seed >> 9);
    }
    HIP(hipMalloc((void **)&src_dev[k], px));
    HIP(hipMemcpy(src_dev[k], h, px, hipMemcpyHostToDevice));
    src_c[k] = src_dev[k];
    free(h);
  }
  hipStream_t st;
  HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  uint8_t id[ASCIICHAT_HIP_COMM_ID_BYTES];
  asciichat_hip_comm_t *comm = NULL;
  CHECK(asciichat_hip_comm_unique_id(id, sizeof id) == 0, "comm_unique_id");
  CHECK(asciichat_hip_comm_init(&comm, 1, 0, id, sizeof id) == 0 && asciichat_hip_comm_world(comm) == 1, "comm_init");
  asciichat_hip_grid_t *grid = NULL;
  CHECK(asciichat_hip_grid_create(&grid, comm, sw, sh, NULL, SOURCES, TW, TH) == 0, "grid_create");
  for (int k = 0; k < SOURCES; k++)
    CHECK(asciichat_hip_grid_owner(grid, k) == 0, "one rank owns every source");
  CHECK(asciichat_hip_grid_exchange(grid, src_c, st) == 0, "grid_exchange");

  achip_composite_t local, *local_dev = NULL; /* the single-GPU path: the same grid sampled straight from the sources */
  achip_composite_setup(&local, src_c, sw, sh, SOURCES, TW, TH);
  CHECK(asciichat_hip_composite_upload(&local, &local_dev) == 0, "composite_upload");

  uint8_t *slab[2] = {NULL, NULL};
  uint32_t *len[2] = {NULL, NULL};
  size_t stride = 0;
  for (int path = 0; path < 2; path++) {
    achip_frame_t f[TARGETS];
    for (int t = 0; t < TARGETS; t++) { /* every target looks at the W x 2H canvas (stream.c:790-854: aspect + padding on) */
      CHECK(achip_frame_setup(&f[t], NULL, TW, 2 * TH, TW - 8 * t, TH - 3 * t, RENDER_MODE_FOREGROUND, true, true, false) == 0,
            "frame_setup target %d", t);
      f[t].comp = path == 0 ? asciichat_hip_grid_composite_dev(grid) : local_dev;
    }
    asciichat_hip_plan_t *plan = NULL;
    CHECK(asciichat_hip_plan_create(&plan, achip_mode_from_caps(3, RENDER_MODE_FOREGROUND), "   ...',;:clodxkO0KXNWM", f, TARGETS) == 0,
          "plan_create (grid, path %d)", path);
    stride = asciichat_hip_plan_out_stride(plan);
    HIP(hipMalloc((void **)&slab[path], stride * TARGETS));
    HIP(hipMalloc((void **)&len[path], 4 * TARGETS));
    CHECK(asciichat_hip_plan_render(plan, slab[path], stride, len[path], st) == 0, "plan_render (grid, path %d)", path);
    void *streams[1] = {st};
    CHECK(asciichat_hip_streams_wait(streams, 1) == 0, "streams_wait");
    asciichat_hip_plan_destroy(plan);
  }
  uint8_t *a = (uint8_t *)malloc(stride * TARGETS), *b = (uint8_t *)malloc(stride * TARGETS);
  uint32_t la[TARGETS], lb[TARGETS];
  CHECK(a && b, "malloc");
  HIP(hipMemcpy(a, slab[0], stride * TARGETS, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(b, slab[1], stride * TARGETS, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(la, len[0], sizeof la, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(lb, len[1], sizeof lb, hipMemcpyDeviceToHost));
  for (int t = 0; t < TARGETS; t++) {
    CHECK(la[t] < 0xFFFFFFF0u && la[t] == lb[t] && la[t] > 100u, "target %d: lengths %u / %u", t, la[t], lb[t]);
    CHECK(memcmp(a + (size_t)t * stride, b + (size_t)t * stride, la[t]) == 0, "target %d: exchanged tiles == direct composite", t);
  }
  printf("grid tick: %d sources -> tiles through one ncclAllGather (world 1) -> %d targets, identical to the direct composite\n", SOURCES,
         TARGETS);
  free(a), free(b);
  for (int p = 0; p < 2; p++)
    (void)hipFree(slab[p]), (void)hipFree(len[p]);
  asciichat_hip_free(local_dev);
  asciichat_hip_grid_destroy(grid);
  asciichat_hip_comm_destroy(comm);
  for (int k = 0; k < SOURCES; k++)
    (void)hipFree(src_dev[k]);
  HIP(hipStreamDestroy(st));
}

int main(void) {
  if (asciichat_hip_device_count() <= 0) {
    fprintf(stderr, "no HIP device: %s\n", asciichat_hip_last_error());
    return 2;
  }
  run_tick(-1); /* the plan's own choice: row bands for twelve frames, the stand-alone wire kernel behind them */
  run_tick(17); /* whole frames on the stream kernel: CRC, headers and packet CRCs leave the render launch */
  run_grid_tick();
  printf("ok: %d checks\n", g_checks);
  return 0;
}
