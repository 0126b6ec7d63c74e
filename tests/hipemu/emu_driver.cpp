/*
 * emu_driver.cpp -- runs the product kernels (render_kernels.hpp, unmodified) under the CPU fiber
 * emulator.  TESTS ONLY; built by tests/emu.py with g++ -DACHIP_HIPEMU.
 */
#define ACHIP_HIPEMU 1
#ifndef ACHIP_TEST_GEOMETRY
#define ACHIP_TEST_GEOMETRY 1 /* the tiny stream geometry exists in emulator builds only */
#endif
#include <vector>
#include "render_rows.hpp"
#include "render_variants.h"
#include "achip_host.h"

static int g_parts = 1, g_rows_per_part = 0;
static unsigned long long *g_part_sync = nullptr;
static uint32_t g_epoch = 1;
/* the prebuilt tables of crc32c_frame_kernel<1024>, as the product's launcher builds them once per process */
static const uint4 *frame_crc_tab_1024() {
  static std::vector<uint32_t> tab;
  if (tab.empty()) {
    tab.resize(ACHIP_FRAME_CRC_TAB_BYTES / 4 + 4);
    uint32_t *t = tab.data();
    hipemu::launch(dim3(1), dim3(256), ACHIP_FRAME_CRC_TAB_BYTES, [&] { achip::crc_frame_tables_init_kernel<1024>(t); });
  }
  return reinterpret_cast<const uint4 *>(tab.data());
}
/* ... and of the span kernel's 256-thread workgroups */
static const uint4 *frame_crc_tab_256() {
  static std::vector<uint32_t> tab;
  if (tab.empty()) {
    tab.resize(ACHIP_FRAME_CRC_TAB_BYTES / 4 + 4);
    uint32_t *t = tab.data();
    hipemu::launch(dim3(1), dim3(256), ACHIP_FRAME_CRC_TAB_BYTES, [&] { achip::crc_frame_tables_init_kernel<256>(t); });
  }
  return reinterpret_cast<const uint4 *>(tab.data());
}
static int g_uniform = 0; /* 1: pass the batch's common descriptor by value when it has one (as plan.c does) */

template <int MODE, int BLOCK, int CAP, int RING>
static void run(const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out, uint64_t stride,
                uint32_t *len) {
  using L = achip::Lds<MODE, BLOCK, CAP, RING>;
  achip_uniform_t uni = {};
  if (g_uniform)
    (void)achip_frames_uniform(frames, n, &uni);
  if (g_parts > 1) /* the two instantiations the product launches: row bands / whole frames (render_inst.hip) */
    hipemu::launch(dim3((unsigned)(n * g_parts)), dim3(BLOCK), (size_t)L::bytes, [&] {
      achip::render_frames_kernel<MODE, BLOCK, CAP, RING, true, true>(frames, lut, out, stride, len, n, nullptr, g_parts,
                                                                      g_rows_per_part, g_part_sync, g_epoch, uni);
    });
  else
    hipemu::launch(dim3((unsigned)n), dim3(BLOCK), (size_t)L::bytes, [&] {
      achip::render_frames_kernel<MODE, BLOCK, CAP, RING, true, false>(frames, lut, out, stride, len, n, nullptr, 1, 0,
                                                                       nullptr, g_epoch, uni);
    });
}

extern "C" int emu_set_uniform(int on) {
  const int was = g_uniform;
  g_uniform = on;
  return was;
}

/* multi-workgroup frames: set before emu_render_batch (parts == 1 restores the default) */
extern "C" void emu_set_parts(int parts, int rows_per_part, unsigned long long *part_sync, uint32_t epoch) {
  g_parts = parts;
  g_rows_per_part = rows_per_part;
  g_part_sync = part_sync;
  g_epoch = epoch;
}

template <int BLOCK, int CAP, int RING>
static int by_mode(int mode, const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out, uint64_t stride,
                   uint32_t *len) {
  switch (mode) {
#define M(m)                                                                                                           \
  case m:                                                                                                              \
    run<m, BLOCK, CAP, RING>(frames, n, lut, out, stride, len);                                                        \
    return 0;
    M(ACHIP_MODE_MONO)
    M(ACHIP_MODE_TRUE_FG)
    M(ACHIP_MODE_256_FG)
    M(ACHIP_MODE_16_FG)
    M(ACHIP_MODE_TRUE_BG)
    M(ACHIP_MODE_HB_TRUE)
    M(ACHIP_MODE_HB_256)
    M(ACHIP_MODE_HB_16)
    M(ACHIP_MODE_HB_MONO)
    M(ACHIP_MODE_16_DITHER_BG)
#undef M
  }
  return -1;
}

/* As the product's launchers: the fast-sampler instantiation (GENERIC = false) unless a frame needs the full repertoire --
 * a virtual composite or a 1x1 source.  (Until round 5 this driver always took GENERIC = true: the fast sampler, and with it
 * the rows kernel's scalar-row path, ran on the GPU only.) */
static bool needs_generic(const achip_frame_t *frames, int n) {
  for (int i = 0; i < n; i++)
    if (frames[i].comp || frames[i].src_w * frames[i].src_h == 1)
      return true;
  return false;
}

/* the stream kernel (render_stream.hpp): per-cell modes, whole frames */
template <int MODE, int WAVES, int CPL>
static void run_stream(const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out, uint64_t stride,
                       uint32_t *len) {
  using L = achip::SLds<MODE, WAVES, CPL>;
  achip_uniform_t uni = {};
  if (g_uniform)
    (void)achip_frames_uniform(frames, n, &uni);
  uni.flags = ((lut->flags & ACHIP_LUT_MULTIBYTE) ? 0u : ACHIP_UNIFORM_PALETTE_ASCII) |
              ACHIP_UNIFORM_MAX_CELLS(achip_max_cells(frames, n)); /* what plan.c / dropin.c pass */
  const size_t lds = (size_t)((L::bytes_for(achip::stream_maxblk(uni.flags, L::EFF)) + 15) & ~15);
  if (g_parts > 1) { /* PARTS: a frame's blocks shared out over g_parts workgroups (the product instantiates geometry 18) */
    const achip_partsdev_t ps = {g_parts, g_epoch, g_part_sync};
    hipemu::launch(dim3((unsigned)(n * g_parts)), dim3(WAVES * 64), lds, [&] {
      achip::render_stream_kernel<MODE, WAVES, CPL, true, false, 0, true>(frames, lut, out, stride, len, n, uni, nullptr, achip_wire_t{},
                                                                          nullptr, achip_packdev_t{}, ps);
    });
    return;
  }
  if (!needs_generic(frames, n)) {
    if constexpr (MODE == ACHIP_MODE_TRUE_FG) { /* as the product's launcher: the palette picks the instantiation */
      if (lut->flags & ACHIP_LUT_MULTIBYTE) {
        using LU = achip::SLds<ACHIP_STREAM_MODE_TRUE_FG_U8, WAVES, CPL>;
        const size_t ldsu = (size_t)((LU::bytes_for(achip::stream_maxblk(uni.flags, LU::EFF)) + 15) & ~15);
        hipemu::launch(dim3((unsigned)n), dim3(WAVES * 64), ldsu, [&] {
          achip::render_stream_kernel<ACHIP_STREAM_MODE_TRUE_FG_U8, WAVES, CPL, false>(frames, lut, out, stride, len, n, uni, nullptr, achip_wire_t{}, nullptr, achip_packdev_t{}, achip_partsdev_t{});
        });
        return;
      }
    }
    hipemu::launch(dim3((unsigned)n), dim3(WAVES * 64), lds, [&] {
      achip::render_stream_kernel<MODE, WAVES, CPL, false>(frames, lut, out, stride, len, n, uni, nullptr, achip_wire_t{}, nullptr, achip_packdev_t{}, achip_partsdev_t{});
    });
    return;
  }
  hipemu::launch(dim3((unsigned)n), dim3(WAVES * 64), lds, [&] {
    achip::render_stream_kernel<MODE, WAVES, CPL, true>(frames, lut, out, stride, len, n, uni, nullptr, achip_wire_t{}, nullptr, achip_packdev_t{}, achip_partsdev_t{});
  });
}
template <int WAVES, int CPL>
static int stream_by_mode(int mode, const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out,
                          uint64_t stride, uint32_t *len) {
  switch (mode) {
#define M(m)                                                                                                           \
  case m:                                                                                                              \
    run_stream<m, WAVES, CPL>(frames, n, lut, out, stride, len);                                                       \
    return 0;
    M(ACHIP_MODE_TRUE_FG)
    M(ACHIP_MODE_256_FG)
    M(ACHIP_MODE_16_FG)
    M(ACHIP_MODE_TRUE_BG)
#undef M
  }
  return -1;
}

/* the stream kernel with the frame CRC riding its drain (what asciichat_hip_plan_render_crc launches) */
template <int MODE, int WAVES, int CPL>
static void run_stream_crc(const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out, uint64_t stride,
                           uint32_t *len, const achip_wire_t &wire) {
  using L = achip::SLds<MODE, WAVES, CPL, true>;
  achip_uniform_t uni = {};
  /* as the product's launcher: tables built once by the init kernel, per-block words sized by the largest frame */
  uni.flags = ((lut->flags & ACHIP_LUT_MULTIBYTE) ? 0u : ACHIP_UNIFORM_PALETTE_ASCII) |
              ACHIP_UNIFORM_MAX_CELLS(achip_max_cells(frames, n));
  static std::vector<uint32_t> tab;
  if (tab.empty()) {
    tab.resize(L::TAB_BYTES / 4 + 4);
    uint32_t *t = tab.data();
    hipemu::launch(dim3(1), dim3(256), 0, [&] { achip::crc_tables_init_kernel<L>(t); });
  }
  const uint4 *tabv = reinterpret_cast<const uint4 *>(tab.data());
  const size_t lds = (size_t)((L::bytes_for(achip::stream_maxblk(uni.flags, L::EFF)) + 15) & ~15);
  hipemu::launch(dim3((unsigned)n), dim3(WAVES * 64), lds, [&] {
    achip::render_stream_kernel<MODE, WAVES, CPL, true, true>(frames, lut, out, stride, len, n, uni, nullptr, wire, tabv, achip_packdev_t{}, achip_partsdev_t{});
  });
}
extern "C" int emu_render_stream_crc(int mode, int variant, const achip_frame_t *frames, int n, const achip_lut_t *lut,
                                     uint8_t *out, uint64_t stride, uint32_t *len, uint32_t *crc, const uint32_t *dims,
                                     uint8_t *hdr, uint32_t *pkt) {
  const achip_wire_t wire = {crc, dims, hdr, pkt};
#define M(m, W, C)                                                                                                     \
  if (mode == m) {                                                                                                     \
    run_stream_crc<m, W, C>(frames, n, lut, out, stride, len, wire);                                                    \
    return 0;                                                                                                          \
  }
  if (variant == 20) {
    M(ACHIP_MODE_TRUE_FG, 2, 1) M(ACHIP_MODE_256_FG, 2, 1) M(ACHIP_MODE_16_FG, 2, 1) M(ACHIP_MODE_TRUE_BG, 2, 1)
  } else if (variant == 17) {
    M(ACHIP_MODE_TRUE_FG, 8, 2) M(ACHIP_MODE_256_FG, 8, 2) M(ACHIP_MODE_16_FG, 8, 2) M(ACHIP_MODE_TRUE_BG, 8, 2)
  } else if (variant == 16) {
    M(ACHIP_MODE_TRUE_FG, 16, 2) M(ACHIP_MODE_256_FG, 16, 2)
  }
#undef M
  return -1;
}

/* the stream kernel's PACK instantiations: frames at their exact lengths straight into `dst` (no slab), with or without
 * the wire stage (the frame checksummed from its LDS image).  Geometries 16 / 17 are the product's; 20 (two waves, one cell
 * per lane) makes tiny frames multi-block. */
template <int MODE, int WAVES, int CPL, bool WIRE>
static void run_stream_pack(const achip_frame_t *frames, int n, const achip_lut_t *lut, uint64_t stride, uint32_t *len,
                            const achip_wire_t &wire, const achip_packdev_t &pack) {
  constexpr int PACK = WIRE ? 2 : 1;
  using L = achip::SLds<MODE, WAVES, CPL, false, PACK>;
  achip_uniform_t uni = {};
  if (g_uniform)
    (void)achip_frames_uniform(frames, n, &uni);
  uni.flags = ((lut->flags & ACHIP_LUT_MULTIBYTE) ? 0u : ACHIP_UNIFORM_PALETTE_ASCII) |
              ACHIP_UNIFORM_MAX_CELLS(achip_max_cells(frames, n));
  static std::vector<uint32_t> tab; /* the Horner table of the checksumming waves */
  if (WIRE && tab.empty()) {
    tab.resize(ACHIP_FRAME_CRC_TAB_BYTES / 4 + 4);
    uint32_t *t = tab.data();
    hipemu::launch(dim3(1), dim3(256), ACHIP_FRAME_CRC_TAB_BYTES, [&] { achip::crc_frame_tables_init_kernel<64 * achip::pack_crc_waves(WAVES)>(t); });
  }
  const uint4 *tabv = WIRE ? reinterpret_cast<const uint4 *>(tab.data()) : nullptr;
  const size_t lds = (size_t)((L::bytes_for_pack(achip::stream_maxblk(uni.flags, L::EFF), (int)stride) + 15) & ~15);
  hipemu::launch(dim3((unsigned)n), dim3(WAVES * 64), lds, [&] {
    achip::render_stream_kernel<MODE, WAVES, CPL, false, false, PACK>(frames, lut, nullptr, stride, len, n, uni, nullptr, wire, tabv, pack, achip_partsdev_t{});
  });
}
extern "C" int emu_render_stream_pack(int mode, int variant, const achip_frame_t *frames, int n, const achip_lut_t *lut,
                                      uint64_t stride, uint32_t *len, uint32_t *crc, const uint32_t *dims, uint8_t *hdr,
                                      uint32_t *pkt, uint8_t *dst, uint64_t capacity, uint64_t *off_out, uint32_t *len_out,
                                      unsigned long long *cursor) {
  const achip_wire_t wire = {crc, dims, hdr, pkt};
  const achip_packdev_t pack = {dst, capacity, off_out, len_out, cursor};
#define M(m, W, C)                                                                                                     \
  if (mode == m) {                                                                                                     \
    if (crc)                                                                                                           \
      run_stream_pack<m, W, C, true>(frames, n, lut, stride, len, wire, pack);                                          \
    else                                                                                                               \
      run_stream_pack<m, W, C, false>(frames, n, lut, stride, len, wire, pack);                                         \
    return 0;                                                                                                          \
  }
  if (variant == 20) {
    M(ACHIP_MODE_TRUE_FG, 2, 1) M(ACHIP_MODE_256_FG, 2, 1) M(ACHIP_MODE_16_FG, 2, 1)
  } else if (variant == 16) {
    M(ACHIP_MODE_TRUE_FG, 16, 2) M(ACHIP_MODE_256_FG, 16, 2) M(ACHIP_MODE_16_FG, 16, 2)
  } else if (variant == 17) {
    M(ACHIP_MODE_TRUE_FG, 8, 2) M(ACHIP_MODE_256_FG, 8, 2) M(ACHIP_MODE_16_FG, 8, 2)
  }
#undef M
  return -1;
}

/* the stream kernel's LENGTH-FIRST instantiation (render_stream.hpp LF): exact-length truecolor frames of any size in one
 * launch, the lean loop run twice; `stride` only bounds a frame's length */
template <int WAVES, int CPL>
static void run_stream_lenfirst(const achip_frame_t *frames, int n, const achip_lut_t *lut, uint64_t stride, uint32_t *len,
                                const achip_packdev_t &pack) {
  using L = achip::SLds<ACHIP_MODE_TRUE_FG, WAVES, CPL>;
  achip_uniform_t uni = {};
  if (g_uniform)
    (void)achip_frames_uniform(frames, n, &uni);
  uni.flags = ACHIP_UNIFORM_PALETTE_ASCII | ACHIP_UNIFORM_MAX_CELLS(achip_max_cells(frames, n));
  const size_t lds = (size_t)((L::bytes_for(achip::stream_maxblk(uni.flags, L::EFF)) + 15) & ~15);
  hipemu::launch(dim3((unsigned)n), dim3(WAVES * 64), lds, [&] {
    achip::render_stream_kernel<ACHIP_MODE_TRUE_FG, WAVES, CPL, false, false, 0, false, true>(frames, lut, nullptr, stride, len, n, uni, nullptr,
                                                                                            achip_wire_t{}, nullptr, pack, achip_partsdev_t{});
  });
}
extern "C" int emu_render_stream_lenfirst(int variant, const achip_frame_t *frames, int n, const achip_lut_t *lut, uint64_t stride,
                                          uint32_t *len, uint8_t *dst, uint64_t capacity, uint64_t *off_out, uint32_t *len_out,
                                          unsigned long long *cursor) {
  const achip_packdev_t pack = {dst, capacity, off_out, len_out, cursor};
  if (needs_generic(frames, n) || (lut->flags & ACHIP_LUT_MULTIBYTE))
    return -1;
  if (variant == 20)
    run_stream_lenfirst<2, 1>(frames, n, lut, stride, len, pack);
  else if (variant == 16)
    run_stream_lenfirst<16, 2>(frames, n, lut, stride, len, pack);
  else if (variant == 17)
    run_stream_lenfirst<8, 2>(frames, n, lut, stride, len, pack);
  else
    return -1;
  return 0;
}

/* the rows kernel (render_rows.hpp): run-structured modes, whole frames; wire != nullptr: its CRC instantiation */
template <int MODE, int WAVES, int CPL, bool CRC, bool WIDE, bool PARTS>
static void run_rows(int variant, const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out, uint64_t stride,
                     uint32_t *len, const achip_wire_t &wire) {
  using L = achip::RLds<MODE, WAVES, CRC, WIDE>;
  achip_uniform_t uni = {};
  if (g_uniform)
    (void)achip_frames_uniform(frames, n, &uni);
  uni.flags = ((lut->flags & ACHIP_LUT_MULTIBYTE) ? 0u : ACHIP_UNIFORM_PALETTE_ASCII) |
              ACHIP_UNIFORM_MAX_CELLS(achip_uniform_extent(MODE, variant, frames, n)); /* what plan.c passes */
  if constexpr (WIDE) { /* rows cut into segments: fast sampler only, no fused checksum (render_rows.hpp) */
    const achip_partsdev_t ps = {PARTS ? g_parts : 1, g_epoch, g_part_sync}; /* (WIDE + PARTS: whole rows per workgroup) */
    if (!needs_generic(frames, n))
      hipemu::launch(dim3((unsigned)(n * ps.parts)), dim3(WAVES * 64), (size_t)((L::bytes_for(achip::stream_maxblk(uni.flags, 1)) + 15) & ~15), [&] {
        achip::render_rows_kernel<MODE, WAVES, CPL, false, false, true, PARTS>(frames, lut, out, stride, len, n, uni, wire, nullptr, ps);
      });
    else
      for (int i = 0; i < n; i++)
        len[i] = ACHIP_LEN_BADDESC;
    return;
  } else if constexpr (PARTS) { /* a frame's blocks shared out over g_parts workgroups: fast sampler only, no fused checksum */
    const achip_partsdev_t ps = {g_parts, g_epoch, g_part_sync};
    if (!needs_generic(frames, n))
      hipemu::launch(dim3((unsigned)(n * g_parts)), dim3(WAVES * 64), (size_t)((L::bytes_for(achip::stream_maxblk(uni.flags, 1)) + 15) & ~15), [&] {
        achip::render_rows_kernel<MODE, WAVES, CPL, false, false, false, true>(frames, lut, out, stride, len, n, uni, wire, nullptr, ps);
      });
    else
      for (int i = 0; i < n; i++)
        len[i] = ACHIP_LEN_BADDESC;
    return;
  } else {
  static std::vector<uint32_t> tab;
  const uint4 *tabv = nullptr;
  if (CRC) {
    if (tab.empty()) {
      tab.resize(L::TAB_BYTES / 4 + 4);
      uint32_t *t = tab.data();
      hipemu::launch(dim3(1), dim3(256), 0, [&] { achip::crc_tables_init_kernel<L>(t); });
    }
    tabv = reinterpret_cast<const uint4 *>(tab.data());
  }
  const size_t lds = (size_t)((L::bytes_for(achip::stream_maxblk(uni.flags, 1)) + 15) & ~15);
  if (!needs_generic(frames, n)) {
    hipemu::launch(dim3((unsigned)n), dim3(WAVES * 64), lds, [&] {
      achip::render_rows_kernel<MODE, WAVES, CPL, false, CRC>(frames, lut, out, stride, len, n, uni, wire, tabv, achip_partsdev_t{});
    });
    return;
  }
  hipemu::launch(dim3((unsigned)n), dim3(WAVES * 64), lds, [&] {
    achip::render_rows_kernel<MODE, WAVES, CPL, true, CRC>(frames, lut, out, stride, len, n, uni, wire, tabv, achip_partsdev_t{});
  });
  }
}
template <int WAVES, int CPL, bool CRC, bool WIDE, bool PARTS>
static int rows_by_mode(int mode, int variant, const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out,
                        uint64_t stride, uint32_t *len, const achip_wire_t &wire) {
  switch (mode) {
#define M(m)                                                                                                           \
  case m:                                                                                                              \
    run_rows<m, WAVES, CPL, CRC && !WIDE && !PARTS, WIDE, PARTS>(variant, frames, n, lut, out, stride, len, wire);                                     \
    return 0;
    M(ACHIP_MODE_MONO)
    M(ACHIP_MODE_HB_TRUE)
    M(ACHIP_MODE_HB_256)
    M(ACHIP_MODE_HB_16)
    M(ACHIP_MODE_HB_MONO)
#undef M
  }
  return -1;
}
extern "C" int emu_render_rows_crc(int mode, int variant, const achip_frame_t *frames, int n, const achip_lut_t *lut,
                                   uint8_t *out, uint64_t stride, uint32_t *len, uint32_t *crc, const uint32_t *dims,
                                   uint8_t *hdr, uint32_t *pkt) {
  const achip_wire_t wire = {crc, dims, hdr, pkt};
  switch (variant) {
#define X(id, W, C)                                                                                                    \
  case id:                                                                                                             \
    return ACHIP_ROWS_VARIANT_WIDE(id) || ACHIP_ROWS_VARIANT_PARTS(id) ? -1 : rows_by_mode<W, C, true, ACHIP_ROWS_VARIANT_WIDE(id), ACHIP_ROWS_VARIANT_PARTS(id)>(mode, variant, frames, n, lut, out, stride, len, wire);
    ACHIP_ROWS_VARIANTS(X)
#undef X
  }
  return -1;
}

extern "C" int emu_render_batch(int mode, int variant, const achip_frame_t *frames, int n, const achip_lut_t *lut,
                                uint8_t *out, uint64_t stride, uint32_t *len) {
  switch (variant) { /* (stream geometries take g_parts themselves) */
#define X(id, W, C)                                                                                                    \
  case id:                                                                                                             \
    return stream_by_mode<W, C>(mode, frames, n, lut, out, stride, len);
    ACHIP_STREAM_VARIANTS(X)
#undef X
  }
  if (g_parts == 1 || ACHIP_ROWS_VARIANT_PARTS(variant)) /* (the PARTS geometries take g_parts themselves) */
    switch (variant) {
#define X(id, W, C)                                                                                                    \
  case id:                                                                                                             \
    return rows_by_mode<W, C, false, ACHIP_ROWS_VARIANT_WIDE(id), ACHIP_ROWS_VARIANT_PARTS(id)>(mode, variant, frames, n, lut, out, stride, len, achip_wire_t{});
      ACHIP_ROWS_VARIANTS(X)
#undef X
    }
  switch (variant) {
#define X(id, B, C, R)                                                                                                 \
  case id:                                                                                                             \
    return by_mode<B, C, R>(mode, frames, n, lut, out, stride, len);
    ACHIP_VARIANTS(X)
#undef X
  }
  return -1;
}

extern "C" void emu_resize_nn(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh, uint32_t xr,
                              uint32_t yr) {
  hipemu::launch(dim3(3), dim3(256), 0, [&] { achip::resize_nn_kernel(src, sw, sh, 3 * sw, dst, dw, dh, xr, yr); });
}

extern "C" void emu_resize_batch(const achip_resize_batch_t *b) {
  hipemu::launch(dim3(2, (unsigned)b->n), dim3(256), 0, [&] { achip::resize_nn_batch_kernel(*b); });
}

extern "C" void emu_composite(const achip_composite_t *comp, uint8_t *dst) {
  hipemu::launch(dim3(2), dim3(256), 0, [&] { achip::composite_kernel(comp, dst); });
}

#include "stream_kernels.hpp"

extern "C" void emu_tint(uint8_t *px, int w, int h, int stride, uint32_t ops, int vector_path) {
  if (vector_path)
    hipemu::launch(dim3(3), dim3(256), 0, [&] { achip::tint_stream_kernel(px, (uint64_t)w * h * 3u, ops); });
  else
    hipemu::launch(dim3(3), dim3(256), 0, [&] { achip::tint_pixels_kernel(px, w, h, stride, ops); });
}

extern "C" void emu_flip(const uint8_t *src, uint8_t *dst, int w, int h, uint32_t ops, int vector_path) {
  if (vector_path)
    hipemu::launch(dim3(3), dim3(256), 0, [&] { achip::flip_stream_kernel(src, dst, w, h, ops); });
  else
    hipemu::launch(dim3(3), dim3(256), 0, [&] { achip::flip_pixels_kernel(src, dst, w, h, 3 * w, 3 * w, ops); });
}

/* compaction of a slab (stream_kernels.hpp), with the launcher's (n, slices) grid */
extern "C" void emu_pack(const uint8_t *slab, uint64_t stride, const uint32_t *len, int n, uint8_t *dst, uint64_t cap,
                         uint64_t *off_out, uint32_t *len_out, int slices) {
  hipemu::launch(dim3((unsigned)n, (unsigned)slices), dim3(256), 64,
                 [&] { achip::pack_frames_kernel(slab, stride, len, n, dst, cap, off_out, len_out); });
}

/* the batched row / pixel scatter of ingest (stream_kernels.hpp), with the launcher's (slices, rows, clients) grid */
extern "C" void emu_scatter_rows_batch(const uint8_t *staged, uint32_t n_clients, uint32_t max_rows, int slices) {
  hipemu::launch(dim3((unsigned)slices, max_rows, n_clients), dim3(256), 0,
                 [&] { achip::scatter_rows_batch_kernel(staged, n_clients); });
}

#include <vector>
#include "crc_kernels.hpp"

/* the span powers the product's launcher computes on the host */
static achip::CrcSpanPows span_pows(int rounds) { return achip::crc_span_pows((uint64_t)rounds * 4096u); }
/* 1: the span form finishes its frames in the same launch (the last span of a frame to arrive combines the registers), as
 * the product does behind a plan; 0: spans + crc32c_finish_kernel (the stand-alone entry points) */
static int g_crc_one_launch = 0;
extern "C" void emu_set_crc_one_launch(int on) { g_crc_one_launch = on; }
static achip::CrcFinish span_finish(uint32_t *counters, int rounds, uint64_t v_bytes, const uint32_t *dims, uint32_t *crc_out, uint8_t *hdr_out,
                                    uint32_t *pkt_out) {
  achip::CrcFinish fin;
  fin.counters = counters;
  fin.cp = span_pows(rounds);
  fin.xinv_v = achip::crc_pow(achip::CRC_XINV8, v_bytes);
  fin.dims = dims;
  fin.crc_out = crc_out;
  fin.hdr_out = hdr_out;
  fin.pkt_crc_out = pkt_out;
  return fin;
}

/* quant16 / ansi16_rgb (render_kernels.hpp: the 16-colour table's structure instead of a walk over it) against the walk itself --
 * rgb_to_16color's loop, ansi.c:437-477, restated here -- for ALL 2^24 colours; returns the number of colours that differ */
extern "C" long emu_quant16_check(uint32_t *first_bad) {
  static const uint8_t tbl[16][3] = {{0, 0, 0},       {128, 0, 0},   {0, 128, 0},   {128, 128, 0}, {0, 0, 128},   {128, 0, 128},
                                     {0, 128, 128},   {192, 192, 192}, {128, 128, 128}, {255, 0, 0},   {0, 255, 0},   {255, 255, 0},
                                     {0, 0, 255},     {255, 0, 255}, {0, 255, 255}, {255, 255, 255}};
  long bad = 0;
  for (uint32_t i = 0; i < 16; i++) {
    const uint32_t want = (uint32_t)tbl[i][0] | ((uint32_t)tbl[i][1] << 8) | ((uint32_t)tbl[i][2] << 16);
    if (achip::ansi16_rgb(i) != want && !bad++ && first_bad)
      *first_bad = 0xFF000000u | i;
  }
  for (uint32_t p = 0; p < (1u << 24); p++) {
    const int r = (int)(p & 0xFF), g = (int)((p >> 8) & 0xFF), b = (int)(p >> 16);
    int best = 0, best_d = 0x7FFFFFFF;
    for (int i = 0; i < 16; i++) {
      const int dr = r - tbl[i][0], dg = g - tbl[i][1], db = b - tbl[i][2];
      const int d = dr * dr + dg * dg + db * db;
      if (d < best_d) {
        best_d = d;
        best = i;
      }
    }
    if (achip::quant16(p) != (uint32_t)best && !bad++ && first_bad)
      *first_bad = p;
  }
  return bad;
}

/* rep_profitable (render_kernels.hpp: `run >= 6`) against the rule as the reference writes it (output_buffer.c:148-155) over the
 * first 2^22 runs and around every power of ten and of two up to 2^32 - 1; returns the number of runs that differ */
extern "C" long emu_rep_rule_check(void) {
  long bad = 0;
  for (uint32_t run = 0; run < (1u << 22); run++)
    bad += achip::rep_profitable(run) != achip::rep_profitable_as_written(run);
  for (uint64_t p10 = 1; p10 <= 10000000000ull; p10 *= 10)
    for (int64_t d = -12; d <= 12; d++) {
      const int64_t v = (int64_t)p10 + d;
      if (v >= 0 && v <= 0xFFFFFFFFll)
        bad += achip::rep_profitable((uint32_t)v) != achip::rep_profitable_as_written((uint32_t)v);
    }
  for (int sh = 0; sh < 32; sh++)
    for (int64_t d = -3; d <= 3; d++) {
      const int64_t v = ((int64_t)1 << sh) + d;
      if (v >= 0 && v <= 0xFFFFFFFFll)
        bad += achip::rep_profitable((uint32_t)v) != achip::rep_profitable_as_written((uint32_t)v);
    }
  bad += achip::rep_profitable(0xFFFFFFFFu) != achip::rep_profitable_as_written(0xFFFFFFFFu);
  return bad;
}

/* the launcher's geometry (hip_launch.hip: achip_launch_crc32c) restated for the emulator; force_parts > 1
 * sends small buffers through the multi-span path with spans of force_rounds * 4 KB */
extern "C" void emu_crc32c(const uint8_t *base, uint64_t stride, const uint32_t *len, uint32_t fixed_len, uint32_t max_len,
                           int n, int force_parts, int force_rounds, const uint32_t *dims, uint32_t *crc_out,
                           uint8_t *hdr_out, uint32_t *pkt_out) {
  int parts = max_len <= 32u * 4096u ? 1 : (int)(((uint64_t)max_len + 65535u) / 65536u);
  int rounds = 16;
  if (force_parts > 0) {
    parts = force_parts;
    rounds = force_rounds;
  }
  if (parts == 1) {
    const uint4 *ftab = frame_crc_tab_1024(); /* (a launch of its own: not from inside the one below) */
    hipemu::launch(dim3((unsigned)n), dim3(1024), achip::CrcLds::bytes, [&] {
      achip::crc32c_frame_kernel<1024>(base, stride, len, fixed_len, n, dims, crc_out, hdr_out, pkt_out,
                                       achip::CrcPack{nullptr, 0, nullptr, nullptr}, ftab);
    });
    return;
  }
  const uint64_t v_bytes = (uint64_t)parts * rounds * 4096u;
  const uint4 *stab = frame_crc_tab_256(); /* (a launch of its own: not from inside the ones below) */
  std::vector<uint32_t> partial((size_t)n * parts);
  std::vector<uint32_t> counters((size_t)n, 0u);
  const achip::CrcFinish fin = span_finish(g_crc_one_launch ? counters.data() : nullptr, rounds, v_bytes, dims, crc_out, hdr_out, pkt_out);
  hipemu::launch(dim3((unsigned)(n * parts)), dim3(256), achip::CrcLds::bytes, [&] {
    achip::crc32c_span_kernel<false>(base, stride, len, fixed_len, n, parts, rounds, partial.data(), stab, fin);
  });
  if (g_crc_one_launch) {
    for (uint32_t c : counters)
      if (c != 0u)
        abort(); /* every frame's counter is re-armed by its last arrival */
    return;
  }
  hipemu::launch(dim3((unsigned)n), dim3(64), ACHIP_FRAME_CRC_TAB_BYTES, [&] {
    achip::crc32c_finish_kernel(partial.data(), parts, span_pows(rounds), achip::crc_pow(achip::CRC_XINV8, v_bytes), len, fixed_len, n, dims,
                                crc_out, hdr_out, pkt_out, stab);
  });
}

/* ... with the slab compacted in the same pass (COPY instantiations); geometry as above */
extern "C" void emu_crc32c_pack(const uint8_t *base, uint64_t stride, const uint32_t *len, uint32_t max_len, int n,
                                int force_parts, int force_rounds, const uint32_t *dims, uint32_t *crc_out, uint8_t *hdr_out,
                                uint32_t *pkt_out, uint8_t *dst, uint64_t cap, uint64_t *off_out, uint32_t *len_out) {
  int parts = max_len <= 32u * 4096u ? 1 : (int)(((uint64_t)max_len + 65535u) / 65536u);
  int rounds = 16;
  if (force_parts > 0) {
    parts = force_parts;
    rounds = force_rounds;
  }
  const achip::CrcPack pack = {dst, cap, off_out, len_out};
  if (parts == 1) {
    const uint4 *ftab = frame_crc_tab_1024();
    hipemu::launch(dim3((unsigned)n), dim3(1024), achip::CrcLds::bytes, [&] {
      achip::crc32c_frame_kernel<1024, true>(base, stride, len, 0u, n, dims, crc_out, hdr_out, pkt_out, pack, ftab);
    });
    return;
  }
  const uint64_t v_bytes = (uint64_t)parts * rounds * 4096u;
  const uint4 *stab = frame_crc_tab_256(); /* (a launch of its own: not from inside the ones below) */
  std::vector<uint32_t> partial((size_t)n * parts);
  std::vector<uint32_t> counters((size_t)n, 0u);
  const achip::CrcFinish fin = span_finish(g_crc_one_launch ? counters.data() : nullptr, rounds, v_bytes, dims, crc_out, hdr_out, pkt_out);
  hipemu::launch(dim3((unsigned)(n * parts)), dim3(256), achip::CrcLds::bytes, [&] {
    achip::crc32c_span_kernel<true>(base, stride, len, 0u, n, parts, rounds, partial.data(), stab, fin, pack);
  });
  if (g_crc_one_launch) {
    for (uint32_t c : counters)
      if (c != 0u)
        abort();
    return;
  }
  hipemu::launch(dim3((unsigned)n), dim3(64), ACHIP_FRAME_CRC_TAB_BYTES, [&] {
    achip::crc32c_finish_kernel(partial.data(), parts, span_pows(rounds), achip::crc_pow(achip::CRC_XINV8, v_bytes), len, 0u, n, dims, crc_out,
                                hdr_out, pkt_out, stab);
  });
}

extern "C" uint32_t emu_crc_mulmod(uint32_t a, uint32_t b) { return achip::crc_mulmod(a, b); }
extern "C" uint32_t emu_crc_pow(uint32_t base, uint64_t n) { return achip::crc_pow(base, n); }
extern "C" uint32_t emu_crc_x8_pow2(int k) { return achip::CRC_X8_POW2[k]; }
