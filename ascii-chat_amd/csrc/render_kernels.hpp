/*
 * render_kernels.hpp -- MI355X (gfx950) kernels of the image -> ASCII/ANSI render path.
 *
 * One workgroup renders one frame end to end, fused:
 *
 *   A  gather    nearest-neighbour point samples straight from the source frame in HBM
 *                (reference: image_resize_interpolation, lib/video/rgba/image.c:267-328) -- the
 *                resized image is never materialised; samples + run keys are parked in LDS
 *   B  heads     run-head / ASCII-glyph bitmasks via wave64 ballots
 *   C  lengths   every cell computes the exact byte length of the token it owns
 *   D  scan      exclusive scan of token lengths (per-thread segment + wave64 shuffle scan + LDS carry)
 *   E  emit      every cell writes its token into an LDS ring; the ring is drained to HBM with
 *                16-byte coalesced stores
 *
 * The reference emits bytes with a sequential state machine (colour-change-only SGR, REP run-length
 * sequences, per-row resets, transparent half-block runs).  Here every piece of that state is
 * re-derived per cell from its own sample, its left neighbour / previous run head and the distance
 * to the next run head (bit scans over the ballot masks), so all cells of a chunk work in parallel
 * and the output is byte-identical.  Per-mode grammar and the reference lines it restates are cited
 * at each token function.
 *
 * Frames are processed in chunks of whole text rows (<= CAP cells); left padding is modelled as
 * pseudo-cells that emit one space each, top padding as a cooperative fill, so
 * ascii_pad_frame_width/_height (ascii.c:457-517, 902-941) cost no extra pass.
 *
 * The file is plain HIP C++; tests compile it against tests/hipemu/hip_emu.h (-DACHIP_HIPEMU) to run
 * the same source on the CPU test box.  The product library only ever contains the hipcc build.
 */
#pragma once

#ifdef ACHIP_HIPEMU
#include "hip_emu.h"
#define ACHIP_SMEM (hipemu::g_smem.data())
#else
#include <hip/hip_runtime.h>
/* The one dynamic-LDS block of the frame kernel.  Every LDS access below is derived from this symbol
 * (never from a pointer stored in a struct), so the compiler keeps the accesses in the LDS address
 * space: ds_read/ds_write instead of flat_load/flat_store. */
extern __shared__ __attribute__((aligned(16))) unsigned char achip_smem[];
#define ACHIP_SMEM achip_smem
#endif

#include <stdint.h>

#include "achip_types.h"

namespace achip {

/* pointers read out of descriptors are generic; tell the compiler they are global memory so that it
 * emits global_load (vmcnt only) rather than flat_load (vmcnt + lgkmcnt, shared with the LDS queue) */
#ifdef ACHIP_HIPEMU
#define ACHIP_GLOBAL
#else
#define ACHIP_GLOBAL __attribute__((address_space(1)))
#endif
struct __attribute__((packed)) unaligned_u32 {
  uint32_t v;
};

template <class T> __device__ inline T *lds_ptr(int byte_off) { return reinterpret_cast<T *>(ACHIP_SMEM + byte_off); }

/* one LDS byte store at (LDS byte address `addr`) + OFF; HI selects bits 23..16 of `v` instead of 7..0.
 * Written as asm so that neighbouring byte stores are never fused into a misaligned wide store. */
template <int OFF, bool HI> __device__ inline void lds_store_byte(uint32_t addr, uint32_t v) {
#ifdef ACHIP_HIPEMU
  ACHIP_SMEM[addr + OFF] = (unsigned char)(HI ? v >> 16 : v);
#else
  if (HI)
    asm volatile("ds_write_b8_d16_hi %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
  else
    asm volatile("ds_write_b8 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
#endif
}
/* LDS byte address of ACHIP_SMEM[0] (0 for a kernel without static LDS, but do not assume) */
__device__ inline uint32_t lds_base_addr() {
#ifdef ACHIP_HIPEMU
  return 0u;
#else
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(achip_smem);
#endif
}
/* all DS operations issued by inline asm must have landed before other waves read the ring */
__device__ inline void lds_store_fence() {
#ifndef ACHIP_HIPEMU
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

/* ------------------------------------------------------------------------------------------- */
/* wave64 primitives                                                                             */
/* ------------------------------------------------------------------------------------------- */
#ifdef ACHIP_HIPEMU
__device__ inline uint64_t wave_ballot(bool p) { return hipemu::ballot(p); }
__device__ inline uint32_t wave_shfl_up(uint32_t v, int d) {
  int l = hipemu::lane();
  return hipemu::shfl_from(v, l - d); /* src < 0 -> own value, like __shfl_up */
}
#else
__device__ inline uint64_t wave_ballot(bool p) { return __ballot(p); }
__device__ inline uint32_t wave_shfl_up(uint32_t v, int d) { return __shfl_up(v, d, 64); }
#endif

/* ------------------------------------------------------------------------------------------- */
/* per-pixel integer maps                                                                        */
/* ------------------------------------------------------------------------------------------- */
/* pixels are carried as 0x00BBGGRR (+ a mode-specific key in bits 31..24) */
__device__ inline uint32_t px_r(uint32_t p) { return p & 0xFFu; }
__device__ inline uint32_t px_g(uint32_t p) { return (p >> 8) & 0xFFu; }
__device__ inline uint32_t px_b(uint32_t p) { return (p >> 16) & 0xFFu; }
__device__ inline uint32_t px_rgb(uint32_t p) { return p & 0x00FFFFFFu; }
__device__ inline uint32_t px_key(uint32_t p) { return p >> 24; }

/* Y = (77R + 150G + 29B + 128) >> 8  (foreground.c:93; LUMA_* common.h:80-86) */
__device__ inline uint32_t luma601(uint32_t p) { return (77u * px_r(p) + 150u * px_g(p) + 29u * px_b(p) + 128u) >> 8; }

/* rgb_to_256color, lib/video/terminal/ansi.c:360-379 */
__device__ inline uint32_t quant256(uint32_t p) {
  const int r = (int)px_r(p), g = (int)px_g(p), b = (int)px_b(p);
  const int avg = (r + g + b) / 3;
  int dr = r - avg, dg = g - avg, db = b - avg;
  dr = dr < 0 ? -dr : dr;
  dg = dg < 0 ? -dg : dg;
  db = db < 0 ? -db : db;
  if (dr + dg + db < 30)
    return (uint32_t)(232 + (avg * 23) / 255);
  return (uint32_t)(16 + 36 * ((r * 5) / 255) + 6 * ((g * 5) / 255) + ((b * 5) / 255));
}

/* rgb_to_16color, ansi.c:437-477: first minimum of the squared distance to the 16 fixed colours */
__device__ inline uint32_t quant16(uint32_t p) {
  const int r = (int)px_r(p), g = (int)px_g(p), b = (int)px_b(p);
  /* packed 0xBBGGRR of the table at ansi.c:442-459 */
  const uint32_t tbl[16] = {0x000000u, 0x000080u, 0x008000u, 0x008080u, 0x800000u, 0x800080u, 0x808000u, 0xC0C0C0u,
                            0x808080u, 0x0000FFu, 0x00FF00u, 0x00FFFFu, 0xFF0000u, 0xFF00FFu, 0xFFFF00u, 0xFFFFFFu};
  uint32_t best = 0;
  int best_d = 0x7FFFFFFF;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int dr = r - (int)(tbl[i] & 0xFF), dg = g - (int)((tbl[i] >> 8) & 0xFF), db = b - (int)(tbl[i] >> 16);
    const int d = dr * dr + dg * dg + db * db;
    if (d < best_d) {
      best_d = d;
      best = (uint32_t)i;
    }
  }
  return best;
}

/* UTF-8 sequence length from the lead byte -- the palette parser's rule (common.c:397-410) */
__device__ inline uint32_t glyph_len(uint32_t g) {
  const uint32_t c = g & 0xFFu;
  return (c & 0xE0u) == 0xC0u ? 2u : (c & 0xF0u) == 0xE0u ? 3u : (c & 0xF8u) == 0xF0u ? 4u : 1u;
}

__device__ inline uint32_t digits_u32(uint32_t v) {
  uint32_t d = 1;
  if (v >= 10u) d = 2;
  if (v >= 100u) d = 3;
  if (v >= 1000u) d = 4;
  if (v >= 10000u) d = 5;
  if (v >= 100000u) d = 6;
  if (v >= 1000000u) d = 7;
  if (v >= 10000000u) d = 8;
  if (v >= 100000000u) d = 9;
  if (v >= 1000000000u) d = 10;
  return d;
}

/* rep_is_profitable, lib/video/ascii/output_buffer.c:148-155 */
__device__ inline bool rep_profitable(uint32_t run) {
  if (run <= 2u)
    return false;
  const uint32_t k = run - 1u;
  return k > digits_u32(k) + 3u;
}

/* ------------------------------------------------------------------------------------------- */
/* token sinks.  A token is a short sequence of FIELDS of <= 4 bytes (packed little-endian in a   */
/* u32).  One token body drives three sinks: CountSink (length pass), FastSink (token lies wholly  */
/* inside the ring window: plain LDS byte stores, constant-length fields fold to immediate-offset  */
/* stores) and ClipSink (token straddles a window edge or the ring's wrap point: per-byte checks). */
/* Measured on gfx950 (scripts/ubench/lds_unaligned.hip): a ds_write_b8 costs ~1.5 cycles per      */
/* wave-instruction per CU, a MISALIGNED ds_write_b32 ~16 -- so bytes are stored one by one        */
/* through a volatile pointer, which also stops the compiler from fusing them into wide stores.    */
/* ------------------------------------------------------------------------------------------- */
/* dec[v], v in 0..255: ASCII digits of v without leading zeros, first digit in the low byte, digit
 * count in bits 31..24 (the reference's dec3 table, lib/video/ascii/common.c:546-570) */
__device__ inline uint32_t dec_entry(uint32_t v) {
  const uint32_t d2 = (v * 41u) >> 12, r = v - 100u * d2, d1 = (r * 205u) >> 11, d0 = r - 10u * d1;
  if (d2)
    return (0x30u + d2) | ((0x30u + d1) << 8) | ((0x30u + d0) << 16) | (3u << 24);
  if (d1)
    return (0x30u + d1) | ((0x30u + d0) << 8) | (2u << 24);
  return (0x30u + d0) | (1u << 24);
}

template <class L> struct CountSink {
  uint32_t n;
  template <int K> __device__ inline void c(uint32_t) { n += (uint32_t)K; }
  __device__ inline void v4(uint32_t, uint32_t k) { n += k; }
  __device__ inline void num(uint32_t value, uint32_t) { n += (lds_ptr<const uint32_t>(L::o_dec)[value] >> 24) + 1u; }
};

template <class L> struct FastSink {
  uint32_t a; /* LDS byte address of the next byte; the token neither wraps nor leaves the window */
  template <int K> __device__ inline void c(uint32_t v) {
    const uint32_t w = v >> 8;
    lds_store_byte<0, false>(a, v);
    if (K > 1) lds_store_byte<1, false>(a, w);
    if (K > 2) lds_store_byte<2, true>(a, v);
    if (K > 3) lds_store_byte<3, true>(a, w);
    a += K;
  }
  __device__ inline void v4(uint32_t v, uint32_t k) { /* k in 1..4 */
    const uint32_t w = v >> 8;
    lds_store_byte<0, false>(a, v);
    if (k > 1u) lds_store_byte<1, false>(a, w);
    if (k > 2u) lds_store_byte<2, true>(a, v);
    if (k > 3u) lds_store_byte<3, true>(a, w);
    a += k;
  }
  __device__ inline void num(uint32_t value, uint32_t term) { /* 1-3 digits + terminator */
    const uint32_t e = lds_ptr<const uint32_t>(L::o_dec)[value], d = e >> 24;
    const uint32_t v = (e & 0x00FFFFFFu) | (term << (8u * d)), w = v >> 8;
    lds_store_byte<0, false>(a, v);
    lds_store_byte<1, false>(a, w);
    if (d > 1u) lds_store_byte<2, true>(a, v);
    if (d > 2u) lds_store_byte<3, true>(a, w);
    a += d + 1u;
  }
};

template <class L, uint32_t RING> struct ClipSink {
  uint32_t pos;    /* absolute stream offset of the next byte */
  uint32_t lo, hi; /* window of the stream currently backed by the ring */
  __device__ inline void put(uint32_t b) {
    if (pos >= lo && pos < hi)
      lds_ptr<unsigned char>(L::o_ring)[pos & (RING - 1u)] = (unsigned char)b;
    pos++;
  }
  template <int K> __device__ inline void c(uint32_t v) {
#pragma unroll
    for (int k = 0; k < K; k++)
      put((v >> (8 * k)) & 0xFFu);
  }
  __device__ inline void v4(uint32_t v, uint32_t k) {
    for (uint32_t j = 0; j < k; j++)
      put((v >> (8u * j)) & 0xFFu);
  }
  __device__ inline void num(uint32_t value, uint32_t term) {
    const uint32_t e = lds_ptr<const uint32_t>(L::o_dec)[value], d = e >> 24;
    v4((e & 0x00FFFFFFu) | (term << (8u * d)), d + 1u);
  }
};

/* ESC[38;2;R;G;Bm / ESC[48;2;R;G;Bm  (append_truecolor_fg/bg ansi.c:143-193; emit_set_fg/bg output_buffer.c:186-214) */
template <class S> __device__ inline void put_sgr_true(S &s, bool bg, uint32_t p) {
  s.template c<4>(bg ? 0x38345B1Bu : 0x38335B1Bu); /* ESC [ 3|4 8 */
  s.template c<3>(0x003B323Bu);                    /* ; 2 ;       */
  s.num(px_r(p), ';');
  s.num(px_g(p), ';');
  s.num(px_b(p), 'm');
}
/* ESC[38;5;Nm / ESC[48;5;Nm  (ansi.c:326-357) */
template <class S> __device__ inline void put_sgr_256(S &s, bool bg, uint32_t idx) {
  s.template c<4>(bg ? 0x38345B1Bu : 0x38335B1Bu);
  s.template c<3>(0x003B353Bu); /* ; 5 ; */
  s.num(idx, 'm');
}
/* fg 30-37/90-97, bg 40-47/100-107  (ansi.c:384-435) */
template <class S> __device__ inline void put_sgr_16(S &s, bool bg, uint32_t idx) {
  const uint32_t code = bg ? (idx < 8u ? 40u + idx : 92u + idx) : (idx < 8u ? 30u + idx : 82u + idx);
  s.template c<2>(0x5B1Bu);
  s.num(code, 'm');
}
template <class S> __device__ inline void put_reset(S &s) { s.template c<4>(0x6D305B1Bu); } /* ESC[0m */
/* emit_rep: ESC[<extra>b, extra <= 4095 (a run never exceeds one chunk row) */
template <class S> __device__ inline void put_rep(S &s, uint32_t extra) {
  s.template c<2>(0x5B1Bu);
  if (extra < 256u) {
    s.num(extra, 'b');
  } else {
    const uint32_t d3 = extra / 1000u, r3 = extra - d3 * 1000u, d2 = r3 / 100u, r2 = r3 - d2 * 100u, d1 = r2 / 10u,
                   d0 = r2 - d1 * 10u;
    if (d3) {
      s.template c<4>((0x30u + d3) | ((0x30u + d2) << 8) | ((0x30u + d1) << 16) | ((0x30u + d0) << 24));
      s.template c<1>('b');
    } else {
      s.template c<4>((0x30u + d2) | ((0x30u + d1) << 8) | ((0x30u + d0) << 16) | ((uint32_t)'b' << 24));
    }
  }
}
template <class S> __device__ inline void put_glyph(S &s, uint32_t g) { s.v4(g, glyph_len(g)); }

/* ------------------------------------------------------------------------------------------- */
/* bit scans over the ballot masks (64 cells per word)                                           */
/* ------------------------------------------------------------------------------------------- */
/* largest set bit index < i, or -1 */
__device__ inline int prev_set(const uint64_t *m, int i) {
  int w = i >> 6;
  uint64_t v = m[w] & ((1ull << (i & 63)) - 1ull);
  for (;;) {
    if (v)
      return (w << 6) + 63 - __clzll((long long)v);
    if (--w < 0)
      return -1;
    v = m[w];
  }
}
/* smallest set bit index > i; the caller guarantees a sentinel bit at index n */
__device__ inline int next_set(const uint64_t *m, int i) {
  int w = i >> 6;
  const int sh = (i & 63) + 1;
  uint64_t v = sh < 64 ? (m[w] >> sh) << sh : 0ull;
  for (;;) {
    if (v)
      return (w << 6) + __ffsll((unsigned long long)v) - 1;
    v = m[++w];
  }
}

/* ------------------------------------------------------------------------------------------- */
/* LDS carve-up                                                                                  */
/* ------------------------------------------------------------------------------------------- */
__host__ __device__ constexpr bool mode_is_halfblock(int m) {
  return m == ACHIP_MODE_HB_TRUE || m == ACHIP_MODE_HB_256 || m == ACHIP_MODE_HB_16 || m == ACHIP_MODE_HB_MONO;
}
__host__ __device__ constexpr bool mode_has_runs(int m) { return m == ACHIP_MODE_MONO || mode_is_halfblock(m); }
/* modes that end every text row with ESC[0m (P256/P16/PB/HT/H256/H16) */
__host__ __device__ constexpr bool mode_row_reset(int m) {
  return m == ACHIP_MODE_256_FG || m == ACHIP_MODE_16_FG || m == ACHIP_MODE_TRUE_BG || m == ACHIP_MODE_HB_TRUE ||
         m == ACHIP_MODE_HB_256 || m == ACHIP_MODE_HB_16;
}

template <int MODE, int BLOCK, int CAP, int RING> struct Lds {
  static constexpr int MASKW = CAP / 64 + 1;
  static constexpr int o_ring = 0;
  static constexpr int o_pixT = o_ring + RING;
  static constexpr int o_pixB = o_pixT + CAP * 4;
  static constexpr int o_off = o_pixB + (mode_is_halfblock(MODE) ? CAP * 4 : 0);
  static constexpr int o_hmask = o_off + (CAP + 4) * 4;
  static constexpr int o_amask = o_hmask + MASKW * 8;
  static constexpr int o_glyph = o_amask + MASKW * 8;
  static constexpr int o_glyph64 = o_glyph + 256 * 4;
  static constexpr int o_ramp = o_glyph64 + 64 * 4;
  static constexpr int o_dec = o_ramp + 64;
  static constexpr int o_wsum = o_dec + 256 * 4;
  static constexpr int bytes = o_wsum + (BLOCK / 64) * 4 + 16;
};

/* ------------------------------------------------------------------------------------------- */
/* sampling (R1) and the fused pixel-space composite (C2)                                        */
/* ------------------------------------------------------------------------------------------- */
__device__ inline uint32_t load_rgb(const uint8_t *__restrict__ src, int32_t stride_bytes, uint32_t x, uint32_t y) {
  const size_t a = (size_t)y * (size_t)stride_bytes + (size_t)x * 3u;
  const ACHIP_GLOBAL uint8_t *p = (const ACHIP_GLOBAL uint8_t *)src + a;
  if (a == 0) /* first pixel of the buffer: nothing in front of it to borrow a byte from */
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
  /* one (unaligned) dword covering the byte before the pixel and the pixel: never reads past the
   * last pixel of the buffer, and costs one VMEM instruction instead of three */
  return ((const ACHIP_GLOBAL unaligned_u32 *)(p - 1))->v >> 8;
}

/* pixel (X,Y) of the virtual W x 2H composite canvas (stream.c:664-779): the tile of the cell that
 * contains it, nearest-neighbour resized on the fly; black outside every tile. */
__device__ inline uint32_t sample_composite(const achip_composite_t *__restrict__ cgen, uint32_t X, uint32_t Y) {
  const ACHIP_GLOBAL achip_composite_t *c = (const ACHIP_GLOBAL achip_composite_t *)cgen;
  const int col = (int)X / c->cell_w, row = (int)Y / c->cell_h;
  if (col >= c->cols || row >= c->rows)
    return 0u;
  const int idx = row * c->cols + col;
  if (idx >= c->n_src)
    return 0u;
  const ACHIP_GLOBAL achip_comp_src_t *s = &c->s[idx];
  if (!s->src)
    return 0u;
  const int lx = (int)X - s->org_x, ly = (int)Y - s->org_y;
  if (lx < 0 || ly < 0 || lx >= s->tile_w || ly >= s->tile_h)
    return 0u;
  uint32_t sx = ((uint32_t)lx * s->x_ratio) >> 16, sy = ((uint32_t)ly * s->y_ratio) >> 16;
  sx = min(sx, (uint32_t)s->src_w - 1u);
  sy = min(sy, (uint32_t)s->src_h - 1u);
  return load_rgb(s->src, s->src_stride, sx, sy);
}

/* sample (x, y) of the out_w x out_h resized image that the reference would have built */
__device__ inline uint32_t sample_frame(const achip_frame_t &f, uint32_t x, uint32_t y) {
  uint32_t sx = (x * f.x_ratio) >> 16, sy = (y * f.y_ratio) >> 16;
  sx = min(sx, (uint32_t)f.src_w - 1u);
  sy = min(sy, (uint32_t)f.src_h - 1u);
  if (f.comp)
    return sample_composite(f.comp, sx, sy);
  return load_rgb(f.src, f.src_stride, sx, sy);
}

/* ------------------------------------------------------------------------------------------- */
/* per-chunk view handed to the token bodies                                                     */
/* ------------------------------------------------------------------------------------------- */
struct Chunk {
  int n;           /* cells in this chunk (pad pseudo-cells included) */
  int wp;          /* cells per text row = pad_left + out_w           */
  int pad_left;
  int r0;          /* first text row of the chunk                     */
  int rows;        /* text rows in the frame                          */
  bool carry_have; /* PT: an ASCII-glyph pixel exists before this chunk */
  uint32_t carry_rgb;
};

/* The token owned by cell i (text row r, column xp inside the padded row). */
template <int MODE, class L, class S> __device__ inline void emit_token(S &s, const Chunk &c, int i, int r, int xp) {
  const uint32_t *pixT = lds_ptr<const uint32_t>(L::o_pixT);
  const uint32_t *pixB = lds_ptr<const uint32_t>(L::o_pixB);
  const uint64_t *hmask = lds_ptr<const uint64_t>(L::o_hmask);
  const uint64_t *amask = lds_ptr<const uint64_t>(L::o_amask);
  const uint32_t *glyph = lds_ptr<const uint32_t>(L::o_glyph);
  const uint32_t *glyph64 = lds_ptr<const uint32_t>(L::o_glyph64);
  const uint8_t *ramp = lds_ptr<const uint8_t>(L::o_ramp);
  (void)pixB; (void)hmask; (void)amask; (void)glyph; (void)glyph64; (void)ramp;
  if (xp < c.pad_left) { /* ascii_pad_frame_width: pad_left spaces in front of every row */
    s.template c<1>(' ');
    return;
  }
  const uint32_t pt = pixT[i];

  if (MODE == ACHIP_MODE_TRUE_FG) {
    /* image_print_color + ansi_rle_add_pixel (foreground.c:268-303, ansi.c:261-300): ASCII glyph ->
     * SGR only when the colour differs from the previous ASCII-glyph pixel (state survives row ends);
     * any other glyph -> SGR always, state untouched. */
    const uint32_t g = glyph[luma601(pt)];
    bool sgr = true;
    if ((g & 0xFFu) < 128u) {
      const int j = prev_set(amask, i);
      if (j >= 0)
        sgr = px_rgb(pixT[j]) != px_rgb(pt);
      else if (c.carry_have)
        sgr = c.carry_rgb != px_rgb(pt);
    }
    if (sgr)
      put_sgr_true(s, false, pt);
    put_glyph(s, g);
  } else if (MODE == ACHIP_MODE_256_FG) { /* foreground.c:475-500 */
    put_sgr_256(s, false, quant256(pt));
    put_glyph(s, glyph[luma601(pt)]);
  } else if (MODE == ACHIP_MODE_16_FG) { /* foreground.c:584-612: glyph = cache[ramp[Y>>2]] (sic) */
    put_sgr_16(s, false, quant16(pt));
    put_glyph(s, glyph[ramp[luma601(pt) >> 2]]);
  } else if (MODE == ACHIP_MODE_TRUE_BG) { /* background.c:49-68 */
    const uint32_t Y = luma601(pt);
    put_sgr_true(s, true, pt);
    put_sgr_true(s, false, Y < 128u ? 0x00FFFFFFu : 0u);
    put_glyph(s, glyph[Y]);
  } else {
    /* run-structured modes: head h, end e, run = e - h */
    const bool is_head = (hmask[i >> 6] >> (i & 63)) & 1ull;
    const int h = is_head ? i : prev_set(hmask, i);
    const int e = next_set(hmask, i);
    const uint32_t run = (uint32_t)(e - h);
    const bool rep = rep_profitable(run);

    if (MODE == ACHIP_MODE_MONO) {
      /* image_print (foreground.c:86-127): key = ramp[Y>>2], glyph = cache64[key] (double mapping) */
      const uint32_t g = glyph64[px_key(pt)];
      if (is_head) {
        put_glyph(s, g);
        if (rep)
          put_rep(s, run - 1u);
      } else if (!rep) {
        put_glyph(s, g);
      }
    } else if (MODE == ACHIP_MODE_HB_MONO) {
      /* rgb_to_halfblocks_scalar (halfblock.c:203-275): 76/150/29 luminance, no rounding term */
      const uint32_t pb = pixB[i];
      const uint32_t lt = (76u * px_r(pt) + 150u * px_g(pt) + 29u * px_b(pt)) >> 8;
      const uint32_t lb = (76u * px_r(pb) + 150u * px_g(pb) + 29u * px_b(pb)) >> 8;
      if (lt < 16u && lb < 16u) {
        s.template c<1>(' ');
      } else if (is_head || !rep) {
        const uint32_t sh = lt >> 6; /* U+2591 U+2592 U+2593 U+2588 = E2 96 91|92|93|88 */
        s.template c<3>(0x0096E2u | ((sh == 3u ? 0x88u : 0x91u + sh) << 16));
        if (is_head && rep)
          put_rep(s, run - 1u);
      }
    } else {
      /* HT / H256 / H16 (halfblock.c:48-165, 297-524): transparency is decided by the run HEAD's raw
       * rgb; fg/bg SGRs only when they differ from the state left by the previous run in this row
       * (unset at row start and after a transparent run). */
      const uint32_t hT = pixT[h], hB = pixB[h];
      const bool transparent = (px_rgb(hT) | px_rgb(hB)) == 0u;
      bool state_set = false;
      uint32_t pT = 0, pB = 0;
      if (is_head && xp > c.pad_left) { /* not the first pixel cell of its row */
        const int p = prev_set(hmask, h);
        pT = pixT[p];
        pB = pixB[p];
        state_set = (px_rgb(pT) | px_rgb(pB)) != 0u;
      }
      if (transparent) {
        if (is_head && state_set)
          put_reset(s);
        s.template c<1>(' ');
      } else {
        if (is_head) {
          if (MODE == ACHIP_MODE_HB_TRUE) {
            if (!state_set || px_rgb(pT) != px_rgb(hT))
              put_sgr_true(s, false, hT);
            if (!state_set || px_rgb(pB) != px_rgb(hB))
              put_sgr_true(s, true, hB);
          } else if (MODE == ACHIP_MODE_HB_256) {
            if (!state_set || px_key(pT) != px_key(hT))
              put_sgr_256(s, false, px_key(hT));
            if (!state_set || px_key(pB) != px_key(hB))
              put_sgr_256(s, true, px_key(hB));
          } else {
            if (!state_set || px_key(pT) != px_key(hT))
              put_sgr_16(s, false, px_key(hT));
            if (!state_set || px_key(pB) != px_key(hB))
              put_sgr_16(s, true, px_key(hB));
          }
        }
        if (is_head || !rep) /* U+2580 upper half block = E2 96 80 */
          s.template c<3>(0x8096E2u);
        if (is_head && rep)
          put_rep(s, run - 1u);
      }
    }
  }

  /* end of a text row */
  if (xp == c.wp - 1) {
    if (mode_row_reset(MODE))
      put_reset(s);
    if (r < c.rows - 1)
      s.template c<1>('\n');
    else if (MODE == ACHIP_MODE_TRUE_FG)
      put_reset(s); /* ansi_rle_finish: the single trailing ESC[0m */
  }
}

/* i / wp via the per-frame magic multiplier (magic == 0 encodes wp == 1) */
__device__ inline int row_of(int i, uint32_t magic) { return magic ? (int)__umulhi((uint32_t)i, magic) : i; }

/* run key of a cell for head detection */
template <int MODE> __device__ inline bool same_run(const uint32_t *pixT, const uint32_t *pixB, int a, int b) {
  if (MODE == ACHIP_MODE_MONO)
    return px_key(pixT[a]) == px_key(pixT[b]);
  if (MODE == ACHIP_MODE_HB_TRUE || MODE == ACHIP_MODE_HB_MONO)
    return px_rgb(pixT[a]) == px_rgb(pixT[b]) && px_rgb(pixB[a]) == px_rgb(pixB[b]);
  return px_key(pixT[a]) == px_key(pixT[b]) && px_key(pixB[a]) == px_key(pixB[b]); /* HB_256 / HB_16 */
}

/* ------------------------------------------------------------------------------------------- */
/* the frame kernel                                                                              */
/* ------------------------------------------------------------------------------------------- */
template <int MODE, int BLOCK, int CAP, int RING>
__device__ inline void drain_ring(unsigned char *ring, uint8_t *__restrict__ out, uint32_t from, uint32_t to) {
  /* [from, to) are stream offsets, from is 16-byte aligned; full 16-byte groups go out as uint4 */
  const uint32_t vec_end = to & ~15u;
  for (uint32_t o = from + 16u * threadIdx.x; o < vec_end; o += 16u * BLOCK)
    *reinterpret_cast<uint4 *>(out + o) = *reinterpret_cast<const uint4 *>(ring + (o & (RING - 1u)));
}

/* optional per-phase cycle accounting (diagnostics: prof == NULL in production launches).
 * prof[frame*8 + k]: 0 setup+pad_top, 1 gather, 2 heads, 3 lengths, 4 scan, 5 emit tokens, 6 drain, 7 total */
#ifdef ACHIP_HIPEMU
__device__ inline unsigned long long cycle_now() { return 0ull; }
#else
__device__ inline unsigned long long cycle_now() { return (unsigned long long)clock64(); }
#endif
#define ACHIP_STAMP(slot)                                                                                              \
  do {                                                                                                                 \
    if (prof) {                                                                                                        \
      const unsigned long long t_now = cycle_now();                                                                    \
      t_acc[slot] += t_now - t_prev;                                                                                   \
      t_prev = t_now;                                                                                                  \
    }                                                                                                                  \
  } while (0)

template <int MODE, int BLOCK, int CAP, int RING>
__global__ void __launch_bounds__(BLOCK)
    render_frames_kernel(const achip_frame_t *__restrict__ frames, const achip_lut_t *__restrict__ lut,
                         uint8_t *__restrict__ out, uint64_t out_stride, uint32_t *__restrict__ out_len, int n_frames,
                         unsigned long long *__restrict__ prof) {
  using L = Lds<MODE, BLOCK, CAP, RING>;
  constexpr bool HB = mode_is_halfblock(MODE);
  constexpr int NW = BLOCK / 64;
  constexpr int SEG = CAP / BLOCK;
  static_assert(CAP % BLOCK == 0 && (RING & (RING - 1)) == 0 && RING % 16 == 0, "geometry");

  unsigned char *smem = ACHIP_SMEM;
  unsigned char *ring = smem + L::o_ring;
  uint32_t *pixT = reinterpret_cast<uint32_t *>(smem + L::o_pixT);
  uint32_t *pixB = reinterpret_cast<uint32_t *>(smem + L::o_pixB);
  uint32_t *off = reinterpret_cast<uint32_t *>(smem + L::o_off);
  uint64_t *hmask = reinterpret_cast<uint64_t *>(smem + L::o_hmask);
  uint64_t *amask = reinterpret_cast<uint64_t *>(smem + L::o_amask);
  uint32_t *glyph = reinterpret_cast<uint32_t *>(smem + L::o_glyph);
  uint32_t *glyph64 = reinterpret_cast<uint32_t *>(smem + L::o_glyph64);
  uint8_t *ramp = smem + L::o_ramp;
  uint32_t *dec = reinterpret_cast<uint32_t *>(smem + L::o_dec);
  uint32_t *wsum = reinterpret_cast<uint32_t *>(smem + L::o_wsum);

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int fidx = (int)blockIdx.x;
  if (fidx >= n_frames)
    return;
  achip_frame_t f = frames[fidx];
  if (f.src_stride == 0)
    f.src_stride = 3 * f.src_w;
  uint8_t *dst = out + (size_t)fidx * out_stride;

  const int wp = f.pad_left + f.out_w;
  const int rows = HB ? (f.out_h + 1) / 2 : f.out_h;
  if (f.out_w <= 0 || f.out_h <= 0 || f.src_w <= 0 || f.src_h <= 0 || f.pad_left < 0 || f.pad_top < 0 || wp > CAP ||
      (!f.src && !f.comp)) {
    if (tid == 0)
      out_len[fidx] = ACHIP_LEN_BADDESC;
    return;
  }

  /* glyph tables -> LDS; decimal table generated in place */
  for (int k = tid; k < 256; k += BLOCK) {
    glyph[k] = lut->glyph[k];
    dec[k] = dec_entry((uint32_t)k);
  }
  for (int k = tid; k < 64; k += BLOCK) {
    glyph64[k] = lut->glyph64[k];
    ramp[k] = lut->ramp[k];
  }

  /* i / wp == umulhi(i, magic) for i, wp <= CAP (i * wp < 2^32); wp == 1 would need magic 2^32 */
  const uint32_t wp_magic = wp > 1 ? (uint32_t)(0x100000000ull / (uint32_t)wp) + 1u : 0u;
  const int rows_per_chunk = max(1, CAP / wp);
  const uint32_t cap_bytes = out_stride > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)out_stride;
  const uint32_t ring_addr = lds_base_addr() + (uint32_t)L::o_ring;

  unsigned long long t_acc[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = prof ? cycle_now() : 0ull;
  const unsigned long long t_start = t_prev;

  uint32_t base = 0;    /* stream bytes produced before the current chunk */
  uint32_t flushed = 0; /* stream bytes already in HBM (multiple of 16)   */
  bool overflow = false;
  bool carry_have = false;
  uint32_t carry_rgb = 0;

  /* ascii_pad_frame_height: pad_top bare newlines */
  if (f.pad_top > 0) {
    const uint32_t total = (uint32_t)f.pad_top;
    if (total > cap_bytes)
      overflow = true;
    while (!overflow && base < total) {
      const uint32_t hi = min(total, flushed + (uint32_t)RING);
      for (uint32_t o = base + (uint32_t)tid; o < hi; o += BLOCK)
        ring[o & (RING - 1u)] = '\n';
      __syncthreads();
      drain_ring<MODE, BLOCK, CAP, RING>(ring, dst, flushed, hi);
      __syncthreads();
      flushed = hi & ~15u;
      base = hi;
    }
  }
  __syncthreads();
  ACHIP_STAMP(0);

  for (int r0 = 0; r0 < rows; r0 += rows_per_chunk) {
    const int r1 = min(rows, r0 + rows_per_chunk);
    const int n = (r1 - r0) * wp;

    /* ---- A: gather: all of a thread's samples are requested before any is consumed, so a thread
     * keeps up to 2*SEG sparse 64-byte-sector fetches in flight ---------------------------------- */
    {
      uint32_t gt[SEG], gb[SEG];
#pragma unroll
      for (int k = 0; k < SEG; k++) {
        const int i = tid + k * BLOCK;
        gt[k] = 0;
        gb[k] = 0;
        if (i < n) {
          const int rr = row_of(i, wp_magic);
          const int xp = i - rr * wp;
          if (xp >= f.pad_left) {
            const uint32_t x = (uint32_t)(xp - f.pad_left);
            const uint32_t r = (uint32_t)(r0 + rr);
            if (HB) {
              const uint32_t yt = 2u * r, yb = 2u * r + 1u;
              gt[k] = sample_frame(f, x, yt);
              /* odd height: the last text row's bottom half repeats the top (halfblock.c:81-88) */
              gb[k] = yb < (uint32_t)f.out_h ? sample_frame(f, x, yb) : 0xFFFFFFFFu;
            } else {
              gt[k] = sample_frame(f, x, r);
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < SEG; k++) {
        const int i = tid + k * BLOCK;
        if (i < n) {
          uint32_t pt = gt[k], pb = gb[k] == 0xFFFFFFFFu ? gt[k] : gb[k];
          const int rr = row_of(i, wp_magic);
          const bool is_pixel = (i - rr * wp) >= f.pad_left;
          if (is_pixel) {
            if (MODE == ACHIP_MODE_HB_256) {
              pt |= quant256(pt) << 24;
              pb |= quant256(pb) << 24;
            } else if (MODE == ACHIP_MODE_HB_16) {
              pt |= quant16(pt) << 24;
              pb |= quant16(pb) << 24;
            } else if (MODE == ACHIP_MODE_MONO) {
              pt |= (uint32_t)ramp[luma601(pt) >> 2] << 24;
            }
          }
          pixT[i] = pt;
          if (HB)
            pixB[i] = pb;
        }
      }
    }
    __syncthreads();
    ACHIP_STAMP(1);

    /* ---- B: run heads / ASCII-glyph mask (one 64-cell word per wave step) -------------- */
    if (mode_has_runs(MODE) || MODE == ACHIP_MODE_TRUE_FG) {
      for (int w0 = wave; w0 <= (n >> 6); w0 += NW) {
        const int i = (w0 << 6) + lane;
        bool bit = false;
        if (i < n) {
          const int rr = row_of(i, wp_magic);
          const int xp = i - rr * wp;
          if (MODE == ACHIP_MODE_TRUE_FG)
            bit = xp >= f.pad_left && (glyph[luma601(pixT[i])] & 0xFFu) < 128u;
          else
            bit = xp <= f.pad_left || !same_run<MODE>(pixT, pixB, i, i - 1);
        } else if (i == n) {
          bit = mode_has_runs(MODE); /* sentinel head closes the last run */
        }
        const uint64_t m = wave_ballot(bit);
        if (lane == 0) {
          if (MODE == ACHIP_MODE_TRUE_FG)
            amask[w0] = m;
          else
            hmask[w0] = m;
        }
      }
      __syncthreads();
    }

    ACHIP_STAMP(2);
    Chunk c;
    c.n = n;
    c.wp = wp;
    c.pad_left = f.pad_left;
    c.r0 = r0;
    c.rows = rows;
    c.carry_have = carry_have;
    c.carry_rgb = carry_rgb;

    /* ---- C: token lengths ------------------------------------------------------------- */
    for (int i = tid; i < CAP; i += BLOCK) {
      uint32_t len = 0;
      if (i < n) {
        const int rr = row_of(i, wp_magic);
        CountSink<L> cs{0u};
        emit_token<MODE, L>(cs, c, i, r0 + rr, i - rr * wp);
        len = cs.n;
      }
      off[i] = len;
    }
    __syncthreads();
    ACHIP_STAMP(3);

    /* ---- D: exclusive scan of off[0..CAP) -------------------------------------------------- */
    uint32_t total;
    {
      uint32_t v[SEG];
      uint32_t sum = 0;
#pragma unroll
      for (int k = 0; k < SEG; k++) {
        v[k] = off[tid * SEG + k];
        sum += v[k];
      }
      uint32_t inc = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = wave_shfl_up(inc, d);
        if (lane >= d)
          inc += t;
      }
      if (lane == 63)
        wsum[wave] = inc;
      __syncthreads();
      uint32_t wbase = 0;
      total = 0;
#pragma unroll
      for (int k = 0; k < NW; k++) {
        const uint32_t t = wsum[k];
        if (k < wave)
          wbase += t;
        total += t;
      }
      uint32_t run = wbase + inc - sum;
#pragma unroll
      for (int k = 0; k < SEG; k++) {
        off[tid * SEG + k] = run;
        run += v[k];
      }
      if (tid == 0)
        off[CAP] = total;
      __syncthreads();
    }

    ACHIP_STAMP(4);
    if ((uint64_t)base + total > cap_bytes)
      overflow = true;

    /* ---- E: emit through the ring, one window at a time ---------------------------------- */
    const uint32_t chunk_end = base + total;
    const bool last_chunk = r1 >= rows;
    while (!overflow) {
      const uint32_t lo = flushed, hi = flushed + (uint32_t)RING;
      for (int i = tid; i < n; i += BLOCK) {
        const uint32_t a = base + off[i];
        const uint32_t b = base + off[i + 1];
        if (b > a && a < hi && b > lo) {
          const int rr = row_of(i, wp_magic);
          const uint32_t ra = a & (RING - 1u);
          if (a >= lo && b <= hi && ra + (b - a) <= (uint32_t)RING) {
            FastSink<L> fs{ring_addr + ra};
            emit_token<MODE, L>(fs, c, i, r0 + rr, i - rr * wp);
          } else {
            ClipSink<L, RING> cs{a, lo, hi};
            emit_token<MODE, L>(cs, c, i, r0 + rr, i - rr * wp);
          }
        }
      }
      lds_store_fence();
      __syncthreads();
      ACHIP_STAMP(5);
      const uint32_t avail = min(chunk_end, hi);
      drain_ring<MODE, BLOCK, CAP, RING>(ring, dst, flushed, avail);
      if (last_chunk && avail == chunk_end) { /* frame tail: < 16 bytes, byte stores */
        for (uint32_t o = (avail & ~15u) + (uint32_t)tid; o < avail; o += BLOCK)
          dst[o] = ring[o & (RING - 1u)];
      }
      __syncthreads();
      ACHIP_STAMP(6);
      flushed = avail & ~15u;
      if (chunk_end <= hi)
        break;
    }

    /* PT: colour of the last ASCII-glyph pixel seen so far (RLE state crosses rows and chunks) */
    if (MODE == ACHIP_MODE_TRUE_FG) {
      const int j = prev_set(amask, n);
      if (j >= 0) {
        carry_have = true;
        carry_rgb = px_rgb(pixT[j]);
      }
    }
    base = chunk_end;
    __syncthreads(); /* pixT/off/masks are rewritten by the next chunk */
    ACHIP_STAMP(6);
  }
  if (prof && tid == 0) {
#pragma unroll
    for (int k = 0; k < 7; k++)
      prof[(size_t)fidx * 8u + (size_t)k] = t_acc[k];
    prof[(size_t)fidx * 8u + 7u] = cycle_now() - t_start;
  }

  if (tid == 0) {
    out_len[fidx] = overflow ? ACHIP_LEN_OVERFLOW : base;
    if (!overflow && (uint64_t)base < out_stride)
      dst[base] = 0; /* NUL after the frame when the slot has room, as the reference's strings carry */
  }
}

/* ------------------------------------------------------------------------------------------- */
/* stand-alone image_resize (lib/video/rgba/image.c:256-328): writes the resized RGB24 image       */
/* ------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
    resize_nn_kernel(const uint8_t *__restrict__ src, int sw, int sh, int src_stride, uint8_t *__restrict__ dst, int dw,
                     int dh, uint32_t x_ratio, uint32_t y_ratio) {
  const uint32_t total = (uint32_t)dw * (uint32_t)dh;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t y = i / (uint32_t)dw, x = i - y * (uint32_t)dw;
    uint32_t sx = (x * x_ratio) >> 16, sy = (y * y_ratio) >> 16;
    sx = min(sx, (uint32_t)sw - 1u);
    sy = min(sy, (uint32_t)sh - 1u);
    const uint32_t p = load_rgb(src, src_stride, sx, sy);
    uint8_t *d = dst + (size_t)i * 3u;
    d[0] = (uint8_t)p;
    d[1] = (uint8_t)(p >> 8);
    d[2] = (uint8_t)(p >> 16);
  }
}

/* materialise the W x 2H composite canvas (only needed by callers that want the image itself) */
__global__ void __launch_bounds__(256)
    composite_kernel(const achip_composite_t *__restrict__ comp, uint8_t *__restrict__ dst) {
  const uint32_t total = (uint32_t)comp->canvas_w * (uint32_t)comp->canvas_h;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t y = i / (uint32_t)comp->canvas_w, x = i - y * (uint32_t)comp->canvas_w;
    const uint32_t p = sample_composite(comp, x, y);
    uint8_t *d = dst + (size_t)i * 3u;
    d[0] = (uint8_t)p;
    d[1] = (uint8_t)(p >> 8);
    d[2] = (uint8_t)(p >> 16);
  }
}

} // namespace achip
