/*
 * render_kernels.hpp -- MI355X (gfx950) kernels of the image -> ASCII/ANSI render path.
 *
 * One workgroup renders one frame -- or one band of text rows of it (small batches) -- end to end, fused:
 *
 *   A  gather    nearest-neighbour point samples straight from the source frame in HBM
 *                (reference: image_resize_interpolation, lib/video/rgba/image.c:267-328) -- the
 *                resized image is never materialised; samples are requested as raw dwords (all of a
 *                thread's requests in flight together, the next chunk's during this chunk's B-E) and
 *                parked in LDS with their run keys
 *   B  heads     run-head / ASCII-glyph bitmasks via wave64 ballots
 *   C  tokens    every cell builds the register-resident descriptor of the token it owns and its
 *                exact byte length
 *   D  scan      exclusive scan of token lengths (DPP wave scans + one LDS exchange of wave totals);
 *                bands publish / look back their byte counts here
 *   E  emit      every cell writes its token into a linear LDS staging buffer (byte stores, or aligned
 *                atomic ORs of register-built dwords for 41-byte half-block tokens); windows cut at
 *                token boundaries are drained to HBM with 16-byte non-temporal stores
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
 * Everything written for the machine rather than in portable HIP C++ (inline DS instructions, DPP wave operations,
 * scoped atomics, cache-policy loads and stores) lives in gfx950_ops.hpp; this file has no conditional compilation.
 * The CPU tests compile the same source against tests/hipemu/gfx950_ops.hpp (a fiber emulation of that interface,
 * found first on their include path); the product library only ever contains the hipcc build.
 */
#pragma once

#include <gfx950_ops.hpp>
#include <stddef.h>
#include <stdint.h>

#include "achip_types.h"

#ifndef ACHIP_EMIT_OR_MODES
#define ACHIP_EMIT_OR_MODES (1 << 5) /* bit m set: mode m emits through PackSink; measured to pay for the 41-byte
                                         half-block truecolor tokens only (profiles/r01_emit_or.txt) */
#endif

namespace achip {

template <class T> __device__ inline T *lds_ptr(int byte_off) { return reinterpret_cast<T *>(ACHIP_SMEM + byte_off); }

/* one LDS byte store at (LDS byte address `addr`) + OFF; HI selects bits 23..16 of `v` instead of 7..0 */
template <int OFF, bool HI> __device__ inline void lds_store_byte(uint32_t addr, uint32_t v) {
#if defined(ACHIP_ABLATE) && ACHIP_ABLATE == 1
  keep_alive(addr, v); /* diagnostics build: keep the operands alive, issue no store */
  return;
#endif
  ds_store_byte<OFF, HI>(addr, v);
}
/* LDS atomic OR of an aligned dword (no return value): tokens built in registers are OR-ed into a pre-zeroed
 * staging buffer, so that neighbouring tokens can share a dword without byte stores */
__device__ inline void lds_or_u32(uint32_t addr, uint32_t v) {
#if defined(ACHIP_ABLATE) && ACHIP_ABLATE == 1
  keep_alive(addr, v);
  return;
#endif
  ds_or_u32(addr, v);
}

/* ------------------------------------------------------------------------------------------- */
/* inter-workgroup hand-off for frames rendered by several workgroups ("parts").  One naturally    */
/* aligned 8-byte word per part carries {launch epoch : 32, byte length of the part : 32}; it is     */
/* written by ONE agent-scope atomic store and polled with relaxed agent-scope loads (which bypass    */
/* the reader's L1), so the word is its own payload and needs no fence (MI355X guide, G16 form R2).  */
/* ------------------------------------------------------------------------------------------- */
__device__ inline void part_publish(unsigned long long *slot, uint32_t epoch, uint32_t value) {
  agent_store_u64(slot, ((unsigned long long)epoch << 32) | (unsigned long long)value);
}
/* value published for this launch, or 0xFFFFFFFF after ~0.2 s of polling (never spins unbounded) */
__device__ inline uint32_t part_wait(const unsigned long long *slot, uint32_t epoch) {
  for (int spin = 0; spin < (1 << 21); spin++) {
    const unsigned long long w = agent_load_u64(slot);
    if ((uint32_t)(w >> 32) == epoch)
      return (uint32_t)w;
    spin_nap<2>();
  }
  return 0xFFFFFFFFu;
}
#define ACHIP_PART_POISON 0xFFFFFFF1u /* a predecessor failed / timed out: propagate, do not emit */

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

/* rgb_to_16color, ansi.c:437-477: first minimum of the squared distance to the 16 fixed colours (table at ansi.c:442-459:
 *   0-6, 8: {0, 128}^3 with index = R | G << 1 | B << 2 (all three set: 8); 7: (192, 192, 192); 9-15: {0, 255}^3, 8 + the same bits)
 * -- found without walking the table (round 6: 144 -> ~45 vector instructions; the 16-colour modes run this once or twice per
 * cell, the Floyd-Steinberg renderer once per step of its serial chain).  The nearest colour of {0, 128}^3 and of {0, 255}^3 is
 * decided channel by channel (the squared distance is a sum over channels), and so is the lowest-index one among equals: a
 * channel exactly between 0 and 128 (64) keeps its bit clear, 127.5 is no byte.  The three candidates -- the dark cube's, the
 * bright cube's, colour 7 -- then compete as (distance << 4 | index): the smallest key is the smallest distance and, among
 * equals, the smallest index, which is what the reference's strict `<` over ascending indices keeps.  (Black belongs to both
 * cubes with the same key.)  Checked against the table walk for all 2^24 colours: tests/test_kernels_emulated.py. */
__device__ inline uint32_t quant16(uint32_t p) {
  const int r = (int)px_r(p), g = (int)px_g(p), b = (int)px_b(p);
  const bool r1 = r > 64, g1 = g > 64, b1 = b > 64;       /* nearer to 128 than to 0 */
  const bool r2 = r >= 128, g2 = g >= 128, b2 = b >= 128; /* nearer to 255 than to 0 */
  const int dr1 = r1 ? r - 128 : r, dg1 = g1 ? g - 128 : g, db1 = b1 ? b - 128 : b;
  const int dr2 = r2 ? r - 255 : r, dg2 = g2 ? g - 255 : g, db2 = b2 ? b - 255 : b;
  const int dr7 = r - 192, dg7 = g - 192, db7 = b - 192;
  const uint32_t bits1 = (r1 ? 1u : 0u) | (g1 ? 2u : 0u) | (b1 ? 4u : 0u), bits2 = (r2 ? 1u : 0u) | (g2 ? 2u : 0u) | (b2 ? 4u : 0u);
  const uint32_t k1 = ((uint32_t)(dr1 * dr1 + dg1 * dg1 + db1 * db1) << 4) | (bits1 == 7u ? 8u : bits1);
  const uint32_t k2 = ((uint32_t)(dr2 * dr2 + dg2 * dg2 + db2 * db2) << 4) | (bits2 ? 8u + bits2 : 0u);
  const uint32_t k7 = ((uint32_t)(dr7 * dr7 + dg7 * dg7 + db7 * db7) << 4) | 7u;
  return min(min(k1, k2), k7) & 15u;
}

/* packed 0xBBGGRR of ANSI colour idx (get_16color_rgb, ansi.c:480-509): the table's structure again */
__device__ inline uint32_t ansi16_rgb(uint32_t idx) {
  const uint32_t bits = idx & 7u; /* 1-6, 9-15: the channels that are set; 7: all at 192; 8: all at 128 */
  const uint32_t spread = (bits & 1u) | ((bits & 2u) << 7) | ((bits & 4u) << 14);
  return idx == 7u ? 0xC0C0C0u : idx == 8u ? 0x808080u : spread * (idx > 8u ? 0xFFu : 0x80u);
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

/* rep_is_profitable, lib/video/ascii/output_buffer.c:148-155: `run > 2 && k > digits(k) + 3` with k = run - 1 -- which is
 * `run >= 6` for every 32-bit run: k <= 4 has one digit and fails k > 4; k in 5..9 passes it; from there on k grows by a factor of ten
 * per digit and the bound by one (k >= 10^(d-1) > d + 3 for every d >= 2).  Twenty vector instructions per token less than
 * counting the digits (round 6; tests/test_kernels_emulated.py walks both over the boundaries and the first 2^22 runs). */
__device__ inline bool rep_profitable(uint32_t run) { return run >= 6u; }
/* (the rule as the reference writes it: the emulator's check compares the two) */
__device__ inline bool rep_profitable_as_written(uint32_t run) {
  if (run <= 2u)
    return false;
  const uint32_t k = run - 1u;
  return k > digits_u32(k) + 3u;
}

/* ------------------------------------------------------------------------------------------- */
/* token sinks.  A token is a short sequence of FIELDS of <= 4 bytes (packed little-endian in a   */
/* u32).  One field walk drives two sinks: CountSink (length) and FastSink (plain LDS byte stores,  */
/* constant-length fields fold to immediate-offset stores).  Windows of the output stream are cut   */
/* at token boundaries (phase E), so a token never straddles anything and no store is checked.       */
/* Measured on gfx950 (scripts/ubench/lds_unaligned.hip): a ds_write_b8 costs ~1.5 cycles per      */
/* wave-instruction per CU, a MISALIGNED ds_write_b32 ~16 -- so bytes are stored one by one, in    */
/* asm, which also stops the compiler from fusing them into misaligned wide stores.                */
/* ------------------------------------------------------------------------------------------- */
/* ASCII digits of v (0..255) without leading zeros, first digit in the low byte, digit count in bits
 * 31..24 -- the content of the reference's dec3 table (lib/video/ascii/common.c:546-570), computed in
 * ~10 VALU ops instead of being looked up: no LDS dependency in the store pass, no registers held */
__device__ inline uint32_t dec_entry(uint32_t v) {
  const uint32_t d2 = (v * 41u) >> 12, r = v - 100u * d2, d1 = (r * 205u) >> 11, d0 = r - 10u * d1;
  if (d2)
    return (0x30u + d2) | ((0x30u + d1) << 8) | ((0x30u + d0) << 16) | (3u << 24);
  if (d1)
    return (0x30u + d1) | ((0x30u + d0) << 8) | (2u << 24);
  return (0x30u + d0) | (1u << 24);
}

/* Decimal fields ("<1-3 digits><terminator>") are the variable-length part of every SGR.  The store
 * sink takes the digits from a 256-entry LDS table (entry = digits, first in the low byte, zero padded;
 * bits 31..24 = 8 * digit count, i.e. the shift that places the terminator) and always stores at least
 * the bytes that are certain to be overwritten later BY THE SAME LANE: DS operations of one wave execute
 * in order, so a 4-byte store whose tail is garbage is harmless as long as the following fields of the
 * same token cover that tail.  ROOM = number of bytes guaranteed to follow the field inside its token;
 * only tails that could reach past the token are predicated, and then by redirecting the store to a
 * dummy LDS byte (one v_cndmask) instead of branching on the exec mask. */
__device__ inline uint32_t dec_table_entry(uint32_t v) {
  const uint32_t e = dec_entry(v);
  return (e & 0x00FFFFFFu) | ((e >> 24) << 27);
}

struct CountSink {
  static constexpr bool FAST_DEC = false; /* (render_rows.hpp's sinks take a truecolor SGR's numbers from wider tables) */
  uint32_t n;
  __device__ inline uint32_t lookup(uint32_t v) const { return 1u + (v >= 10u) + (v >= 100u); } /* digit count */
  template <int K> __device__ inline void c(uint32_t) { n += (uint32_t)K; }
  __device__ inline void v4(uint32_t, uint32_t k) { n += k; }
  template <int ROOM> __device__ inline void num(uint32_t digits, uint32_t) { n += digits + 1u; }
};

template <int DEC_OFF, int DUMMY_OFF> struct FastSink {
  static constexpr bool FAST_DEC = false;
  uint32_t a;     /* LDS byte address of the next byte; the token neither wraps nor leaves the window */
  uint32_t dummy; /* LDS byte address that swallows predicated-off stores */
  __device__ inline uint32_t lookup(uint32_t v) const { return lds_ptr<const uint32_t>(DEC_OFF)[v]; }
  template <int K> __device__ inline void c(uint32_t v) {
    const uint32_t w = v >> 8;
    lds_store_byte<0, false>(a, v);
    if (K > 1) lds_store_byte<1, false>(a, w);
    if (K > 2) lds_store_byte<2, true>(a, v);
    if (K > 3) lds_store_byte<3, true>(a, w);
    a += K;
  }
  __device__ inline void v4(uint32_t v, uint32_t k) { /* k in 1..4, nothing guaranteed to follow */
    const uint32_t w = v >> 8;
    lds_store_byte<0, false>(a, v);
    lds_store_byte<0, false>(k > 1u ? a + 1u : dummy, w);
    lds_store_byte<0, true>(k > 2u ? a + 2u : dummy, v);
    lds_store_byte<0, true>(k > 3u ? a + 3u : dummy, w);
    a += k;
  }
  template <int ROOM> __device__ inline void num(uint32_t entry, uint32_t term) { /* 1-3 digits + terminator */
    const uint32_t sh = entry >> 24; /* 8 * digits */
    const uint32_t v = (entry & 0x00FFFFFFu) | (term << sh), w = v >> 8;
    lds_store_byte<0, false>(a, v);
    lds_store_byte<1, false>(a, w);
    if (ROOM >= 1)
      lds_store_byte<2, true>(a, v);
    else
      lds_store_byte<0, true>(sh >= 16u ? a + 2u : dummy, v);
    if (ROOM >= 2)
      lds_store_byte<3, true>(a, w);
    else
      lds_store_byte<0, true>(sh >= (ROOM == 1 ? 16u : 24u) ? a + 3u : dummy, w);
    a += (sh >> 3) + 1u;
  }
};

/* ------------------------------------------------------------------------------------------- */
/* SGRs built as words                                                                           */
/* ------------------------------------------------------------------------------------------- */
/* One truecolor SGR -- ESC[38;2;R;G;Bm / ESC[48;2;R;G;Bm, LONGB: with the half block U+2580 behind it -- built in registers
 * and OR-ed into a ZEROED staging area as aligned dwords; `p` = LDS byte address of its first byte, `pre` = its first four
 * bytes.  Returns its length in BITS.  The three fields come out of LDS as {text, 8 x length} (the G table's term carries
 * -32, the B tables' +88, so that the shifts and the total are single additions): X = R | G << 8 lr | B << 8 (lr + lg) in
 * three or four dwords, moved up behind the 7-byte prefix by constant funnel shifts, then moved up by the 1..4 bytes
 * between the dword in front of `p` and `p` by funnel shifts of one per-lane amount (1..4 and not 0..3: v_alignbit_b32
 * takes its amount modulo 32, and a move by a whole dword is the amount 0 -- the dword in front of an aligned `p` gets a
 * zero OR-ed in).  6 / 7 LDS instructions instead of 19 / 22 byte stores: the byte stores of a wave's tokens lie 20-40
 * bytes apart at pseudo-random banks -- in the rows kernel the LDS pipe was active for 65 % of a launch, more than half of
 * it bank conflicts (profiles/r05_k5_sampled_sq_counters.txt; A/B: profiles/r05_rows_word_emit_ab.txt). */
struct WordFields {
  uint2 r, g, b;
};
template <int WR, int WG, int WB> __device__ inline WordFields word_fields(uint32_t rgb) {
  return WordFields{lds_ptr<const uint2>(WR)[px_r(rgb)], lds_ptr<const uint2>(WG)[px_g(rgb)], lds_ptr<const uint2>(WB)[px_b(rgb)]};
}
/* the tables' entries for the value v: R fields, G fields, the B field with its 'm', the same with the half block behind it */
__device__ inline void word_table_entries(uint32_t v, uint2 &wr, uint2 &wg, uint2 &wm, uint2 &wmg) {
  const uint32_t e = dec_entry(v), nd = e >> 24, dg = e & 0x00FFFFFFu, l8 = 8u * (nd + 1u);
  const uint32_t fm = dg | ((uint32_t)'m' << (8u * nd));
  const uint64_t fmg = (uint64_t)fm | (0x8096E2ull << l8); /* + U+2580 = E2 96 80 */
  wr = make_uint2(dg | ((uint32_t)';' << (8u * nd)), l8);
  wg = make_uint2(wr.x, l8 - 32u);
  wm = make_uint2(fm, l8 + 88u);
  wmg = make_uint2((uint32_t)fmg, (uint32_t)(fmg >> 32) | ((l8 + 24u + 88u) << 24));
}
/* `n` dwords a[0..n) of a string at LDS byte address p, OR-ed in as n + 1 aligned dwords */
template <int N> __device__ inline void word_string(uint32_t p, const uint32_t (&a)[N]) {
  const uint32_t t = p - 1u, base = t & ~3u, sh = (t << 3) ^ 24u;
  static_assert(N >= 2 && N <= 6, "one OR per dword and one behind them");
  ds_or_u32_at<0>(base, alignbit(a[0], 0u, sh));
  ds_or_u32_at<4>(base, alignbit(a[1], a[0], sh));
  if constexpr (N > 2) ds_or_u32_at<8>(base, alignbit(a[N > 2 ? 2 : 0], a[1], sh));
  if constexpr (N > 3) ds_or_u32_at<12>(base, alignbit(a[N > 3 ? 3 : 0], a[N > 2 ? 2 : 0], sh));
  if constexpr (N > 4) ds_or_u32_at<16>(base, alignbit(a[N > 4 ? 4 : 0], a[N > 3 ? 3 : 0], sh));
  if constexpr (N > 5) ds_or_u32_at<20>(base, alignbit(a[N > 5 ? 5 : 0], a[N > 4 ? 4 : 0], sh));
  ds_or_u32_at<4 * N>(base, alignbit(0u, a[N - 1], sh));
}
/* the two halves of word_sgr<false> for callers that need an SGR's length long before they store it (the stream kernel's
 * lean loop reads the tables ONCE, in its length pass): the body "R;G;Bm" as up to twelve bytes in three dwords + the length
 * of the whole SGR in bits ... */
struct SgrBody {
  uint32_t x0, x1, x2, bits;
};
__device__ inline SgrBody sgr_body(const WordFields &w) {
  const uint2 r = w.r, g = w.g, b = w.b;
  const uint64_t rg = (uint64_t)g.x << r.y; /* r.y = 16, 24, 32 */
  const uint32_t sb = r.y + g.y;            /* 8 (lr + lg) - 32 = 0 .. 32 */
  const uint64_t bb = (uint64_t)b.x << sb;
  return SgrBody{r.x | (uint32_t)rg, (uint32_t)(rg >> 32) | (uint32_t)bb, (uint32_t)(bb >> 32), sb + b.y};
}
/* ... and its placement at LDS byte address p behind the prefix `pre` (ESC [ 3|4 8): six aligned dword ORs */
__device__ inline void sgr_place(uint32_t p, uint32_t pre, const SgrBody &s) {
  const uint32_t a[5] = {pre, (s.x0 << 24) | 0x003B323Bu /* ; 2 ; + the first digit */, alignbit(s.x1, s.x0, 8u),
                         alignbit(s.x2, s.x1, 8u), s.x2 >> 8};
  word_string<5>(p, a);
}
template <bool LONGB> __device__ inline uint32_t word_sgr(uint32_t p, const WordFields &w, uint32_t pre) {
  if (!LONGB) {
    const SgrBody s = sgr_body(w);
    sgr_place(p, pre, s);
    return s.bits;
  }
  const uint2 r = w.r, g = w.g, b = w.b;
  const uint64_t rg = (uint64_t)g.x << r.y; /* r.y = 16, 24, 32 */
  const uint32_t sb = r.y + g.y;            /* 8 (lr + lg) - 32 = 0 .. 32 */
  const uint32_t x0 = r.x | (uint32_t)rg;
  const uint64_t b0 = (uint64_t)b.x << sb, b1 = (uint64_t)(b.y & 0x00FFFFFFu) << sb;
  const uint32_t x1 = (uint32_t)(rg >> 32) | (uint32_t)b0;
  const uint32_t x2 = (uint32_t)(b0 >> 32) | (uint32_t)b1;
  const uint32_t x3 = (uint32_t)(b1 >> 32);
  const uint32_t a1 = (x0 << 24) | 0x003B323Bu; /* ; 2 ; + the first digit */
  const uint32_t a[6] = {pre, a1, alignbit(x1, x0, 8u), alignbit(x2, x1, 8u), alignbit(x3, x2, 8u), x3 >> 8};
  word_string<6>(p, a);
  return sb + (b.y >> 24);
}
/* FastSink whose truecolor SGRs leave as words (tables at WR / WG / WM); everything else of a token as FastSink's bytes.
 * No byte field with room behind it (FastSink::num<ROOM >= 1>) may precede an SGR in a token: its spill would be OR-ed
 * into.  (The 256-colour SGRs -- 9-11 bytes -- were measured the same way and stay bytes: 1080p -> 80x24 ANSI-256 6.49
 * against 6.55 us, 256-colour half blocks level too, profiles/r05_rows_word_emit_256_ab.txt.) */
template <int DEC_OFF, int DUMMY_OFF, int WR, int WG, int WM> struct WordSink : FastSink<DEC_OFF, DUMMY_OFF> {
  static constexpr bool FAST_DEC = WR >= 0;
  using FastSink<DEC_OFF, DUMMY_OFF>::a;
  template <int ROOM> __device__ inline void sgr_true(bool bg, uint32_t rgb) {
    const WordFields w = word_fields<(WR >= 0 ? WR : 0), (WG >= 0 ? WG : 0), (WM >= 0 ? WM : 0)>(rgb);
    a += word_sgr<false>(a, w, bg ? 0x38345B1Bu : 0x38335B1Bu) >> 3;
  }
};

/* PackSink: the token is assembled in a 64-bit register window and leaves it one ALIGNED dword at a time as
 * an LDS atomic OR into the pre-zeroed staging buffer (the drain re-zeroes what it flushes).  Compared with
 * FastSink a 41-byte half-block token costs 12 LDS instructions instead of 41 -- token stores are bound by LDS
 * bank conflicts (64 lanes land 19-41 bytes apart), so fewer instructions win even though each field costs a
 * few more VALU operations.  Every field is at most 4 bytes, so at most one dword completes per field. */
template <int DEC_OFF> struct PackSink {
  static constexpr bool FAST_DEC = false;
  uint32_t a;   /* LDS byte address (4-byte aligned) of the window's first byte */
  uint32_t nb;  /* bytes pending in the window, 0..3 between fields                */
  uint64_t acc; /* pending bytes, first in the low byte                            */
  __device__ inline explicit PackSink(uint32_t addr) : a(addr & ~3u), nb(addr & 3u), acc(0ull) {}
  __device__ inline uint32_t lookup(uint32_t v) const { return lds_ptr<const uint32_t>(DEC_OFF)[v]; }
  __device__ inline void put(uint32_t v, uint32_t k) { /* v holds exactly k <= 4 bytes (zero above) */
    acc |= (uint64_t)v << (8u * nb);
    nb += k;
    /* OR-ing the window's low dword is idempotent, so it goes out after every field whether complete or not
     * (no select); the window only advances when the dword is complete (nb >= 4, and nb < 8 always) */
    lds_or_u32(a, (uint32_t)acc);
    const uint32_t step = nb & 4u;
    acc >>= 8u * step;
    a += step;
    nb &= 3u;
  }
  template <int K> __device__ inline void c(uint32_t v) { put(K >= 4 ? v : v & ((1u << (8 * (K & 3))) - 1u), (uint32_t)K); }
  __device__ inline void v4(uint32_t v, uint32_t k) { put(k >= 4u ? v : v & ((1u << (8u * k)) - 1u), k); }
  template <int ROOM> __device__ inline void num(uint32_t entry, uint32_t term) { /* 1-3 digits + terminator */
    const uint32_t sh = entry >> 24; /* 8 * digits */
    put((entry & 0x00FFFFFFu) | (term << sh), (sh >> 3) + 1u);
  }
  __device__ inline void finish() { /* the last 0..3 bytes; the zeros above them leave the next token's bytes alone */
    lds_or_u32(a, (uint32_t)acc);
  }
};

/* ------------------------------------------------------------------------------------------- */
/* bit scans over the head / ASCII masks (64 cells per word)                                     */
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
         m == ACHIP_MODE_HB_256 || m == ACHIP_MODE_HB_16 || m == ACHIP_MODE_16_DITHER_BG;
}

#define ACHIP_COMP_LDS_BYTES 480 /* achip_composite_t (464 bytes) + the two reciprocals of cell_w / cell_h */

template <int MODE, int BLOCK, int CAP, int RING> struct Lds {
  static constexpr int MASKW = CAP / 64 + 1;
  static constexpr int SEG = CAP / BLOCK; /* cells per thread per chunk */
  static constexpr int NW = BLOCK / 64;
  static constexpr int o_ring = 0;
  static constexpr int o_pixT = o_ring + RING;
  static constexpr int o_pixB = o_pixT + CAP * 4;
  static constexpr int o_hmask = o_pixB + (mode_is_halfblock(MODE) ? CAP * 4 : 0);
  static constexpr int o_amask = o_hmask + MASKW * 8;
  static constexpr int o_glyph = o_amask + MASKW * 8;
  static constexpr int o_glyph64 = o_glyph + 256 * 4;
  static constexpr int o_ramp = o_glyph64 + 64 * 4;
  static constexpr int o_dec = o_ramp + 64;     /* 256 decimal-field entries */
  static constexpr int o_wsum = o_dec + 256 * 4; /* SEG*NW wave totals (<= 64) */
  static constexpr int o_flags = o_wsum + 64 * 4; /* [0] unused, [1],[2] window cut (ping-pong), [3] predecessors' byte
                                                     count (multi-part frames), [4] dummy store target */
  static constexpr int o_prof = o_flags + 32;     /* 8 x u64 diagnostics accumulators + the kernel's start stamp */
  static constexpr int o_carry = o_prof + 10 * 8;            /* dither: error sums entering the next row, 3 x int per column */
  static constexpr int o_comp = o_carry + (MODE == ACHIP_MODE_16_DITHER_BG ? CAP * 12 : 0); /* composite descriptor (COMP launches) */
  static constexpr int bytes = o_comp + ACHIP_COMP_LDS_BYTES;
  static_assert(SEG * NW <= 64, "wave-total table must fit one wave");
};

/* ------------------------------------------------------------------------------------------- */
/* sampling (R1) and the fused pixel-space composite (C2)                                        */
/* ------------------------------------------------------------------------------------------- */
/* A sample is requested as ONE unaligned dword and finished later (finish_rgb), so that nothing between the
 * requests of a thread consumes loaded data and all of them are in flight together (a use right behind the load
 * makes the compiler wait for every sample in turn).  kind says how the dword maps to the pixel. */
enum : uint32_t {
  RAW_FINAL = 0u, /* v is the pixel already (0x00BBGGRR)                                  */
  RAW_BACK = 1u,  /* v = byte before the pixel + the pixel: never past the last pixel      */
  RAW_FIRST = 2u, /* v = the buffer's first pixel + the byte behind it (buffer >= 2 pixels) */
  RAW_TOP = 3u    /* half-block bottom sample of an odd last row: repeats the top sample    */
};
__device__ inline uint32_t load_rgb_raw(const uint8_t *__restrict__ src, int32_t stride_bytes, uint32_t x, uint32_t y,
                                        bool single_pixel, uint32_t &kind, bool stream = false) {
  /* a frame is at most 3840x2160x3 = 24.9 MB: 32-bit offsets keep the address in (SGPR base + VGPR offset) form */
  const uint32_t a = y * (uint32_t)stride_bytes + x * 3u;
  const ACHIP_GLOBAL uint8_t *p = (const ACHIP_GLOBAL uint8_t *)src + a;
  if (single_pixel) { /* a 1x1 source is 3 bytes: no dword to read (uniform per frame) */
    kind = RAW_FINAL;
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
  }
  kind = a != 0u ? RAW_BACK : RAW_FIRST;
  /* samples a cache line apart or more share no line with their neighbours: a non-temporal load keeps them out of
   * the L2 (1080p -> 80 columns, 72-byte stride: 13.1 -> 12.0 us per 256 frames); closer samples do share lines
   * and want the cache (4K -> 200 / 400 columns get 3-5 % slower without it) -- profiles/r01_nontemporal.txt */
  if (stream) /* opaque(): keeps this load distinct -- merged with the cached one it would lose its hint */
    return load_u32_unaligned_nt((const uint8_t *)p - opaque(a != 0u ? 1u : 0u));
  return ((const ACHIP_GLOBAL unaligned_u32 *)(p - (a != 0u ? 1u : 0u)))->v;
}
__device__ inline uint32_t finish_rgb(uint32_t v, uint32_t kind) {
  return kind == RAW_BACK ? v >> 8 : (kind == RAW_FIRST ? v & 0x00FFFFFFu : v);
}
__device__ inline uint32_t load_rgb(const uint8_t *__restrict__ src, int32_t stride_bytes, uint32_t x, uint32_t y,
                                    bool single_pixel = false) {
  uint32_t kind;
  const uint32_t v = load_rgb_raw(src, stride_bytes, x, y, single_pixel, kind);
  return finish_rgb(v, kind);
}

/* apply_color_filter for one pixel (lib/video/rgba/color_filter.c:246-345; rgb_to_grayscale color_filter.h:172:
 * (77R + 150G + 29B) >> 8, no rounding term) */
__device__ inline uint32_t tint_pixel(uint32_t p, uint32_t ops) {
  const uint32_t gray = (77u * px_r(p) + 150u * px_g(p) + 29u * px_b(p)) >> 8;
  const uint32_t t = ops >> ACHIP_OP_TINT_SHIFT;
  uint32_t r, g, b;
  if (ops & ACHIP_OP_TINT_ON_WHITE) {
    r = (px_r(t) * (255u - gray) + 255u * gray) / 255u;
    g = (px_g(t) * (255u - gray) + 255u * gray) / 255u;
    b = (px_b(t) * (255u - gray) + 255u * gray) / 255u;
  } else {
    r = (px_r(t) * gray) / 255u;
    g = (px_g(t) * gray) / 255u;
    b = (px_b(t) * gray) / 255u;
  }
  return r | (g << 8) | (b << 16);
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
  return load_rgb(s->src, s->src_stride, sx, sy, s->src_w * s->src_h == 1);
}

/* The same sampler with the descriptor in LDS (byte offset O of the kernel's LDS block).  The global version above costs
 * every sample two integer divisions and a chain of DEPENDENT global loads (c->cell_w -> c->s[idx] -> pixel): three L2
 * round trips where the single-source sampler has one (r02: 42 G cells/s on the grid against 63-70 G on plain frames).
 * Here the workgroup copies the 464-byte descriptor into LDS once, in front of its first barrier (comp_stage), together
 * with the reciprocals of the two cell sizes; a sample is then two multiply-highs, three ds_read_b128 and ONE global
 * request, returned raw (kind) like every other sample so that nothing waits between requests. */
static_assert(sizeof(achip_composite_t) == 464 && sizeof(achip_comp_src_t) == 48, "LDS image of the composite descriptor");
template <int O, int BLOCK> __device__ inline void comp_stage(const achip_composite_t *__restrict__ cgen, int tid) {
  const ACHIP_GLOBAL uint32_t *g = (const ACHIP_GLOBAL uint32_t *)cgen;
  for (int k = tid; k < (int)(sizeof(achip_composite_t) / 4u); k += BLOCK)
    lds_ptr<uint32_t>(O)[k] = g[k];
  if (tid == 32 || tid == 33) { /* floor(2^32 / d) + 1: X / d == umulhi(X, m) or one less, fixed up by the user */
    const ACHIP_GLOBAL achip_composite_t *c = (const ACHIP_GLOBAL achip_composite_t *)cgen;
    const uint32_t d = (uint32_t)(tid == 32 ? c->cell_w : c->cell_h);
    lds_ptr<uint32_t>(O + (int)sizeof(achip_composite_t))[tid - 32] = d > 1u ? (uint32_t)(0x100000000ull / d) + 1u : 0u;
  }
}
__device__ inline uint32_t div_by_magic(uint32_t x, uint32_t d, uint32_t m) {
  if (m == 0u) /* d <= 1 */
    return x;
  const uint32_t q = __umulhi(x, m); /* floor(x / d) or one more (m overshoots 2^32 / d by less than one unit) */
  return q * d > x ? q - 1u : q;
}
/* the wave-uniform head of the staged descriptor, read ONCE per frame behind the staging barrier and kept in scalar
 * registers: read per sample, every field was its own dependent LDS round trip in front of the request */
struct CompHead {
  uint32_t cell_w, cell_h, cols, rows, n_src, m_w, m_h;
};
template <int O> __device__ inline CompHead comp_head() {
  const achip_composite_t *c = lds_ptr<const achip_composite_t>(O);
  const uint32_t *magic = lds_ptr<const uint32_t>(O + (int)sizeof(achip_composite_t));
  CompHead h;
  h.cell_w = (uint32_t)wave_uniform(c->cell_w);
  h.cell_h = (uint32_t)wave_uniform(c->cell_h);
  h.cols = (uint32_t)wave_uniform(c->cols);
  h.rows = (uint32_t)wave_uniform(c->rows);
  h.n_src = (uint32_t)wave_uniform(c->n_src);
  h.m_w = (uint32_t)wave_uniform((int)magic[0]);
  h.m_h = (uint32_t)wave_uniform((int)magic[1]);
  return h;
}
template <int O>
__device__ inline uint32_t sample_composite_lds(const CompHead &h, const achip_composite_t *__restrict__ cgen, uint32_t X,
                                                uint32_t Y, uint32_t &kind) {
  /* Straight-line: every lane runs the same instructions and issues exactly one request; the tile's 48-byte record
   * comes out of LDS as three 16-byte reads issued together.  A sample that falls outside every tile (the black
   * margins, empty cells) reads the descriptor's own zero padding word instead of branching around the load. */
  const uint32_t col = div_by_magic(X, h.cell_w, h.m_w), row = div_by_magic(Y, h.cell_h, h.m_h);
  const uint32_t idx = __umul24(row, h.cols) + col;
  const bool in_grid = (col < h.cols) & (row < h.rows) & (idx < h.n_src);
  const uint4 *t = lds_ptr<const uint4>(O + (int)offsetof(achip_composite_t, s)) + 3u * (in_grid ? idx : 0u);
  const uint4 q0 = t[0], q1 = t[1], q2 = t[2]; /* {src, src_w, src_h} {stride, -, tile_w, tile_h} {org_x, org_y, xr, yr} */
  const uint8_t *src = reinterpret_cast<const uint8_t *>((uint64_t)q0.x | ((uint64_t)q0.y << 32));
  /* unsigned: a sample left of / above the tile wraps to a huge value and fails the same comparison */
  const uint32_t lx = X - q2.x, ly = Y - q2.y;
  const bool valid = in_grid & ((q0.x | q0.y) != 0u) & (lx < q1.z) & (ly < q1.w);
  const uint32_t sx = min((lx * q2.z) >> 16, q0.z - 1u);
  const uint32_t sy = min((ly * q2.w) >> 16, q0.w - 1u);
  if (valid && q0.z * q0.w == 1u) { /* a 1x1 source is 3 bytes: no dword to read (practically never taken) */
    kind = RAW_FINAL;
    const ACHIP_GLOBAL uint8_t *p = (const ACHIP_GLOBAL uint8_t *)src;
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
  }
  const uint32_t a = sy * q1.x + __umul24(sx, 3u);
  const uint32_t back = a != 0u ? 1u : 0u;
  kind = (valid & (back != 0u)) ? RAW_BACK : RAW_FIRST;
  /* outside: the descriptor's _pad word (zero: achip_composite_setup clears it, composite_upload enforces it),
   * finished as RAW_FIRST = its low 24 bits */
  const uint8_t *zero = reinterpret_cast<const uint8_t *>(cgen) + offsetof(achip_composite_t, _pad);
  const ACHIP_GLOBAL uint8_t *p = (const ACHIP_GLOBAL uint8_t *)(valid ? src + (a - back) : zero);
  return ((const ACHIP_GLOBAL unaligned_u32 *)p)->v;
}

/* sample (x, y) of the out_w x out_h resized image that the reference would have built.
 * COMP selects the virtual-composite sampler at compile time (its own kernel instantiation), so the
 * common single-source kernels carry none of its address arithmetic. */
/* O_COMP >= 0: the composite descriptor was staged at that LDS offset (comp_stage, behind a barrier) */
template <bool COMP, int O_COMP = -1>
__device__ inline uint32_t sample_frame_raw(const achip_frame_t &f, uint32_t x, uint32_t y, uint32_t &kind,
                                            const CompHead *head = nullptr) {
  uint32_t sx = (x * f.x_ratio) >> 16, sy = (y * f.y_ratio) >> 16;
  sx = min(sx, (uint32_t)f.src_w - 1u);
  sy = min(sy, (uint32_t)f.src_h - 1u);
  kind = RAW_FINAL;
#if defined(ACHIP_ABLATE) && ACHIP_ABLATE == 3
  return (sx * 2654435761u + sy * 40503u) & 0x00FFFFFFu; /* diagnostics: no memory access */
#endif
  if (COMP && f.comp) {
    if (O_COMP >= 0)
      return sample_composite_lds<(O_COMP >= 0 ? O_COMP : 0)>(*head, f.comp, sx, sy, kind);
    return sample_composite(f.comp, sx, sy);
  }
  /* display-path flips folded in (uniform per frame): an index map */
  if (f.ops & ACHIP_OP_FLIP_X)
    sx = (uint32_t)f.src_w - 1u - sx;
  if (f.ops & ACHIP_OP_FLIP_Y)
    sy = (uint32_t)f.src_h - 1u - sy;
  return load_rgb_raw(f.src, f.src_stride, sx, sy, f.src_w * f.src_h == 1, kind,
                      f.x_ratio >= ((64u << 16) + 2u) / 3u /* horizontal sample stride >= 64 bytes */);
}
/* raw dword -> pixel, then the display path's colour filter (a per-sample map); composite samples come back
 * final and are not filtered, like the reference's server path */
template <bool COMP> __device__ inline uint32_t sample_finish(const achip_frame_t &f, uint32_t v, uint32_t kind) {
  uint32_t p = finish_rgb(v, kind);
  if ((f.ops & ACHIP_OP_TINT) && !(COMP && f.comp))
    p = tint_pixel(p, f.ops);
  return p;
}
template <bool COMP> __device__ inline uint32_t sample_frame(const achip_frame_t &f, uint32_t x, uint32_t y) {
  uint32_t kind;
  const uint32_t v = sample_frame_raw<COMP>(f, x, y, kind);
  return sample_finish<COMP>(f, v, kind);
}

/* ------------------------------------------------------------------------------------------- */
/* tokens.  build_token() takes every decision the reference's sequential emitters take for one   */
/* cell (all LDS look-ups happen here) and leaves a register-resident descriptor; token_fields()   */
/* walks the descriptor into a sink.  The descriptor is built once per cell and used for the       */
/* length pass and for the store pass(es).                                                         */
/* ------------------------------------------------------------------------------------------- */
enum : uint32_t {
  TF_PAD = 1u << 0,         /* padding pseudo-cell: one space                          */
  TF_RESET_PRE = 1u << 1,   /* ESC[0m before a transparent run (half-block)            */
  TF_SGR_FG = 1u << 2,      /* foreground SGR                                          */
  TF_SGR_BG = 1u << 3,      /* background SGR                                          */
  TF_SPACE = 1u << 4,       /* a space instead of a glyph (transparent half-block)     */
  TF_GLYPH = 1u << 5,       /* the glyph                                               */
  TF_REP = 1u << 6,         /* ESC[<rep>b after the glyph                              */
  TF_ROW_RESET = 1u << 7,   /* ESC[0m at the end of the text row                       */
  TF_NL = 1u << 8,          /* newline after the row                                   */
  TF_FINAL_RESET = 1u << 9, /* the single trailing ESC[0m of image_print_color         */
  TF_FG_WHITE = 1u << 10,   /* PB: contrasting foreground is white (else black)        */
  TF_FG_GIVEN = 1u << 11,   /* PB: foreground colour in Tok::fg (rainbow override)     */
};

struct Tok {
  uint32_t flags;
  uint32_t fg; /* 0x00BBGGRR (truecolor) or palette index / SGR code (256/16-colour) */
  uint32_t bg;
  uint32_t glyph;
  uint32_t rep;
};

struct Chunk {
  int n;           /* cells in this chunk (pad pseudo-cells included) */
  int wp;          /* cells per text row = pad_left + out_w           */
  int pad_left;
  int rows;        /* text rows in the frame                          */
  bool all_ascii;  /* PT: every glyph of the palette is a single ASCII byte */
  bool carry_have; /* PT: an ASCII-glyph pixel exists before this chunk     */
  uint32_t ops;    /* achip_frame_t.ops (the dithered renderer's style bits)  */
  uint32_t carry_rgb;
};

__device__ inline uint32_t sgr16_code(bool bg, uint32_t idx) { /* fg 30-37/90-97, bg 40-47/100-107 (ansi.c:384-435) */
  return bg ? (idx < 8u ? 40u + idx : 92u + idx) : (idx < 8u ? 30u + idx : 82u + idx);
}

/* the token owned by cell i of the chunk (text row r of the frame, rr within the chunk, column xp of the padded row) */
template <int MODE, class L> __device__ inline Tok build_token(const Chunk &c, int i, int r, int rr, int xp) {
  const uint32_t *pixT = lds_ptr<const uint32_t>(L::o_pixT);
  const uint32_t *pixB = lds_ptr<const uint32_t>(L::o_pixB);
  const uint64_t *hmask = lds_ptr<const uint64_t>(L::o_hmask);
  const uint64_t *amask = lds_ptr<const uint64_t>(L::o_amask);
  const uint32_t *glyph = lds_ptr<const uint32_t>(L::o_glyph);
  const uint32_t *glyph64 = lds_ptr<const uint32_t>(L::o_glyph64);
  const uint8_t *ramp = lds_ptr<const uint8_t>(L::o_ramp);
  (void)pixB; (void)hmask; (void)amask; (void)glyph; (void)glyph64; (void)ramp; (void)rr;

  Tok t;
  t.flags = 0;
  t.fg = t.bg = t.glyph = t.rep = 0;
  if (xp < c.pad_left) { /* ascii_pad_frame_width: pad_left spaces in front of every row */
    t.flags = TF_PAD;
    return t;
  }
  const uint32_t pt = pixT[i];

  if (MODE == ACHIP_MODE_TRUE_FG) {
    /* image_print_color + ansi_rle_add_pixel (foreground.c:268-303, ansi.c:261-300): ASCII glyph ->
     * SGR only when the colour differs from the previous ASCII-glyph pixel (state survives row ends);
     * any other glyph -> SGR always, state untouched. */
    const uint32_t g = glyph[luma601(pt)];
    bool sgr = true;
    if (c.all_ascii) {
      /* previous pixel in raster order: left neighbour, or the last pixel of the previous row */
      int j = xp > c.pad_left ? i - 1 : (rr > 0 ? i - 1 - c.pad_left : -1);
      if (j >= 0)
        sgr = px_rgb(pixT[j]) != px_rgb(pt);
      else if (c.carry_have)
        sgr = c.carry_rgb != px_rgb(pt);
    } else if ((g & 0xFFu) < 128u) {
      const int j = prev_set(amask, i);
      if (j >= 0)
        sgr = px_rgb(pixT[j]) != px_rgb(pt);
      else if (c.carry_have)
        sgr = c.carry_rgb != px_rgb(pt);
    }
    if (sgr) {
      t.flags |= TF_SGR_FG;
      t.fg = px_rgb(pt);
    }
    t.flags |= TF_GLYPH;
    t.glyph = g;
  } else if (MODE == ACHIP_MODE_256_FG) { /* foreground.c:475-500 */
    t.flags |= TF_SGR_FG | TF_GLYPH;
    t.fg = quant256(pt);
    t.glyph = glyph[luma601(pt)];
  } else if (MODE == ACHIP_MODE_16_FG) { /* foreground.c:584-612: glyph = cache[ramp[Y>>2]] (sic) */
    t.flags |= TF_SGR_FG | TF_GLYPH;
    t.fg = sgr16_code(false, quant16(pt));
    t.glyph = glyph[ramp[luma601(pt) >> 2]];
  } else if (MODE == ACHIP_MODE_TRUE_BG) { /* background.c:49-68 */
    const uint32_t Y = luma601(pt);
    t.flags |= TF_SGR_BG | TF_SGR_FG | TF_GLYPH | (Y < 128u ? TF_FG_WHITE : 0u);
    t.bg = px_rgb(pt);
    t.glyph = glyph[Y];
  } else if (MODE == ACHIP_MODE_16_DITHER_BG) {
    /* foreground.c:787-827: background = dithered colour (index left in bits 31..24 by the dither pass),
     * foreground = white on dark / black on bright, glyph = cache[Y] of the ORIGINAL pixel */
    const uint32_t idx = px_key(pt), pal = ansi16_rgb(idx);
    const uint32_t bl = (77u * px_r(pal) + 150u * px_g(pal) + 29u * px_b(pal)) / 256u;
    if (c.ops & ACHIP_OP_DITHER_FG) { /* the exported foreground-only forms (foreground.c:712-723, 809-819) */
      t.flags |= TF_SGR_FG | TF_GLYPH;
      t.fg = sgr16_code(false, idx);
      t.glyph = (c.ops & ACHIP_OP_DITHER_RAMP) ? glyph[ramp[luma601(pt) >> 2]] : glyph[luma601(pt)];
    } else {
      t.flags |= TF_SGR_BG | TF_SGR_FG | TF_GLYPH;
      t.bg = sgr16_code(true, idx);
      t.fg = bl < 127u ? 97u : 30u; /* append_16color_fg(15) / append_16color_fg(0) */
      t.glyph = glyph[luma601(pt)];
    }
  } else {
    /* run-structured modes: head h, end e, run = e - h */
    const bool is_head = (hmask[i >> 6] >> (i & 63)) & 1ull;
    const int h = is_head ? i : prev_set(hmask, i);
    const int e = next_set(hmask, i);
    const uint32_t run = (uint32_t)(e - h);
    const bool rep = rep_profitable(run);
    if (is_head && rep) {
      t.flags |= TF_REP;
      t.rep = run - 1u;
    }

    if (MODE == ACHIP_MODE_MONO) {
      /* image_print (foreground.c:86-127): key = ramp[Y>>2], glyph = cache64[key] (double mapping) */
      /* palettes of more than 64 characters: the reference reads past cache64[64] here (undefined); clamped */
      t.glyph = glyph64[min(px_key(pt), 63u)];
      if (is_head || !rep)
        t.flags |= TF_GLYPH;
    } else if (MODE == ACHIP_MODE_HB_MONO) {
      /* rgb_to_halfblocks_scalar (halfblock.c:203-275): 76/150/29 luminance, no rounding term */
      const uint32_t pb = pixB[i];
      const uint32_t lt = (76u * px_r(pt) + 150u * px_g(pt) + 29u * px_b(pt)) >> 8;
      const uint32_t lb = (76u * px_r(pb) + 150u * px_g(pb) + 29u * px_b(pb)) >> 8;
      if (lt < 16u && lb < 16u) {
        t.flags = TF_SPACE; /* no REP for padding */
      } else {
        const uint32_t sh = lt >> 6; /* U+2591 U+2592 U+2593 U+2588 = E2 96 91|92|93|88 */
        t.glyph = 0x0096E2u | ((sh == 3u ? 0x88u : 0x91u + sh) << 16);
        if (is_head || !rep)
          t.flags |= TF_GLYPH;
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
        t.flags = TF_SPACE | ((is_head && state_set) ? TF_RESET_PRE : 0u);
      } else {
        t.glyph = 0x8096E2u; /* U+2580 upper half block = E2 96 80 */
        if (is_head || !rep)
          t.flags |= TF_GLYPH;
        if (is_head) {
          if (MODE == ACHIP_MODE_HB_TRUE) {
            if (!state_set || px_rgb(pT) != px_rgb(hT)) {
              t.flags |= TF_SGR_FG;
              t.fg = px_rgb(hT);
            }
            if (!state_set || px_rgb(pB) != px_rgb(hB)) {
              t.flags |= TF_SGR_BG;
              t.bg = px_rgb(hB);
            }
          } else {
            const bool is256 = MODE == ACHIP_MODE_HB_256;
            if (!state_set || px_key(pT) != px_key(hT)) {
              t.flags |= TF_SGR_FG;
              t.fg = is256 ? px_key(hT) : sgr16_code(false, px_key(hT));
            }
            if (!state_set || px_key(pB) != px_key(hB)) {
              t.flags |= TF_SGR_BG;
              t.bg = is256 ? px_key(hB) : sgr16_code(true, px_key(hB));
            }
          }
        }
      }
    }
  }

  /* rainbow_replace_ansi_colors (color_filter.c:348-408) rewrites every ESC[38;2;..m of the finished frame */
  if ((MODE == ACHIP_MODE_TRUE_FG || MODE == ACHIP_MODE_HB_TRUE || MODE == ACHIP_MODE_TRUE_BG) &&
      (c.ops & ACHIP_OP_FG_OVERRIDE) && (t.flags & TF_SGR_FG)) {
    t.fg = c.ops >> ACHIP_OP_TINT_SHIFT;
    if (MODE == ACHIP_MODE_TRUE_BG)
      t.flags |= TF_FG_GIVEN;
  }

  /* end of a text row */
  if (xp == c.wp - 1) {
    if (mode_row_reset(MODE))
      t.flags |= TF_ROW_RESET;
    if (r < c.rows - 1)
      t.flags |= TF_NL;
    else if (MODE == ACHIP_MODE_TRUE_FG)
      t.flags |= TF_FINAL_RESET; /* ansi_rle_finish: the single trailing ESC[0m */
  }
  return t;
}

/* The payload of a token whose flags and REP count are known: colours and glyph follow from the cell's OWN sample (a
 * run head's SGRs carry its own colours).  The geometries that hold four or more cells per thread keep ONE packed word per
 * cell {flags:12, rep:12, length:6} between the length pass and the store pass and rebuild the rest here -- five
 * registers per cell put them over their 128-VGPR budget (scratch spills; scripts/isa_stats.py allows none). */
template <int MODE, class L> __device__ inline Tok token_payload(uint32_t flags, uint32_t rep, int i, uint32_t ops) {
  const uint32_t *glyph = lds_ptr<const uint32_t>(L::o_glyph);
  const uint32_t *glyph64 = lds_ptr<const uint32_t>(L::o_glyph64);
  const uint8_t *ramp = lds_ptr<const uint8_t>(L::o_ramp);
  (void)glyph; (void)glyph64; (void)ramp;
  Tok t;
  t.flags = flags;
  t.rep = rep;
  t.fg = t.bg = t.glyph = 0;
  if (flags & TF_PAD)
    return t;
  const uint32_t pt = lds_ptr<const uint32_t>(L::o_pixT)[i];
  const uint32_t pb = mode_is_halfblock(MODE) ? lds_ptr<const uint32_t>(L::o_pixB)[i] : 0u;
  const bool given = (ops & ACHIP_OP_FG_OVERRIDE) != 0u;
  if (MODE == ACHIP_MODE_TRUE_FG) {
    t.glyph = glyph[luma601(pt)];
    t.fg = given ? ops >> ACHIP_OP_TINT_SHIFT : px_rgb(pt);
  } else if (MODE == ACHIP_MODE_256_FG) {
    t.fg = quant256(pt);
    t.glyph = glyph[luma601(pt)];
  } else if (MODE == ACHIP_MODE_16_FG) {
    t.fg = sgr16_code(false, quant16(pt));
    t.glyph = glyph[ramp[luma601(pt) >> 2]];
  } else if (MODE == ACHIP_MODE_TRUE_BG) {
    t.bg = px_rgb(pt);
    t.glyph = glyph[luma601(pt)];
    t.fg = given ? ops >> ACHIP_OP_TINT_SHIFT : 0u;
  } else if (MODE == ACHIP_MODE_16_DITHER_BG) {
    const uint32_t idx = px_key(pt), pal = ansi16_rgb(idx);
    const uint32_t bl = (77u * px_r(pal) + 150u * px_g(pal) + 29u * px_b(pal)) / 256u;
    if (ops & ACHIP_OP_DITHER_FG) {
      t.fg = sgr16_code(false, idx);
      t.glyph = (ops & ACHIP_OP_DITHER_RAMP) ? glyph[ramp[luma601(pt) >> 2]] : glyph[luma601(pt)];
    } else {
      t.bg = sgr16_code(true, idx);
      t.fg = bl < 127u ? 97u : 30u;
      t.glyph = glyph[luma601(pt)];
    }
  } else if (MODE == ACHIP_MODE_MONO) {
    t.glyph = glyph64[min(px_key(pt), 63u)];
  } else if (MODE == ACHIP_MODE_HB_MONO) {
    const uint32_t lt = (76u * px_r(pt) + 150u * px_g(pt) + 29u * px_b(pt)) >> 8;
    const uint32_t sh = lt >> 6;
    t.glyph = 0x0096E2u | ((sh == 3u ? 0x88u : 0x91u + sh) << 16);
  } else {
    t.glyph = 0x8096E2u;
    if (MODE == ACHIP_MODE_HB_TRUE) {
      t.fg = given ? ops >> ACHIP_OP_TINT_SHIFT : px_rgb(pt);
      t.bg = px_rgb(pb);
    } else if (MODE == ACHIP_MODE_HB_256) {
      t.fg = px_key(pt);
      t.bg = px_key(pb);
    } else {
      t.fg = sgr16_code(false, px_key(pt));
      t.bg = sgr16_code(true, px_key(pb));
    }
  }
  return t;
}

/* ESC[38;2;R;G;Bm / ESC[48;2;R;G;Bm  (append_truecolor_fg/bg ansi.c:143-193; emit_set_fg/bg output_buffer.c:186-214).
 * ROOM = bytes of the same token guaranteed to follow the SGR. */
template <int ROOM, class S> __device__ inline void put_sgr_true(S &s, bool bg, uint32_t rgb) {
  if constexpr (S::FAST_DEC) {
    s.template sgr_true<ROOM>(bg, rgb);
    return;
  }
  const uint32_t e0 = s.lookup(px_r(rgb)), e1 = s.lookup(px_g(rgb)), e2 = s.lookup(px_b(rgb));
  s.template c<4>(bg ? 0x38345B1Bu : 0x38335B1Bu); /* ESC [ 3|4 8 */
  s.template c<3>(0x003B323Bu);                    /* ; 2 ;       */
  s.template num<2>(e0, ';');
  s.template num<2>(e1, ';');
  s.template num<ROOM>(e2, 'm');
}
/* ESC[38;5;Nm / ESC[48;5;Nm  (ansi.c:326-357) */
template <int ROOM, class S> __device__ inline void put_sgr_256(S &s, bool bg, uint32_t idx) {
  const uint32_t e = s.lookup(idx);
  s.template c<4>(bg ? 0x38345B1Bu : 0x38335B1Bu);
  s.template c<3>(0x003B353Bu); /* ; 5 ; */
  s.template num<ROOM>(e, 'm');
}
/* ESC[<code>m */
template <int ROOM, class S> __device__ inline void put_sgr_16(S &s, uint32_t code) {
  const uint32_t e = s.lookup(code);
  s.template c<2>(0x5B1Bu);
  s.template num<ROOM>(e, 'm');
}
template <class S> __device__ inline void put_reset(S &s) { s.template c<4>(0x6D305B1Bu); } /* ESC[0m */
/* emit_rep (output_buffer.c:157-164): ESC[<extra>b, extra <= 4095 (a run never exceeds one chunk row) */
template <class S> __device__ inline void put_rep(S &s, uint32_t extra) {
  s.template c<2>(0x5B1Bu);
  if (extra < 256u) {
    const uint32_t e = s.lookup(extra);
    s.template num<0>(e, 'b');
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

/* ascii_only: every glyph of the palette is one byte (uniform per frame) */
template <int MODE, class S> __device__ inline void token_fields(S &s, const Tok &t, bool ascii_only) {
  const uint32_t f = t.flags;
  if (f & TF_PAD) {
    s.template c<1>(' ');
    return;
  }
  if (mode_is_halfblock(MODE) && (f & TF_RESET_PRE))
    put_reset(s);
  if (MODE == ACHIP_MODE_TRUE_BG) { /* background SGR first, then the contrasting foreground (background.c:53-62) */
    put_sgr_true<3>(s, true, t.bg); /* the fixed foreground SGR follows */
    if (f & TF_FG_GIVEN) {
      put_sgr_true<1>(s, false, t.fg);
    } else if (f & TF_FG_WHITE) {
      s.template c<4>(0x38335B1Bu); /* ESC[38;2;255;255;255m */
      s.template c<4>(0x323B323Bu);
      s.template c<4>(0x323B3535u);
      s.template c<4>(0x323B3535u);
      s.template c<3>(0x006D3535u);
    } else {
      s.template c<4>(0x38335B1Bu); /* ESC[38;2;0;0;0m */
      s.template c<4>(0x303B323Bu);
      s.template c<4>(0x303B303Bu);
      s.template c<1>('m');
    }
  } else if (MODE == ACHIP_MODE_16_DITHER_BG) {
    if (f & TF_SGR_BG)
      put_sgr_16<3>(s, t.bg); /* the 5-byte foreground SGR follows */
    put_sgr_16<1>(s, t.fg);
  } else {
    if (f & TF_SGR_FG) {
      /* what is certain to follow a foreground SGR: the glyph -- 3 bytes in half-block modes, >= 1 otherwise */
      constexpr int ROOM = mode_is_halfblock(MODE) ? 3 : 1;
      if (MODE == ACHIP_MODE_TRUE_FG || MODE == ACHIP_MODE_HB_TRUE)
        put_sgr_true<ROOM>(s, false, t.fg);
      else if (MODE == ACHIP_MODE_256_FG || MODE == ACHIP_MODE_HB_256)
        put_sgr_256<ROOM>(s, false, t.fg);
      else if (MODE == ACHIP_MODE_16_FG || MODE == ACHIP_MODE_HB_16)
        put_sgr_16<ROOM>(s, t.fg);
    }
    if (mode_is_halfblock(MODE) && (f & TF_SGR_BG)) {
      if (MODE == ACHIP_MODE_HB_TRUE) /* a background SGR is always followed by the 3-byte half block */
        put_sgr_true<3>(s, true, t.bg);
      else if (MODE == ACHIP_MODE_HB_256)
        put_sgr_256<3>(s, true, t.bg);
      else if (MODE == ACHIP_MODE_HB_16)
        put_sgr_16<3>(s, t.bg);
    }
  }
  if (mode_is_halfblock(MODE) && (f & TF_SPACE))
    s.template c<1>(' ');
  if (f & TF_GLYPH) {
    if (mode_is_halfblock(MODE))
      s.template c<3>(t.glyph);
    else if (ascii_only)
      s.template c<1>(t.glyph);
    else
      s.v4(t.glyph, glyph_len(t.glyph));
  }
  if (mode_has_runs(MODE) && (f & TF_REP))
    put_rep(s, t.rep);
  if (f & TF_ROW_RESET)
    put_reset(s);
  if (f & TF_NL)
    s.template c<1>('\n');
  if (f & TF_FINAL_RESET)
    put_reset(s);
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
/* 16 output bytes to HBM: written once and never read back by the kernel, hence non-temporal (gfx950_ops.hpp) */
__device__ inline void store_out16(uint8_t *__restrict__ p, uint4 v) { store_u4_nt(p, v); }
/* the drains' lane -> group mapping starts at a multiple of this (16 = at the first whole group: the round-3 form, kept
 * for A/B builds: make EXTRA=-DACHIP_DRAIN_ALIGN=16u) */
#ifndef ACHIP_DRAIN_ALIGN
#define ACHIP_DRAIN_ALIGN 128u
#endif

/* the thread of drain_ring that reads the staging buffer's group 0 -- the one that may overwrite it (with the carry)
 * right behind the call without a barrier: thread 0 when the window starts with a partial head group (copied bytewise
 * by thread 0), else the thread whose group sits (from + dmis) mod 128 behind a line boundary */
__device__ inline uint32_t drain_group0_tid(const uint8_t *out, uint32_t from, uint32_t own_from) {
  const uint32_t dmis = (uint32_t)(uintptr_t)out & (ACHIP_DRAIN_ALIGN - 1u) & ~15u;
  return own_from > from ? 0u : ((from + dmis) & (ACHIP_DRAIN_ALIGN - 1u)) >> 4;
}

template <int BLOCK, bool REZERO = false>
__device__ inline void drain_ring(int ring_off, uint8_t *__restrict__ out, uint32_t from, uint32_t to, uint32_t own_from,
                                  bool flush_tail) {
  /* [from, to) are stream offsets, `from` is 16-byte aligned and sits at byte 0 of the staging buffer.  This
   * workgroup owns the bytes >= own_from (own_from > from only for the first window of a frame part whose
   * predecessor ended mid-group).  Whole 16-byte groups go out as uint4 (group g by thread g mod BLOCK); the
   * partial head group and, when flush_tail, the partial tail group go out as bytes. */
  /* REZERO: the staging buffer is filled by atomic ORs (PackSink) and must read as zero wherever the next
   * window's tokens land: every flushed byte is cleared behind the read */
  unsigned char *ring = lds_ptr<unsigned char>(ring_off);
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  const uint32_t vec_begin = (own_from + 15u) & ~15u;
  const uint32_t vec_end = to & ~15u;
  /* thread t takes the group 16 t bytes behind a 128-byte LINE boundary of the address (at most seven threads sit out
   * the first trip): every wave's store instruction covers whole lines (profiles/r04_rows_floor.txt) */
  const uint32_t dmis = (uint32_t)(uintptr_t)out & (ACHIP_DRAIN_ALIGN - 1u) & ~15u; /* (slots are 16-byte aligned on the GPU; the emulator's need not be) */
  uint32_t o = ((vec_begin + dmis) & ~(ACHIP_DRAIN_ALIGN - 1u)) + 16u * threadIdx.x; /* (+ dmis: never below zero) */
  const uint32_t vb = vec_begin + dmis, ve = vec_end + dmis;
  uint8_t *const outm = out - dmis;
  const unsigned char *const ringm = ring - dmis - from; /* ringm + o = ring + (o - dmis - from) */
  for (; o + 16u * BLOCK < ve; o += 32u * BLOCK) { /* two groups per trip: both LDS reads in flight */
    const bool first = o >= vb; /* (the second group of a trip lies 16 BLOCK bytes further on: always inside) */
    uint4 v0 = zero4;
    if (first)
      v0 = *reinterpret_cast<const uint4 *>(ringm + o);
    const uint4 v1 = *reinterpret_cast<const uint4 *>(ringm + o + 16u * BLOCK);
    if (REZERO) {
      if (first)
        *reinterpret_cast<uint4 *>(const_cast<unsigned char *>(ringm + o)) = zero4;
      *reinterpret_cast<uint4 *>(const_cast<unsigned char *>(ringm + o) + 16u * BLOCK) = zero4;
    }
#if defined(ACHIP_ABLATE) && ACHIP_ABLATE == 4
    asm volatile("" ::"v"(v0.x), "v"(v0.y), "v"(v1.z), "v"(v1.w)); /* diagnostics: no HBM writes */
#else
    if (first)
      store_out16(outm + o, v0);
    store_out16(outm + o + 16u * BLOCK, v1);
#endif
  }
  if (o >= vb && o < ve) {
    store_out16(outm + o, *reinterpret_cast<const uint4 *>(ringm + o));
    if (REZERO)
      *reinterpret_cast<uint4 *>(const_cast<unsigned char *>(ringm + o)) = zero4;
  }
  /* head: [own_from, min(vec_begin, to)), < 16 bytes of the buffer's group 0.  ONE thread touches group 0 in this call
   * (drain_group0_tid: it moves the carry there right after the call): with a head, thread 0, which copies it itself. */
  if (threadIdx.x == 0)
    for (uint32_t h = own_from; h < min(vec_begin, to); h++) {
      out[h] = ring[h - from];
      if (REZERO)
        ring[h - from] = 0;
    }
  if (flush_tail) /* tail: what lies beyond the last whole group (and beyond the head) */
    for (uint32_t t = max(vec_end, vec_begin) + threadIdx.x; t < to; t += BLOCK) {
      out[t] = ring[t - from];
      if (REZERO)
        ring[t - from] = 0;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* Floyd-Steinberg 16-colour dithering (rgb_to_16color_dithered, lib/video/terminal/ansi.c:511-583)  */
/* as a register-resident wavefront.  Error diffusion is serial along a row and from row to row, but  */
/* pixel (x, y) only needs (x-1, y) and (x+1, y-1): lane y of ONE wave walks row y two columns behind */
/* lane y-1.  The error a row sends to the row below never touches memory: after pixel x the lane     */
/* knows the complete sum entering (x-1, y+1) -- (e(x-2)*1)/16 + (e(x-1)*5)/16 + (e(x)*3)/16, each    */
/* term truncated like the reference's integer divisions -- and hands it to lane y+1 with one shuffle */
/* per channel, exactly when that lane starts (x-1, y+1).  Sweeps of 64 rows are chained through a    */
/* 3 x int per column LDS buffer, which also carries the state from chunk to chunk.                   */
/* ------------------------------------------------------------------------------------------- */
__device__ inline int fs_term(int e, int w) { return (e * w) / 16; } /* C division: truncates toward zero */

template <class L>
__device__ inline void dither16_rows(int lane, int chunk_rows, int wp, int pad_left, int out_w, int first_row,
                                     int frame_rows) {
  uint32_t *pixT = lds_ptr<uint32_t>(L::o_pixT);
  int *carry = lds_ptr<int>(L::o_carry);
  for (int rb = 0; rb < chunk_rows; rb += 64) {
    const int nrows = min(64, chunk_rows - rb);
    const bool row_ok = lane < nrows;
    const int row_base = (rb + lane) * wp + pad_left;
    int right[3] = {0, 0, 0};              /* (e(x-1)*7)/16 entering the next pixel of this row            */
    int pa[3] = {0, 0, 0}, pb[3] = {0, 0, 0}; /* partial sums entering columns x-1 and x of the row below  */
    int in[3] = {0, 0, 0};                 /* complete sum entering the pixel this lane processes next      */
    const int steps = out_w + 1 + 2 * (nrows - 1);
    /* (requesting a step's LDS reads -- the pixel, the first row's carried sums -- one step ahead was measured and is slower:
     * 41.6 -> 44.0 us per 256 frames, one launch at a time 78.8 -> 83.2; the chain is dependent arithmetic, not LDS latency) */
    for (int t = 0; t < steps; t++) {
      const int x = t - 2 * lane;
      const bool active = row_ok && x >= 0 && x <= out_w;
      int out[3] = {0, 0, 0};
      if (active) {
        int e[3] = {0, 0, 0};
        if (x < out_w) {
          if (lane == 0) { /* first row of the sweep: what the previous sweep / chunk left for this column */
            in[0] = carry[3 * x + 0];
            in[1] = carry[3 * x + 1];
            in[2] = carry[3 * x + 2];
          }
          const uint32_t p = pixT[row_base + x];
          const int v0 = (int)px_r(p) + in[0] + right[0];
          const int v1 = (int)px_g(p) + in[1] + right[1];
          const int v2 = (int)px_b(p) + in[2] + right[2];
          const uint32_t c0 = (uint32_t)min(255, max(0, v0)), c1 = (uint32_t)min(255, max(0, v1)),
                         c2 = (uint32_t)min(255, max(0, v2));
          const uint32_t idx = quant16(c0 | (c1 << 8) | (c2 << 16));
          const uint32_t pal = ansi16_rgb(idx);
          e[0] = v0 - (int)px_r(pal); /* the error uses the UNclamped value (ansi.c:541-543) */
          e[1] = v1 - (int)px_g(pal);
          e[2] = v2 - (int)px_b(pal);
          pixT[row_base + x] = px_rgb(p) | (idx << 24);
        }
        const bool has_right = x + 1 < out_w; /* the reference drops contributions that leave the image */
#pragma unroll
        for (int k = 0; k < 3; k++) {
          out[k] = pa[k] + fs_term(e[k], 3);                    /* completes column x-1 of the row below */
          pa[k] = pb[k] + fs_term(e[k], 5);                     /* column x   */
          pb[k] = has_right ? fs_term(e[k], 1) : 0;             /* column x+1 */
          right[k] = has_right ? fs_term(e[k], 7) : 0;
        }
        /* last row of the sweep: park the completed sums for the next sweep / chunk */
        if (lane == nrows - 1 && x >= 1 && first_row + rb + lane + 1 < frame_rows) {
          carry[3 * (x - 1) + 0] = out[0];
          carry[3 * (x - 1) + 1] = out[1];
          carry[3 * (x - 1) + 2] = out[2];
        }
      }
      /* hand the completed sums to the lane below: it starts column x-1 in the next step */
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int got = (int)wave_shfl_up((uint32_t)out[k], 1);
        if (lane > 0)
          in[k] = got;
      }
    }
  }
}

/* optional per-phase cycle accounting (diagnostics: prof == NULL in production launches).
 * prof[frame*8 + k]: 0 setup+pad_top, 1 gather, 2 heads, 3 tokens+lengths, 4 scan, 5 token stores, 6 drain, 7 total */
#define ACHIP_STAMP(slot)                                                                                              \
  do {                                                                                                                 \
    if (prof && tid == 0) { /* accumulators live in LDS: no registers are held for diagnostics */                     \
      unsigned long long *pa = lds_ptr<unsigned long long>(L::o_prof);                                                 \
      const unsigned long long t_now = cycle_now();                                                                    \
      pa[slot] += t_now - pa[7];                                                                                       \
      pa[7] = t_now;                                                                                                   \
    }                                                                                                                  \
  } while (0)

/* The 512- and 256-thread geometries exist to put two (four) workgroups on a CU; that needs <= 128 VGPRs, and the
 * non-half-block kernels sit right at that edge (125-130 depending on unrelated code motion: at 130 only ONE 512-thread
 * workgroup fits and the step time doubles).  Pin it.  The half-block kernels need 160-185 there and are never
 * launched in these geometries by the host policy. */
template <int MODE, int BLOCK, int CAP> struct MinWaves { /* (not the wide geometry: eight cells per thread, one workgroup per CU) */
  static constexpr int value = (BLOCK <= 512 && BLOCK >= 256 && CAP <= 2048 && !mode_is_halfblock(MODE)) ? 4 : 1;
};
#define ACHIP_PIN_OCCUPANCY(M, B, C) ACHIP_WAVES_PER_EU((MinWaves<M, B, C>::value))

template <int MODE, int BLOCK, int CAP, int RING, bool COMP, bool SPLIT = true>
__global__ void __launch_bounds__(BLOCK) ACHIP_PIN_OCCUPANCY(MODE, BLOCK, CAP)
    render_frames_kernel(const achip_frame_t *__restrict__ frames, const achip_lut_t *__restrict__ lut,
                         uint8_t *__restrict__ out, uint64_t out_stride, uint32_t *__restrict__ out_len, int n_frames,
                         unsigned long long *__restrict__ prof, int parts, int rows_per_part,
                         unsigned long long *__restrict__ part_sync, uint32_t epoch, achip_uniform_t uni) {
  /* parts == 1: one workgroup renders the whole frame.  parts > 1: workgroup (frame, part) renders text rows
   * [part*rows_per_part, ...) -- exactly one chunk -- and learns where its bytes start from the lengths its
   * predecessors publish in part_sync[frame*parts + q] (see part_publish / part_wait). */
  /* SPLIT = false instantiations serve whole-frame launches: the band plumbing folds away (measured: 3 % of
   * the 1080p -> 80x24 kernel, profiles/r01_ablation.txt) */
  if (!SPLIT)
    parts = 1;

  using L = Lds<MODE, BLOCK, CAP, RING>;
  constexpr bool HB = mode_is_halfblock(MODE);
  /* Samples of chunk c+1 are requested while chunk c is tokenised -- in the 1024 x 2 geometry only: its 2(+2)
   * samples per thread fit the 128-VGPR budget (4K -> 200x60: 58 -> 47 us, 4K -> 400x120 half-block: 314 -> 287 us,
   * profiles/r01_prefetch.txt); the smaller geometries would drop to half the waves per CU, and a row band is a
   * single chunk anyway. */
  /* (the emulated test build takes the request-ahead order in every geometry, so that tiny inputs exercise it --
   * except in the 512 x 4 geometry, which keeps its product configuration: PRE_ISSUE without PREFETCH) */
  /* (not the composite half-block instantiations: their four requests-ahead per thread are the registers that would
   * spill, and their sources -- composite tiles -- come out of the L2) */
  constexpr bool PREFETCH = ACHIP_EMULATED ? !SPLIT && !(BLOCK == 512 && CAP == 2048)
                                           : BLOCK == 1024 && CAP == 2048 && !SPLIT && !(COMP && HB);
  /* A wave waits AT the request until the memory pipeline has room for its lines, so the requests of a chunk are
   * not issued in one burst behind phase A's barrier but at four points of B/C: in the half-block modes every
   * thread issues one of its four requests per point (4K -> 400x120: 281 -> 266 us); in the other modes a
   * quarter of the waves (wave & 3) issues both of its requests per point while the rest tokenises
   * (4K -> 200x60: 46.9 -> 45.1 us; profiles/r01_prefetch.txt) */
  constexpr bool SPREAD = PREFETCH && HB;
  /* the first chunk's samples are requested in the prologue, ahead of the glyph-table wait: in the 1024 x 2 geometry
   * (same register budget argument) and in the 512 x 4 one, where it is worth 4-6 % with three launches in flight
   * (profiles/r01_overlap.txt) and costs no registers (125 instead of 128 VGPRs in truecolor-fg) */
  constexpr bool PRE_ISSUE = PREFETCH || (BLOCK == 512 && CAP == 2048 && !SPLIT);
  /* token stores as aligned atomic ORs of register-built dwords (PackSink) instead of byte stores (FastSink) */
  constexpr bool EMIT_OR = ((ACHIP_EMIT_OR_MODES) >> MODE) & 1;
  constexpr int NW = L::NW;
  constexpr int SEG = L::SEG;
  static_assert(CAP % BLOCK == 0 && RING % 16 == 0 && RING >= 256, "geometry");

  unsigned char *ring = lds_ptr<unsigned char>(L::o_ring);
  uint32_t *pixT = lds_ptr<uint32_t>(L::o_pixT);
  uint32_t *pixB = lds_ptr<uint32_t>(L::o_pixB);
  uint64_t *hmask = lds_ptr<uint64_t>(L::o_hmask);
  uint64_t *amask = lds_ptr<uint64_t>(L::o_amask);
  uint32_t *glyph = lds_ptr<uint32_t>(L::o_glyph);
  uint32_t *glyph64 = lds_ptr<uint32_t>(L::o_glyph64);
  uint8_t *ramp = lds_ptr<uint8_t>(L::o_ramp);
  uint32_t *wsum = lds_ptr<uint32_t>(L::o_wsum);
  uint32_t *flags = lds_ptr<uint32_t>(L::o_flags);

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int fidx = parts > 1 ? (int)blockIdx.x / parts : (int)blockIdx.x;
  const int part = parts > 1 ? (int)blockIdx.x - fidx * parts : 0;
  if (fidx >= n_frames)
    return;
  /* the glyph tables are requested before the descriptor is: the two fetches are one overlapped latency.  In the
   * 1024 x 2 geometry the first chunk's pixels are requested in the prologue as well (PRE_ISSUE below); the extra
   * live registers would push the small geometries over 128 VGPRs, i.e. to half the waves per CU. */
  constexpr int LUTN = (256 + BLOCK - 1) / BLOCK;
  uint32_t lut_g[LUTN];
#pragma unroll
  for (int k = 0; k < LUTN; k++)
    lut_g[k] = tid + k * BLOCK < 256 ? lut->glyph[tid + k * BLOCK] : 0u;
  const uint32_t lut_g64 = tid < 64 ? lut->glyph64[tid] : 0u;
  const uint32_t lut_ramp = tid < 64 ? lut->ramp[tid] : 0u;
  const bool ascii_only = (lut->flags & ACHIP_LUT_MULTIBYTE) == 0u;
  /* uniform batch: the descriptor came with the kernel arguments -- the first gather does not wait for a fetch */
  achip_frame_t f = uni.f;
  if (uni.enabled)
    f.src = uni.f.src + (int64_t)fidx * uni.src_pitch;
  else
    f = frames[fidx];
#if defined(ACHIP_ABLATE_NOOPS) /* diagnostics: cost of the folded display ops */
  f.ops = 0;
#endif
  if (f.src_stride == 0)
    f.src_stride = 3 * f.src_w;
  uint8_t *dst = out + (size_t)fidx * out_stride;

  const int wp = f.pad_left + f.out_w;
  const int rows = HB ? (f.out_h + 1) / 2 : f.out_h;
  if (f.out_w <= 0 || f.out_h <= 0 || f.src_w <= 0 || f.src_h <= 0 || f.pad_left < 0 || f.pad_top < 0 || wp > CAP ||
      (!f.src && !f.comp) || (!COMP && f.comp) || (parts > 1 && rows_per_part * wp > CAP)) {
    if (tid == 0 && part == 0)
      out_len[fidx] = ACHIP_LEN_BADDESC;
    if (tid == 0 && parts > 1)
      part_publish(&part_sync[(size_t)fidx * parts + part], epoch, ACHIP_PART_POISON);
    return;
  }
  const int row_begin = parts > 1 ? part * rows_per_part : 0;
  const int row_end = parts > 1 ? min(rows, row_begin + rows_per_part) : rows;
  if (row_begin >= rows)
    return; /* this frame has fewer parts than the launch provides: nobody waits for them */
  const bool last_part = row_end >= rows;

  if (prof && tid == 0) { /* (the start stamp lives in LDS too: two registers less across the whole kernel) */
    unsigned long long *pa = lds_ptr<unsigned long long>(L::o_prof);
    for (int k = 0; k < 7; k++)
      pa[k] = 0ull;
    pa[8] = pa[7] = cycle_now();
  }

  /* i / wp == umulhi(i, magic) for i, wp <= CAP (i * wp < 2^32); wp == 1 would need magic 2^32 */
  /* One float division instead of a 64-bit integer one (a ~150-instruction dependent chain in front of the
   * first gather): any m with 2^32/wp < m < 2^32/wp + 256 divides exactly for i * wp <= 2^24; the float quotient
   * is within 64 of the true one, so +65 lands inside (checked exhaustively for wp <= 4096, i <= 4096 in
   * tests/test_cabi_host.py). */
  const uint32_t wp_magic = wp > 1 ? (uint32_t)(4294967296.0f / (float)wp) + 65u : 0u;
  const int rows_per_chunk = max(1, row_of(CAP, wp_magic));
  const uint32_t cap_bytes = out_stride > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)out_stride;
  const uint32_t ring_addr = lds_base_addr() + (uint32_t)L::o_ring;
  const uint32_t dummy_addr = lds_base_addr() + (uint32_t)L::o_flags + 16u;

  uint32_t window_no = 0; /* parity selects the LDS slot that carries the window cut */
  uint32_t base = 0;    /* stream bytes produced before the current chunk */
  uint32_t flushed = 0; /* stream bytes already in HBM (multiple of 16)   */
  bool overflow = false;

  Chunk c;
  c.wp = wp;
  c.pad_left = f.pad_left;
  c.rows = rows;
  c.all_ascii = MODE == ACHIP_MODE_TRUE_FG && ascii_only;
  c.ops = f.ops;
  c.carry_have = false;
  c.carry_rgb = 0;
  if (MODE == ACHIP_MODE_TRUE_FG && row_begin > 0) {
    /* the RLE state entering this part is the colour of the last pixel of the previous text row
     * (all-ASCII palettes only: the host does not split frames otherwise) */
    c.carry_have = true;
    c.carry_rgb = px_rgb(sample_frame<COMP>(f, (uint32_t)f.out_w - 1u, (uint32_t)row_begin - 1u));
  }
  /* gather: all of a thread's samples of a chunk are requested before any is consumed, so a thread keeps
   * up to 2*SEG sparse fetches in flight; cell i_k = tid + k*BLOCK (lane <-> consecutive cells) */
  uint32_t gt[SEG], gb[SEG];
  uint32_t gkind = 0; /* 2 bits per sample: top of cell k at bit 2k, bottom at bit 2(SEG+k) */
  CompHead chead = {}; /* composite frames: filled behind the prologue's barrier, before any request */
  /* request ONE sample of cell k (top, or the half-block bottom row) of the chunk starting at text row row0 */
  auto gather_one = [&](int k, bool bottom, int row0, int cells) {
    const int i = tid + k * BLOCK;
    const int rr = row_of(i, wp_magic);
    const int xp = i - rr * wp;
    const int sh = 2 * (bottom ? SEG + k : k);
    gkind &= ~(3u << sh);
    if (bottom)
      gb[k] = 0;
    else
      gt[k] = 0;
    if (i < cells && xp >= f.pad_left) {
      const uint32_t x = (uint32_t)(xp - f.pad_left);
      const uint32_t r = (uint32_t)(row0 + rr);
      uint32_t kind = RAW_FINAL;
      if (!bottom) {
        gt[k] = sample_frame_raw<COMP, COMP ? L::o_comp : -1>(f, x, HB ? 2u * r : r, kind, &chead);
      } else if (2u * r + 1u < (uint32_t)f.out_h) {
        gb[k] = sample_frame_raw<COMP, COMP ? L::o_comp : -1>(f, x, 2u * r + 1u, kind, &chead);
      } else {
        kind = RAW_TOP; /* odd height: the last text row's bottom half repeats the top (halfblock.c:81-88) */
      }
      gkind |= kind << sh;
    }
  };
  auto gather_issue = [&](int row0, int cells) {
#pragma unroll
    for (int k = 0; k < SEG; k++) {
      gather_one(k, false, row0, cells);
      if (HB)
        gather_one(k, true, row0, cells);
    }
  };
  /* the first chunk's samples are requested before the glyph tables are waited for: with the descriptor in the
   * kernel arguments (uniform batches) nothing but address arithmetic stands between the launch and these requests,
   * and the table fetch, the LDS set-up and the first barrier all run under their latency */
  /* without request-ahead (512 x 4 geometry) only single-chunk frames do this: with the prologue request in front of
   * it the chunk loop of a multi-chunk frame runs 1.5x slower (4K -> 200x60, profiles/r01_overlap.txt) */
  /* (composite frames sample through the LDS copy of their descriptor, staged below: their first requests follow the
   * prologue's barrier) */
  const bool pre_issued = PRE_ISSUE && (PREFETCH || row_end - row_begin <= rows_per_chunk) && !(COMP && f.comp);
  if (COMP && f.comp)
    comp_stage<L::o_comp, BLOCK>(f.comp, tid);
  if (pre_issued)
    gather_issue(row_begin, (min(row_end, row_begin + rows_per_chunk) - row_begin) * wp);
  /* glyph tables -> LDS */
#pragma unroll
  for (int k = 0; k < LUTN; k++)
    if (tid + k * BLOCK < 256) {
      glyph[tid + k * BLOCK] = lut_g[k];
      lds_ptr<uint32_t>(L::o_dec)[tid + k * BLOCK] = dec_table_entry((uint32_t)(tid + k * BLOCK));
    }
  if (tid < 64) {
    glyph64[tid] = lut_g64;
    ramp[tid] = (uint8_t)lut_ramp;
  }
  if (tid == 0) {
    flags[1] = 0xFFFFFFFFu;
    flags[2] = 0xFFFFFFFFu;
  }
  if (MODE == ACHIP_MODE_16_DITHER_BG) { /* no error enters the first row */
    int *carry = lds_ptr<int>(L::o_carry);
    for (int k = tid; k < 3 * CAP; k += BLOCK)
      carry[k] = 0;
  }

  if (EMIT_OR) { /* the OR-filled staging buffer starts out zero (as far as this frame can reach); the drain keeps it so */
    const int groups = (int)(min((uint64_t)RING, out_stride + 32u) / 16u);
    for (int k = tid; k < groups; k += BLOCK)
      reinterpret_cast<uint4 *>(ring)[k] = make_uint4(0u, 0u, 0u, 0u);
    if (f.pad_top > 0 && part == 0)
      __syncthreads(); /* the newline fill below writes other threads' groups */
  }
  /* ascii_pad_frame_height: pad_top bare newlines, through the same staging buffer */
  if (f.pad_top > 0 && part == 0) {
    const uint32_t total = (uint32_t)f.pad_top;
    if (total > cap_bytes)
      overflow = true;
    while (!overflow && base < total) {
      const uint32_t lo = flushed;
      const uint32_t hi = min(total, lo + (uint32_t)RING);
      for (uint32_t o = base + (uint32_t)tid; o < hi; o += BLOCK)
        ring[o - lo] = '\n';
      __syncthreads();
      drain_ring<BLOCK, EMIT_OR>(L::o_ring, dst, lo, hi, lo, false);
      flushed = hi & ~15u;
      if ((uint32_t)tid == drain_group0_tid(dst, lo, lo) && flushed > lo) {
        for (uint32_t j = 0; j < hi - flushed; j++) {
          ring[j] = ring[flushed - lo + j];
          if (EMIT_OR)
            ring[flushed - lo + j] = 0;
        }
      }
      base = hi;
      __syncthreads();
    }
  }
  __syncthreads();
  if (COMP && f.comp)
    chead = comp_head<L::o_comp>();

  ACHIP_STAMP(0);

  uint32_t own_from = 0; /* first stream byte this workgroup owns (parts > 1: set once its predecessors are known) */
  for (int r0 = row_begin; r0 < row_end; r0 += rows_per_chunk) {
    const int r1 = min(row_end, r0 + rows_per_chunk);
    const int n = (r1 - r0) * wp;
    c.n = n;

    /* per-thread cell coordinates: cell i_k = tid + k*BLOCK (lane <-> consecutive cells keeps the LDS
     * byte stores of a wave spread over the banks) */
#define ACHIP_CELL_RR(k) row_of(tid + (k)*BLOCK, wp_magic)
#define ACHIP_CELL_XP(k) (tid + (k)*BLOCK - ACHIP_CELL_RR(k) * wp)

    /* ---- A: commit the chunk's samples (requested one chunk ahead) to LDS with the mode's run key in
     * bits 31..24 ------------------------------------------------------------------------------- */
    /* who requests this chunk's samples: the prologue (first chunk, PRE_ISSUE), the previous chunk (PREFETCH), or here */
    if (r0 == row_begin ? !pre_issued : !PREFETCH)
      gather_issue(r0, n);
#pragma unroll
    for (int k = 0; k < SEG; k++) {
      const int i = tid + k * BLOCK;
      if (i < n) {
        const uint32_t kt = (gkind >> (2 * k)) & 3u, kb = (gkind >> (2 * (SEG + k))) & 3u;
        uint32_t pt = gt[k], pb = gb[k];
        if (ACHIP_CELL_XP(k) >= f.pad_left) {
          pt = sample_finish<COMP>(f, pt, kt);
          pb = kb == RAW_TOP ? pt : sample_finish<COMP>(f, pb, kb);
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
    __syncthreads();
    /* request the next chunk's samples: the sparse fetches stay in flight during B-E of this chunk */
    const bool pf = PREFETCH && r1 < row_end;
    const int pf_cells = (min(row_end, r1 + rows_per_chunk) - r1) * wp;
    if (pf && SPREAD)
      gather_one(0, false, r1, pf_cells);
    else if (pf && (wave & 3) == 0)
      gather_issue(r1, pf_cells);
    ACHIP_STAMP(1);
    if (MODE == ACHIP_MODE_16_DITHER_BG) { /* one wave diffuses the errors and leaves the colour index in the key byte */
      if (wave == 0)
        dither16_rows<L>(lane, r1 - r0, wp, f.pad_left, f.out_w, r0, rows);
      __syncthreads();
    }

    /* ---- B: run heads / ASCII-glyph mask (one 64-cell word per wave step); skipped entirely for the
     * per-cell modes and for truecolor-fg with an all-ASCII palette ------------------------------- */
    if (mode_has_runs(MODE) || (MODE == ACHIP_MODE_TRUE_FG && !c.all_ascii)) {
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
    if (pf && SPREAD)
      gather_one(0, true, r1, pf_cells);
    else if (pf && (wave & 3) == 1)
      gather_issue(r1, pf_cells);

    /* ---- C: build the tokens (registers) and their lengths ------------------------------------ */
    /* four or more cells per thread: one packed word per cell instead of the token (token_payload rebuilds the rest) */
    constexpr bool PACK_TOK = SEG >= 4 || (HB && (COMP || MODE == ACHIP_MODE_HB_16)); /* (.. and the two-cell half-block
                                        instantiations that would otherwise spill a handful of registers) */
    Tok tok[PACK_TOK ? 1 : SEG];
    uint32_t meta[PACK_TOK ? SEG : 1];
    uint32_t len[SEG];
#pragma unroll
    for (int k = 0; k < SEG; k++) {
      const int i = tid + k * BLOCK;
      len[k] = 0;
      if (PACK_TOK)
        meta[k] = 0;
      else
        tok[k].flags = 0;
      if (i < n) {
        const Tok t = build_token<MODE, L>(c, i, r0 + ACHIP_CELL_RR(k), ACHIP_CELL_RR(k), ACHIP_CELL_XP(k));
        CountSink cs{0u};
        token_fields<MODE>(cs, t, ascii_only);
        len[k] = cs.n;
        if (PACK_TOK)
          meta[k] = t.flags | (t.rep << 12);
        else
          tok[k] = t;
      }
      if (pf && SPREAD && k + 1 < SEG)
        gather_one(k + 1, false, r1, pf_cells);
      if (pf && SPREAD && k + 1 == SEG) {
#pragma unroll
        for (int q = 1; q < SEG; q++)
          gather_one(q, true, r1, pf_cells);
      }
      if (pf && !SPREAD && k == 0 && (wave & 3) == 2)
        gather_issue(r1, pf_cells);
      if (pf && !SPREAD && k + 1 == SEG && (wave & 3) == 3)
        gather_issue(r1, pf_cells);
    }
    /* PT: colour of the last ASCII-glyph pixel seen so far (the RLE state crosses rows and chunks);
     * read before the barrier below, after which pixT may be overwritten by the next chunk's gather */
    if (MODE == ACHIP_MODE_TRUE_FG) {
      const int j = c.all_ascii ? n - 1 : prev_set(amask, n);
      if (j >= 0) {
        c.carry_have = true;
        c.carry_rgb = px_rgb(pixT[j]);
      }
    }
    ACHIP_STAMP(3);

    /* ---- D: exclusive scan over the chunk's cells in cell order.  Round k covers the contiguous cells
     * [k*BLOCK, (k+1)*BLOCK): a DPP wave scan per round, one LDS exchange of the SEG*NW wave totals,
     * then every wave scans that 64-entry table itself ------------------------------------------- */
    uint32_t off[SEG];
    uint32_t total;
    {
      uint32_t incl[SEG];
#pragma unroll
      for (int k = 0; k < SEG; k++) {
        incl[k] = wave_inclusive_scan(len[k]);
        if (lane == 63)
          wsum[k * NW + wave] = incl[k];
      }
      __syncthreads();
      const uint32_t t = lane < SEG * NW ? wsum[lane] : 0u;
      const uint32_t ts = wave_inclusive_scan(t);
      total = wave_read_lane(ts, SEG * NW - 1);
#pragma unroll
      for (int k = 0; k < SEG; k++) {
        const int slot = k * NW + wave;
        const uint32_t before = wave_read_lane(ts, slot) - wave_read_lane(t, slot);
        off[k] = before + incl[k] - len[k];
      }
    }
    ACHIP_STAMP(4);
    if (parts > 1) {
      /* publish this part's byte count, then add up the predecessors' (one lane per predecessor) */
      unsigned long long *sync = part_sync + (size_t)fidx * parts;
      if (tid == 0)
        part_publish(&sync[part], epoch, base + total); /* part 0: base = pad_top, else 0 so far */
      if (part > 0) {
        if (wave == 0) {
          uint32_t acc = 0;
          bool bad = false;
          for (int q0 = 0; q0 < part; q0 += 64) {
            const int q = q0 + lane;
            uint32_t v = 0;
            if (q < part) {
              v = part_wait(&sync[q], epoch);
              bad |= v >= 0xFFFFFFF0u;
            }
            acc += wave_read_lane(wave_inclusive_scan(v), 63);
          }
          const uint64_t any_bad = wave_ballot(bad);
          if (lane == 0)
            flags[3] = any_bad != 0ull ? 0xFFFFFFFFu : acc;
        }
        __syncthreads();
        const uint32_t before = flags[3];
        if (before == 0xFFFFFFFFu)
          overflow = true; /* a predecessor failed or never published: emit nothing, report below */
        base = before;
        own_from = base; /* the bytes below belong to the predecessor, even inside our first 16-byte group */
        flushed = base & ~15u;
      }
    }
    if ((uint64_t)base + total > cap_bytes)
      overflow = true;

    /* ---- E: store the tokens into the LDS staging buffer and drain it to HBM, one window of the stream
     * at a time.  The buffer is linear: its byte 0 is stream offset `flushed` (16-byte aligned).  A window is
     * cut at a TOKEN boundary -- the first token that does not fit entirely waits for the next window -- so a
     * token never straddles anything and every store is a plain, unchecked LDS byte store.  After a drain the
     * < 16 not-yet-flushable tail bytes are moved to the front by the one thread that owns 16-byte group 0. */
    const uint32_t chunk_end = base + total;
    const bool last_chunk = r1 >= row_end;
    uint32_t done = base; /* tokens starting below `done` are already in the buffer or in HBM */
    while (!overflow) {
      const uint32_t lo = flushed, hi = flushed + (uint32_t)RING;
      uint32_t *cut_slot = &flags[1 + (window_no & 1)];
#pragma unroll
      for (int k = 0; k < SEG; k++) {
        const uint32_t a = base + off[k];
        const uint32_t b = a + len[k];
        if (len[k] != 0u && a >= done) {
          if (b <= hi) {
            const Tok tk = PACK_TOK ? token_payload<MODE, L>(meta[PACK_TOK ? k : 0] & 0xFFFu, meta[PACK_TOK ? k : 0] >> 12,
                                                             tid + k * BLOCK, f.ops)
                                    : tok[PACK_TOK ? 0 : k];
#if !defined(ACHIP_ABLATE) || ACHIP_ABLATE != 2
            if (EMIT_OR) {
              PackSink<L::o_dec> ps(ring_addr + (a - lo));
              token_fields<MODE>(ps, tk, ascii_only);
              ps.finish();
            } else {
              FastSink<L::o_dec, L::o_flags + 16> fs{ring_addr + (a - lo), dummy_addr};
              token_fields<MODE>(fs, tk, ascii_only);
            }
#else
            asm volatile("" ::"v"(a), "v"(tk.flags), "v"(tk.fg), "v"(tk.glyph));
#endif
          } else if (a <= hi) {
            *cut_slot = a; /* the unique token that starts inside the window but does not fit: the cut */
          }
        }
      }
      lds_store_fence();
      __syncthreads();
      ACHIP_STAMP(5);
      const uint32_t cut = min(*cut_slot, chunk_end);
      if (tid == 0)
        flags[1 + ((window_no + 1) & 1)] = 0xFFFFFFFFu; /* next window's slot; its last readers are a barrier behind */
      window_no++;
      /* the bytes of a part's last window are flushed completely (its successor owns the rest of the group) */
      const bool final_flush = last_chunk && cut == chunk_end; /* the tail bytes go out too: nothing to carry */
      drain_ring<BLOCK, EMIT_OR>(L::o_ring, dst, lo, cut, max(own_from, lo), final_flush);
      flushed = cut & ~15u;
      if ((uint32_t)tid == drain_group0_tid(dst, lo, max(own_from, lo)) && flushed > lo && cut > flushed && !final_flush) {
        /* the < 16 bytes behind the last flushed group move to the front as ONE 16-byte group (a byte loop here
         * is a chain of dependent LDS round trips between every two chunks).  This thread drained group 0 itself
         * (program order), so it may overwrite it; the bytes behind `cut` in the group are stale (FastSink: the
         * next window overwrites them) or zero (PackSink: they must stay zero, and the source group is cleared) */
        uint4 *src = reinterpret_cast<uint4 *>(ring + (flushed - lo));
        *reinterpret_cast<uint4 *>(ring) = *src;
        if (EMIT_OR)
          *src = make_uint4(0u, 0u, 0u, 0u);
      }
      done = cut;
      if (cut >= chunk_end) {
        ACHIP_STAMP(6);
        break; /* the next buffer writes are several barriers away (next chunk's A/B/D) */
      }
      __syncthreads(); /* the next window overwrites bytes that are being drained */
      ACHIP_STAMP(6);
    }
    base = chunk_end;
  }

  if (prof && tid == 0) {
    const unsigned long long *pa = lds_ptr<const unsigned long long>(L::o_prof);
    for (int k = 0; k < 7; k++)
      prof[(size_t)fidx * 8u + (size_t)k] = pa[k];
    prof[(size_t)fidx * 8u + 7u] = cycle_now() - pa[8];
  }
  if (tid == 0 && last_part) {
    out_len[fidx] = overflow ? ACHIP_LEN_OVERFLOW : base;
    if (!overflow && (uint64_t)base < out_stride)
      dst[base] = 0; /* NUL after the frame when the slot has room, as the reference's strings carry */
  }
}

#ifndef ACHIP_FRAME_KERNEL_ONLY /* render_inst.hip: the non-template kernels below live in hip_launch.hip only */
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
    const uint32_t p = load_rgb(src, src_stride, sx, sy, sw * sh == 1);
    uint8_t *d = dst + (size_t)i * 3u;
    d[0] = (uint8_t)p;
    d[1] = (uint8_t)(p >> 8);
    d[2] = (uint8_t)(p >> 16);
  }
}

/* the same for up to ACHIP_RESIZE_BATCH_MAX images in one launch: blockIdx.y selects the image (the grid path's per-tick
 * tile resizes: nine ~5 KB outputs, where nine launches cost more than the work) */
__global__ void __launch_bounds__(256) resize_nn_batch_kernel(achip_resize_batch_t b) {
  const achip_resize_item_t it = b.item[blockIdx.y];
  const uint32_t total = (uint32_t)it.dw * (uint32_t)it.dh;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t y = i / (uint32_t)it.dw, x = i - y * (uint32_t)it.dw;
    uint32_t sx = (x * it.x_ratio) >> 16, sy = (y * it.y_ratio) >> 16;
    sx = min(sx, (uint32_t)it.sw - 1u);
    sy = min(sy, (uint32_t)it.sh - 1u);
    const uint32_t p = load_rgb(it.src, 3 * it.sw, sx, sy, it.sw * it.sh == 1);
    uint8_t *d = it.dst + (size_t)i * 3u;
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

#endif /* ACHIP_FRAME_KERNEL_ONLY */

} // namespace achip
