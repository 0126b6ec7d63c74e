/*
 * render_rows.hpp -- the wave-autonomous frame kernel for the RUN-STRUCTURED renderers of the path:
 *   image_print (monochrome, REP runs)                             lib/video/ascii/scalar/foreground.c:27-138
 *   rgb_to_truecolor_halfblocks_scalar                             lib/video/ascii/scalar/halfblock.c:48-165
 *   rgb_to_256color_halfblocks_scalar / rgb_to_16color_...         halfblock.c:416-524 / 297-405
 *   rgb_to_halfblocks_scalar (monochrome half blocks)              halfblock.c:184-286
 * whose tokens depend on runs of equal cells: a run's head carries the SGRs and the REP count, the cells behind it
 * are swallowed by the REP or repeat the glyph, a head's SGRs are suppressed against the state the previous run left.
 *
 * render_stream.hpp takes the per-cell modes through the path one wave at a time; what kept the run modes on the
 * barrier-fenced phase kernel (render_kernels.hpp) was that a cell needs the NEXT run head, which may belong to
 * another wave.  The observation that removes the problem: every renderer starts a new run at the first cell of a
 * text row (the `x == 0` / state-reset rules of the reference's row loops), so run structure never crosses a row.  A
 * block here is therefore a whole number of text rows -- floor(64 * CPL / row width), at least one -- and ONE WAVE
 * owns it: run heads are wave ballots (CPL 64-bit masks in scalar registers), a cell finds its head, the next head and
 * the previous run's head with bit scans over those masks, the left neighbour arrives by one DPP move.  No LDS pixel
 * arrays, no head-mask phase, no barrier: everything the stream kernel does -- request-ahead gather, wave scan,
 * decoupled look-back over LDS words, private staging, 16-byte non-temporal drain, the frame CRC riding the drain -- is
 * the same code (this file includes render_stream.hpp and shares its helpers).  A row that does not fit 64 * CPL cells
 * is the phase kernel's (the host decides: achip_choose_geometry).
 *
 * Staging is per SLICE (the 64 cells of one lane slot k), not per block: a 448-cell block of 53-byte half-block tokens
 * would need 25 KB per wave; a slice needs 3.6 KB, so sixteen waves fit a CU's LDS and the CRC tables besides.  Slices
 * drain like the stream kernel's blocks do (whole 16-byte groups as uint4, the shared first / last group as bytes).
 *
 * Round 5 (docs/history/round5.md): half the vector instructions per block.  What a cell's POSITION decides is
 * decided once per frame (one record per slot: the sample's byte offset in its source row, pad / first-pixel / row-end
 * flags, row inside the block); a block that is one text row takes its source rows from scalar registers, so a sample's
 * address costs no vector arithmetic; two slots share a length scan, kept for the store pass; a truecolor SGR's
 * decimal fields come out of LDS ready to store.  The kernel then stopped being bound by instruction issue alone -- the token
 * byte stores' LDS bank conflicts were what was left -- so the truecolor SGRs are built in registers and leave as aligned
 * dword ORs (render_kernels.hpp word_sgr; the store pass below), and a word of cells that all start a run takes its tokens
 * without the head bit scans (make_tok_heads).  What was costed or measured and rejected is in the round's log.
 */
#pragma once

#include "render_stream.hpp"

namespace achip {

/* a cell's record keeps its sample's byte offset in the source row in 16 bits: 3 * (src_w - 1) < 65 536.  (The reference's
 * entry points take sources up to 10 000 pixels wide, ascii.c:204; wider hand-built descriptors go to the phase kernel:
 * achip_choose_geometry.) */
#define ACHIP_ROWS_MAX_SRC_W 21845
#ifndef ACHIP_ROWS_SLOT_EMIT
#define ACHIP_ROWS_SLOT_EMIT 0 /* 1 (A/B builds): tokens through per-lane LDS slots and straight to their place in the frame
                                  (store pass, below) instead of the packed staging area with its line-wise drain.  Measured
                                  and NOT adopted: the byte stores stop conflicting, but a frame's lines then reach the L2 in
                                  ~16-byte pieces from seven store instructions per slice, and the write path takes that worse
                                  than the LDS took the conflicts -- sampled 400x240 half blocks 161.6 -> 201.8 us, from 4K
                                  sources 201 -> 235 (profiles/r05_rows_slot_emit_ab.txt) */
#endif
#ifndef ACHIP_ROWS_WORD_EMIT
#define ACHIP_ROWS_WORD_EMIT 1 /* truecolor half blocks: a token's SGRs are put together in registers from ready-made table
                                  pieces and leave them as 6 / 7 aligned dword ORs each instead of 19 / 22 byte stores
                                  (render_kernels.hpp word_sgr; the store pass below); 0 (A/B builds): byte stores only */
#endif
#ifndef ACHIP_ROWS_EMIT_OR_MODES
#define ACHIP_ROWS_EMIT_OR_MODES 0 /* bit m set: mode m stores its tokens through PackSink here.  Off: this kernel is bound
                                      by VALU issue (profiles/r03_k5_sq_counters.txt: 130 M VALU instructions per 256-frame
                                      launch with the OR sink, 110 M with byte stores; 250 vs 231 us), unlike the phase
                                      kernel, whose store phase is fenced by barriers and bound by the LDS pipe */
#endif

/* longest token of a run-structured mode, bytes, rounded up (SURVEY 8a "per-token byte lengths") */
__host__ __device__ constexpr int rows_max_token(int m) {
  return m == ACHIP_MODE_HB_TRUE  ? 56   /* 19 + 19 + 3-byte half block + ESC[4095b (7) + reset 4 + NL */
         : m == ACHIP_MODE_HB_256 ? 40   /* 11 + 11 + 3 + 7 + 4 + 1                                     */
         : m == ACHIP_MODE_HB_16  ? 32   /* 5 + 6 + 3 + 7 + 4 + 1                                       */
                                  : 16;  /* mono / mono half blocks: glyph <= 4 + 7 + NL                */
}

template <int MODE, int WAVES, bool CRC = false, bool WIDE = false> struct RLds {
  /* one slice + the 16-byte group it starts in; plain instantiations: + the 128-byte line it starts in (the tail of
   * the slice before it, carried: whole lines leave the wave) + the 128 bytes the carry's move may read behind it */
  static constexpr int STAGE = 64 * rows_max_token(MODE) + (CRC ? 16 : 256);
  static constexpr int GPL = (STAGE / 16 + 63) / 64;
  static constexpr int o_stage = 16; /* (a word-built token whose first byte is the area's first ORs a zero dword in front of it) */
  static constexpr int o_glyph = o_stage + WAVES * STAGE;
  static constexpr int o_glyph64 = o_glyph + 256 * 4;
  static constexpr int o_ramp = o_glyph64 + 64 * 4;
  static constexpr int o_dec = o_ramp + 64;
  /* truecolor half blocks: a channel's decimal field ready to store -- {digits + terminator, zero padded; byte count} per
   * value for ';' (R, G) and for 'm' (B) -- and the byte counts alone for the length pass (RowsFastSink / RowsCountSink) */
  static constexpr bool NUM8 = MODE == ACHIP_MODE_HB_TRUE;
  static constexpr int o_num_semi = o_dec + 256 * 4;
  static constexpr int o_num_m = o_num_semi + (NUM8 ? 256 * 8 : 0);
  static constexpr int o_num_len = o_num_m + (NUM8 ? 256 * 8 : 0);
  /* ... and for tokens built as words (word_sgr): {text, shift / length terms in bits} of the R, G, B fields and of the
   * background's B field with the half block behind it */
  static constexpr bool WORDS = NUM8 && !CRC && (ACHIP_ROWS_WORD_EMIT != 0);
  static constexpr int o_wr = o_num_len + (NUM8 ? 256 : 0);
  static constexpr int o_wg = o_wr + (WORDS ? 256 * 8 : 0);
  static constexpr int o_wm = o_wg + (WORDS ? 256 * 8 : 0);
  static constexpr int o_wmg = o_wm + (WORDS ? 256 * 8 : 0);
  static constexpr int o_flags = o_wmg + (WORDS ? 256 * 8 : 0); /* [+16 ..] swallows predicated-off byte stores */
  static constexpr int o_comp = o_flags + 32 + 64 * 4;
  static constexpr int o_tab = o_comp + ACHIP_COMP_LDS_BYTES;
  static constexpr int o_slice = o_tab;
  static constexpr int o_lanek = o_slice + (CRC ? 16 * 1024 : 0);
  static constexpr int base_crc = o_lanek + (CRC ? 3 * 1024 + 256 + 16 + 2 * ACHIP_STREAM_MAXBLK * 4 : 0);
  static constexpr int WIN = base_crc + 32 * 1024 <= 160 * 1024 ? 4 : 2;
  static constexpr int NWIN = 32 / WIN;
  static constexpr int o_pow = o_lanek + (CRC ? NWIN * (1 << WIN) * 64 * 4 : 0);
  static constexpr int o_xk = o_pow + (CRC ? 3 * 1024 : 0);
  static constexpr int TAB_BYTES = (CRC ? o_xk + 256 : o_tab) - o_tab;
  static constexpr int o_crcacc = o_tab + TAB_BYTES;
  static constexpr int o_slots = o_crcacc + (CRC ? 16 : 0);
  /* per block: the look-back word (+ the CRC's) and, WIDE, the segment's summary word (rows_seg_*, below) */
  static constexpr int bytes_for(int maxblk) { return o_slots + maxblk * 4 * ((CRC ? 2 : 1) + (WIDE ? 1 : 0)); }
  static constexpr int bytes = bytes_for(ACHIP_STREAM_MAXBLK);
  static_assert(STAGE % 16 == 0 && o_tab % 16 == 0 && TAB_BYTES % 16 == 0, "16-byte aligned areas");
  static_assert(bytes <= 160 * 1024, "one workgroup's LDS");
};

/* The rows kernel is bound by VALU issue, not by the LDS pipe (profiles/r03_k5_sq_counters.txt), so its sinks trade vector
 * arithmetic for table reads: a truecolor SGR's three decimal fields come out of LDS ready to store (text with its
 * terminator + byte count: one ds_read_b64, one shift, one add per field instead of the six operations FastSink::num
 * assembles it with), and the length pass adds three byte counts instead of running six compare / add-carry pairs. */
template <int LEN_OFF> struct RowsCountSink : CountSink {
  static constexpr bool FAST_DEC = LEN_OFF >= 0;
  template <int ROOM> __device__ inline void sgr_true(bool, uint32_t rgb) {
    const uint8_t *len = lds_ptr<const uint8_t>(LEN_OFF >= 0 ? LEN_OFF : 0);
    n += 7u + len[px_r(rgb)] + len[px_g(rgb)] + len[px_b(rgb)];
  }
};
template <int DEC_OFF, int DUMMY_OFF, int SEMI_OFF, int M_OFF> struct RowsFastSink : FastSink<DEC_OFF, DUMMY_OFF> {
  static constexpr bool FAST_DEC = SEMI_OFF >= 0;
  using FastSink<DEC_OFF, DUMMY_OFF>::a;
  __device__ inline void field4(uint2 e) { /* <= 4 bytes; whatever lies behind the field's own is overwritten by the token's next field */
    const uint32_t w = e.x >> 8;
    lds_store_byte<0, false>(a, e.x);
    lds_store_byte<1, false>(a, w);
    lds_store_byte<2, true>(a, e.x);
    lds_store_byte<3, true>(a, w);
    a += e.y;
  }
  template <int ROOM> __device__ inline void sgr_true(bool bg, uint32_t rgb) {
    if constexpr (ROOM < 2) { /* (the per-cell modes' SGRs, compiled for no mode of this kernel: the last field would store four bytes) */
      const uint32_t e0 = this->lookup(px_r(rgb)), e1 = this->lookup(px_g(rgb)), e2 = this->lookup(px_b(rgb));
      this->template c<4>(bg ? 0x38345B1Bu : 0x38335B1Bu);
      this->template c<3>(0x003B323Bu);
      this->template num<2>(e0, ';');
      this->template num<2>(e1, ';');
      this->template num<ROOM>(e2, 'm');
      return;
    }
    const uint2 e0 = lds_ptr<const uint2>(SEMI_OFF >= 0 ? SEMI_OFF : 0)[px_r(rgb)];
    const uint2 e1 = lds_ptr<const uint2>(SEMI_OFF >= 0 ? SEMI_OFF : 0)[px_g(rgb)];
    const uint2 e2 = lds_ptr<const uint2>(M_OFF >= 0 ? M_OFF : 0)[px_b(rgb)];
    this->template c<4>(bg ? 0x38345B1Bu : 0x38335B1Bu); /* ESC [ 3|4 8 */
    this->template c<3>(0x003B323Bu);                    /* ; 2 ;       */
    field4(e0);
    field4(e1);
    field4(e2);
  }
};

/* run key comparison of two cells (render_kernels.hpp same_run, on registers) */
template <int MODE> __device__ inline bool rows_same_run(uint32_t aT, uint32_t aB, uint32_t bT, uint32_t bB) {
  if (MODE == ACHIP_MODE_MONO)
    return px_key(aT) == px_key(bT);
  if (MODE == ACHIP_MODE_HB_TRUE || MODE == ACHIP_MODE_HB_MONO)
    return px_rgb(aT) == px_rgb(bT) && px_rgb(aB) == px_rgb(bB);
  return px_key(aT) == px_key(bT) && px_key(aB) == px_key(bB); /* HB_256 / HB_16 */
}

/* What a cell of a run-structured mode needs to know about its surroundings -- all of it derived from the wave's head
 * masks (rows kernel) exactly as build_token derives it from the LDS masks (phase kernel). */
struct RunCtx {
  bool is_head;          /* the cell starts a run                                                          */
  uint32_t run;          /* cells in the cell's run                                                        */
  bool head_transparent; /* HT/H256/H16: the run's HEAD is raw black over raw black (halfblock.c:357,476)  */
  bool state_set;        /* head, not first in its row: the previous run left fg/bg state (was not transparent) */
  uint32_t prevT, prevB; /* head, not first in its row: a cell of the previous run (same keys / rgb as its head) */
};

/* the token of one cell: the run-structured branch of build_token (render_kernels.hpp), cited there */
template <int MODE>
__device__ inline Tok rows_token(const RunCtx &c, uint32_t pt, uint32_t pb, uint32_t ops, const uint32_t *glyph64, bool pad,
                                 bool row_end, bool last_row) {
  Tok t;
  t.flags = 0;
  t.fg = t.bg = t.glyph = t.rep = 0;
  if (pad) { /* ascii_pad_frame_width: pad_left spaces in front of every row */
    t.flags = TF_PAD;
    return t;
  }
  const bool rep = rep_profitable(c.run);
  if (c.is_head && rep) {
    t.flags |= TF_REP;
    t.rep = c.run - 1u;
  }
  if (MODE == ACHIP_MODE_MONO) {
    /* image_print (foreground.c:86-127): key = ramp[Y>>2], glyph = cache64[key] (double mapping; clamped as in the
     * phase kernel for palettes of more than 64 characters) */
    t.glyph = glyph64[min(px_key(pt), 63u)];
    if (c.is_head || !rep)
      t.flags |= TF_GLYPH;
  } else if (MODE == ACHIP_MODE_HB_MONO) {
    /* rgb_to_halfblocks_scalar (halfblock.c:203-275): 76/150/29 luminance, no rounding term */
    const uint32_t lt = (76u * px_r(pt) + 150u * px_g(pt) + 29u * px_b(pt)) >> 8;
    const uint32_t lb = (76u * px_r(pb) + 150u * px_g(pb) + 29u * px_b(pb)) >> 8;
    if (lt < 16u && lb < 16u) {
      t.flags = TF_SPACE; /* no REP for padding */
    } else {
      const uint32_t sh = lt >> 6; /* U+2591 U+2592 U+2593 U+2588 = E2 96 91|92|93|88 */
      t.glyph = 0x0096E2u | ((sh == 3u ? 0x88u : 0x91u + sh) << 16);
      if (c.is_head || !rep)
        t.flags |= TF_GLYPH;
    }
  } else {
    /* HT / H256 / H16 (halfblock.c:48-165, 297-524): transparency is decided by the run HEAD's raw rgb; fg/bg SGRs only
     * when they differ from the state left by the previous run in this row (unset at row start and after a transparent
     * run) */
    if (c.head_transparent) {
      t.flags = TF_SPACE | ((c.is_head && c.state_set) ? TF_RESET_PRE : 0u);
    } else {
      t.glyph = 0x8096E2u; /* U+2580 upper half block = E2 96 80 */
      if (c.is_head || !rep)
        t.flags |= TF_GLYPH;
      if (c.is_head) {
        if (MODE == ACHIP_MODE_HB_TRUE) {
          if (!c.state_set || px_rgb(c.prevT) != px_rgb(pt)) {
            t.flags |= TF_SGR_FG;
            t.fg = px_rgb(pt);
          }
          if (!c.state_set || px_rgb(c.prevB) != px_rgb(pb)) {
            t.flags |= TF_SGR_BG;
            t.bg = px_rgb(pb);
          }
        } else {
          const bool is256 = MODE == ACHIP_MODE_HB_256;
          if (!c.state_set || px_key(c.prevT) != px_key(pt)) {
            t.flags |= TF_SGR_FG;
            t.fg = is256 ? px_key(pt) : sgr16_code(false, px_key(pt));
          }
          if (!c.state_set || px_key(c.prevB) != px_key(pb)) {
            t.flags |= TF_SGR_BG;
            t.bg = is256 ? px_key(pb) : sgr16_code(true, px_key(pb));
          }
        }
      }
    }
  }
  /* rainbow_replace_ansi_colors (color_filter.c:348-408) rewrites every ESC[38;2;..m of the finished frame */
  if (MODE == ACHIP_MODE_HB_TRUE && (ops & ACHIP_OP_FG_OVERRIDE) && (t.flags & TF_SGR_FG))
    t.fg = ops >> ACHIP_OP_TINT_SHIFT;
  if (row_end) {
    if (mode_row_reset(MODE))
      t.flags |= TF_ROW_RESET;
    if (!last_row)
      t.flags |= TF_NL;
  }
  return t;
}

/* the payload of a token whose flags are known: colours and glyph follow from the cell's own pixels (a head's SGRs carry
 * its own colours; every cell of a run shows its own glyph).  The store pass rebuilds tokens with this from one packed
 * word per slot {flags:12, rep:12, length:6} instead of holding five registers per slot or deciding everything twice. */
template <int MODE>
__device__ inline Tok rows_token_payload(uint32_t flags, uint32_t rep, uint32_t pt, uint32_t pb, uint32_t ops,
                                         const uint32_t *glyph64) {
  Tok t;
  t.flags = flags;
  t.rep = rep;
  t.fg = t.bg = t.glyph = 0;
  if (MODE == ACHIP_MODE_MONO) {
    t.glyph = glyph64[min(px_key(pt), 63u)];
  } else if (MODE == ACHIP_MODE_HB_MONO) {
    const uint32_t lt = (76u * px_r(pt) + 150u * px_g(pt) + 29u * px_b(pt)) >> 8;
    const uint32_t sh = lt >> 6;
    t.glyph = 0x0096E2u | ((sh == 3u ? 0x88u : 0x91u + sh) << 16);
  } else {
    t.glyph = 0x8096E2u;
    if (MODE == ACHIP_MODE_HB_TRUE) {
      t.fg = (ops & ACHIP_OP_FG_OVERRIDE) ? ops >> ACHIP_OP_TINT_SHIFT : px_rgb(pt);
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

/* the bits strictly below / above a lane's own, made once per kernel: the scans below then cost two ANDs with the
 * (scalar) mask instead of 64-bit shifts by the lane number */
struct LaneMasks {
  uint32_t below_lo, below_hi, above_lo, above_hi;
};
__device__ inline LaneMasks lane_masks(int lane) {
  const uint64_t below = (1ull << lane) - 1ull, above = ~(below | (1ull << lane));
  return LaneMasks{(uint32_t)below, (uint32_t)(below >> 32), (uint32_t)above, (uint32_t)(above >> 32)};
}
/* index of the highest set bit of m strictly below the lane's bit position, or -1 */
__device__ inline int rows_prev_bit(uint64_t m, const LaneMasks &lm) {
  const uint32_t lo = (uint32_t)m & lm.below_lo, hi = (uint32_t)(m >> 32) & lm.below_hi;
  return hi ? 63 - __clz((int)hi) : (lo ? 31 - __clz((int)lo) : -1);
}
/* index of the lowest set bit of m strictly above the lane's bit position, or -1 */
__device__ inline int rows_next_bit(uint64_t m, const LaneMasks &lm) {
  const uint32_t lo = (uint32_t)m & lm.above_lo, hi = (uint32_t)(m >> 32) & lm.above_hi;
  return lo ? __ffs((int)lo) - 1 : (hi ? 31 + __ffs((int)hi) : -1);
}

/* ---- WIDE: rows beyond one block (round 6) -------------------------------------------------------------------------------
 * A padded row of more than 64 * CPL cells is cut into ceil(row / (64 * CPL)) SEGMENTS of equal width (the last one
 * shorter), block b = segment b % nseg of text row b / nseg, taken by consecutive waves.  A segment decides its own run
 * heads: beside its cells it samples the cell in front of it and the cell behind it (a GHOST slot: lane 0 / lane 1 of one
 * more register per array), so "does my first cell continue your last run" is a comparison of two of its own pixels.  What
 * it cannot know by itself is how far the run it starts in reaches back (its head, the head's transparency) and how far
 * its last run goes on: every segment publishes ONE LDS word -- cells in front of its first head, cells from its last head
 * on, that head's transparency, whether its last cell's run goes on behind it -- BEFORE it waits for anything, and only a
 * segment whose first cell continues a run / whose last run goes on reads its neighbours' words (a frame without flat
 * areas never waits).  No deadlock: a word of block x is published at the top of x's turn, which follows the turn of
 * block x - WAVES of the same wave; that turn waits for words of its own row only (blocks < x - WAVES + nseg <= x for
 * nseg <= WAVES -- the kernel refuses wider rows, the host never sends them) and for byte counts of earlier blocks.
 * ascii.c:204 admits rows of up to 10 000 cells; a run's repeat count is kept in 12 bits here, so 4 096 (the phase
 * kernel's limit as well). */
#ifndef ACHIP_ROWS_WIDE_MAX_ROW
#define ACHIP_ROWS_WIDE_MAX_ROW 4096 /* (render_variants.h states it for the host as well) */
#endif
#define ACHIP_SEG_PUB (1u << 31)
#define ACHIP_SEG_HAS_HEAD (1u << 18)
#define ACHIP_SEG_LAST_T (1u << 19)  /* the segment's last head is transparent                        */
#define ACHIP_SEG_CONT (1u << 20)    /* the run of the segment's last cell goes on in the next segment */
__host__ __device__ constexpr uint32_t rows_seg_word(uint32_t lead, uint32_t tail, bool has_head, bool last_t, bool cont) {
  return ACHIP_SEG_PUB | lead | (tail << 9) | (has_head ? ACHIP_SEG_HAS_HEAD : 0u) | (last_t ? ACHIP_SEG_LAST_T : 0u) |
         (cont ? ACHIP_SEG_CONT : 0u);
}
/* a published word of the row (wave-uniform); 0 if it never comes (bounded) */
__device__ inline uint32_t rows_seg_wait(const uint32_t *sumw, int j) {
  for (int spin = 0; spin < (1 << 22); spin++) {
    const uint32_t w = wave_read_lane(slot_load(&sumw[j]), 0); /* (a wave operation: where the emulator's fibers take turns) */
    if (w & ACHIP_SEG_PUB)
      return w;
    spin_nap<1>();
  }
  return 0u;
}
/* the run that is open where block b starts: {cells of it in front of the block: bits 15..0, its head transparent: bit 16};
 * 0xFFFFFFFF if a word never comes.  (Never walks past the row's first segment: that one starts with a head.) */
__device__ inline uint32_t rows_seg_back(const uint32_t *sumw, int b, uint32_t segw) {
  uint32_t acc = 0;
  for (int j = b - 1;; j--) {
    const uint32_t w = rows_seg_wait(sumw, j);
    if (!w)
      return 0xFFFFFFFFu;
    if (w & ACHIP_SEG_HAS_HEAD)
      return (acc + ((w >> 9) & 0x1FFu)) | ((w & ACHIP_SEG_LAST_T) ? 1u << 16 : 0u);
    acc += segw;
  }
}
/* cells behind block b that go on with its last cell's run (the caller knows that the run goes on); 0xFFFFFFFF if a word
 * never comes.  (Ends at the row's last segment at the latest: that one never sets ACHIP_SEG_CONT.) */
__device__ inline uint32_t rows_seg_ahead(const uint32_t *sumw, int b) {
  uint32_t acc = 0;
  for (int j = b + 1;; j++) {
    const uint32_t w = rows_seg_wait(sumw, j);
    if (!w)
      return 0xFFFFFFFFu;
    acc += w & 0x1FFu;
    if ((w & ACHIP_SEG_HAS_HEAD) || !(w & ACHIP_SEG_CONT))
      return acc;
  }
}

/* two 512-thread workgroups per CU need <= 128 VGPRs (4 waves per SIMD): left alone the compiler spreads the 7-slot
 * geometry over 140-180 registers for scheduling freedom it has no use for (the kernel waits on memory, not on issue) */
/* (not the CRC instantiations: their tables leave room for one workgroup per CU anyway, so they may spread out; nor the
 * one composite instantiation that would have to spill to get there) */
template <int MODE, int CPL, bool GENERIC, bool CRC> struct RowsMinWaves {
#ifdef ACHIP_ROWS_MIN_WAVES /* A/B builds: fewer resident waves, more registers each */
  static constexpr int value = ACHIP_ROWS_MIN_WAVES;
#else
  static constexpr int value = (!CRC && !(GENERIC && MODE == ACHIP_MODE_HB_16 && CPL > 4)) ? 4 : 1;
#endif
};
/* PARTS (round 6; small launches -- a lone mono frame, a handful of half-block frames): a frame's blocks shared out over
 * ps.parts four-wave workgroups exactly as the stream kernel shares out a per-cell frame (render_stream.hpp PARTS): the grid
 * is n_frames * parts, workgroup f * parts + p takes the p-th run of ceil(blocks / parts) blocks, the look-back inside a
 * workgroup stays in LDS, and every workgroup publishes the bytes of its blocks (ps.sync[workgroup] = {epoch, bytes}, agent
 * scope) when its last block is counted; the first block of a workgroup waits for the words of the parts in front of it. */
template <int MODE, int WAVES, int CPL, bool GENERIC, bool CRC = false, bool WIDE = false, bool PARTS = false>
__global__ void __launch_bounds__(WAVES * 64) ACHIP_WAVES_PER_EU((RowsMinWaves<MODE, CPL + (WIDE ? 1 : 0), GENERIC, CRC>::value))
    render_rows_kernel(const achip_frame_t *__restrict__ frames, const achip_lut_t *__restrict__ lut,
                       uint8_t *__restrict__ out, uint64_t out_stride, uint32_t *__restrict__ out_len, int n_frames,
                       achip_uniform_t uni, achip_wire_t wire, const uint4 *__restrict__ crc_tab, achip_partsdev_t ps) {
  /* uni.flags bits 31..8 (ACHIP_UNIFORM_MAX_CELLS' field) carry the BLOCKS of the launch's largest frame here: the host
   * knows every frame's row width and rows, and the per-block LDS words are sized by it */
  static_assert(mode_has_runs(MODE), "per-cell modes use render_stream_kernel");
  static_assert(!WIDE || (!GENERIC && !CRC), "rows cut into segments: fast sampler, no fused checksum");
  static_assert(!PARTS || (!GENERIC && !CRC), "shared-out frames: fast sampler, no fused checksum");
  using L = RLds<MODE, WAVES, CRC, WIDE>;
  constexpr int CPG = CPL + (WIDE ? 1 : 0); /* registers per array: WIDE keeps the segment's two ghost cells in one more */
  constexpr bool HB = mode_is_halfblock(MODE);
  constexpr bool HBC = MODE == ACHIP_MODE_HB_TRUE || MODE == ACHIP_MODE_HB_256 || MODE == ACHIP_MODE_HB_16;
  constexpr int BLOCK = WAVES * 64;
  constexpr int SLOTS = 64 * CPL;
  /* token stores as aligned atomic ORs of register-built dwords (PackSink) into a pre-zeroed staging area instead of
   * byte stores (FastSink): 12 instead of 41 LDS instructions for a half-block truecolor token -- the byte stores of 64
   * lanes land 41 bytes apart and serialise on bank conflicts (profiles/r01_emit_or.txt; the same switch as the phase
   * kernel's) */
  constexpr bool EMIT_OR = ((ACHIP_ROWS_EMIT_OR_MODES) >> MODE) & 1;

  uint32_t *slots = lds_ptr<uint32_t>(L::o_slots);
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int wg = (int)blockIdx.x;
  const int parts = PARTS ? ps.parts : 1;
  const int fidx = PARTS ? wg / parts : wg;
  const int part = PARTS ? wg - fidx * parts : 0;
  ACHIP_DEVICE_ONLY(
      if (PARTS) asm volatile("" ::"s"(ps.parts), "s"(ps.epoch), "s"(ps.sync));
      asm volatile("" ::"s"(n_frames), "s"(lut), "s"(out), "s"(out_stride), "s"(out_len), "s"(frames), "s"(uni.enabled),
                   "s"(uni.flags), "s"(uni.src_pitch), "s"(uni.f.src), "s"(uni.f.comp));
      asm volatile("" ::"s"(uni.f.src_w), "s"(uni.f.src_h), "s"(uni.f.out_w), "s"(uni.f.out_h), "s"(uni.f.pad_left),
                   "s"(uni.f.pad_top), "s"(uni.f.x_ratio), "s"(uni.f.y_ratio), "s"(uni.f.src_stride), "s"(uni.f.ops));
      if (CRC) asm volatile("" ::"s"(wire.crc), "s"(wire.dims), "s"(wire.hdr), "s"(wire.pkt_crc), "s"(crc_tab));)
  if (fidx >= n_frames)
    return;
  constexpr int TABV = L::TAB_BYTES / 16, TABN = (TABV + BLOCK - 1) / BLOCK;
  typedef uint32_t tab4_t __attribute__((vector_size(16)));
  tab4_t tabv[TABN > 0 ? TABN : 1];
  uint32_t dim_w = 0, dim_h = 0;
  if (CRC && wire.dims) {
    dim_w = wire.dims[2 * fidx];
    dim_h = wire.dims[2 * fidx + 1];
  }
  if (CRC) {
#pragma unroll
    for (int k = 0; k < TABN; k++)
      tabv[k] = reinterpret_cast<const tab4_t *>(crc_tab)[tid + k * BLOCK < TABV ? tid + k * BLOCK : TABV - 1];
  }
  constexpr int LUTN = (256 + BLOCK - 1) / BLOCK;
  uint32_t lut_g[LUTN];
#pragma unroll
  for (int k = 0; k < LUTN; k++)
    lut_g[k] = tid + k * BLOCK < 256 ? lut->glyph[tid + k * BLOCK] : 0u;
  const uint32_t lut_g64 = tid < 64 ? lut->glyph64[tid] : 0u;
  const uint32_t lut_ramp = tid < 64 ? lut->ramp[tid] : 0u;
  const bool ascii_only = (uni.flags & ACHIP_UNIFORM_PALETTE_ASCII) != 0u;
  achip_frame_t f = uni.f;
  if (uni.enabled)
    f.src = uni.f.src + (int64_t)fidx * uni.src_pitch;
  else
    f = frames[fidx];
  if (f.src_stride == 0)
    f.src_stride = 3 * f.src_w;
  uint8_t *dst = out + (size_t)fidx * out_stride;
  const uint32_t dmis = (uint32_t)(uintptr_t)dst & (ACHIP_DRAIN_ALIGN - 1u) & ~15u; /* the slot's own offset inside a line */

  const int wp = f.pad_left + f.out_w;
  const int rows = HB ? (f.out_h + 1) / 2 : f.out_h;
  const int nblk_cap = stream_maxblk(uni.flags, 1); /* words in each per-block LDS array of this launch */
  /* WIDE: a row is nseg segments of segw cells (the last one shorter, never empty), a block is one segment */
  const int nseg0 = WIDE && wp > 0 ? (wp + SLOTS - 1) / SLOTS : 1;
  const int segw = WIDE && wp > 0 ? (wp + nseg0 - 1) / nseg0 : wp;
  const int nseg = WIDE && wp > 0 ? (wp + segw - 1) / segw : 1;
  const int rpb = WIDE ? (wp > 0 && wp <= ACHIP_ROWS_WIDE_MAX_ROW && nseg <= WAVES ? 1 : 0) : (wp > 0 ? SLOTS / wp : 0); /* text rows per block */
  const int nblk = WIDE ? rows * nseg : (rpb > 0 ? (rows + rpb - 1) / rpb : 0);
  auto bad_frame = [&]() {
    if (tid == 0) {
      out_len[fidx] = ACHIP_LEN_BADDESC;
      if (CRC) {
        wire.crc[fidx] = 0u;
        if (wire.hdr)
          for (int j = 0; j < 24; j++)
            wire.hdr[(size_t)fidx * 24u + j] = 0;
        if (wire.pkt_crc)
          wire.pkt_crc[fidx] = ~crc_mulmod(0xFFFFFFFFu, crc_pow(CRC_X8, 24u));
      }
    }
  };
  if (f.out_w <= 0 || f.out_h <= 0 || f.src_w <= 0 || f.src_h <= 0 || f.pad_left < 0 || f.pad_top < 0 ||
      (!f.src && !f.comp) || (!GENERIC && (f.comp || f.src_w * f.src_h == 1 || f.src_w > ACHIP_ROWS_MAX_SRC_W)) || rpb < 1 || nblk > nblk_cap ||
      out_stride > (uint64_t)ACHIP_STREAM_MAX_STRIDE) {
    bad_frame();
    return;
  }
  /* PARTS: this workgroup's run of blocks [b0, b1); a part behind the frame's last block only reports in */
  /* (WIDE && PARTS: whole text rows per workgroup -- the segments of a row talk through LDS words) */
  const int bpp = !PARTS ? nblk : WIDE ? ((rows + parts - 1) / parts) * nseg : (nblk + parts - 1) / parts;
  const int b0 = PARTS ? min(part * bpp, nblk) : 0, b1 = PARTS ? min(b0 + bpp, nblk) : nblk;
  uint32_t *partacc = lds_ptr<uint32_t>(L::o_flags); /* PARTS: [0] bytes of this workgroup's blocks so far, [1] blocks counted */
  if (PARTS && b0 >= b1) {
    if (tid == 0)
      agent_store_u64(&ps.sync[wg], ((unsigned long long)ps.epoch << 32));
    return;
  }
  const uint32_t cap_bytes = (uint32_t)out_stride;
  const uint32_t pad_left = (uint32_t)f.pad_left, uwp = (uint32_t)wp;
  StreamSrc src;
  src.base = f.src;
  src.stride = (uint32_t)f.src_stride;
  src.xr = f.x_ratio;
  src.yr = f.y_ratio;
  src.w1 = (uint32_t)f.src_w - 1u;
  src.h1 = (uint32_t)f.src_h - 1u;
  src.flip_x = (f.ops & ACHIP_OP_FLIP_X) != 0u;
  src.flip_y = (f.ops & ACHIP_OP_FLIP_Y) != 0u;
  src.nt = f.x_ratio >= ((64u << 16) + 2u) / 3u;
  CompHead chead = {};

  /* slot s = k * 64 + lane of a block is cell s of its rows: row s / wp, column s % wp -- the same for every block.  So
   * everything a cell's position decides is decided ONCE per frame and kept in one register per slot (the passes below
   * recomputed row and column five times per block: ~200 of the block's 4 400 VALU instructions, and the sampler's
   * horizontal index ~300 more):
   *   bits 15..0   the sample's byte offset in its source row, 3 * min(x * x_ratio >> 16, src_w - 1) with the flip folded
   *                in (image.c:293-325; < 30 000: the host checks the source size) -- GENERIC: the column x itself
   *   bit  16      padding pseudo-cell (column < pad_left)         bit 17   first pixel of the row, or in front of it
   *   bit  18      last cell of the row                            bits 31..23   text row inside the block
   * A cell exists in block b iff its row is below the block's row count: ONE unsigned compare of the whole word. */
  constexpr uint32_t CM_PAD = 1u << 16, CM_FIRST = 1u << 17, CM_END = 1u << 18;
  constexpr int CM_ROW = 23;
  uint32_t cm[CPG];
  /* WIDE: a record per slot of ONE segment -- made again for every block (a block's segment changes from turn to turn):
   * columns x0 + slot, the slots behind the segment's last cell marked as a row the block does not have; the ghost slot's
   * lane 0 is the cell in front of the segment, lane 1 the cell behind it (where the row has them) */
  auto seg_records = [&](int seg, uint32_t (&c)[CPG]) {
    const uint32_t x0 = (uint32_t)(seg * segw), n = min((uint32_t)segw, uwp - x0);
    auto record = [&](uint32_t xp, bool exists) {
      uint32_t lo = 0;
      if (xp >= pad_left) {
        const uint32_t sx = min(((xp - pad_left) * src.xr) >> 16, src.w1);
        lo = __umul24(src.flip_x ? src.w1 - sx : sx, 3u);
      }
      return lo | (xp < pad_left ? CM_PAD : 0u) | (xp <= pad_left ? CM_FIRST : 0u) | (xp == uwp - 1u ? CM_END : 0u) |
             (exists ? 0u : 1u << CM_ROW);
    };
#pragma unroll
    for (int k = 0; k < CPL; k++)
      c[k] = record(x0 + (uint32_t)(64 * k + lane), (uint32_t)(64 * k + lane) < n);
    const bool gl = lane == 0 && seg > 0, gr = lane == 1 && x0 + n < uwp;
    c[CPG - 1] = record(gl ? x0 - 1u : gr ? x0 + n : 0u, gl || gr);
  };
  if constexpr (!WIDE) {
    const uint32_t q64 = 64u / uwp, r64 = 64u - q64 * uwp;
    uint32_t rr = (uint32_t)lane / uwp, xp = (uint32_t)lane - rr * uwp; /* one division per lane per frame, then constant steps */
#pragma unroll
    for (int k = 0; k < CPL; k++) {
      uint32_t lo = 0;
      if (xp >= pad_left) {
        lo = xp - pad_left;
        if (!GENERIC) {
          const uint32_t sx = min((lo * src.xr) >> 16, src.w1);
          lo = __umul24(src.flip_x ? src.w1 - sx : sx, 3u);
        }
      }
      cm[k] = lo | (xp < pad_left ? CM_PAD : 0u) | (xp <= pad_left ? CM_FIRST : 0u) | (xp == uwp - 1u ? CM_END : 0u) |
              (min(rr, 511u) << CM_ROW);
      xp += r64;
      rr += q64;
      const bool wrap = xp >= uwp;
      xp -= wrap ? uwp : 0u;
      rr += wrap ? 1u : 0u;
    }
  }
  auto block_rows = [&](int blk) { return WIDE ? 1u : (uint32_t)min(rpb, rows - blk * rpb); };
  /* ROW1: every block is exactly one text row (4K -> 400x120 half blocks in 448 slots, 200x60 in 256): the source rows of
   * a block are wave-uniform, so a sample's address is (scalar row base) + (the slot's byte offset) and costs no vector
   * arithmetic at all.  The dword is requested one byte early (finish: >> 8) so that the last pixel's request stays inside
   * the buffer; only the buffer's very first pixel cannot be (row offset 0, byte offset 0): it asks for offset 1 of the
   * early base -- the pixel itself -- and is finished by masking. */
#ifdef ACHIP_ROWS_COUNT_ROW1 /* diagnostics (scripts/isa_lines.py): only the one-row path with cached loads is compiled, so that
                               the listing's static instruction counts are the counts a 4K -> 400x120 block executes */
  const bool row1 = true;
#else
  const bool row1 = WIDE || (!GENERIC && rpb == 1);
#endif
  auto src_row = [&](uint32_t y) { /* wave-uniform */
    const uint32_t sy = min((y * src.yr) >> 16, src.h1);
    return (src.flip_y ? src.h1 - sy : sy) * src.stride;
  };

  /* request the samples of block `blk`: nothing here consumes loaded data */
  /* (row0 = the block's first text row, cm = its records: the frame's, or WIDE the records of the block's segment) */
  auto issue = [&](auto nt_tag, int blk, uint32_t row0, const uint32_t (&cm)[CPG], uint32_t (&rawT)[CPG], uint32_t (&rawB)[CPG], uint32_t &kinds) {
    constexpr bool NT = decltype(nt_tag)::value;
    kinds = 0;
    const uint32_t vlim = block_rows(blk) << CM_ROW;
    if (row1) {
      const uint32_t yt = HB ? 2u * row0 : row0;
      const bool two = HB && yt + 1u < (uint32_t)f.out_h; /* odd height: the last row's bottom half repeats the top (halfblock.c:81-88) */
      const uint32_t rot = src_row(yt), rob = two ? src_row(yt + 1u) : rot;
      const uint8_t *bt = src.base + rot - 1, *bb = src.base + rob - 1;
      const uint32_t ft = rot == 0u ? 1u : 0u, fb = rob == 0u ? 1u : 0u;
#pragma unroll
      for (int k = 0; k < CPG; k++) {
        rawT[k] = 0;
        rawB[k] = 0;
        if (cm[k] < vlim && !(cm[k] & CM_PAD)) {
          const uint32_t xo = cm[k] & 0xFFFFu;
          const ACHIP_GLOBAL uint8_t *pt_ = (const ACHIP_GLOBAL uint8_t *)bt + max(xo, ft);
          rawT[k] = NT ? load_u32_unaligned_nt((const uint8_t *)pt_) : ((const ACHIP_GLOBAL unaligned_u32 *)pt_)->v;
          if (two) {
            const ACHIP_GLOBAL uint8_t *pb_ = (const ACHIP_GLOBAL uint8_t *)bb + max(xo, fb);
            rawB[k] = NT ? load_u32_unaligned_nt((const uint8_t *)pb_) : ((const ACHIP_GLOBAL unaligned_u32 *)pb_)->v;
          }
        }
      }
      return;
    }
    if constexpr (!WIDE) {
#pragma unroll
    for (int k = 0; k < CPL; k++) {
      rawT[k] = 0;
      rawB[k] = 0;
      if (cm[k] < vlim && !(cm[k] & CM_PAD)) {
        const uint32_t lo = cm[k] & 0xFFFFu, r = row0 + (cm[k] >> CM_ROW);
        auto request = [&](uint32_t y, uint32_t &kind) {
          if (GENERIC)
            return stream_request<true, NT, L::o_comp>(f, src, lo, y, kind, chead);
          uint32_t sy = min((y * src.yr) >> 16, src.h1);
          sy = src.flip_y ? src.h1 - sy : sy;
          const uint32_t a = __umul24(sy, src.stride) + lo; /* stream_request's address, the horizontal part from the record */
          const uint32_t back = a != 0u ? 1u : 0u;
          kind = back ? RAW_BACK : RAW_FIRST;
          const ACHIP_GLOBAL uint8_t *q = (const ACHIP_GLOBAL uint8_t *)src.base + (a - back);
          return NT ? load_u32_unaligned_nt((const uint8_t *)q) : ((const ACHIP_GLOBAL unaligned_u32 *)q)->v;
        };
        uint32_t kind = RAW_FINAL;
        rawT[k] = request(HB ? 2u * r : r, kind);
        kinds |= kind << (2 * k);
        if (HB) {
          uint32_t kb = RAW_TOP; /* odd height: the last text row's bottom half repeats the top (halfblock.c:81-88) */
          if (2u * r + 1u < (uint32_t)f.out_h)
            rawB[k] = request(2u * r + 1u, kb);
          kinds |= kb << (2 * (CPL + k));
        }
      }
    }
    }
  };
  auto issue_any = [&](int blk, uint32_t row0, const uint32_t (&cmx)[CPG], uint32_t (&rawT)[CPG], uint32_t (&rawB)[CPG], uint32_t &kinds) {
#ifndef ACHIP_ROWS_COUNT_ROW1
    if (!GENERIC && src.nt)
      issue(StreamTagNT{}, blk, row0, cmx, rawT, rawB, kinds);
    else
#endif
      issue(StreamTagCached{}, blk, row0, cmx, rawT, rawB, kinds);
  };
  static_assert(4 * CPL <= 32, "two bits per sample in one word");

  /* WIDE: (text row, segment) of the wave's block, and the step to its next one (blk + WAVES) */
  int row_c = 0, seg_c = 0;
  const int step_q = WIDE ? WAVES / nseg : 0, step_r = WIDE ? WAVES - step_q * nseg : 0;
  if constexpr (WIDE) { /* (PARTS: b0 is a multiple of nseg) */
    row_c = wave / nseg;
    seg_c = wave - row_c * nseg;
    row_c += b0 / nseg;
    seg_records(seg_c, cm);
  }
  uint32_t rawT[CPG], rawB[CPG], kinds = 0;
  const bool late_first = GENERIC && f.comp != nullptr;
  if (late_first)
    comp_stage<L::o_comp, BLOCK>(f.comp, tid);
  else if (b0 + wave < b1)
    issue_any(b0 + wave, (uint32_t)(WIDE ? row_c : (b0 + wave) * rpb), cm, rawT, rawB, kinds);

  /* tables -> LDS; look-back words of this frame cleared */
  uint32_t *glyph = lds_ptr<uint32_t>(L::o_glyph);
  uint32_t *glyph64 = lds_ptr<uint32_t>(L::o_glyph64);
  uint8_t *ramp = lds_ptr<uint8_t>(L::o_ramp);
#pragma unroll
  for (int k = 0; k < LUTN; k++)
    if (tid + k * BLOCK < 256) {
      glyph[tid + k * BLOCK] = lut_g[k];
      lds_ptr<uint32_t>(L::o_dec)[tid + k * BLOCK] = dec_table_entry((uint32_t)(tid + k * BLOCK));
      if (L::NUM8) {
        const uint32_t e = dec_entry((uint32_t)(tid + k * BLOCK)), nd = e >> 24, dg = e & 0x00FFFFFFu;
        lds_ptr<uint2>(L::o_num_semi)[tid + k * BLOCK] = make_uint2(dg | ((uint32_t)';' << (8u * nd)), nd + 1u);
        lds_ptr<uint2>(L::o_num_m)[tid + k * BLOCK] = make_uint2(dg | ((uint32_t)'m' << (8u * nd)), nd + 1u);
        lds_ptr<uint8_t>(L::o_num_len)[tid + k * BLOCK] = (uint8_t)(nd + 1u);
        if (L::WORDS) { /* word_sgr's pieces */
          uint2 wr, wg, wm, wmg;
          word_table_entries((uint32_t)(tid + k * BLOCK), wr, wg, wm, wmg);
          lds_ptr<uint2>(L::o_wr)[tid + k * BLOCK] = wr;
          lds_ptr<uint2>(L::o_wg)[tid + k * BLOCK] = wg;
          lds_ptr<uint2>(L::o_wm)[tid + k * BLOCK] = wm;
          lds_ptr<uint2>(L::o_wmg)[tid + k * BLOCK] = wmg;
        }
      }
    }
  if (tid < 64) {
    glyph64[tid] = lut_g64;
    ramp[tid] = (uint8_t)lut_ramp;
  }
  for (int k = tid; k < b1 - b0; k += BLOCK) /* (indexed from the workgroup's first block) */
    slots[k] = 0u;
  if (PARTS && tid < 2)
    partacc[tid] = 0u;
  uint32_t *sumw = slots + nblk_cap; /* WIDE: the segments' summary words (indexed from the workgroup's first block too) */
  if (WIDE)
    for (int k = tid; k < b1 - b0; k += BLOCK)
      sumw[k] = 0u;
  if (EMIT_OR) /* the OR-filled staging areas start out zero; every slice clears what it used */
    for (int k = tid; k < WAVES * L::STAGE / 16; k += BLOCK)
      lds_ptr<uint4>(L::o_stage)[k] = make_uint4(0u, 0u, 0u, 0u);
  if (CRC) {
#pragma unroll
    for (int k = 0; k < TABN; k++)
      if (tid + k * BLOCK < TABV)
        lds_ptr<tab4_t>(L::o_tab)[tid + k * BLOCK] = tabv[k];
    for (int k = tid; k < nblk; k += BLOCK)
      slots[nblk_cap + k] = 0u;
    if (tid < 4)
      lds_ptr<uint32_t>(L::o_crcacc)[tid] = 0u;
  }
  const uint32_t first_base = (uint32_t)f.pad_top;
  if (first_base > 0u && first_base <= cap_bytes && part == 0)
    for (uint32_t o = (uint32_t)tid; o < first_base; o += BLOCK)
      dst[o] = '\n';
  __syncthreads(); /* the only workgroup barrier */
  if (late_first) {
    chead = comp_head<L::o_comp>();
    if (b0 + wave < b1)
      issue_any(b0 + wave, (uint32_t)((b0 + wave) * rpb), cm, rawT, rawB, kinds);
  }

  const LaneMasks lm = lane_masks(lane);
  const uint32_t stage_off = (uint32_t)(L::o_stage + wave * L::STAGE);
  const uint32_t stage_addr = lds_base_addr() + stage_off;
  const uint32_t dummy_addr = lds_base_addr() + (uint32_t)L::o_flags + 16u + 4u * (uint32_t)lane;
  const unsigned char *stage = lds_ptr<const unsigned char>((int)stage_off);
  if (L::WORDS) { /* word-built tokens are OR-ed into zeros: every wave clears its own area (its samples are on their way meanwhile) */
    for (int g = lane; g < L::STAGE / 16; g += 64)
      lds_ptr<uint4>((int)stage_off)[g] = make_uint4(0u, 0u, 0u, 0u);
    wave_lockstep();
  }

  /* ---- samples -> pixels with the mode's run key in bits 31..24 (as the phase kernel parks them in LDS), in place */
  auto to_pixels = [&](int blk, uint32_t row0, const uint32_t (&cm)[CPG], uint32_t (&pt)[CPG], uint32_t (&pb)[CPG], uint32_t kinds) {
    const uint32_t vlim = block_rows(blk) << CM_ROW;
    const uint32_t yt = HB ? 2u * row0 : row0;
    const bool two = HB && yt + 1u < (uint32_t)f.out_h;
    const bool ft = row1 && src_row(yt) == 0u, fb = row1 && two && src_row(yt + 1u) == 0u; /* wave-uniform */
    const bool tint = !GENERIC && (f.ops & ACHIP_OP_TINT) != 0u;
    /* (the buffer's first pixel was requested AT its address, not one byte early: moved up a byte here, for the blocks that
     * read source row 0 and no others, every sample is finished by the same shift below) */
    if (row1 && (ft || fb)) {
#pragma unroll
      for (int k = 0; k < CPG; k++)
        if ((cm[k] & 0xFFFFu) == 0u) {
          pt[k] = ft ? pt[k] << 8 : pt[k];
          pb[k] = fb ? pb[k] << 8 : pb[k];
        }
    }
#pragma unroll
    for (int k = 0; k < CPG; k++) {
      const bool pix = cm[k] < vlim && !(cm[k] & CM_PAD);
      const uint32_t rawT = pt[k], rawB = pb[k];
      uint32_t t = 0, b = 0;
      if (pix) {
        if (row1) { /* requested one byte early (issue) */
          t = rawT >> 8;
          if (tint)
            t = tint_pixel(t, f.ops);
          b = t;
          if (two) {
            b = rawB >> 8;
            if (tint)
              b = tint_pixel(b, f.ops);
          }
        } else {
          const uint32_t kt = (kinds >> (2 * k)) & 3u, kb = (kinds >> (2 * (CPL + k))) & 3u;
          t = sample_finish<GENERIC>(f, rawT, kt);
          if (HB)
            b = kb == RAW_TOP ? t : sample_finish<GENERIC>(f, rawB, kb);
        }
        if (MODE == ACHIP_MODE_HB_256) {
          t |= quant256(t) << 24;
          b |= quant256(b) << 24;
        } else if (MODE == ACHIP_MODE_HB_16) {
          t |= quant16(t) << 24;
          b |= quant16(b) << 24;
        } else if (MODE == ACHIP_MODE_MONO) {
          t |= (uint32_t)ramp[luma601(t) >> 2] << 24;
        }
      }
      pt[k] = t;
      pb[k] = HB ? b : 0u;
    }
  };

  /* The loop holds a block's PIXELS (pt / pb), not its samples: the samples of block b + WAVES are requested at the top of
   * block b's turn and turned into pixels in the middle of it, behind the length pass and in front of the drain.  (In the
   * product build the wave still waits right behind the requests: the copies of the request loop -- one row / several rows,
   * cached / non-temporal -- define the samples in different registers, and the phi moves that join them are moves of
   * loaded values.  A build with one load per sample has no such wait, measured: no faster here, slower on the metric's
   * shape in the stream kernel -- docs/history/round5.md 2b.) */
  uint32_t (&pt)[CPG] = rawT, (&pb)[CPG] = rawB;
  if (b0 + wave < b1)
    to_pixels(b0 + wave, (uint32_t)(WIDE ? row_c : (b0 + wave) * rpb), cm, pt, pb, kinds);
  for (int blk = b0 + wave; blk < b1; blk += WAVES) {
    uint32_t pt_n[CPG], pb_n[CPG], kinds_n = 0;
    uint32_t cm_n[WIDE ? CPG : 1];
    int row_n = 0, seg_n = 0;
    const bool more = blk + WAVES < b1;
    if constexpr (WIDE) {
      seg_n = seg_c + step_r;
      row_n = row_c + step_q;
      if (seg_n >= nseg) {
        seg_n -= nseg;
        row_n++;
      }
      if (more) {
        seg_records(seg_n, cm_n);
        issue_any(blk + WAVES, (uint32_t)row_n, cm_n, pt_n, pb_n, kinds_n);
      }
    } else {
      if (more)
        issue_any(blk + WAVES, (uint32_t)((blk + WAVES) * rpb), cm, pt_n, pb_n, kinds_n);
    }

    /* (WIDE: a block's cells are its segment's: ncb of them from column seg_x0 on) */
    const uint32_t seg_x0 = (uint32_t)(seg_c * segw);
    const uint32_t nrb = block_rows(blk), ncb = WIDE ? min((uint32_t)segw, uwp - seg_x0) : nrb * uwp,
                   row0 = (uint32_t)(WIDE ? row_c : blk * rpb), vlim = nrb << CM_ROW;
    auto next_to_pixels = [&]() {
      if constexpr (WIDE)
        to_pixels(blk + WAVES, (uint32_t)row_n, cm_n, pt_n, pb_n, kinds_n);
      else
        to_pixels(blk + WAVES, (uint32_t)((blk + WAVES) * rpb), cm, pt_n, pb_n, kinds_n);
    };
    /* the records do not change, but the compiler must not know: it would hoist every flag test out of the loop as a lane
     * mask (four scalar pairs per slot) and spill most of them -- two lane reads per use instead of the AND + compare */
#ifndef ACHIP_ROWS_HOIST_FLAGS /* A/B builds: make EXTRA=-DACHIP_ROWS_HOIST_FLAGS */
#pragma unroll
    for (int k = 0; k < CPL; k++)
      cm[k] = opaque(cm[k]);
#endif
    /* ---- run heads: one ballot per slot; a cell starts a run at or in front of its row's first pixel, or where its
     * key differs from its left neighbour's (the lane below; lane 0 takes lane 63 of the slot before) */
    uint64_t hm[CPL];
    /* (WIDE: the cell in front of the segment is lane 0 of the ghost slot) */
    auto left_T = [&](int k) { return wave_shift_up1(pt[k], k > 0 ? wave_read_lane(pt[k > 0 ? k - 1 : 0], 63) : WIDE ? wave_read_lane(pt[CPG - 1], 0) : 0u); };
    auto left_B = [&](int k) { return HB ? wave_shift_up1(pb[k], k > 0 ? wave_read_lane(pb[k > 0 ? k - 1 : 0], 63) : WIDE ? wave_read_lane(pb[CPG - 1], 0) : 0u) : 0u; };
#pragma unroll
    for (int k = 0; k < CPL; k++) {
      const bool valid = cm[k] < vlim;
      const uint32_t lT = left_T(k), lB = left_B(k); /* wave operations: outside the short-circuit below */
      const bool head = valid && ((cm[k] & CM_FIRST) != 0u || !rows_same_run<MODE>(pt[k], pb[k], lT, lB));
      hm[k] = wave_ballot(head);
    }
    /* what the words behind slot k contribute (wave-uniform): the first head above word k (the block's end closes the last
     * run), nine bits per slot in one scalar pair.  What the words in FRONT of slot k contribute -- the last head below
     * word k and its transparency -- is carried along by the length pass below.  (Held as arrays of scalars, these and
     * the transparency masks were 42 of the 96 scalar registers the seven-slot geometry spilled.) */
    static_assert(64 * CPL < 512 && 9 * CPL <= 64, "nine bits per slot");
    /* (WIDE: "the block's end" is the value 511 here -- a segment's last run may end thousands of cells further on, at
     * e_end, which takes its place where the field is read) */
    constexpr uint32_t E_BLOCK_END = 511u;
    uint64_t e_pack = 0;
    uint32_t first_head; /* the block's first head, or where the scan started: no head at all */
    {
      uint32_t next = WIDE ? E_BLOCK_END : ncb;
#pragma unroll
      for (int k = CPL - 1; k >= 0; k--) {
        e_pack |= (uint64_t)next << (9 * k);
        if (hm[k] != 0ull)
          next = (uint32_t)(64 * k + __ffsll((unsigned long long)hm[k]) - 1);
      }
      first_head = next;
    }
    /* ---- WIDE: what the segment's neighbours contribute.  Publish first, then read (the file's header) */
    uint32_t e_end = ncb;  /* where the run of the block's last cell ends, in cells from the block's first */
    int h_open = -1;       /* the head of the run that is open where the block starts (<= 0: cells in front of the block) */
    bool t_open = false;   /* ... transparent */
    bool seg_lost = false; /* a neighbour's word never came */
    if constexpr (WIDE) {
      const bool has_head = first_head != E_BLOCK_END;
      const uint32_t lead = has_head ? first_head : ncb;
      uint32_t last_head = 0, lastT = 0, lastB = 0, endT = 0, endB = 0; /* the last head's / the last cell's pixels */
#pragma unroll
      for (int k = 0; k < CPL; k++) {
        if (hm[k] != 0ull) {
          const int b = 63 - __clzll((long long)hm[k]);
          last_head = (uint32_t)(64 * k + b);
          lastT = wave_read_lane(pt[k], b);
          lastB = HB ? wave_read_lane(pb[k], b) : 0u;
        }
        if ((int)((ncb - 1u) >> 6) == k) {
          endT = wave_read_lane(pt[k], (int)((ncb - 1u) & 63u));
          endB = HB ? wave_read_lane(pb[k], (int)((ncb - 1u) & 63u)) : 0u;
        }
      }
      const bool last_t = HBC && has_head && (px_rgb(lastT) | px_rgb(lastB)) == 0u;
      /* the cell behind the segment continues the last cell's run: it exists, lies behind its row's first pixel, same key */
      const uint32_t xg = seg_x0 + ncb;
      const uint32_t gRT = wave_read_lane(pt[CPG - 1], 1), gRB = HB ? wave_read_lane(pb[CPG - 1], 1) : 0u;
      const bool cont = xg < uwp && xg > pad_left && rows_same_run<MODE>(gRT, gRB, endT, endB);
      if (lane == 0)
        slot_store(&sumw[blk - b0], rows_seg_word(lead, ncb - (has_head ? last_head : 0u), has_head, last_t, cont));
      /* the open run: only a segment whose first cell continues it needs its length; its head's transparency decides the
       * first head's SGRs -- equal rgb in the truecolor mode (the ghost cell tells), equal KEYS in the 256- / 16-colour
       * modes, where only a ghost cell with black's keys can belong to a run with a raw-black head (halfblock.c:357,476) */
      const uint32_t gLT = wave_read_lane(pt[CPG - 1], 0), gLB = HB ? wave_read_lane(pb[CPG - 1], 0) : 0u;
      const bool joined = seg_c > 0 && seg_x0 > pad_left; /* the first cell has a left neighbour in its row's image */
      bool look_back = joined && lead > 0u;
      if (HBC && joined) {
        if (MODE == ACHIP_MODE_HB_TRUE) {
          t_open = (px_rgb(gLT) | px_rgb(gLB)) == 0u;
        } else {
          const uint32_t kb = MODE == ACHIP_MODE_HB_256 ? quant256(0u) : quant16(0u);
          look_back |= px_key(gLT) == kb && px_key(gLB) == kb;
        }
      }
      if (look_back) {
        const uint32_t w = rows_seg_back(sumw, blk - b0, (uint32_t)segw);
        seg_lost |= w == 0xFFFFFFFFu;
        h_open = -(int)(w & 0xFFFFu);
        t_open = HBC && (w >> 16) != 0u;
      }
      if (cont) {
        const uint32_t w = rows_seg_ahead(sumw, blk - b0);
        seg_lost |= w == 0xFFFFFFFFu;
        e_end = ncb + (w & 0xFFFFu);
      }
    }
    /* the end of the run that is open behind word k (wave-uniform) */
    auto e_field = [&](int k) {
      const uint32_t e = (uint32_t)(e_pack >> (9 * k)) & 0x1FFu;
      return (int)(WIDE && e == E_BLOCK_END ? e_end : e);
    };
    /* ---- the token of slot k.  Built twice -- once for its length, once for the store pass -- from the pixels and the
     * masks, instead of being held: five registers per slot would put the 7-slot geometry beyond 128 VGPRs, i.e. at one
     * workgroup per CU instead of two */
    const uint32_t lastlim = (uint32_t)min(rows - 1 - (int)row0, 511) << CM_ROW; /* cells of the frame's last text row: record >= this */
    /* tmk: the transparent heads of word k (raw black over raw black: halfblock.c:357,476), remade from the pixels */
    auto make_tok = [&](int k, uint64_t tmk, int h_in, bool t_in) {
      const int s = 64 * k + lane;
      RunCtx c;
      c.is_head = lane_bit(hm[k]);
      const int below = rows_prev_bit(hm[k], lm); /* nearest head strictly below in this word */
      const int h = c.is_head ? s : (below >= 0 ? 64 * k + below : h_in);
      const int above = rows_next_bit(hm[k], lm);
      const int e = above >= 0 ? 64 * k + above : e_field(k);
      c.run = (uint32_t)(e - h);
      /* transparency of the run's head; and, for a head behind its row's first pixel, of the previous run's head */
      const bool t_below = below >= 0 ? ((tmk >> below) & 1ull) != 0ull : t_in;
      c.head_transparent = HBC && (c.is_head ? lane_bit(tmk) : t_below);
      c.state_set = HBC && c.is_head && (cm[k] & CM_FIRST) == 0u && !t_below;
      c.prevT = left_T(k); /* one DPP move each: cheaper than holding 2 x CPL registers across the block */
      c.prevB = left_B(k);
      const bool row_end = cm[k] < vlim && (cm[k] & CM_END) != 0u;
      return rows_token<MODE>(c, pt[k], pb[k], f.ops, glyph64, (cm[k] & CM_PAD) != 0u, row_end, cm[k] >= lastlim);
    };
    /* ... of a word in which EVERY cell starts a run (hm[k] = the word's valid cells: a frame without flat areas -- camera
     * noise alone does that): run = 1 but for the word's last cell, the previous run's head is the left neighbour, and none
     * of the bit scans above is needed (35 of a slot's ~320 vector instructions).  t_in as above:
     * lane 0's left neighbour lies in the word below, where it need not be a head (the 256 / 16-colour modes compare keys,
     * a run's cells may differ in raw rgb, and the HEAD's decides transparency). */
    auto make_tok_heads = [&](int k, bool t_in) {
      RunCtx c;
      c.is_head = true;
      /* (the word's LAST head may start a run that goes on in the words above: its end is the first head there) */
      c.run = lane_bit(hm[k] >> 1) ? 1u : (uint32_t)(e_field(k) - (64 * k + lane));
      c.prevT = left_T(k);
      c.prevB = left_B(k);
      c.head_transparent = HBC && (px_rgb(pt[k]) | px_rgb(pb[k])) == 0u;
      const bool t_left = lane == 0 ? t_in : (px_rgb(c.prevT) | px_rgb(c.prevB)) == 0u;
      c.state_set = HBC && (cm[k] & CM_FIRST) == 0u && !t_left;
      const bool row_end = cm[k] < vlim && (cm[k] & CM_END) != 0u;
      return rows_token<MODE>(c, pt[k], pb[k], f.ops, glyph64, (cm[k] & CM_PAD) != 0u, row_end, cm[k] >= lastlim);
    };
    /* ---- the slices' byte totals (wave-uniform).  Two slots' lengths share one scan (a slice is at most 64 x 56 bytes:
     * 16 bits each); the scans are KEPT for the store pass, which takes a cell's offset inside its slice from them */
    auto tok_len = [&](int k, const Tok &t) {
      uint32_t n = 0;
      if (cm[k] < vlim) {
        RowsCountSink<L::NUM8 ? L::o_num_len : -1> cs{{0u}};
        token_fields<MODE>(cs, t, ascii_only);
        n = cs.n;
      }
      return n;
    };
    constexpr int NPK = (CPL + 1) / 2;
    uint32_t meta[CPL]; /* meta = {flags:12, rep:12, length:6}: what the store pass cannot re-derive cheaply */
    uint32_t pk[NPK], tot2[NPK];
    uint32_t total = 0;
    {
      int last = WIDE ? h_open : -1; /* the last head below the word being processed, and whether it is transparent */
      bool lt = WIDE ? t_open : false;
#pragma unroll
      for (int j = 0; j < NPK; j++) {
        uint32_t two_n = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int k = 2 * j + h;
          if (k < CPL) {
            const uint64_t tmk = HBC ? hm[k] & wave_ballot((px_rgb(pt[k]) | px_rgb(pb[k])) == 0u) : 0ull;
#ifdef ACHIP_ROWS_NO_HEADS_PATH /* A/B builds */
            const Tok t = make_tok(k, tmk, last, lt);
#elif defined(ACHIP_ROWS_COUNT_WORDS) /* diagnostics: only the all-heads form, as the straight-line word path below */
            const Tok t = make_tok_heads(k, lt);
#else
            const Tok t = hm[k] == wave_ballot(cm[k] < vlim) ? make_tok_heads(k, lt) : make_tok(k, tmk, last, lt);
#endif
            const uint32_t n = tok_len(k, t);
            meta[k] = (n ? t.flags : 0u) | (t.rep << 12) | (n << 24);
            two_n |= n << (16 * h);
            if (hm[k] != 0ull) {
              const int b = 63 - __clzll((long long)hm[k]);
              last = 64 * k + b;
              lt = ((tmk >> b) & 1ull) != 0ull;
            }
          }
        }
        pk[j] = wave_inclusive_scan(two_n);
        tot2[j] = wave_read_lane(pk[j], 63);
        total += (tot2[j] & 0xFFFFu) + (tot2[j] >> 16);
      }
    }
    auto stot = [&](int k) { return (tot2[k / 2] >> (16 * (k & 1))) & 0xFFFFu; };
    /* ---- where the block starts in the frame (decoupled look-back over LDS words, as the stream kernel) */
    uint32_t base = first_base;
    const int lb = blk - b0; /* the block's look-back word */
    if (PARTS) {
      /* the workgroup's bytes, for the parts behind it: whoever counts the workgroup's last block publishes the sum (its own
       * add to [0] precedes its add to [1] in the LDS queue, like everybody's: the sum is complete) */
      if (lane == 0) {
        (void)slot_fetch_add(&partacc[0], total);
        const uint32_t counted = slot_fetch_add(&partacc[1], 1u);
        if ((int)counted == b1 - b0 - 1)
          agent_store_u64(&ps.sync[wg], ((unsigned long long)ps.epoch << 32) | (unsigned long long)slot_load(&partacc[0]));
      }
    }
    if (lb > 0) {
      if (lane == 0)
        slot_store(&slots[lb], ACHIP_SLOT_AGG | total);
      base = stream_lookback(slots, lb, lane);
    } else if (PARTS && part > 0) {
      /* the bytes of the parts in front: lane q waits for part q's word of this launch (bounded, like the look-back) */
      unsigned long long w = 0ull;
      bool got = lane >= part;
      if (!got)
        for (int spin = 0; spin < (1 << 22); spin++) {
          w = agent_load_u64(&ps.sync[wg - part + lane]);
          got = (uint32_t)(w >> 32) == ps.epoch;
          if (got)
            break;
          spin_nap<1>();
        }
      const bool lost = wave_ballot(!got) != 0ull;
      const uint32_t sum = wave_read_lane(wave_inclusive_scan(got ? (uint32_t)w & ACHIP_SLOT_VALUE : 0u), 63);
      base = lost ? 0xFFFFFFFFu : min(first_base + sum, cap_bytes + 1u);
    }
    const bool ok = base != 0xFFFFFFFFu && (uint64_t)base + total <= cap_bytes && !seg_lost;
    if (lane == 0)
      slot_store(&slots[lb], ACHIP_SLOT_PREFIX | (ok ? base + total : cap_bytes + 1u));

    /* (the 16-colour quantiser's temporaries do not fit next to the store pass's registers in the seven-slot geometry:
     * there the conversion waits for the end of the turn) */
    constexpr bool MID_TURN = !(MODE == ACHIP_MODE_HB_16 && CPL > 4);
    if (MID_TURN && more)
      next_to_pixels();

    uint32_t braw = 0; /* CRC: register after the block's bytes, starting from 0 */
    if (ok) {
      uint32_t a = base; /* stream offset of the slice */
      /* CARRY (plain instantiations): a slice hands the bytes behind its last whole 128-byte LINE to the next slice of the
       * block, which finds them at the front of the staging area -- between a block's first and last byte only whole
       * lines leave the wave.  Slices are ~2.6 KB: flushed one by one, every tenth line of the frame reached memory in
       * two pieces (16-byte groups and single bytes on either side of the seam), and the memory side takes lines written
       * in pieces at 4.4 instead of 5.7 TB/s (profiles/r04_rows_floor.txt).  Coordinates q = stream offset + dmis: line
       * boundaries of the ADDRESS are the multiples of 128. */
      constexpr bool CARRY = !CRC && !EMIT_OR && ACHIP_DRAIN_ALIGN >= 128u;
      constexpr bool WORDS = CARRY && L::WORDS && !(ACHIP_ROWS_SLOT_EMIT != 0);
      int last_k = 0;
#pragma unroll
      for (int k = 0; k < CPL; k++)
        last_k = stot(k) != 0u ? k : last_k;
      uint32_t own_q = base + dmis; /* first byte of the block that has not left the wave yet */
#pragma unroll
      for (int k = 0; k < CPL; k++) {
        const uint32_t n = stot(k);
        if (n == 0u)
          continue;
        const uint32_t len_k = meta[k] >> 24;
        const uint32_t off_k = ((pk[k / 2] >> (16 * (k & 1))) & 0xFFFFu) - len_k; /* within the slice */
        const Tok tk = rows_token_payload<MODE>(meta[k] & 0xFFFu, (meta[k] >> 12) & 0xFFFu, pt[k], pb[k], f.ops, glyph64);
        /* SLOTS: every lane builds its token in a slot of its OWN, an odd number of dwords from its neighbours' -- the byte
         * stores of a half-wave then fall into distinct banks (packed at the stream's offsets they land ~38 bytes apart at
         * pseudo-random banks: 56 % of the launch's LDS-active cycles were conflicts, profiles/r05_k5_sampled_sq_counters.txt)
         * -- reads it back as dwords and stores it at its place in the frame itself: whole 16-byte pieces, then 8 / 4 / 2 / 1
         * by the bits of its length.  No packed image of the slice, no drain, no carry; the L2 merges the pieces of a line. */
        constexpr bool SLOTS = !CRC && !EMIT_OR && (ACHIP_ROWS_SLOT_EMIT != 0);
        if (SLOTS) {
          constexpr uint32_t PITCH = 4u * (uint32_t)((rows_max_token(MODE) / 4 + 1) | 1); /* odd dwords; >= token + the 3 bytes a field may store past it */
          static_assert(!SLOTS || 64 * (int)PITCH <= L::STAGE, "slots fit the wave's staging area");
          const uint32_t slot = PITCH * (uint32_t)lane;
          if (len_k != 0u) {
            RowsFastSink<L::o_dec, L::o_flags + 16, L::NUM8 ? L::o_num_semi : -1, L::NUM8 ? L::o_num_m : -1> fs{{stage_addr + slot, dummy_addr}};
            token_fields<MODE>(fs, tk, ascii_only);
          }
          lds_store_fence();
          if (len_k != 0u) {
            uint8_t *g = dst + (a + off_k);
            const unsigned char *sl = stage + slot;
            const uint32_t *sw = reinterpret_cast<const uint32_t *>(sl);
            constexpr int BODY = rows_max_token(MODE) / 16; /* whole 16-byte pieces a token can have */
#pragma unroll
            for (int j = 0; j < BODY; j++)
              if (len_k >= 16u * (uint32_t)(j + 1))
                store_u4_unaligned(g + 16 * j, make_uint4(sw[4 * j], sw[4 * j + 1], sw[4 * j + 2], sw[4 * j + 3]));
            uint32_t o = len_k & ~15u;
            if (len_k & 8u) {
              store_u2_unaligned(g + o, sw[o >> 2], sw[(o >> 2) + 1]);
              o += 8u;
            }
            if (len_k & 4u) {
              store_u1_unaligned(g + o, sw[o >> 2]);
              o += 4u;
            }
            if (len_k & 2u) {
              store_u16_unaligned(g + o, *reinterpret_cast<const uint16_t *>(sl + o));
              o += 2u;
            }
            if (len_k & 1u)
              g[o] = sl[o];
          }
          wave_lockstep(); /* the next slice's tokens go where these were read from (a wave's DS operations complete in order) */
          a = a + n;
          continue;
        }
        if (CARRY) {
          /* the staging area's byte 0 is the line the slice starts in (q0); [own_q, qa) is already there */
          const uint32_t qa = a + dmis, q0 = qa & ~127u, qend = qa + n;
          /* WORDS: the SGRs of a token go out as 6 / 7 dword ORs each (word_sgr; the background's carries the half
           * block), whatever else a token holds -- the reset in front of a transparent run, its space, a lone half block,
           * a repeat count, a row's reset and newline, a padding space -- as bytes around them: token_fields' order.  The
           * area is zero wherever no token has been written (cleared behind every flush, below). */
          if (WORDS) {
            const uint32_t fl = meta[k] & 0xFFFu;
            constexpr uint32_t FULL = (uint32_t)(TF_SGR_FG | TF_SGR_BG | TF_GLYPH);
            constexpr uint32_t TAIL = (uint32_t)(TF_ROW_RESET | TF_NL);
#ifdef ACHIP_ROWS_COUNT_WORDS /* diagnostics (scripts/isa_lines.py): only the straight-line form is compiled, so that the listing's
                                 static counts are what a slice of a frame without flat areas executes */
            if (true) {
#else
            if (wave_ballot(len_k != 0u && (fl & ~TAIL) != FULL) == 0ull) {
#endif
              /* every token of the slice has both SGRs and nothing in front of the row's end (a frame without flat
               * areas): straight through, no lane branches */
              if (len_k != 0u) {
                FastSink<L::o_dec, L::o_flags + 16> fs{stage_addr + (qa + off_k - q0), dummy_addr};
                const WordFields wf = word_fields<L::o_wr, L::o_wg, L::o_wm>(tk.fg), wb = word_fields<L::o_wr, L::o_wg, L::o_wmg>(tk.bg);
                fs.a += word_sgr<false>(fs.a, wf, 0x38335B1Bu) >> 3;
                fs.a += word_sgr<true>(fs.a, wb, 0x38345B1Bu) >> 3;
                if (fl & TAIL) {
                  if (fl & TF_ROW_RESET)
                    put_reset(fs);
                  if (fl & TF_NL)
                    fs.template c<1>('\n');
                }
              }
            } else if (len_k != 0u) {
              FastSink<L::o_dec, L::o_flags + 16> fs{stage_addr + (qa + off_k - q0), dummy_addr};
              /* (the table reads of both SGRs in front of the lanes' branches: they overlap, and a lane without the SGR reads harmlessly) */
              const WordFields wf = word_fields<L::o_wr, L::o_wg, L::o_wm>(tk.fg), wb = word_fields<L::o_wr, L::o_wg, L::o_wmg>(tk.bg);
              if (fl & TF_PAD) {
                fs.template c<1>(' ');
              } else {
                if (fl & TF_RESET_PRE)
                  put_reset(fs);
                if (fl & TF_SGR_FG)
                  fs.a += word_sgr<false>(fs.a, wf, 0x38335B1Bu) >> 3;
                if (fl & TF_SGR_BG) { /* (only in front of a half block: rows_token) */
                  fs.a += word_sgr<true>(fs.a, wb, 0x38345B1Bu) >> 3;
                } else {
                  if (fl & TF_SPACE)
                    fs.template c<1>(' ');
                  if (fl & TF_GLYPH)
                    fs.template c<3>(0x8096E2u);
                }
                if (fl & TF_REP)
                  put_rep(fs, tk.rep);
                if (fl & TF_ROW_RESET)
                  put_reset(fs);
                if (fl & TF_NL)
                  fs.template c<1>('\n');
              }
            }
          } else if (len_k != 0u) {
            RowsFastSink<L::o_dec, L::o_flags + 16, L::NUM8 ? L::o_num_semi : -1, L::NUM8 ? L::o_num_m : -1> fs{{stage_addr + (qa + off_k - q0), dummy_addr}};
            token_fields<MODE>(fs, tk, ascii_only);
          }
          lds_store_fence();
          const bool last = k == last_k;
          const uint32_t q0n = qend & ~127u;           /* the line the next slice starts in */
          const uint32_t lim = last ? qend : q0n;      /* bytes [own_q, lim) leave now */
          if (lim > own_q) {
            const uint32_t vb = (own_q + 15u) & ~15u, ve = lim & ~15u;
            for (uint32_t q = (vb & ~127u) + 16u * (uint32_t)lane; q < ve; q += 1024u)
              if (q >= vb)
                store_out16(dst + (q - dmis), *reinterpret_cast<const uint4 *>(stage + (q - q0)));
            const uint32_t head_end = min(vb, lim); /* (own_q is a line boundary from the block's second flush on) */
            if (own_q + (uint32_t)lane < head_end)
              dst[own_q + (uint32_t)lane - dmis] = stage[own_q + (uint32_t)lane - q0];
            const uint32_t tail_begin = max(ve, head_end);
            if (lane >= 32 && tail_begin + (uint32_t)(lane - 32) < lim)
              dst[tail_begin + (uint32_t)(lane - 32) - dmis] = stage[tail_begin + (uint32_t)(lane - 32) - q0];
            own_q = lim;
          }
          wave_lockstep(); /* the next slice's tokens (and the move below) go where these bytes were read from */
          if (!last && q0n != q0) { /* [q0n, qend) moves to the front: < 128 bytes, from at least 128 bytes further up */
            uint4 c = make_uint4(0u, 0u, 0u, 0u);
            if (lane < 8)
              c = *reinterpret_cast<const uint4 *>(stage + (q0n - q0) + 16u * (uint32_t)lane);
            if (WORDS) { /* (a wave's DS operations complete in order: read, clear, write) */
              wave_lockstep();
              for (uint32_t g = (uint32_t)lane; g < (qend - q0 + 15u) >> 4; g += 64u)
                lds_ptr<uint4>((int)stage_off)[g] = make_uint4(0u, 0u, 0u, 0u);
              wave_lockstep();
            }
            if (lane < 8)
              *lds_ptr<uint4>((int)stage_off + 16 * lane) = c;
            wave_lockstep();
          } else if (WORDS && last) {
            for (uint32_t g = (uint32_t)lane; g < (qend - q0 + 15u) >> 4; g += 64u)
              lds_ptr<uint4>((int)stage_off)[g] = make_uint4(0u, 0u, 0u, 0u);
            wave_lockstep();
          }
          a = a + n;
          continue;
        }
        /* ---- the slice's tokens into this wave's staging area: stream offset g0 (16-byte aligned) sits at its byte 0 */
        const uint32_t g0 = a & ~15u;
        if (CRC && !EMIT_OR && lane == 0) /* the bytes in front of the slice inside its first group read as zero for the checksum */
          *lds_ptr<uint4>((int)stage_off) = make_uint4(0u, 0u, 0u, 0u);
        if (len_k != 0u) {
          if (EMIT_OR) {
            PackSink<L::o_dec> ps(stage_addr + (a + off_k - g0));
            token_fields<MODE>(ps, tk, ascii_only);
            ps.finish();
          } else {
            RowsFastSink<L::o_dec, L::o_flags + 16, L::NUM8 ? L::o_num_semi : -1, L::NUM8 ? L::o_num_m : -1> fs{{stage_addr + (a + off_k - g0), dummy_addr}};
            token_fields<MODE>(fs, tk, ascii_only);
          }
        }
        lds_store_fence();
        /* ---- staging -> HBM: whole 16-byte groups as uint4, the shared first / last group as bytes */
        const uint32_t end = a + n;
        const uint32_t vec_begin = (a + 15u) & ~15u, vec_end = end & ~15u;
        /* (whole 128-byte lines per store instruction, as the stream kernel's drain) */
        for (uint32_t q = ((vec_begin + dmis) & ~(ACHIP_DRAIN_ALIGN - 1u)) + 16u * (uint32_t)lane; q < vec_end + dmis; q += 1024u)
          if (q >= vec_begin + dmis)
            store_out16(dst + (q - dmis), *reinterpret_cast<const uint4 *>(stage + (q - dmis - g0)));
        const uint32_t head_end = min(vec_begin, end);
        if (a + (uint32_t)lane < head_end)
          dst[a + (uint32_t)lane] = stage[a + (uint32_t)lane - g0];
        const uint32_t tail_begin = max(vec_end, head_end);
        if (lane >= 32 && tail_begin + (uint32_t)(lane - 32) < end)
          dst[tail_begin + (uint32_t)(lane - 32)] = stage[tail_begin + (uint32_t)(lane - 32) - g0];
        wave_lockstep(); /* the next slice's tokens go where these bytes were read from */
        if (CRC) { /* raw(block so far || slice) = raw(block so far) * x^(8 n) xor raw(slice): both wave-uniform */
          const uint32_t xk = lds_ptr<const uint32_t>(L::o_xk)[lane];
          const uint32_t sraw = stream_crc_staged<L>(stage, (a & 15u) + n, lane);
          braw = braw ? wave_mulmod_uniform(braw, wave_x8_pow_uniform(lds_ptr<const uint32_t>(L::o_pow), n, lane, xk), lane, xk) ^ sraw
                      : sraw;
        }
        if (EMIT_OR) { /* the next slice ORs into zeros */
          const uint32_t used = ((a & 15u) + n + 15u) >> 4;
          for (uint32_t g = (uint32_t)lane; g < used; g += 64u)
            lds_ptr<uint4>((int)stage_off)[g] = make_uint4(0u, 0u, 0u, 0u);
          wave_lockstep();
        }
        a = end;
      }
    }
    if (blk == nblk - 1 && lane == 0) {
      out_len[fidx] = ok ? base + total : ACHIP_LEN_OVERFLOW;
      if (ok && (uint64_t)base + total < out_stride)
        dst[base + total] = 0; /* NUL behind the frame when the slot has room, as the reference's strings carry */
    }
    if (CRC) {
      if (ok)
        stream_crc_place<L>(slots, nblk, nblk_cap, blk, braw, base + total, cap_bytes, lane);
      stream_crc_finish<L>(slots, nblk, nblk_cap, cap_bytes, first_base, fidx, dim_w, dim_h, wire, lane);
    }

    if (!MID_TURN && more)
      next_to_pixels();
#pragma unroll
    for (int k = 0; k < CPG; k++) {
      pt[k] = pt_n[k];
      pb[k] = pb_n[k];
      if constexpr (WIDE)
        cm[k] = cm_n[k];
    }
    row_c = row_n;
    seg_c = seg_n;
  }
  (void)glyph;
}

} // namespace achip
