/*
 * render_stream.hpp -- the wave-autonomous ("stream") frame kernel for the per-cell renderers of the path:
 *   image_print_color (truecolor foreground, all-ASCII palette)   lib/video/ascii/scalar/foreground.c:195-308
 *   image_print_256color                                          foreground.c:433-509
 *   image_print_16color                                           foreground.c:535-624
 *   image_print_color_background                                  lib/video/ascii/scalar/background.c:17-84
 * whose tokens depend on a cell and (truecolor-fg only) its raster predecessor -- no run structure.
 *
 * Why a second kernel.  render_frames_kernel (render_kernels.hpp) walks a frame in workgroup-wide phases fenced by
 * barriers: every wave waits at the barrier behind the gather until the SLOWEST wave's samples have arrived, and the
 * chip alternates between a fabric-bound burst (all CUs gathering) and latency-bound token work (memory idle).  Here a
 * frame is cut into blocks of 64 x CPL consecutive cells and every WAVE takes a block through the whole path on its
 * own -- gather -> tokens -> wave scan -> look-back -> token stores into a wave-private LDS staging area -> 16-byte
 * stores to HBM -- with ONE workgroup barrier, in the prologue (glyph / decimal tables and the look-back words in
 * LDS).  Waves whose samples have arrived tokenise and drain while the others' requests are still in flight; the next
 * block's samples are requested before the current block is tokenised.  41-58 VGPRs: two 1024-thread (or four
 * 512-thread) workgroups share a CU where the phase kernel needs 125.
 *
 * What the timeline of a launch showed (profiles/r02_stream_timeline.txt) and the code answers:
 *   * the prologue was six DEPENDENT scalar-load round trips to the kernarg segment (the compiler loads an argument at
 *     its first use, behind the branches above it): all arguments are now requested in one burst at the top;
 *   * a wave alone in the tail of a launch runs at one instruction per ~5 cycles, so instruction COUNT is latency:
 *     the sampler is branch-free (flips folded into the index, cached / non-temporal chosen once per frame) and the
 *     per-cell decisions are selects, not exec-mask branches;
 *   * requests are issued only as fast as lines return (the load instruction itself stalls once the CU's miss queue
 *     is full): the gather is bound by the ~7 TB/s at which the fabric fills 128-byte lines, HBM or Infinity Cache
 *     alike (profiles/r02_ubench_sparse_policy.txt: no load flavour or memory type fetches less than a whole line).
 *
 * The only cross-wave dependency is the byte offset of a block in the frame (variable-length output):
 * a decoupled look-back over one LDS word per block -- {state:2, bytes:30}; state 1 = this block's own byte count
 * (published right after its wave scan), 2 = inclusive prefix -- read 64 predecessors per poll, one lane each.
 * All waves of a frame live in one workgroup, so the words are workgroup-scope LDS atomics; no global traffic.
 *
 * Everything byte-level (token grammar, sinks, sampler, quantisers) is shared with render_kernels.hpp.
 */
#pragma once

#include "render_kernels.hpp"
#include "crc_math.hpp" /* the GF(2) toolbox of the wire stage: the frame CRC can ride the drain */

namespace achip {

/* The truecolor-foreground renderer with a palette that holds multi-byte glyphs (three of the reference's five built-in
 * palettes: BLOCKS, DIGITAL, COOL -- palette.h:161-197) is an instantiation of its own (the all-ASCII one keeps its registers):
 * this tag in the MODE parameter of SLds / render_stream_kernel.  foreground.c:281-296: such a cell always carries its SGR
 * and leaves the RLE state alone, so an ASCII cell is compared with the nearest EARLIER ASCII cell of the frame. */
#define ACHIP_STREAM_MODE_TRUE_FG_U8 (16 + ACHIP_MODE_TRUE_FG)
__host__ __device__ constexpr bool mode_is_true_fg(int m) { return m == ACHIP_MODE_TRUE_FG || m == ACHIP_STREAM_MODE_TRUE_FG_U8; }
__host__ __device__ constexpr bool mode_is_cell(int m) {
  return mode_is_true_fg(m) || m == ACHIP_MODE_256_FG || m == ACHIP_MODE_16_FG || m == ACHIP_MODE_TRUE_BG;
}
/* longest token of a mode, bytes (SURVEY 8a "per-token byte lengths"): SGR(s) + glyph + row reset + newline */
__host__ __device__ constexpr int stream_max_token(int m) {
  return m == ACHIP_MODE_TRUE_FG ? 24    /* 19 + 1-byte glyph (all-ASCII palettes only) + max(NL, final reset 4) */
         : m == ACHIP_STREAM_MODE_TRUE_FG_U8 ? 28 /* 19 + glyph <= 4 + max(NL, final reset 4), rounded up */
         : m == ACHIP_MODE_256_FG ? 20   /* 11 + glyph <= 4 + reset 4 + NL                                      */
         : m == ACHIP_MODE_16_FG ? 16    /* 5 + 4 + 4 + 1 (rounded up)                                           */
                                 : 48;   /* background: 19 + 19 + 4 + 4 + 1 (rounded up)                         */
}

#define ACHIP_STREAM_MAXBLK 2048            /* blocks per frame the look-back table holds              */
#define ACHIP_STREAM_MAX_STRIDE 0x3F000000u /* block prefixes are 30-bit: slab slots up to ~1 GB        */

/* PACK instantiations (frames written at their exact length, below): the whole frame is staged in LDS -- this many bytes
 * at most -- instead of one block per wave.  PACK = 1: frames only; 2: + frame CRC, packet header, packet CRC, computed
 * from the frame's LDS image by the whole workgroup (crc_kernels.hpp's scheme: 20 KB of tables instead of the 52 KB of
 * the per-block checksum, so two 8-wave workgroups still share a CU). */
#ifndef ACHIP_STREAM_WORD_EMIT
#define ACHIP_STREAM_WORD_EMIT 1 /* 0 (A/B builds): every token byte as a byte store */
#endif
#define ACHIP_PACK_FRAME_CAP (48 * 1024)
/* PACK == 2: how many waves of the workgroup checksum the frame's image (the Horner table of the prebuilt image is the one
 * of that many threads: crc_frame_tables_init_kernel<64 * pack_crc_waves(WAVES)>) */
#ifndef ACHIP_PACK_CRCW
#define ACHIP_PACK_CRCW 4
#endif
constexpr int pack_crc_waves(int waves) { return waves >= 8 ? ACHIP_PACK_CRCW : 1; }

template <int MODE, int WAVES, int CPL, bool CRC = false, int PACK = 0> struct SLds {
  static_assert(!(CRC && PACK), "exact-length instantiations checksum the frame's LDS image as a whole");
  static constexpr int BLK = 64 * CPL;
  /* cells a block OWNS.  Truecolor-fg decides its SGR against the raster predecessor (ansi_rle_add_pixel): slot
   * (k = 0, lane 0) of every block is a ghost that samples the cell in front of the block and owns no token, so that
   * every cell finds its predecessor one lane down (one DPP move) and no cell needs a second sampler pass. */
  static constexpr bool U8 = MODE == ACHIP_STREAM_MODE_TRUE_FG_U8;
  static_assert(!U8 || (!CRC && !PACK), "multi-byte palettes: the plain instantiation only");
  static constexpr int EFF = BLK - (mode_is_true_fg(MODE) ? 1 : 0);
  static constexpr int STAGE = BLK * stream_max_token(MODE) + 16; /* + the 16-byte group the block starts in */
  static constexpr int GPL = (STAGE / 16 + 63) / 64; /* 16-byte groups of a block per lane when it is checksummed */
  static constexpr int o_stage = 16; /* (a word-built SGR at the area's first byte ORs a zero into the dword in front of it) */
  /* PACK: [16 bytes: the frame's offset in the destination] here; the frame's image lies BEHIND everything else
   * (frame_off(), as long as the launch's largest frame can be: a small footprint lets workgroups share a CU) */
  static constexpr int o_packoff = 0;
  static constexpr int o_glyph = PACK ? 16 : o_stage + WAVES * STAGE;
  static constexpr int o_ramp = o_glyph + 256 * 4;
  static constexpr int o_dec = o_ramp + 64;
  /* WORDS: truecolor-fg SGRs leave the registers as aligned dword ORs (render_kernels.hpp word_sgr): their tables; the
   * staging areas start out zero and are cleared behind every drain.  (Not the instantiations that checksum the staged
   * bytes or keep the whole frame in LDS: their staging is read again, or shared by the waves; nor truecolor backgrounds:
   * one register more than the shared-out form's seven waves per SIMD leave; nor the 256-colour SGRs: level, WordSink.) */
  static constexpr bool WORDS = !CRC && !PACK && (ACHIP_STREAM_WORD_EMIT != 0 || U8) && mode_is_true_fg(MODE);
  static constexpr int o_wr = o_dec + 256 * 4;
  static constexpr int o_wg = o_wr + (WORDS ? 256 * 8 : 0);
  static constexpr int o_wm = o_wg + (WORDS ? 256 * 8 : 0);
  static constexpr int o_g8 = o_wm + (WORDS ? 256 * 8 : 0);    /* the lean loop's glyphs as bytes (all-ASCII palettes; U8: length | ASCII << 7) */
  static constexpr int o_flags = o_g8 + (WORDS ? 256 : 0);     /* [+16 ..] swallows predicated-off byte stores */
  /* CRC instantiations: constant tables, copied from global memory where crc_tables_init_kernel put them -- the 16
   * slicing tables; window tables of the lanes' multipliers; x^(8v), x^(8*256v), x^(8*65536v); x^k (k = 0..62) -- then
   * accumulator, deferred and finished counts */
  static constexpr int o_comp = o_flags + 32 + 64 * 4; /* composite descriptor of a GENERIC launch (comp_stage) */
  static constexpr int o_tab = o_comp + ACHIP_COMP_LDS_BYTES;
  static constexpr int o_fcrc = o_tab; /* PACK == 2: crc_kernels.hpp's tables (CrcLds layout), built by the workgroup */
  static constexpr int o_slice = o_tab + (PACK == 2 ? CrcLds::bytes : 0);
  static constexpr int o_lanek = o_slice + (CRC ? 16 * 1024 : 0);
  static constexpr int base_crc = o_lanek + (CRC ? 3 * 1024 + 256 + 16 + 2 * ACHIP_STREAM_MAXBLK * 4 : 0);
  static constexpr int WIN = base_crc + 32 * 1024 <= 160 * 1024 ? 4 : 2; /* bits per window of the lane multiply */
  static constexpr int NWIN = 32 / WIN;
  static constexpr int o_pow = o_lanek + (CRC ? NWIN * (1 << WIN) * 64 * 4 : 0);
  static constexpr int o_xk = o_pow + (CRC ? 3 * 1024 : 0);
  static constexpr int TAB_BYTES = (CRC ? o_xk + 256 : o_tab) - o_tab; /* the image crc_tables_init_kernel writes */
  static constexpr int o_crcacc = o_tab + TAB_BYTES + (PACK == 2 ? CrcLds::bytes : 0); /* [0] acc [1] deferred [2] finished */
  /* per-block words, as many as the launch's largest frame has blocks: look-back words {state:2, bytes:30}, and
   * (CRC) behind them the raw CRC of a block that could not be placed yet */
  static constexpr int o_slots = o_crcacc + (CRC ? 16 : 0);
  /* (U8: behind them the blocks' {state:2, has:1, rgb:24} words of the second look-back, the RLE state) */
  static constexpr int bytes_for(int maxblk) { return o_slots + maxblk * 4 * (CRC || U8 ? 2 : 1); }
  /* PACK: where the frame's image starts, and the LDS of a launch whose frames are at most `bound` bytes */
  static constexpr int frame_off(int maxblk) { return (bytes_for(maxblk) + 15) & ~15; }
  static constexpr int bytes_for_pack(int maxblk, int bound) { return frame_off(maxblk) + ((bound + 15) & ~15) + 64; }
  static constexpr int bytes = PACK ? bytes_for_pack(ACHIP_STREAM_MAXBLK, ACHIP_PACK_FRAME_CAP) : bytes_for(ACHIP_STREAM_MAXBLK);
  static_assert(STAGE % 16 == 0 && o_tab % 16 == 0 && TAB_BYTES % 16 == 0 && CrcLds::bytes % 16 == 0, "16-byte aligned areas");
  static_assert(bytes <= 160 * 1024, "one workgroup's LDS");
};
/* blocks the per-block LDS words of a launch hold: from the largest frame's cells when the host states them */
__host__ __device__ constexpr int stream_maxblk(uint32_t uniform_flags, int blk_cells) {
  const uint32_t cells = uniform_flags >> ACHIP_UNIFORM_MAX_CELLS_SHIFT;
  const uint32_t nb = (cells + (uint32_t)blk_cells - 1u) / (uint32_t)blk_cells;
  return cells == 0u || nb > (uint32_t)ACHIP_STREAM_MAXBLK ? ACHIP_STREAM_MAXBLK : (int)nb;
}

/* ---- workgroup-scope LDS words of the look-back (gfx950_ops.hpp: relaxed workgroup-scope atomics) ---------------- */
__device__ inline void slot_store(uint32_t *p, uint32_t v) { wg_store_u32(p, v); }
__device__ inline uint32_t slot_load(const uint32_t *p) { return wg_load_u32(p); }
__device__ inline uint32_t slot_fetch_add(uint32_t *p, uint32_t v) { return wg_fetch_add_u32(p, v); }
__device__ inline void slot_xor(uint32_t *p, uint32_t v) { wg_xor_u32(p, v); }
#define ACHIP_SLOT_AGG (1u << 30)
#define ACHIP_SLOT_PREFIX (2u << 30)
#define ACHIP_SLOT_VALUE 0x3FFFFFFFu

/* bytes of the frame in front of block `blk` (blk >= 1): sums the predecessors' words, newest first, 64 per
 * poll, until an inclusive prefix is met.  Returns 0xFFFFFFFF if a predecessor never publishes (bounded). */
__device__ inline uint32_t stream_lookback(const uint32_t *slots, int blk, int lane) {
  uint32_t acc = 0;
  int hi = blk - 1;
  for (int spin = 0; spin < (1 << 22);) {
    const int j = hi - lane;
    const uint32_t v = j >= 0 ? slot_load(&slots[j]) : 0u;
    const uint32_t st = j >= 0 ? v >> 30 : 3u; /* lanes in front of block 0 neither block nor contribute */
    const uint64_t pm = wave_ballot(st == 2u), am = wave_ballot(st != 0u);
    if (pm != 0ull) {
      const int P = __ffsll((unsigned long long)pm) - 1; /* nearest predecessor that knows its prefix */
      const uint64_t need = P ? ((1ull << P) - 1ull) : 0ull;
      if ((am & need) == need) {
        const uint32_t c = lane <= P ? (v & ACHIP_SLOT_VALUE) : 0u;
        return acc + wave_read_lane(wave_inclusive_scan(c), 63);
      }
    } else if (am == ~0ull) { /* 64 own counts and no prefix yet: take them and look further back */
      acc += wave_read_lane(wave_inclusive_scan(v & ACHIP_SLOT_VALUE), 63);
      hi -= 64;
      continue;
    }
    spin++;
    spin_nap<1>();
  }
  return 0xFFFFFFFFu;
}

/* ---- the RLE state of truecolor foreground with multi-byte palettes (ACHIP_STREAM_MODE_TRUE_FG_U8) -------------------- */
/* For every lane: the value of the nearest LOWER lane whose flag is set (found = there is one).  lt_lo / lt_hi = the lane's
 * "lanes below me" masks. */
struct NearestLower {
  bool found;
  uint32_t val;
};
__device__ inline NearestLower nearest_lower(bool flag, uint32_t val, uint32_t lt_lo, uint32_t lt_hi) {
  const uint64_t m = wave_ballot(flag);
  const uint32_t a = (uint32_t)m & lt_lo, b = (uint32_t)(m >> 32) & lt_hi;
  const int src = b ? 63 - __clz((int)b) : 31 - __clz((int)a); /* a == b == 0: lane -1, not taken */
  return NearestLower{(a | b) != 0u, wave_shfl(val, src < 0 ? 0 : src)};
}
/* the flagged lane with the highest number: its value, wave-uniform (m = the ballot of the flags, not zero) */
__device__ inline uint32_t highest_flagged(uint64_t m, uint32_t val) {
  const uint32_t hi = (uint32_t)(m >> 32);
  return wave_read_lane(val, hi ? 63 - __clz((int)hi) : 31 - __clz((int)(uint32_t)m));
}
/* One word per block {state:2, has:1 (bit 24), rgb:24}: state 1 = the block's own last ASCII cell (has = it holds one),
 * 2 = the frame's last ASCII cell up to and including the block.  Returns the state in front of block `lb` (lb >= 1) as
 * {has, rgb}: the nearest predecessor that decides -- an inclusive word, or an own word that holds a cell -- with every
 * block between published (own words without a cell are skipped); 0xFFFFFFFF if a predecessor never publishes. */
#define ACHIP_RLE_HAS (1u << 24)
__device__ inline uint32_t stream_lookback_rle(const uint32_t *words, int lb, int lane) {
  int hi = lb - 1;
  for (int spin = 0; spin < (1 << 22);) {
    const int j = hi - lane;
    const uint32_t v = j >= 0 ? slot_load(&words[j]) : ACHIP_SLOT_PREFIX; /* in front of the frame: no cell yet */
    const uint32_t st = v >> 30;
    const uint64_t dm = wave_ballot(st == 2u || (st == 1u && (v & ACHIP_RLE_HAS) != 0u)), pm = wave_ballot(st != 0u);
    if (dm != 0ull) {
      const int P = __ffsll((unsigned long long)dm) - 1;
      const uint64_t need = P ? ((1ull << P) - 1ull) : 0ull;
      if ((pm & need) == need)
        return wave_read_lane(v, P) & (ACHIP_RLE_HAS | 0x00FFFFFFu);
    } else if (pm == ~0ull) { /* 64 blocks without an ASCII cell: further back */
      hi -= 64;
      continue;
    }
    spin++;
    spin_nap<1>();
  }
  return 0xFFFFFFFFu;
}

struct CellPos {
  uint32_t rr, xp; /* text row, column of the padded row */
};

__device__ inline uint32_t dec_digits(uint32_t v) { return 1u + (v >= 10u ? 1u : 0u) + (v >= 100u ? 1u : 0u); }

/* The frame's source as the sampler needs it -- all wave-uniform (SGPRs) */
struct StreamSrc {
  const uint8_t *base;
  uint32_t stride, xr, yr, w1, h1; /* w1 = src_w - 1 */
  bool flip_x, flip_y, nt;
};

/* GENERIC = false: the fast sampler -- one unaligned dword per sample with the flips folded into the index, no
 * branches (nothing between two requests waits for memory); requires a single source of >= 2 pixels.
 * GENERIC = true: sample_frame_raw's full repertoire (virtual composite canvas, 1x1 sources). */
template <bool GENERIC, bool NT, int O_COMP>
__device__ inline uint32_t stream_request(const achip_frame_t &f, const StreamSrc &s, uint32_t x, uint32_t y,
                                          uint32_t &kind, const CompHead &head) {
  if (GENERIC)
    return sample_frame_raw<true, O_COMP>(f, x, y, kind, &head);
  uint32_t sx = min((x * s.xr) >> 16, s.w1), sy = min((y * s.yr) >> 16, s.h1);
  sx = s.flip_x ? s.w1 - sx : sx;
  sy = s.flip_y ? s.h1 - sy : sy;
  /* sy, sx < 10 000 and the row stride < 2^24 (checked on the host): full-rate 24-bit multiplies; the 32-bit
   * v_mul_lo_u32 the compiler would pick runs at a quarter of the rate */
  const uint32_t a = __umul24(sy, s.stride) + __umul24(sx, 3u);
  const uint32_t back = a != 0u ? 1u : 0u; /* byte before the pixel + the pixel: never past the last pixel */
  kind = back ? RAW_BACK : RAW_FIRST;
  const ACHIP_GLOBAL uint8_t *p = (const ACHIP_GLOBAL uint8_t *)s.base + (a - back);
  if (NT)
    return load_u32_unaligned_nt((const uint8_t *)p);
  return ((const ACHIP_GLOBAL unaligned_u32 *)p)->v;
}

template <int P> struct StreamPhase { /* which pass of the lean loop (LENGTH-FIRST builds: 1 = lengths, 2 = emission; else 0) */
  static constexpr int value = P;
};
struct StreamTagNT {
  static constexpr bool value = true;
};
struct StreamTagCached {
  static constexpr bool value = false;
};

/* ---- GF(2) helpers of the fused frame CRC ----------------------------------------------------- */
/* x^(8n) for wave-uniform n from the LDS copies of CRC_POW_TAB's levels (n < 2^24; above: bit-serial fallback) */
__device__ inline uint32_t wave_x8_pow_uniform(const uint32_t *pw, uint32_t n, int lane, uint32_t xk) {
  if (n >> 24)
    return crc_x8_pow_len(n);
  uint32_t r = pw[n & 0xFFu];
  if (n >> 8)
    r = wave_mulmod_uniform(r, pw[256 + ((n >> 8) & 0xFFu)], lane, xk);
  if (n >> 16)
    r = wave_mulmod_uniform(r, pw[512 + (n >> 16)], lane, xk);
  return r;
}
/* register s after m more zero bytes, 0 <= m <= 16: byte i of s is followed by m-1-i bytes (slicing rows) */
__device__ inline uint32_t crc_advance16(const uint32_t *slice, uint32_t s, int m) {
  uint32_t r = m < 4 ? (m ? s >> (8 * m) : s) : 0u;
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (i < m)
      r ^= slice[(m - 1 - i) * 256 + ((s >> (8 * i)) & 0xFFu)];
  return r;
}

/* The constant tables of a CRC instantiation with LDS layout L (SLds<.., true> here, RLds<.., true> in render_rows.hpp),
 * written once per process into global memory (one workgroup of 256 threads; the launchers run it before the first such
 * launch); every launch copies the image into LDS.  Layout = L from o_tab on. */
template <class L> __global__ void __launch_bounds__(256) crc_tables_init_kernel(uint32_t *tab) {
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t *slice = tab + (L::o_slice - L::o_tab) / 4;
  uint32_t *nib = tab + (L::o_lanek - L::o_tab) / 4;
  uint32_t *pw = tab + (L::o_pow - L::o_tab) / 4;
  uint32_t *xkt = tab + (L::o_xk - L::o_tab) / 4;
  /* slice[k][b] = register after byte b followed by k zero bytes (as crc_build_tables) */
  slice[tid] = crc_byte(0u, (uint32_t)tid);
  for (int k = tid; k < 768; k += 256)
    pw[k] = CRC_POW_TAB.t[k >> 8][k & 0xFF];
  if (wave == 0) /* x^k mod P for the lane-parallel multiply; lane 63 takes no part */
    xkt[lane] = lane == 63 ? 0u : (lane < 32 ? 0x80000000u >> lane : crc_mulmod(1u, 0x80000000u >> (lane - 31)));
  if (wave == 1) {
    /* lane l's GPL groups of a block are followed by 16*GPL*(63-l) bytes of the other lanes' groups: the lane's
     * multiplier K_l = x^(8*16*GPL*(63-l)) as window tables, W[j][v][l] = (v placed at window j) * K_l, so that
     * s * K_l is NWIN conflict-free lookups (the lane index is the bank).  Built bit by bit: K_l * x^i, i = 0..31. */
    uint32_t kx = crc_x8_pow_len(16u * (uint32_t)L::GPL * (uint32_t)(63 - lane));
    constexpr int WIN = L::WIN, NV = 1 << L::WIN;
#pragma unroll
    for (int j = 0; j < L::NWIN; j++) {
      /* window j = bits [32-WIN*(j+1), 32-WIN*j) of the operand = coefficients x^(WIN*j) .. x^(WIN*j+WIN-1); bit t
       * of the window value is the coefficient of x^(WIN*j + WIN-1-t) */
      uint32_t basis[WIN];
#pragma unroll
      for (int t = WIN - 1; t >= 0; t--) {
        basis[t] = kx;
        kx = (kx & 1u) ? (kx >> 1) ^ CRC32C_POLY : kx >> 1; /* * x */
      }
      uint32_t w[NV];
      w[0] = 0u;
#pragma unroll
      for (int v = 1; v < NV; v++) {
        const int low = __builtin_ctz((unsigned)v);
        w[v] = w[v & (v - 1)] ^ basis[low];
      }
#pragma unroll
      for (int v = 0; v < NV; v++)
        nib[(j * NV + v) * 64 + lane] = w[v];
    }
  }
  __syncthreads();
  uint32_t v = slice[tid];
  for (int k = 1; k < 16; k++) {
    v = (v >> 8) ^ slice[v & 0xFFu];
    slice[k * 256 + tid] = v;
  }
}

/* ---- the fused frame CRC, shared by the wave-autonomous kernels (this file and render_rows.hpp).  L = the kernel's
 * LDS layout: o_slice / o_lanek / o_pow / o_xk (constant tables), o_crcacc ([0] accumulator, [1] deferred blocks,
 * [2] finished blocks), GPL / WIN / NWIN ------------------------------------------------------------------------------ */
/* Raw CRC (the register after the bytes, starting from 0) of the bytes a wave has staged at [p0, end_off) of `stage`;
 * [0, p0) reads as zero (leading zeros do not move a zero register).  Whole 16-byte groups: lane l folds GPL consecutive
 * groups Horner-style, the groups aligned to the END of the bytes so that absent ones are leading zeros; sreg * K_l
 * through the window tables, one xor reduction over the lanes; the < 16 tail bytes come in through the slicing rows. */
template <class L> __device__ inline uint32_t stream_crc_staged(const unsigned char *stage, uint32_t end_off, int lane) {
  const uint32_t *slice = lds_ptr<const uint32_t>(L::o_slice);
  const int m_full = (int)(end_off >> 4), tail = (int)(end_off & 15u);
  constexpr int GPL = L::GPL;
  const int shift = 64 * GPL - m_full;
  uint32_t sreg = 0;
#pragma unroll
  for (int k = 0; k < GPL; k++) {
    const int g = lane * GPL + k - shift;
    uint4 d = make_uint4(0u, 0u, 0u, 0u);
    if (g >= 0)
      d = *reinterpret_cast<const uint4 *>(stage + 16 * g);
    d.x ^= sreg; /* the register so far goes in with the next 16 bytes: slicing-by-16, no multiplication */
    sreg = crc_raw16(slice, d);
  }
  const uint32_t *nib = lds_ptr<const uint32_t>(L::o_lanek);
  uint32_t term = 0;
#pragma unroll
  for (int j = 0; j < L::NWIN; j++)
    term ^= nib[(j * (1 << L::WIN) + (int)((sreg >> (32 - L::WIN * (j + 1))) & ((1u << L::WIN) - 1u))) * 64 + lane];
  const uint32_t full = wave_read_lane(wave_xor_to_last(term), 63);
  /* the < 16 tail bytes: the register moves on by `tail` bytes; tail byte j is followed by tail-1-j bytes */
  uint32_t tb = 0;
  if (lane < tail)
    tb = slice[(tail - 1 - lane) * 256 + stage[16 * m_full + lane]];
  return crc_advance16(slice, full, tail) ^ wave_read_lane(wave_xor_to_last(tb), 63);
}
/* Place a block's raw CRC in the frame: * x^(8 * bytes behind it).  The frame's length is the last block's prefix, known
 * as soon as every wave has tokenised -- usually long before a wave gets here; a block that cannot be placed yet leaves
 * its raw value (in the words behind the look-back words) for the wave that finishes the frame. */
template <class L>
__device__ inline void stream_crc_place(uint32_t *slots, int nblk, int nblk_cap, int blk, uint32_t braw, uint32_t block_end,
                                        uint32_t cap_bytes, int lane) {
  const uint32_t *pw = lds_ptr<const uint32_t>(L::o_pow);
  uint32_t *crcacc = lds_ptr<uint32_t>(L::o_crcacc);
  const uint32_t xk = lds_ptr<const uint32_t>(L::o_xk)[lane];
  const uint32_t lastw = slot_load(&slots[nblk - 1]);
  if ((lastw >> 30) == 2u) {
    const uint32_t n_total = lastw & ACHIP_SLOT_VALUE;
    if (n_total <= cap_bytes) {
      const uint32_t placed = wave_mulmod_uniform(braw, wave_x8_pow_uniform(pw, n_total - block_end, lane, xk), lane, xk);
      if (lane == 0)
        slot_xor(&crcacc[0], placed);
    }
  } else if (lane == 0) {
    (slots + nblk_cap)[blk] = braw;
    (void)slot_fetch_add(&crcacc[1], 1u);
  }
}
/* Every block reports here once it is done; the last one completes the frame: crc(M) = ~(0xFFFFFFFF * x^(8|M|) xor
 * raw(M)), M = pad_top newlines || block 0 || block 1 || ...  (every prefix is in the look-back words by then), and --
 * when asked for -- the frame's 24-byte network-order header and the CRC of header || frame. */
template <class L>
__device__ inline void stream_crc_finish(uint32_t *slots, int nblk, int nblk_cap, uint32_t cap_bytes, uint32_t first_base,
                                         int fidx, uint32_t dim_w, uint32_t dim_h, const achip_wire_t &wire, int lane) {
  const uint32_t *slice = lds_ptr<const uint32_t>(L::o_slice);
  const uint32_t *pw = lds_ptr<const uint32_t>(L::o_pow);
  uint32_t *crcval = slots + nblk_cap;
  uint32_t *crcacc = lds_ptr<uint32_t>(L::o_crcacc);
  const uint32_t xk = lds_ptr<const uint32_t>(L::o_xk)[lane];
  uint32_t arrived = 0;
  if (lane == 0)
    arrived = slot_fetch_add(&crcacc[2], 1u);
  arrived = wave_read_lane(arrived, 0);
  if (arrived != (uint32_t)nblk - 1u)
    return;
  const uint32_t n_total = slot_load(&slots[nblk - 1]) & ACHIP_SLOT_VALUE;
  const bool fits = n_total <= cap_bytes;
  const uint32_t xn = fits ? wave_x8_pow_uniform(pw, n_total, lane, xk) : CRC_X0; /* x^(8 * frame length) */
  uint32_t fraw = 0; /* raw(M): the register after the frame starting from 0 */
  if (fits) {
    if (first_base > 0u) { /* ascii_pad_frame_height's newlines in front: lanes take runs of them */
      const uint32_t per = (first_base + 63u) / 64u;
      const uint32_t lo = (uint32_t)lane * per, hi = lo + per < first_base ? lo + per : first_base;
      uint32_t st = 0;
      for (uint32_t k = lo; k < hi; k++)
        st = (st >> 8) ^ slice[(st ^ (uint32_t)'\n') & 0xFFu];
      if (lo < hi)
        st = crc_mulmod(st, crc_x8_pow_len(n_total - hi));
      fraw ^= wave_read_lane(wave_xor_to_last(lo < hi ? st : 0u), 63);
    }
    if (slot_load(&crcacc[1]) != 0u) { /* blocks that finished before the frame's length was known */
      uint32_t acc = 0;
      for (int b0 = 0; b0 < nblk; b0 += 64) {
        const int b = b0 + lane;
        const uint32_t v = b < nblk ? crcval[b] : 0u;
        if (v != 0u)
          acc ^= crc_mulmod(v, crc_x8_pow_len(n_total - (slot_load(&slots[b]) & ACHIP_SLOT_VALUE)));
      }
      fraw ^= wave_read_lane(wave_xor_to_last(acc), 63);
    }
    fraw ^= slot_load(&crcacc[0]);
  }
  const uint32_t crc = fits ? ~(wave_mulmod_uniform(0xFFFFFFFFu, xn, lane, xk) ^ fraw) : 0u;
  if (lane == 0)
    wire.crc[fidx] = crc;
  if (wire.hdr || wire.pkt_crc) {
    /* ascii_frame_packet_t in network byte order (lib/network/acip/server.c:186-214): {width, height, original_size,
     * compressed_size = 0, checksum, flags = 0}; an unusable frame gets a header of zeros, as the stand-alone kernel
     * reports it.  Lane j < 24 owns header byte j. */
    const int wi = lane >> 2;
    const uint32_t word = !fits ? 0u : wi == 0 ? dim_w : wi == 1 ? dim_h : wi == 2 ? n_total : wi == 4 ? crc : 0u;
    const uint32_t hb = lane < 24 ? (word >> (8 * (3 - (lane & 3)))) & 0xFFu : 0u;
    if (wire.hdr && lane < 24)
      wire.hdr[(size_t)fidx * 24u + (size_t)lane] = (uint8_t)hb;
    if (wire.pkt_crc) {
      /* CRC of header || frame (packet_send_via_transport, send.c:59-69) = ~(S_h * x^(8n) xor raw(M)) with S_h the
       * register after the header from 0xFFFFFFFF: bytes 0..7 through slicing rows 7..0 and on by 16 bytes, bytes
       * 8..23 through rows 15..0 */
      const uint32_t v = lane < 8 ? slice[(7 - lane) * 256 + hb] : lane < 24 ? slice[(23 - lane) * 256 + hb] : 0u;
      const uint32_t ra = wave_read_lane(wave_xor_to_last(lane < 8 ? v : 0u), 63);
      const uint32_t rb = wave_read_lane(wave_xor_to_last(lane >= 8 ? v : 0u), 63);
      constexpr uint32_t INIT24 = crc_mulmod(0xFFFFFFFFu, crc_pow(CRC_X8, 24u));
      const uint32_t sh = INIT24 ^ crc_advance16(slice, ra, 16) ^ rb;
      const uint32_t pkt = ~(wave_mulmod_uniform(sh, xn, lane, xk) ^ fraw);
      if (lane == 0)
        wire.pkt_crc[fidx] = pkt;
    }
  }
}

template <int MODE, int WAVES, int CPL, bool GENERIC, bool CRC = false, int PACK = 0, bool PARTS = false, bool LF = false>
__global__ void __launch_bounds__(WAVES * 64)
    render_stream_kernel(const achip_frame_t *__restrict__ frames, const achip_lut_t *__restrict__ lut,
                         uint8_t *__restrict__ out, uint64_t out_stride, uint32_t *__restrict__ out_len, int n_frames,
                         achip_uniform_t uni, unsigned long long *__restrict__ prof, achip_wire_t wire,
                         const uint4 *__restrict__ crc_tab, achip_packdev_t pack, achip_partsdev_t ps) {
  /* PARTS (small launches: a lone frame, the nine targets of a grid): ONE workgroup per frame puts every wave of the
   * frame on one CU, whose four SIMDs then issue the sixteen waves' instructions one after the other -- ~3 of a lone
   * 80x24 frame's 5.4 us are that queue (profiles/r04_lone_frame_timeline.txt) while 255 CUs idle.  Here a frame's blocks
   * are shared out over ps.parts workgroups (the grid is n_frames * parts; workgroup f * parts + p takes the p-th run of
   * ceil(blocks / parts) blocks), the look-back inside a workgroup stays in LDS, and ONE hand-off crosses workgroups:
   * every workgroup publishes the bytes of its blocks (ps.sync[workgroup] = {epoch, bytes}, agent scope) as soon as its
   * blocks are counted, and the wave that owns a workgroup's first block adds up what the parts in front of it
   * published -- one round trip through memory behind the slowest predecessor, not a chain.  A launch dispatches its
   * workgroups in order and the host only asks for parts when all of them are resident at once, so the workgroups a
   * poller waits for are always running. */
  /* PACK != 0 (lib/network/acip/server.c:190-222 ships exactly frame_size bytes): frames leave
   * the kernel at their EXACT length, back to back in pack.dst, and the fixed-stride slab is never written.  A frame's
   * length is only known once its last block has been tokenised, so the whole frame is staged in LDS (frames up to
   * ACHIP_PACK_FRAME_CAP bytes: 1080p -> 80x24 truecolor is 36 KB), its place in pack.dst is claimed with ONE atomic add of
   * round16(length) on a launch-wide cursor -- no workgroup ever waits for another: the order of the frames in pack.dst is
   * the order in which they finish, pack.off_out[i] says where frame i went -- and the workgroup copies it out with
   * coalesced 16-byte stores.  PACK == 2: while those stores drain, the workgroup checksums the LDS image (wire.crc, and
   * the packet header / packet CRC when asked for) the way crc32c_frame_kernel checksums a slab slot.  The last workgroup
   * to finish leaves the total in pack.off_out[n] and clears the cursor words for the plan's next launch.  `out` /
   * out_stride only bound a frame's length here. */
  /* CRC = true: the frame's CRC-32C (asciichat_crc32, lib/network/crc32.c:95-190 -- what acip_send_ascii_frame puts
   * into ascii_frame_packet_t.checksum, lib/network/acip/server.c:186-214) rides the drain: every wave checksums its
   * block while the bytes are in its staging area, the last wave to finish combines the blocks (a CRC is linear over
   * GF(2): raw(A || B) = raw(A) * x^(8|B|) xor raw(B)) and writes wire.crc[frame] -- and, when asked for, the frame's
   * 24-byte network-order header and the CRC of header || frame (acip_send_ascii_frame).  crc_tab: the constant tables
   * (crc_tables_init_kernel's image, SLds::TAB_BYTES).  Both unused otherwise. */
  /* prof (diagnostics, NULL in production launches): 8 timestamps of the 100 MHz wall clock per wave, for the wave's
   * FIRST block -- prof[(frame*WAVES + wave)*8 + k]: 0 kernel entry, 1 prologue barrier passed, 2 samples requested,
   * 3 samples arrived, 4 tokens + scan done, 5 look-back done, 6 token bytes in LDS, 7 stores issued */
#define ACHIP_SSTAMP(slot)                                                                                             \
  do {                                                                                                                 \
    if (prof && lane == 0 && first_block)                                                                              \
      prof[((size_t)blockIdx.x * WAVES + wave) * 8u + (slot)] = wall_now();                                            \
  } while (0)
  static_assert(mode_is_cell(MODE), "run-structured modes use render_frames_kernel");
  using L = SLds<MODE, WAVES, CPL, CRC, PACK>;
  constexpr bool U8 = L::U8;                               /* truecolor foreground, palette with multi-byte glyphs */
  constexpr int RMODE = U8 ? ACHIP_MODE_TRUE_FG : MODE;    /* the renderer (token grammar) */
  static_assert(!U8 || (!GENERIC && !PARTS), "multi-byte palettes: whole frames of single sources (the host sends the rest to render_frames_kernel)");
  static_assert(!PACK || (!GENERIC && !CRC && MODE != ACHIP_MODE_TRUE_BG), "exact-length frames: single-source per-cell foreground modes");
  static_assert(!PARTS || (!CRC && !PACK), "a frame's checksum and its LDS image belong to one workgroup");
  constexpr bool WIRE = CRC || PACK == 2; /* the launch leaves checksums (and headers) */
  constexpr int BLOCK = WAVES * 64;
  constexpr int BLK = L::BLK;
  constexpr int SH = BLK - L::EFF; /* 1: slot (k = 0, lane 0) is the ghost of the cell in front of the block */
  constexpr int EFF = L::EFF;
  constexpr bool LEAN = L::WORDS && !GENERIC; /* the lean per-block loop below (truecolor foreground, single source) */

  uint32_t *slots = lds_ptr<uint32_t>(L::o_slots);

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int wg = (int)blockIdx.x;
  const int parts = PARTS ? ps.parts : 1;
  const int fidx = PARTS ? wg / parts : wg;
  const int part = PARTS ? wg - fidx * parts : 0;
  /* Every kernel argument the prologue needs is requested HERE, in one burst of scalar loads: left to itself the
   * compiler loads each argument at its first use, behind the branches above it -- six dependent round trips to
   * the kernarg segment in front of the first gather (profiles/r02_stream_timeline.txt). */
  ACHIP_DEVICE_ONLY(
      asm volatile("" ::"s"(n_frames), "s"(lut), "s"(out), "s"(out_stride), "s"(out_len), "s"(prof), "s"(frames),
                   "s"(uni.enabled), "s"(uni.flags), "s"(uni.src_pitch), "s"(uni.f.src), "s"(uni.f.comp));
      asm volatile("" ::"s"(uni.f.src_w), "s"(uni.f.src_h), "s"(uni.f.out_w), "s"(uni.f.out_h), "s"(uni.f.pad_left),
                   "s"(uni.f.pad_top), "s"(uni.f.x_ratio), "s"(uni.f.y_ratio), "s"(uni.f.src_stride), "s"(uni.f.ops));
      if (WIRE) asm volatile("" ::"s"(wire.crc), "s"(wire.dims), "s"(wire.hdr), "s"(wire.pkt_crc), "s"(crc_tab));
      if (PACK) asm volatile("" ::"s"(pack.dst), "s"(pack.capacity), "s"(pack.off_out), "s"(pack.len_out), "s"(pack.cursor));)
  if (fidx >= n_frames)
    return;
  /* PACK: every workgroup of the launch reports in exactly once (thread 0): where its frame went and how long it is (an
   * error code takes no room); the last one to report publishes the total and re-arms the cursor words */
  auto pack_report = [&](uint64_t off, uint32_t lenval) {
    if (pack.off_out)
      pack.off_out[fidx] = off;
    if (pack.len_out)
      pack.len_out[fidx] = lenval;
    const unsigned long long before = agent_fetch_add_u64(&pack.cursor[1], 1ull);
    if (before == (unsigned long long)n_frames - 1ull) {
      const unsigned long long total = agent_fetch_add_u64(&pack.cursor[0], 0ull);
      if (pack.off_out)
        pack.off_out[n_frames] = total;
      agent_store_u64(&pack.cursor[0], 0ull);
      agent_store_u64(&pack.cursor[1], 0ull);
    }
  };
  bool first_block = true;
  ACHIP_SSTAMP(0);
  /* CRC: the constant tables are requested before anything else (L2 hits after a process's first launch) and go to
   * LDS in front of the barrier below; nothing of them is computed here */
  /* (PACK == 2: the 22 KB image crc_frame_tables_init_kernel<256> wrote: slicing tables, the Horner table of the four
   * checksumming waves, the power tables) */
  constexpr bool TABLES = CRC || PACK == 2;
  constexpr int TABV = (PACK == 2 ? ACHIP_FRAME_CRC_TAB_BYTES : L::TAB_BYTES) / 16, TABN = (TABV + BLOCK - 1) / BLOCK;
  typedef uint32_t tab4_t __attribute__((vector_size(16))); /* a native vector: HIP's uint4 class keeps the array in scratch */
  tab4_t tabv[TABN > 0 ? TABN : 1];
  uint32_t dim_w = 0, dim_h = 0; /* header fields of this frame: every wave may be the one that finishes it */
  if (WIRE && wire.dims) {
    dim_w = wire.dims[2 * fidx];
    dim_h = wire.dims[2 * fidx + 1];
  }
  uint32_t lane_k = 0, lane_xk = 0; /* PACK == 2: this lane's constants of the final reduction (crc_reduce_waves) */
  if (PACK == 2) {
    lane_k = CRC_LANE_TAB.k[lane];
    lane_xk = CRC_LANE_TAB.xk[lane];
  }
  if (TABLES) {
#pragma unroll
    for (int k = 0; k < TABN; k++) /* clamped, not predicated: the values stay in registers */
      tabv[k] = reinterpret_cast<const tab4_t *>(crc_tab)[tid + k * BLOCK < TABV ? tid + k * BLOCK : TABV - 1];
  }

  /* glyph tables are requested first, the first block's samples right behind (the descriptor came with the kernel
   * arguments for uniform batches): one overlapped latency in front of the only barrier */
  constexpr int LUTN = (256 + BLOCK - 1) / BLOCK;
  uint32_t lut_g[LUTN];
#pragma unroll
  for (int k = 0; k < LUTN; k++)
    lut_g[k] = tid + k * BLOCK < 256 ? lut->glyph[tid + k * BLOCK] : 0u;
  const uint32_t lut_ramp = (RMODE == ACHIP_MODE_16_FG && tid < 64) ? lut->ramp[tid] : 0u;
  /* whether every glyph of the palette is one ASCII byte travels with the launch (the host knows the palette);
   * reading achip_lut_t.flags here would be one more dependent round trip */
  const bool ascii_only = (uni.flags & ACHIP_UNIFORM_PALETTE_ASCII) != 0u;
  achip_frame_t f = uni.f;
  if (uni.enabled)
    f.src = uni.f.src + (int64_t)fidx * uni.src_pitch;
  else
    f = frames[fidx];
  if (f.src_stride == 0)
    f.src_stride = 3 * f.src_w;
  uint8_t *dst = out + (size_t)fidx * out_stride;
  uint32_t dmis = (uint32_t)(uintptr_t)dst & (ACHIP_DRAIN_ALIGN - 1u) & ~15u; /* the slot's own offset inside a line */
  /* LF = LENGTH-FIRST (round 6; VERDICT r5 next 6, lib/network/acip/server.c:190-222 ships exactly frame_size bytes): exact-length
   * frames of ANY size in one launch -- PACK stages the whole frame in LDS and stops at 48 KB.  The lean loop runs twice: the first
   * pass samples, classifies and scans only, every block publishes its prefix; the frame claims its place in pack.dst with one
   * agent-scope add (completion order, as PACK); the second pass samples again -- out of the caches -- and emits at the claimed
   * place.  An instantiation of its own (four registers more than the plain one).  Measured (profiles/r06_length_first_ab.txt):
   * from sampled images 16.5 us against 34.5 for render + pack pass (256 frames of 200x60); from 4K sources the second gather
   * costs more than the pass (71.9 against 60.4): the plan takes it for dense sources only. */
  constexpr bool lenfirst = LF;
  static_assert(!LF || (L::WORDS && !GENERIC && !PARTS && !PACK && !CRC && !L::U8), "length-first: the plain lean truecolor-foreground form");

  const int wp = f.pad_left + f.out_w;
  const int rows = f.out_h;
  const long long cells_ll = (long long)rows * (long long)wp;
  if (f.out_w <= 0 || f.out_h <= 0 || f.src_w <= 0 || f.src_h <= 0 || f.pad_left < 0 || f.pad_top < 0 ||
      (!f.src && !f.comp) || (!GENERIC && (f.comp || f.src_w * f.src_h == 1)) ||
      cells_ll > (long long)stream_maxblk(uni.flags, EFF) * EFF || out_stride > (uint64_t)ACHIP_STREAM_MAX_STRIDE) {
    if (tid == 0) {
      out_len[fidx] = ACHIP_LEN_BADDESC;
      if (WIRE) {
        wire.crc[fidx] = 0u;
        if (wire.hdr) /* as the stand-alone kernel reports an unusable frame: a header of zeros, its CRC behind it */
          for (int j = 0; j < 24; j++)
            wire.hdr[(size_t)fidx * 24u + j] = 0;
        if (wire.pkt_crc)
          wire.pkt_crc[fidx] = ~crc_mulmod(0xFFFFFFFFu, crc_pow(CRC_X8, 24u));
      }
      if (PACK)
        pack_report(agent_fetch_add_u64(&pack.cursor[0], 0ull), ACHIP_LEN_BADDESC);
    }
    return;
  }
  const uint32_t ncells = (uint32_t)cells_ll;
  const int nblk = (int)((ncells + EFF - 1) / EFF);
  const int nblk_cap = stream_maxblk(uni.flags, EFF); /* words in each per-block LDS array of this launch */
  (void)nblk_cap;
  /* PARTS: this workgroup's run of blocks [b0, b1); a part behind the frame's last block only reports in */
  const int bpp = PARTS ? (nblk + parts - 1) / parts : nblk;
  const int b0 = PARTS ? min(part * bpp, nblk) : 0, b1 = PARTS ? min(b0 + bpp, nblk) : nblk;
  uint32_t *partacc = lds_ptr<uint32_t>(L::o_flags); /* PARTS: [0] bytes of this workgroup's blocks so far, [1] blocks counted */
  if (PARTS && b0 >= b1) {
    if (tid == 0)
      agent_store_u64(&ps.sync[wg], ((unsigned long long)ps.epoch << 32));
    return;
  }
  /* PACK: the frame must also fit its LDS image, which the launch sized for out_stride bytes (the host keeps that below
   * ACHIP_PACK_FRAME_CAP) behind the per-block words */
  const uint32_t cap_bytes = PACK ? min((uint32_t)out_stride, (uint32_t)ACHIP_PACK_FRAME_CAP) : (uint32_t)out_stride;
  const int frame_lds = PACK ? L::frame_off(nblk_cap) : 0;
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
  /* samples a cache line apart or more share no line with their neighbours: non-temporal loads keep them out of the
   * L2; closer samples do share lines and want the cache (profiles/r01_nontemporal.txt) */
  src.nt = f.x_ratio >= ((64u << 16) + 2u) / 3u;

  CompHead chead = {}; /* composite frames: filled behind the barrier, before their first request */
  /* cell -> (row, column) without a division per cell: one division per lane here, then constant steps */
  const uint32_t q64 = 64u / uwp, r64 = 64u - q64 * uwp;                                         /* k -> k+1 */
  const uint32_t qit = (uint32_t)(WAVES * EFF) / uwp, rit = (uint32_t)(WAVES * EFF) - qit * uwp; /* block -> block + WAVES */
  auto advance = [&](CellPos p, uint32_t q, uint32_t r) {
    p.xp += r;
    p.rr += q;
    const bool wrap = p.xp >= uwp;
    p.xp -= wrap ? uwp : 0u;
    p.rr += wrap ? 1u : 0u;
    return p;
  };
  /* truecolor-fg, left padding: the pad cell in front of a row's first pixel has nothing to sample for itself and
   * fetches what that pixel needs instead -- the last pixel of the row above, its raster predecessor */
  auto pred_pad = [&](CellPos p) { return SH != 0 && pad_left > 0u && p.xp == pad_left - 1u && p.rr > 0u; };
  /* request the samples of the block whose cell (k = 0, this lane) is cell0 at p0: nothing here consumes loaded data */
  auto issue = [&](auto nt_tag, uint32_t cell0, CellPos p0, uint32_t (&raw)[CPL], uint32_t &kinds) {
    constexpr bool NT = decltype(nt_tag)::value;
    kinds = 0;
    CellPos p = p0;
#pragma unroll
    for (int k = 0; k < CPL; k++) {
      raw[k] = 0;
      const bool pixc = p.xp >= pad_left;
      if (cell0 + 64u * k < ncells && (pixc || pred_pad(p))) {
        uint32_t kind = RAW_FINAL;
#if defined(ACHIP_STREAM_ABLATE) && ACHIP_STREAM_ABLATE == 2 /* diagnostics: no gather */
        raw[k] = ((p.xp * 2654435761u) ^ (p.rr * 40503u) ^ (uint32_t)fidx) & 0x00FFFFFFu;
#else
        raw[k] = stream_request<GENERIC, NT, L::o_comp>(f, src, pixc ? p.xp - pad_left : (uint32_t)f.out_w - 1u,
                                                        pixc ? p.rr : p.rr - 1u, kind, chead);
#endif
        kinds |= kind << (2 * k);
      }
      p = advance(p, q64, r64);
    }
  };
  auto issue_any = [&](uint32_t cell0, CellPos p0, uint32_t (&raw)[CPL], uint32_t &kinds) {
    if (!GENERIC && src.nt)
      issue(StreamTagNT{}, cell0, p0, raw, kinds);
    else
      issue(StreamTagCached{}, cell0, p0, raw, kinds);
  };

  /* ---- the lean loop (round 6; truecolor foreground with word-built SGRs from a single source: the render of every server
   * tick) walks a block LANE-major -- lane l owns the CPL consecutive cells c0 .. c0 + CPL - 1, slot (lane 0, j = 0) the
   * ghost -- so that a cell's raster predecessor is the lane's own previous slot (one DPP move per BLOCK for slot 0), a
   * block needs ONE wave scan, and positions step by one.  The sampler is multiplications by scalars only: the 16.16
   * ratio split into halves (24-bit multiplies run at full rate, v_mul_lo_u32 at a quarter), the flips folded into signed
   * steps; ratio 1.0 (a sampled image, csrc/frame_dense.c) skips it.  A sample is requested one byte early (`sh` = 8) except
   * the buffer's first pixel (`sh` = 0) and finished by ONE v_bfe_u32.
   * Sources whose samples lie a cache line apart or more (the non-temporal copy of the loop) keep the old order, slot j =
   * cell c0 + 64 j: there the lanes of ONE load instruction fetch neighbouring cells, and neighbours that share a line
   * (1080p -> 80 columns: 72 bytes apart) are one request; lane-major they are two requests of two instructions, and
   * the metric's launch, bound by requests, takes 9.97 instead of 6.89 us (profiles/r06_stream_lean_ab.txt, visit A). */
  const bool lean_dense = f.x_ratio == 65537u && f.y_ratio == 65537u; /* (x * 65537) >> 16 = x for x < 65536 */
  const uint32_t xr_lo = f.x_ratio & 0xFFFFu, xr_hi = f.x_ratio >> 16, yr_lo = f.y_ratio & 0xFFFFu, yr_hi = f.y_ratio >> 16;
  const int32_t lean_mx = src.flip_x ? -3 : 3;
  const uint32_t lean_ax = src.flip_x ? 3u * src.w1 : 0u;
  const uint32_t lean_my = src.flip_y ? ~0u : 0u, lean_ay = src.flip_y ? src.h1 + 1u : 0u; /* h1 - y = (y ^ ~0) + h1 + 1 */
  auto step1 = [&](CellPos p) {
    p.xp += 1u;
    const bool wrap = p.xp == uwp;
    p.xp = wrap ? 0u : p.xp;
    p.rr += wrap ? 1u : 0u;
    return p;
  };
  auto lean_issue = [&](auto nt_tag, uint32_t c0, CellPos p0, uint32_t (&raw)[CPL], uint32_t (&sh)[CPL]) {
    constexpr bool NT = decltype(nt_tag)::value, KM = NT; /* KM: slot j is cell c0 + 64 j (below) */
    CellPos p = p0;
#pragma unroll
    for (int j = 0; j < CPL; j++) {
      raw[j] = 0;
      uint32_t x = p.xp, y = p.rr;
      if (pad_left != 0u) { /* a pad cell fetches the last pixel of the row above: what the row's first pixel is compared with */
        const bool pixc = p.xp >= pad_left;
        x = pixc ? p.xp - pad_left : (uint32_t)f.out_w - 1u;
        y = pixc ? p.rr : p.rr - (p.rr != 0u ? 1u : 0u);
      }
      uint32_t sx = x, sy = y;
      if (!lean_dense) { /* (x * ratio) >> 16 = x * hi + ((x * lo) >> 16), exactly: x < 2^14, lo < 2^16, x * hi <= src_w */
        sx = min(mul_u24(x, xr_hi) + (mul_u24(x, xr_lo) >> 16), src.w1);
        sy = min(mul_u24(y, yr_hi) + (mul_u24(y, yr_lo) >> 16), src.h1);
      }
      const uint32_t syf = (sy ^ lean_my) + lean_ay;
      const uint32_t a = (uint32_t)mad_i24((int32_t)sx, lean_mx, (int32_t)(mul_u24(syf, src.stride) + lean_ax));
      const uint32_t back = a != 0u ? 1u : 0u;
      sh[j] = 8u * back;
      if (c0 + (uint32_t)(KM ? 64 * j : j) < ncells) {
#if defined(ACHIP_STREAM_ABLATE) && ACHIP_STREAM_ABLATE == 2 /* diagnostics: no gather */
        raw[j] = (((p.xp * 2654435761u) ^ (p.rr * 40503u) ^ (uint32_t)fidx) & 0x00FFFFFFu) << sh[j];
#else
        const ACHIP_GLOBAL uint8_t *q = (const ACHIP_GLOBAL uint8_t *)src.base + (a - back);
        if (NT)
          raw[j] = load_u32_unaligned_nt((const uint8_t *)q);
        else
          raw[j] = ((const ACHIP_GLOBAL unaligned_u32 *)q)->v;
#endif
      }
      p = KM ? advance(p, q64, r64) : step1(p);
    }
  };

  /* lane's cell of slot k = 0; the ghost of block 0 is "cell -1" = the last cell of row -1 (modulo 2^32: the steps
   * below carry it to the right place, and it fails every `< ncells` test) */
  uint32_t cell0 = (uint32_t)((b0 + wave) * EFF + (LEAN && !src.nt ? lane * CPL : lane)) - (uint32_t)SH;
  CellPos pos;
  pos.rr = cell0 / uwp;
  pos.xp = cell0 - pos.rr * uwp;
  if (SH && cell0 == 0xFFFFFFFFu) {
    pos.rr = 0xFFFFFFFFu;
    pos.xp = uwp - 1u;
  }
  uint32_t raw[CPL], kinds = 0;
  uint32_t raw_sh[CPL] = {}; /* LEAN: bit offset of the pixel inside its raw dword */
  /* a composite frame samples through the LDS copy of its descriptor: its first requests follow the barrier */
  const bool late_first = GENERIC && f.comp != nullptr;
  if (late_first)
    comp_stage<L::o_comp, BLOCK>(f.comp, tid);
  else if (b0 + wave < b1) {
    if constexpr (LEAN) {
      if (src.nt)
        lean_issue(StreamTagNT{}, cell0, pos, raw, raw_sh);
      else
        lean_issue(StreamTagCached{}, cell0, pos, raw, raw_sh);
    } else {
      issue_any(cell0, pos, raw, kinds);
    }
  }
  ACHIP_SSTAMP(2);

  if (RMODE == ACHIP_MODE_TRUE_FG && ascii_only == U8) { /* (the launchers pick the instantiation by the palette) */
    if (tid == 0) {
      out_len[fidx] = ACHIP_LEN_BADDESC;
      if (PACK)
        pack_report(agent_fetch_add_u64(&pack.cursor[0], 0ull), ACHIP_LEN_BADDESC);
    }
    return;
  }
  /* tables -> LDS; look-back words of this frame cleared */
  uint32_t *glyph = lds_ptr<uint32_t>(L::o_glyph);
  uint8_t *ramp = lds_ptr<uint8_t>(L::o_ramp);
#pragma unroll
  for (int k = 0; k < LUTN; k++)
    if (tid + k * BLOCK < 256) {
      glyph[tid + k * BLOCK] = lut_g[k];
      lds_ptr<uint32_t>(L::o_dec)[tid + k * BLOCK] = dec_table_entry((uint32_t)(tid + k * BLOCK));
      if (L::WORDS) {
        uint2 wr, wg, wm, wmg;
        word_table_entries((uint32_t)(tid + k * BLOCK), wr, wg, wm, wmg);
        lds_ptr<uint2>(L::o_wr)[tid + k * BLOCK] = wr;
        lds_ptr<uint2>(L::o_wg)[tid + k * BLOCK] = wg;
        lds_ptr<uint2>(L::o_wm)[tid + k * BLOCK] = wm;
        { /* U8: the glyph's length, bit 7 = "single ASCII byte" (foreground.c:282) */
          const uint32_t gn = glyph_len(lut_g[k]);
          lds_ptr<uint8_t>(L::o_g8)[tid + k * BLOCK] = U8 ? (uint8_t)(gn | (gn == 1u && (lut_g[k] & 0xFFu) < 128u ? 0x80u : 0u)) : (uint8_t)lut_g[k];
        }
      }
    }
  if (RMODE == ACHIP_MODE_16_FG && tid < 64)
    ramp[tid] = (uint8_t)lut_ramp;
  for (int k = tid; k < b1 - b0; k += BLOCK) /* (indexed from the workgroup's first block) */
    slots[k] = 0u;
  if (U8)
    for (int k = tid; k < b1 - b0; k += BLOCK)
      slots[nblk_cap + k] = 0u; /* the RLE-state words */
  if (PARTS && tid < 2)
    partacc[tid] = 0u;
  (void)ramp;
  if (TABLES) {
#pragma unroll
    for (int k = 0; k < TABN; k++)
      if (tid + k * BLOCK < TABV)
        lds_ptr<tab4_t>(PACK == 2 ? L::o_fcrc : L::o_tab)[tid + k * BLOCK] = tabv[k];
  }
  if (CRC) {
    for (int k = tid; k < nblk; k += BLOCK)
      slots[nblk_cap + k] = 0u; /* raw CRCs of blocks that could not be placed */
    if (tid < 4)
      lds_ptr<uint32_t>(L::o_crcacc)[tid] = 0u;
  }
  /* ascii_pad_frame_height (ascii.c:902-941): pad_top bare newlines in front of the frame */
  const uint32_t first_base = (uint32_t)f.pad_top;
  if (first_base > 0u && first_base <= cap_bytes && part == 0 && !lenfirst)
    for (uint32_t o = (uint32_t)tid; o < first_base; o += BLOCK) {
      if (PACK)
        lds_ptr<uint8_t>(frame_lds)[o] = '\n';
      else
        dst[o] = '\n';
    }
  __syncthreads(); /* the only workgroup barrier (PACK: the first of three) */
  ACHIP_SSTAMP(1);
  if (late_first) {
    chead = comp_head<L::o_comp>();
    if (b0 + wave < b1)
      issue_any(cell0, pos, raw, kinds);
  }

  /* PACK: every wave writes into the frame's own image (stream offset o at its byte o) */
  const uint32_t stage_off = (uint32_t)(PACK ? frame_lds : L::o_stage + wave * L::STAGE);
  const uint32_t stage_addr = lds_base_addr() + stage_off;
  /* predicated-off byte stores land in a per-lane dummy word (one address for all lanes would serialise them) */
  const uint32_t dummy_addr = lds_base_addr() + (uint32_t)L::o_flags + 16u + 4u * (uint32_t)lane;
  if (L::WORDS) { /* every wave clears its own area, behind the barrier: its samples are on their way meanwhile */
    for (int g = lane; g < L::STAGE / 16; g += 64)
      lds_ptr<uint4>((int)stage_off)[g] = make_uint4(0u, 0u, 0u, 0u);
    wave_lockstep();
  }

  /* ---- where a block of `total` bytes starts in the frame (look-back inside the workgroup, the hand-off between the parts
   * of a shared-out frame), published for the blocks behind it.  Words hold absolute stream offsets (pad_top included). */
  auto place_block = [&](int blk, uint32_t total, bool &ok) -> uint32_t {
    uint32_t base = first_base;
    const int lb = blk - b0; /* the block's look-back word */
    if (PARTS) {
      /* the workgroup's bytes, for the parts behind it: whoever counts the workgroup's last block publishes the sum
       * (its own add to [0] precedes its add to [1] in the LDS queue, like everybody's: the sum is complete) */
      uint32_t counted = 0;
      if (lane == 0) {
        (void)slot_fetch_add(&partacc[0], total);
        counted = slot_fetch_add(&partacc[1], 1u);
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
      /* (sums of frames that overflow their slots: every part's bytes are < 2^30, their sum is clamped below) */
      base = lost ? 0xFFFFFFFFu : min(first_base + sum, cap_bytes + 1u);
    }
    ok = base != 0xFFFFFFFFu && (uint64_t)base + total <= cap_bytes;
    /* a frame that overflows its slot (or whose look-back failed) publishes cap+1 from there on, so every later
     * block fails the same test and prefixes stay below 2^30 (stride <= ACHIP_STREAM_MAX_STRIDE) */
    if (lane == 0)
      slot_store(&slots[lb], ACHIP_SLOT_PREFIX | (ok ? base + total : cap_bytes + 1u));
    return base;
  };
  /* ---- a block's bytes, staging -> HBM (g0 = the stream offset that sits at the staging area's byte 0) */
  auto drain_block = [&](uint32_t base, uint32_t total, uint32_t g0) {
    /* ---- staging -> HBM: whole 16-byte groups as uint4, the shared first / last group as bytes (PACK: the frame
     * leaves LDS as a whole, behind the loop) */
    const unsigned char *stage = lds_ptr<const unsigned char>((int)stage_off);
    const uint32_t end = base + total;
    const uint32_t vec_begin = PACK ? end : (base + 15u) & ~15u, vec_end = end & ~15u;
    /* every store instruction of the wave covers whole 128-byte lines (lane l takes the group 16 l bytes behind a LINE
     * boundary of the ADDRESS, not behind vec_begin): a 1 KB wave store that straddles lines leaves two of them half written, and
     * the memory side takes such writes at 4.4 instead of 5.7 TB/s (profiles/r04_rows_floor.txt) */
    for (uint32_t q = ((vec_begin + dmis) & ~(ACHIP_DRAIN_ALIGN - 1u)) + 16u * (uint32_t)lane; q < vec_end + dmis; q += 1024u) {
      if (q < vec_begin + dmis)
        continue;
      const uint32_t o = q - dmis;
#if defined(ACHIP_STREAM_ABLATE) && (ACHIP_STREAM_ABLATE == 1 || ACHIP_STREAM_ABLATE == 3) /* diagnostics: no HBM writes */
      const uint4 v = *reinterpret_cast<const uint4 *>(stage + (o - g0));
      asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
#else
      store_out16(dst + o, *reinterpret_cast<const uint4 *>(stage + (o - g0)));
#endif
    }
    if (!PACK) {
      const uint32_t head_end = min(vec_begin, end);
      if (base + (uint32_t)lane < head_end)
        dst[base + (uint32_t)lane] = stage[base + (uint32_t)lane - g0];
      const uint32_t tail_begin = max(vec_end, head_end);
      if (lane >= 32 && tail_begin + (uint32_t)(lane - 32) < end)
        dst[tail_begin + (uint32_t)(lane - 32)] = stage[tail_begin + (uint32_t)(lane - 32) - g0];
    }
    if (L::WORDS) { /* the next block's SGRs are OR-ed into zeros (a wave's DS operations complete in order) */
      wave_lockstep();
      for (uint32_t g = (uint32_t)lane; g < ((base & 15u) + total + 15u) >> 4; g += 64u)
        lds_ptr<uint4>((int)stage_off)[g] = make_uint4(0u, 0u, 0u, 0u);
      wave_lockstep();
    }
  };

  if constexpr (LEAN) {
    /* Two copies of the loop, chosen once per frame: a shared loop would merge the cached and the non-temporal load of a
     * sample, and the copies of a request meeting in phi moves is a wait for memory right behind the requests
     * (docs/history/round5.md 2b). */
    const CellPos pos_first = pos;
    auto lean_loop = [&](auto nt_tag, auto phase_tag) {
      constexpr bool KM = decltype(nt_tag)::value; /* slot-major cell order (see lean_issue) */
      constexpr int PH = decltype(phase_tag)::value; /* 0: the render; LF: 1 = lengths only, 2 = emit at known prefixes */
      uint32_t c0 = cell0;
      if (PH == 2) { /* the second pass asks for the wave's first block again */
        pos = pos_first;
        if (b0 + wave < b1)
          lean_issue(nt_tag, cell0, pos, raw, raw_sh);
      }
      for (int blk = b0 + wave; blk < b1; blk += WAVES) {
        if (prof) { /* diagnostics only: make "samples arrived" a point in time */
          wait_vmem_all();
          ACHIP_SSTAMP(3);
        }
        /* ---- this block's samples become pixels (0x00BBGGRR); the next block's are requested into the same registers
         * and stay in flight while this block is tokenised and drained */
        uint32_t px[CPL];
#pragma unroll
        for (int j = 0; j < CPL; j++) {
          px[j] = bfe_u32(raw[j], raw_sh[j], 24u);
          if (f.ops & ACHIP_OP_TINT)
            px[j] = tint_pixel(px[j], f.ops);
        }
        const uint32_t c0_next = c0 + (uint32_t)(WAVES * EFF);
        const CellPos pos_next = advance(pos, qit, rit);
        if (blk + WAVES < b1)
          lean_issue(nt_tag, c0_next, pos_next, raw, raw_sh);

        /* ---- image_print_color + ansi_rle_add_pixel (foreground.c:268-303, ansi.c:261-300) per cell: the SGR only when the
         * colour differs from the raster predecessor's (the state survives row ends); its body "R;G;Bm" is put together
         * HERE, from the word tables, and its length falls out of the tables' terms -- the length pass and the store pass
         * share one set of table reads */
        SgrBody body[CPL];
        uint32_t len[CPL], sl[CPL], gl[CPL];
        uint32_t gn[CPL]; /* U8: bytes of the glyph in gl[] */
        bool has_nl[CPL], is_fin[CPL];
        if constexpr (!U8)
        {
          CellPos p = pos;
          uint32_t prev = KM ? 0u : wave_shift_up1(px[CPL - 1], 0u);
#pragma unroll
          for (int j = 0; j < CPL; j++) {
            const uint32_t c = c0 + (uint32_t)(KM ? 64 * j : j);
            if (KM) /* the neighbouring lane's slot j; lane 0: the slot below's last lane (the ghost for lane 1 of slot 0) */
              prev = wave_shift_up1(px[j], j > 0 ? wave_read_lane(px[j > 0 ? j - 1 : 0], 63) : 0u);
            const bool valid = c < ncells && !(j == 0 && lane == 0); /* the ghost owns no token */
            const bool pix = valid && p.xp >= pad_left;
            const bool row_end = valid && p.xp == uwp - 1u; /* always a pixel cell: out_w >= 1 */
            const bool fin = valid && c == ncells - 1u;     /* ansi_rle_finish: the single trailing ESC[0m */
            const bool sgr = c == pad_left || prev != px[j]; /* (the frame's first pixel: ansi_rle_init's first_pixel) */
            const uint32_t Y = dot4_u8(px[j], 0x001D964Du, 128u) >> 8; /* (77 R + 150 G + 29 B + 128) >> 8 */
            /* rainbow_replace_ansi_colors (color_filter.c:348-408) folded in: the colour of every SGR is the given one */
            const uint32_t colour = (f.ops & ACHIP_OP_FG_OVERRIDE) ? f.ops >> ACHIP_OP_TINT_SHIFT : px[j];
            body[j] = sgr_body(word_fields<L::o_wr, L::o_wg, L::o_wm>(colour));
            /* (every lane reads its glyph and takes a select: a cell that is no pixel costs no branch of the wave) */
            const uint32_t gy = lds_ptr<const uint8_t>(L::o_g8)[Y];
            /* ascii_pad_frame_width: a pad cell is one space; cells behind the frame own nothing */
            gl[j] = pix ? gy : (uint32_t)' ';
            sl[j] = pix && sgr ? body[j].bits >> 3 : 0u;
            has_nl[j] = row_end && !fin;
            is_fin[j] = fin;
            /* SGR + glyph + newline (+ the final reset): the predicates go in as carries */
            len[j] = sl[j] + (valid ? 1u : 0u) + (row_end ? 1u : 0u) + (fin ? 3u : 0u);
            prev = px[j];
            p = KM ? advance(p, q64, r64) : step1(p);
          }
        }
        else {
          /* ---- multi-byte palettes (foreground.c:281-296).  A cell whose glyph is not a single ASCII byte always carries
           * its SGR and leaves the RLE state alone; an ASCII cell carries one iff the frame holds no ASCII cell in front of
           * it or the nearest one has another colour.  Inside a block that cell is found with ballots (nearest_lower); in
           * front of it, it is the second look-back's word -- the chain depends on pixels only, never on byte counts, so
           * every block publishes its own word right behind its samples. */
          bool valid[CPL], pix[CPL], asc[CPL], rend[CPL];
          {
            CellPos p = pos;
#pragma unroll
            for (int j = 0; j < CPL; j++) {
              const uint32_t c = c0 + (uint32_t)(KM ? 64 * j : j);
              valid[j] = c < ncells && !(j == 0 && lane == 0); /* (the ghost slot is kept: same blocks as the ASCII form) */
              pix[j] = valid[j] && p.xp >= pad_left;
              rend[j] = valid[j] && p.xp == uwp - 1u;
              is_fin[j] = valid[j] && c == ncells - 1u;
              has_nl[j] = rend[j] && !is_fin[j];
              const uint32_t Y = dot4_u8(px[j], 0x001D964Du, 128u) >> 8;
              const uint32_t info = lds_ptr<const uint8_t>(L::o_g8)[Y];
              const uint32_t gw = glyph[Y];
              asc[j] = pix[j] && (info & 0x80u) != 0u;
              gl[j] = pix[j] ? gw : (uint32_t)' ';
              gn[j] = pix[j] ? info & 7u : 1u;
              p = KM ? advance(p, q64, r64) : step1(p);
            }
          }
          const uint32_t lt_lo = lane < 32 ? (1u << lane) - 1u : 0xFFFFFFFFu, lt_hi = lane < 32 ? 0u : (1u << (lane - 32)) - 1u;
          /* the nearest earlier ASCII cell INSIDE the block, per slot: {found, rgb}; and the block's own last one */
          bool ph[CPL];
          uint32_t pv[CPL];
          bool blk_has = false;
          uint32_t blk_last = 0;
          if (KM) {
#pragma unroll
            for (int j = 0; j < CPL; j++) { /* cell order = slot after slot */
              const NearestLower nl = nearest_lower(asc[j], px[j], lt_lo, lt_hi);
              ph[j] = nl.found || blk_has;
              pv[j] = nl.found ? nl.val : blk_last;
              const uint64_t m = wave_ballot(asc[j]);
              if (m != 0ull) {
                blk_has = true;
                blk_last = highest_flagged(m, px[j]);
              }
            }
          } else {
            bool have = false; /* cell order = lane after lane: first the lane's own earlier slots ... */
            uint32_t last = 0;
#pragma unroll
            for (int j = 0; j < CPL; j++) {
              ph[j] = have;
              pv[j] = last;
              have = have || asc[j];
              last = asc[j] ? px[j] : last;
            }
            const NearestLower nl = nearest_lower(have, last, lt_lo, lt_hi); /* ... then the lanes below */
#pragma unroll
            for (int j = 0; j < CPL; j++) {
              pv[j] = ph[j] ? pv[j] : nl.val;
              ph[j] = ph[j] || nl.found;
            }
            const uint64_t m = wave_ballot(have);
            if (m != 0ull) {
              blk_has = true;
              blk_last = highest_flagged(m, last);
            }
          }
          /* the state in front of the block */
          uint32_t *rle = slots + nblk_cap;
          const int lbk = blk - b0;
          uint32_t in = 0u; /* {has, rgb} */
          if (lbk > 0) {
            if (lane == 0)
              slot_store(&rle[lbk], ACHIP_SLOT_AGG | (blk_has ? ACHIP_RLE_HAS | blk_last : 0u));
            in = stream_lookback_rle(rle, lbk, lane);
          }
          const bool rle_lost = in == 0xFFFFFFFFu; /* (bounded polling ran out: the frame is reported as not fitting) */
          if (lane == 0)
            slot_store(&rle[lbk], ACHIP_SLOT_PREFIX | (blk_has ? ACHIP_RLE_HAS | blk_last : (rle_lost ? 0u : in)));
          const bool in_has = !rle_lost && (in & ACHIP_RLE_HAS) != 0u;
          const uint32_t in_rgb = in & 0x00FFFFFFu;
#pragma unroll
          for (int j = 0; j < CPL; j++) {
            const bool prev_has = ph[j] || in_has;
            const uint32_t prev_rgb = ph[j] ? pv[j] : in_rgb;
            const bool sgr = pix[j] && (!asc[j] || !prev_has || prev_rgb != px[j]);
            const uint32_t colour = (f.ops & ACHIP_OP_FG_OVERRIDE) ? f.ops >> ACHIP_OP_TINT_SHIFT : px[j];
            body[j] = sgr_body(word_fields<L::o_wr, L::o_wg, L::o_wm>(colour));
            sl[j] = sgr ? body[j].bits >> 3 : 0u;
            len[j] = rle_lost ? 0x100000u /* overflows every slot */ : sl[j] + (valid[j] ? gn[j] : 0u) + (rend[j] ? 1u : 0u) + (is_fin[j] ? 3u : 0u);
          }
        }

        /* ---- lane-major: ONE wave scan of the lanes' sums; slot-major: one per slot */
        uint32_t off[CPL], total = 0;
        if (KM) {
#pragma unroll
          for (int j = 0; j < CPL; j++) {
            const uint32_t incl = wave_inclusive_scan(len[j]);
            off[j] = total + incl - len[j];
            total += wave_read_lane(incl, 63);
          }
        } else {
          uint32_t mine = 0;
#pragma unroll
          for (int j = 0; j < CPL; j++) {
            off[j] = mine;
            mine += len[j];
          }
          const uint32_t incl = wave_inclusive_scan(mine);
          total = wave_read_lane(incl, 63);
#pragma unroll
          for (int j = 0; j < CPL; j++)
            off[j] += incl - mine;
        }

        ACHIP_SSTAMP(4);
        bool ok;
        uint32_t base;
        if (PH == 2) { /* every prefix was published by the first pass (a barrier lies between) */
          const uint32_t incl = slot_load(&slots[blk - b0]) & ACHIP_SLOT_VALUE;
          ok = incl <= cap_bytes;
          base = incl - total;
        } else {
          base = place_block(blk, total, ok);
        }

        ACHIP_SSTAMP(5);
        if (ok && PH != 1) {
          /* ---- token bytes into this wave's staging area: stream offset g0 (16-byte aligned) sits at its byte 0 */
          const uint32_t g0 = base & ~15u;
#pragma unroll
          for (int j = 0; j < CPL; j++) {
            const uint32_t a = stage_addr + (base - g0) + off[j];
#if defined(ACHIP_STREAM_ABLATE) && ACHIP_STREAM_ABLATE == 3 /* diagnostics: no token stores either */
            asm volatile("" ::"v"(body[j].x0), "v"(body[j].x1), "v"(body[j].x2), "v"(gl[j]), "v"(a));
#else
            if (sl[j] != 0u)
              sgr_place(a, 0x38335B1Bu, body[j]);
            if constexpr (!U8) {
              /* the glyph (a pad cell's space): cells that own nothing store into the lane's dummy word instead of branching */
              const uint32_t g = len[j] != 0u ? a + sl[j] : dummy_addr;
              lds_store_byte<0, false>(g, gl[j]);
              if (has_nl[j])
                lds_store_byte<1, false>(g, (uint32_t)'\n');
              if (is_fin[j]) {
                lds_store_byte<1, false>(g, 0x1Bu);
                lds_store_byte<2, false>(g, (uint32_t)'[');
                lds_store_byte<3, false>(g, (uint32_t)'0');
                lds_store_byte<4, false>(g, (uint32_t)'m');
              }
            } else if (len[j] != 0u) {
              /* the glyph's one to four bytes: two aligned dword ORs (the staging area is zero where nothing was written) */
              const uint32_t g = a + sl[j];
              const uint32_t gv = gn[j] >= 4u ? gl[j] : gl[j] & ((1u << (8u * gn[j])) - 1u);
              const uint32_t t = g - 1u, wbase = t & ~3u, wsh = (t << 3) ^ 24u;
              ds_or_u32_at<0>(wbase, alignbit(gv, 0u, wsh));
              ds_or_u32_at<4>(wbase, alignbit(0u, gv, wsh));
              const uint32_t e = g + gn[j];
              if (has_nl[j])
                lds_store_byte<0, false>(e, (uint32_t)'\n');
              if (is_fin[j]) {
                lds_store_byte<0, false>(e, 0x1Bu);
                lds_store_byte<1, false>(e, (uint32_t)'[');
                lds_store_byte<2, false>(e, (uint32_t)'0');
                lds_store_byte<3, false>(e, (uint32_t)'m');
              }
            }
#endif
          }
          lds_store_fence(); /* DS operations of one wave complete in order: the reads below see every lane's bytes */
          ACHIP_SSTAMP(6);
          drain_block(base, total, g0);
        }
        ACHIP_SSTAMP(7);
        first_block = false;
        if (PH == 0 && blk == nblk - 1 && lane == 0) {
          out_len[fidx] = ok ? base + total : ACHIP_LEN_OVERFLOW;
          if (ok && (uint64_t)base + total < out_stride)
            dst[base + total] = 0; /* NUL behind the frame when the slot has room, as the reference's strings carry */
        }
        c0 = c0_next;
        pos = pos_next;
      }
    };
#ifdef ACHIP_STREAM_COUNT_LM /* instruction-count builds (scripts/isa_lines.py --loop): the lane-major copy alone */
    lean_loop(StreamTagCached{}, StreamPhase<0>{});
#else
    if constexpr (LF) {
      if (src.nt)
        lean_loop(StreamTagNT{}, StreamPhase<1>{});
      else
        lean_loop(StreamTagCached{}, StreamPhase<1>{});
      __syncthreads(); /* every block's prefix is in its look-back word */
      const uint32_t n_total = nblk > 0 ? slot_load(&slots[nblk - 1]) & ACHIP_SLOT_VALUE : first_base;
      const bool fits = n_total <= cap_bytes;
      const uint32_t room = fits ? (n_total + 15u) & ~15u : 0u;
      unsigned long long *offw = lds_ptr<unsigned long long>(L::o_flags);
      if (tid == 0)
        offw[0] = agent_fetch_add_u64(&pack.cursor[0], (unsigned long long)room);
      __syncthreads();
      const unsigned long long off = offw[0];
      const bool placed = fits && off + room <= pack.capacity;
      if (placed) {
        dst = pack.dst + off;
        dmis = (uint32_t)(uintptr_t)dst & (ACHIP_DRAIN_ALIGN - 1u) & ~15u;
        if (first_base > 0u) /* ascii_pad_frame_height's newlines in front */
          for (uint32_t o = (uint32_t)tid; o < first_base; o += BLOCK)
            dst[o] = '\n';
        if (tid >= 64 && tid < 80 && n_total + (uint32_t)(tid - 64) < room) /* the <= 15 bytes of padding leave as zeros */
          dst[n_total + (uint32_t)(tid - 64)] = 0;
        if (src.nt)
          lean_loop(StreamTagNT{}, StreamPhase<2>{});
        else
          lean_loop(StreamTagCached{}, StreamPhase<2>{});
      }
      if (tid == 0) {
        out_len[fidx] = placed ? n_total : ACHIP_LEN_OVERFLOW;
        pack_report(off, placed ? n_total : ACHIP_LEN_OVERFLOW);
      }
      return;
    }
    if (src.nt)
      lean_loop(StreamTagNT{}, StreamPhase<0>{});
    else
      lean_loop(StreamTagCached{}, StreamPhase<0>{});
#endif
  } else {
    for (int blk = b0 + wave; blk < b1; blk += WAVES) {
      /* ---- request the next block's samples: they stay in flight while this block is tokenised and drained */
      const uint32_t cell0_next = cell0 + (uint32_t)(WAVES * EFF);
      const CellPos pos_next = advance(pos, qit, rit);
      uint32_t raw_n[CPL], kinds_n = 0;
      if (blk + WAVES < b1)
        issue_any(cell0_next, pos_next, raw_n, kinds_n);

      if (prof) { /* diagnostics only: make "samples arrived" a point in time */
        wait_vmem_all();
        ACHIP_SSTAMP(3);
      }
      /* ---- tokens and lengths (registers).  Written select-style: per-lane conditions become v_cndmask, not
       * exec-mask branches -- the tail of a launch is ONE wave's instruction stream, every branch is latency. */
      Tok tok[CPL];
      uint32_t len[CPL], px[CPL];
      bool is_pix[CPL], is_valid[CPL];
      {
        CellPos p = pos;
#pragma unroll
        for (int k = 0; k < CPL; k++) {
          const bool inb = cell0 + 64u * k < ncells;
          is_valid[k] = inb && !(SH && k == 0 && lane == 0); /* the ghost owns no token */
          is_pix[k] = is_valid[k] && p.xp >= pad_left;
          const uint32_t v = sample_finish<GENERIC>(f, raw[k], (kinds >> (2 * k)) & 3u);
          px[k] = inb && (p.xp >= pad_left || pred_pad(p)) ? v : 0u;
          p = advance(p, q64, r64);
        }
      }
      {
        CellPos p = pos;
#pragma unroll
        for (int k = 0; k < CPL; k++) {
          const bool valid = is_valid[k], pix = is_pix[k];
          const uint32_t pt = px[k];
          const uint32_t Y = luma601(pt);
          Tok t;
          t.rep = 0;
          t.bg = 0;
          uint32_t flags, n;
          const bool row_end = valid && p.xp == uwp - 1u; /* always a pixel cell: out_w >= 1 */
          const bool nl = row_end && p.rr < (uint32_t)rows - 1u;
          if (RMODE == ACHIP_MODE_TRUE_FG) {
            /* image_print_color + ansi_rle_add_pixel (foreground.c:268-303, ansi.c:261-300): the SGR only when the
             * colour differs from the previous pixel in raster order (the state survives row ends) */
            /* the raster predecessor sits one slot down: the neighbouring lane (the ghost for lane 1 of slot 0; with left
             * padding the pad cell in front of the row, which fetched the row above's last pixel) */
            const uint32_t prev = wave_shift_up1(pt, k > 0 ? wave_read_lane(px[k > 0 ? k - 1 : 0], 63) : 0u);
            const bool have_prev = !(p.xp == pad_left && p.rr == 0u);
            const bool sgr = !have_prev || px_rgb(prev) != px_rgb(pt);
            t.glyph = glyph[Y];
            t.fg = px_rgb(pt);
            const bool fin = row_end && !nl; /* ansi_rle_finish: the single trailing ESC[0m */
            flags = TF_GLYPH | (sgr ? TF_SGR_FG : 0u) | (nl ? TF_NL : 0u) | (fin ? TF_FINAL_RESET : 0u);
            if (f.ops & ACHIP_OP_FG_OVERRIDE) /* rainbow_replace_ansi_colors (color_filter.c:348-408) folded in */
              t.fg = f.ops >> ACHIP_OP_TINT_SHIFT;
            n = (sgr ? 10u + dec_digits(px_r(t.fg)) + dec_digits(px_g(t.fg)) + dec_digits(px_b(t.fg)) : 0u) + 1u +
                (nl ? 1u : 0u) + (fin ? 4u : 0u);
          } else if (RMODE == ACHIP_MODE_256_FG) { /* foreground.c:475-500 */
            t.fg = quant256(pt);
            t.glyph = glyph[Y];
            flags = TF_SGR_FG | TF_GLYPH | (row_end ? TF_ROW_RESET : 0u) | (nl ? TF_NL : 0u);
            n = 8u + dec_digits(t.fg) + (ascii_only ? 1u : glyph_len(t.glyph)) + (row_end ? 4u : 0u) + (nl ? 1u : 0u);
          } else if (RMODE == ACHIP_MODE_16_FG) { /* foreground.c:584-612: glyph = cache[ramp[Y>>2]] (sic) */
            t.fg = sgr16_code(false, quant16(pt));
            t.glyph = glyph[ramp[Y >> 2]];
            flags = TF_SGR_FG | TF_GLYPH | (row_end ? TF_ROW_RESET : 0u) | (nl ? TF_NL : 0u);
            n = 5u + (ascii_only ? 1u : glyph_len(t.glyph)) + (row_end ? 4u : 0u) + (nl ? 1u : 0u);
          } else { /* background.c:49-68 */
            t.bg = px_rgb(pt);
            t.fg = 0;
            t.glyph = glyph[Y];
            flags = TF_SGR_BG | TF_SGR_FG | TF_GLYPH | (Y < 128u ? TF_FG_WHITE : 0u) | (row_end ? TF_ROW_RESET : 0u) |
                    (nl ? TF_NL : 0u);
            if (f.ops & ACHIP_OP_FG_OVERRIDE) {
              t.fg = f.ops >> ACHIP_OP_TINT_SHIFT;
              flags |= TF_FG_GIVEN;
            }
            t.flags = flags;
            CountSink cs{0u};
            token_fields<RMODE>(cs, t, ascii_only);
            n = cs.n;
          }
          /* ascii_pad_frame_width: a pad cell is one space; cells behind the frame own nothing */
          t.flags = pix ? flags : (valid ? (uint32_t)TF_PAD : 0u);
          len[k] = pix ? n : (valid ? 1u : 0u);
          tok[k] = t;
          p = advance(p, q64, r64);
        }
      }

      /* ---- wave scan: cell order is k-major (cell = k*64 + lane) */
      uint32_t off[CPL];
      uint32_t total = 0;
#pragma unroll
      for (int k = 0; k < CPL; k++) {
        const uint32_t incl = wave_inclusive_scan(len[k]);
        off[k] = total + incl - len[k];
        total += wave_read_lane(incl, 63);
      }

      ACHIP_SSTAMP(4);
      bool ok;
      const uint32_t base = place_block(blk, total, ok);

      ACHIP_SSTAMP(5);
      if (ok) {
        /* ---- token bytes into this wave's staging area: stream offset g0 (16-byte aligned) sits at its byte 0 */
        const uint32_t g0 = PACK ? 0u : base & ~15u;
        if (CRC && lane == 0) /* the bytes in front of the block inside its first 16-byte group read as zero for the
                                 checksum: leading zeros do not move a zero register (program order: before the tokens) */
          *lds_ptr<uint4>((int)stage_off) = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int k = 0; k < CPL; k++)
          if (len[k] != 0u) {
#if defined(ACHIP_STREAM_ABLATE) && ACHIP_STREAM_ABLATE == 3 /* diagnostics: no token stores either */
            asm volatile("" ::"v"(tok[k].flags), "v"(tok[k].fg), "v"(tok[k].glyph), "v"(off[k]));
#else
            WordSink<L::o_dec, L::o_flags + 16, L::WORDS ? L::o_wr : -1, L::WORDS ? L::o_wg : -1, L::WORDS ? L::o_wm : -1> fs{{stage_addr + (base + off[k] - g0), dummy_addr}};
            token_fields<RMODE>(fs, tok[k], ascii_only);
#endif
          }
        lds_store_fence(); /* DS operations of one wave complete in order: the reads below see every lane's bytes */
        if (!CRC)
          ACHIP_SSTAMP(6);

        drain_block(base, total, g0);
      }
      ACHIP_SSTAMP(CRC ? 6 : 7); /* CRC instantiations: 6 = stores issued, 7 = block checksummed and placed */
      const bool stamp_crc = first_block;
      first_block = false;
      if (blk == nblk - 1 && lane == 0) {
        out_len[fidx] = ok ? base + total : ACHIP_LEN_OVERFLOW;
        if (!PACK && ok && (uint64_t)base + total < out_stride)
          dst[base + total] = 0; /* NUL behind the frame when the slot has room, as the reference's strings carry */
      }
      if (CRC) {
        /* ---- raw CRC of the block, from the staging area (after the stores to HBM have been issued), placed in the
         * frame; the last block to finish completes the frame (stream_crc_* above) */
        if (ok) {
          const uint32_t braw = stream_crc_staged<L>(lds_ptr<const unsigned char>((int)stage_off), (base & 15u) + total, lane);
          stream_crc_place<L>(slots, nblk, nblk_cap, blk, braw, base + total, cap_bytes, lane);  /* (CRC: never PARTS, lb == blk) */
        }
        stream_crc_finish<L>(slots, nblk, nblk_cap, cap_bytes, first_base, fidx, dim_w, dim_h, wire, lane);
      }

      if (CRC && prof && lane == 0 && stamp_crc)
        prof[((size_t)fidx * WAVES + wave) * 8u + 7] = wall_now();

      /* ---- next block */
      cell0 = cell0_next;
      pos = pos_next;
      kinds = kinds_n;
#pragma unroll
      for (int k = 0; k < CPL; k++)
        raw[k] = raw_n[k];
    }
  }
  if (PACK) {
    /* ---- the frame is complete in LDS: claim its place, copy it out.  (Every wave gets here exactly once, those without
     * a block too; the look-back words are final: a block publishes its prefix before it stores its bytes.) */
    __syncthreads();
    const uint32_t lastw = slot_load(&slots[nblk - 1]);
    const uint32_t n_total = lastw & ACHIP_SLOT_VALUE;
    const bool fits = n_total <= cap_bytes;
    const uint32_t room = fits ? (n_total + 15u) & ~15u : 0u;
    unsigned long long *offw = lds_ptr<unsigned long long>(L::o_packoff);
    const uint32_t *fslice = lds_ptr<const uint32_t>(L::o_fcrc + CrcLds::o_slice);
    const uint32_t *fpow = lds_ptr<const uint32_t>(L::o_fcrc + CrcLds::o_powtab);
    uint32_t *fpw = lds_ptr<uint32_t>(L::o_fcrc + CrcLds::o_pow);
    const bool want_pkt = PACK == 2 && wire.hdr && wire.pkt_crc;
    if (tid == 0)
      offw[0] = agent_fetch_add_u64(&pack.cursor[0], (unsigned long long)room);
    if (tid >= 64 && tid < 80 && n_total + (uint32_t)(tid - 64) < room) /* the <= 15 bytes of padding leave as zeros */
      lds_ptr<uint8_t>(frame_lds)[n_total + (uint32_t)(tid - 64)] = 0;
    if (PACK == 2 && want_pkt && fits) { /* the header's share of the packet CRC, on two waves that idle here */
      if (wave == WAVES - 1) {
        const uint32_t xl = crc_x8_pow_len_wave(fpow, n_total, lane, lane_xk);
        if (lane == 0)
          fpw[1] = xl;
      } else if (wave == WAVES - 2) {
        const uint32_t hp = crc_header_part_wave(fslice, dim_w, dim_h, n_total, lane);
        if (lane == 0)
          fpw[0] = hp;
      }
    }
    __syncthreads();
    const unsigned long long off = offw[0];
    const uint4 *from = lds_ptr<const uint4>(frame_lds);
    /* PACK == 2: the first CRCW waves checksum the image while the others copy it out (a per-thread register costs one
     * bit-serial multiplication by the lane's constant at the end: few threads with long Horner chains beat many with
     * short ones) */
    constexpr int CRCW = PACK == 2 ? pack_crc_waves(WAVES) : 0, CB = CRCW * 64;
    static_assert(WAVES > CRCW && WAVES >= 2, "waves left for the copy and the header's share");
    if (tid >= CB && fits && off + room <= pack.capacity) {
      uint8_t *to = pack.dst + off;
      /* (frames land at 16-byte granularity: thread t takes the group t groups behind a LINE boundary of the address, so
       * that every wave's store covers whole 128-byte lines; at most seven threads sit out the first trip) */
      const uint32_t shift = (uint32_t)((uintptr_t)to >> 4) & ((ACHIP_DRAIN_ALIGN >> 4) - 1u);
      for (uint32_t a = (uint32_t)(tid - CB); a < room / 16u + shift; a += BLOCK - CB)
        if (a >= shift)
          store_out16(to + 16u * (a - shift), from[a - shift]);
    }
    if (PACK == 2) {
      /* ---- the frame's CRC-32C from its LDS image: thread t < CB folds groups t, t + CB, ... Horner-style (zero groups
       * in FRONT, the initial value folded into the first four bytes), one reduction over the CRCW waves, then wave 0
       * closes the frame: the < 16 tail bytes, the header's checksum field, the packet CRC (crc_close_wave) --
       * crc32c_frame_kernel's scheme, on LDS instead of a slab slot */
      const uint32_t *mulh = lds_ptr<const uint32_t>(L::o_fcrc + CrcLds::o_mulh);
      uint32_t *tree = lds_ptr<uint32_t>(L::o_fcrc + CrcLds::o_tree);
      const int full = fits ? (int)(n_total >> 4) : 0;
      const uint32_t ntail = fits ? n_total & 15u : 0u;
      const int rounds = (full + CB - 1) / CB, lead = rounds * CB - full;
      uint32_t sreg = 0, tail_byte = 0;
      if (tid < CB) {
        if ((uint32_t)tid < ntail)
          tail_byte = lds_ptr<const uint8_t>(frame_lds)[(uint32_t)full * 16u + (uint32_t)tid];
        for (int j = 0; j < rounds; j++) {
          const int g = j * CB + tid - lead;
          uint4 d = make_uint4(0u, 0u, 0u, 0u);
          if (g >= 0) {
            d = from[g];
            if (g == 0)
              d.x = ~d.x;
          }
          sreg = crc_mul_table(mulh, sreg) ^ crc_raw16(fslice, d);
        }
      }
      const uint32_t whole = crc_reduce_waves<BLOCK, CB>(tree, sreg, tid, lane_k, lane_xk);
      if (wave == 0) {
        if (fits) {
          const CrcClose c = crc_close_wave(fslice, fpow, full > 0 ? whole : 0xFFFFFFFFu, ntail, tail_byte, want_pkt, fpw[0],
                                            fpw[1], lane, lane_xk);
          if (lane == 0) {
            wire.crc[fidx] = ~c.st;
            if (wire.hdr)
              crc_store_header(wire.hdr, fidx, dim_w, dim_h, n_total, ~c.st);
            if (want_pkt)
              wire.pkt_crc[fidx] = c.pkt;
          }
        } else if (lane == 0) { /* as the fused stream checksum reports a frame that did not fit: a header of zeros, its CRC behind it */
          wire.crc[fidx] = 0u;
          if (wire.hdr) {
            for (int j = 0; j < 24; j++)
              wire.hdr[(size_t)fidx * 24u + j] = 0;
            if (wire.pkt_crc)
              wire.pkt_crc[fidx] = ~crc_mulmod(0xFFFFFFFFu, crc_pow(CRC_X8, 24u));
          }
        }
      }
    }
    if (tid == 0)
      pack_report(off, fits ? n_total : ACHIP_LEN_OVERFLOW);
  }
}

} // namespace achip
