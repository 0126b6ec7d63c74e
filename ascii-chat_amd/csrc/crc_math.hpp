/*
 * crc_math.hpp -- the GF(2) toolbox of the wire stage: CRC-32C constants, table builders, constant multipliers and the
 * packet-header helpers shared by the stand-alone CRC kernels (crc_kernels.hpp) and by the stream render kernel, whose
 * drain can carry the frame CRC (render_stream.hpp).  Device functions only: safe to include from every translation unit.
 * See crc_kernels.hpp for the algebra.
 */
#pragma once

#include "render_kernels.hpp"

namespace achip {

constexpr uint32_t CRC32C_POLY = 0x82F63B78u; /* reflected 0x1EDC6F41 */
constexpr uint32_t CRC_X0 = 0x80000000u;      /* the polynomial "1" in reflected bit order */
constexpr uint32_t CRC_X8 = 0x00800000u;      /* x^8: the register after one more zero byte      */
constexpr uint32_t CRC_XINV8 = 0xFDE39562u;   /* x^-8: CRC_X8 * CRC_XINV8 == 1 (checked in tests) */
constexpr int CRC_BLOCK = 256;

/* a * b mod P (reflected operands) */
__host__ __device__ constexpr uint32_t crc_mulmod(uint32_t a, uint32_t b) {
  uint32_t p = 0;
#pragma unroll /* with a compile-time b every b*x^i folds to a literal: two instructions per bit */
  for (int i = 0; i < 32; i++) {
    if (a & (0x80000000u >> i))
      p ^= b;
    b = (b & 1u) ? (b >> 1) ^ CRC32C_POLY : b >> 1; /* b *= x */
  }
  return p;
}

/* base^n mod P by square-and-multiply (host side and compile-time constants) */
__host__ __device__ constexpr uint32_t crc_pow(uint32_t base, uint64_t n) {
  uint32_t r = CRC_X0;
  while (n) {
    if (n & 1ull)
      r = crc_mulmod(r, base);
    base = crc_mulmod(base, base);
    n >>= 1;
  }
  return r;
}

/* x^(8 * 2^k), k = 0..31 (generated with crc_pow; checked against it in tests/test_crc_wire.py) */
__device__ const uint32_t CRC_X8_POW2[32] = {
    0x00800000u, 0x00008000u, 0x82F63B78u, 0x6EA2D55Cu, 0x18B8EA18u, 0x510AC59Au, 0xB82BE955u, 0xB8FDB1E7u,
    0x88E56F72u, 0x74C360A4u, 0xE4172B16u, 0x0D65762Au, 0x35D73A62u, 0x28461564u, 0xBF455269u, 0xE2EA32DCu,
    0xFE7740E6u, 0xF946610Bu, 0x3C204F8Fu, 0x538586E3u, 0x59726915u, 0x734D5309u, 0xBC1AC763u, 0x7D0722CCu,
    0xD289CABEu, 0xE94CA9BCu, 0x05B74F3Fu, 0xA51E1F42u, 0x40000000u, 0x20000000u, 0x08000000u, 0x00800000u};

/* x^(8n) from the table: one multiplication per set bit of n */
__device__ inline uint32_t crc_x8_pow(uint32_t n) {
  uint32_t r = CRC_X0;
  for (int k = 0; n; k++, n >>= 1)
    if (n & 1u)
      r = crc_mulmod(r, CRC_X8_POW2[k]);
  return r;
}

/* register after one byte from state s (bitwise) */
__host__ __device__ constexpr uint32_t crc_byte(uint32_t s, uint32_t byte) {
  s ^= byte;
  for (int j = 0; j < 8; j++)
    s = (s & 1u) ? (s >> 1) ^ CRC32C_POLY : s >> 1;
  return s;
}

struct CrcLds {
  static constexpr int o_slice = 0;                  /* uint32 [16][256]: byte b followed by k zero bytes */
  static constexpr int o_mulh = o_slice + 16 * 1024; /* uint32 [4][256]: (v << 8k) * x^(128*256)         */
  static constexpr int o_powtab = o_mulh + 4 * 1024; /* uint32 [2][256]: x^(8i), x^(8*256*i) -- prebuilt images only (CRC_POW_TAB) */
  static constexpr int o_tree = o_powtab + 2 * 1024; /* uint32 [1024]                                     */
  static constexpr int o_pow = o_tree + 4096;        /* uint32 [64]: product trees x^(8*len), x^(-8*surplus) */
  static constexpr int o_pack = o_pow + 256;         /* uint32 [2][16]: wave totals of the packed-offset prefix */
  static constexpr int bytes = o_pack + 128;
};

/* raw() of 16 bytes held little-endian in four dwords */
__device__ inline uint32_t crc_raw16(const uint32_t *slice, uint4 d) {
  uint32_t r = 0;
  const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
  for (int q = 0; q < 4; q++) {
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int m = 4 * q + b; /* byte m of the group is followed by 15 - m bytes */
      r ^= slice[(15 - m) * 256 + ((w[q] >> (8 * b)) & 0xFFu)];
    }
  }
  return r;
}

__device__ inline uint32_t crc_mul_table(const uint32_t *t, uint32_t s) {
  return t[s & 0xFFu] ^ t[256 + ((s >> 8) & 0xFFu)] ^ t[512 + ((s >> 16) & 0xFFu)] ^ t[768 + (s >> 24)];
}

/* x^(8*v), x^(8*256*v), x^(8*65536*v) for v = 0..255: x^(8*len) for len < 2^24 is a product of three entries */
struct CrcPowTab {
  uint32_t t[3][256];
};
__host__ __device__ constexpr CrcPowTab crc_make_pow_tab() {
  CrcPowTab r{};
  uint32_t step = CRC_X8;
  for (int k = 0; k < 3; k++) {
    uint32_t v = CRC_X0;
    for (int i = 0; i < 256; i++) {
      r.t[k][i] = v;
      v = crc_mulmod(v, step);
    }
    step = v; /* step^256 */
  }
  return r;
}
__device__ const CrcPowTab CRC_POW_TAB = crc_make_pow_tab();

__device__ inline uint32_t crc_x8_pow_len(uint32_t len) {
  uint32_t r = crc_mulmod(CRC_POW_TAB.t[0][len & 0xFFu], CRC_POW_TAB.t[1][(len >> 8) & 0xFFu]);
  r = crc_mulmod(r, CRC_POW_TAB.t[2][(len >> 16) & 0xFFu]);
  if (len >> 24)
    r = crc_mulmod(r, crc_x8_pow(len & 0xFF000000u));
  return r;
}

/* Combining the registers of a workgroup whose thread t checksummed groups t, t + BLOCK, ... (so that its last group is
 * followed by BLOCK-1 - t groups) WITHOUT a barrier-fenced tree: lane l of a wave multiplies by x^(128 * (63 - l)) -- a
 * per-lane constant from this table -- one xor reduction gives the wave's register, and the (<= 16) wave registers are
 * folded with the constant x^(128 * 64) by one wave (render_stream.hpp: crc_reduce_waves).  x^k for the wave-uniform
 * multiply (wave_mulmod_uniform) rides in the same table. */
struct CrcLaneTab {
  uint32_t k[64];  /* x^(128 * (63 - l)) */
  uint32_t xk[64]; /* x^l mod P, l < 63; 0 for l = 63 */
};
__host__ __device__ constexpr CrcLaneTab crc_make_lane_tab() {
  CrcLaneTab r{};
  for (int l = 0; l < 64; l++) {
    r.k[l] = crc_pow(CRC_X8, 16ull * (uint64_t)(63 - l));
    r.xk[l] = l == 63 ? 0u : (l < 32 ? 0x80000000u >> l : crc_mulmod(1u, 0x80000000u >> (l - 31)));
  }
  return r;
}
__device__ const CrcLaneTab CRC_LANE_TAB = crc_make_lane_tab();

__device__ inline uint32_t bswap32(uint32_t v) {
  return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24);
}

/* slicing tables + the Horner table for x^(128*BLOCK); needs a barrier afterwards */
template <int BLOCK> __device__ inline void crc_build_tables(uint32_t *slice, uint32_t *mulh, int tid) {
  if (tid < 256)
    slice[tid] = crc_byte(0u, (uint32_t)tid);
  __syncthreads();
  if (tid < 256) {
    uint32_t v = slice[tid];
    for (int k = 1; k < 16; k++) {
      v = (v >> 8) ^ slice[v & 0xFFu]; /* one more zero byte */
      slice[k * 256 + tid] = v;
    }
    constexpr uint32_t CH = crc_pow(CRC_X8, 16u * BLOCK); /* x^(128*BLOCK) */
#pragma unroll
    for (int k = 0; k < 4; k++)
      mulh[k * 256 + tid] = crc_mulmod((uint32_t)tid << (8 * k), CH);
  }
}

/* combine the thread registers: thread t's last group is followed by BLOCK-1 - t groups; result in tree[0] */
template <int BLOCK> __device__ inline void crc_tree(uint32_t *tree, uint32_t s, int tid) {
  tree[tid] = s;
  __syncthreads();
  constexpr uint32_t TC[10] = {crc_pow(CRC_X8, 16ull << 0), crc_pow(CRC_X8, 16ull << 1), crc_pow(CRC_X8, 16ull << 2),
                               crc_pow(CRC_X8, 16ull << 3), crc_pow(CRC_X8, 16ull << 4), crc_pow(CRC_X8, 16ull << 5),
                               crc_pow(CRC_X8, 16ull << 6), crc_pow(CRC_X8, 16ull << 7), crc_pow(CRC_X8, 16ull << 8),
                               crc_pow(CRC_X8, 16ull << 9)};
#pragma unroll
  for (int k = 0; (1 << k) < BLOCK; k++) {
    const int d = 1 << k;
    if ((tid & (2 * d - 1)) == 0)
      tree[tid] = crc_mulmod(tree[tid], TC[k]) ^ tree[tid + d];
    __syncthreads();
  }
}

/* the 24-byte ascii_frame_packet_t of frame i in network byte order + the CRC of header || frame.
 * state16 = CRC register after header bytes 0..15 (from 0xFFFFFFFF), xl = x^(8*len), s = register after the
 * frame (from 0xFFFFFFFF).  byte_table = slice[0] or NULL (bitwise).  One thread. */
__device__ inline void crc_emit_packet(uint32_t state16, uint32_t xl, uint32_t s, uint32_t crc, uint32_t w, uint32_t h,
                                       uint32_t len, bool bad, int i, const uint32_t *byte_table,
                                       uint8_t *__restrict__ hdr_out, uint32_t *__restrict__ pkt_crc_out) {
  uint32_t *hp = reinterpret_cast<uint32_t *>(hdr_out + (size_t)i * 24u); /* 8-byte aligned */
  hp[0] = bswap32(w); /* HOST_TO_NET_U32 */
  hp[1] = bswap32(h);
  hp[2] = bswap32(len);
  hp[3] = 0u;
  hp[4] = bswap32(crc);
  hp[5] = 0u;
  if (!pkt_crc_out)
    return;
  uint32_t st = state16;
  for (int k = 0; k < 8; k++) { /* checksum (big-endian) and flags */
    const uint32_t b = k < 4 ? (crc >> (8 * (3 - k))) & 0xFFu : 0u;
    st = byte_table ? (st >> 8) ^ byte_table[(st ^ b) & 0xFFu] : crc_byte(st, b);
  }
  /* clocking the frame in from register st: st * x^(8 len) + raw(frame), and s = 0xFFFFFFFF * x^(8 len) + raw(frame) */
  pkt_crc_out[i] = bad ? 0u : ~(crc_mulmod(st ^ 0xFFFFFFFFu, xl) ^ s);
}

/* CRC register after the first 16 header bytes {width, height, len, 0} in network byte order */
__device__ inline uint32_t crc_header_state16(uint32_t w, uint32_t h, uint32_t len) {
  const uint32_t f[4] = {w, h, len, 0u};
  uint32_t st = 0xFFFFFFFFu;
  for (int k = 0; k < 16; k++)
    st = crc_byte(st, (f[k >> 2] >> (8 * (3 - (k & 3)))) & 0xFFu);
  return st;
}

/* ---- wave-level GF(2) helpers (shared by the stand-alone frame kernel and the render kernels) ---------------------------- */
/* a * b mod P for WAVE-UNIFORM a, b, by the whole wave: lane k owns coefficient k of the 63-term carry-less product
 * (parity of a's bits against b slid to position k) and contributes x^k mod P (xk, lane 63: 0); one xor reduction.
 * ~14 instructions where the bit-serial crc_mulmod takes ~220.  Result uniform. */
__device__ inline uint32_t wave_mulmod_uniform(uint32_t a, uint32_t b, int lane, uint32_t xk) {
  /* reflected order: bit 31-i of a word is the coefficient of x^i.  Coefficient k of the product pairs bit p of a
   * with bit 62-k-p of b: the parity of a & m_k, m_k = bitreverse(b) slid so that its bit 31 lands on bit 62-k */
  const uint32_t rb = bitreverse32(b);
  const uint32_t m = (uint32_t)((((uint64_t)rb) << 31) >> lane);
  const uint32_t c = (uint32_t)__builtin_popcount(a & m) & 1u;
  return wave_read_lane(wave_xor_to_last((0u - c) & xk), 63);
}
/* Sum over the workgroup's first ACTIVE threads of s_t * x^(128 * (ACTIVE-1 - t)) -- the register after all their groups
 * -- valid in wave 0: one multiplication by the lane's constant, one xor reduction per wave, one hand-off through `scratch`
 * (ACTIVE / 64 words of LDS), ACTIVE/64 - 1 wave-uniform multiplications.  klane / xk: this lane's entries of CRC_LANE_TAB
 * (requested at kernel entry).  Every thread of the workgroup calls it (one barrier); waves behind the ACTIVE ones only
 * pass the barrier.  Replaces crc_tree's log2(BLOCK) barrier-fenced levels. */
template <int BLOCK, int ACTIVE = BLOCK>
__device__ inline uint32_t crc_reduce_waves(uint32_t *scratch, uint32_t s, int tid, uint32_t klane, uint32_t xk) {
  static_assert(ACTIVE % 64 == 0 && ACTIVE <= BLOCK, "whole waves");
  constexpr uint32_t WC = crc_pow(CRC_X8, 16ull * 64ull); /* x^(128 * 64): one wave's worth of groups */
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  if (ACTIVE == BLOCK || wave < ACTIVE / 64) {
    const uint32_t v = wave_read_lane(wave_xor_to_last(crc_mulmod(s, klane)), 63);
    if (lane == 0)
      scratch[wave] = v;
  }
  __syncthreads();
  uint32_t acc = 0;
  if (wave == 0) {
    acc = scratch[0];
    for (int w = 1; w < ACTIVE / 64; w++)
      acc = wave_mulmod_uniform(acc, WC, lane, xk) ^ scratch[w];
  }
  return acc;
}

/* ---- a frame's closing arithmetic by whole waves (uniform operands, all 64 lanes active) instead of one thread's
 * byte-at-a-time loops and bit-serial multiplications: those were ~2 us at the end of an 11 us launch (round 4) ---------- */
__device__ inline uint32_t wave_xor_all(uint32_t v) { return wave_read_lane(wave_xor_to_last(v), 63); }

/* x^(8 len): two look-ups in the prebuilt power tables (CrcLds::o_powtab), one multiplication (one more per set bit of
 * len above 64 KB) */
__device__ inline uint32_t crc_x8_pow_len_wave(const uint32_t *powtab, uint32_t len, int lane, uint32_t xk) {
  uint32_t r = wave_mulmod_uniform(powtab[len & 0xFFu], powtab[256u + ((len >> 8) & 0xFFu)], lane, xk);
  for (int k = 16; (len >> k) != 0u; k++)
    if ((len >> k) & 1u)
      r = wave_mulmod_uniform(r, CRC_X8_POW2[k], lane, xk);
  return r;
}

/* What the 24-byte header {w, h, len, 0, crc, 0} (network order) contributes to the packet CRC before the frame's own CRC
 * is known: (register after the 8 bytes {w, h}) clocked through the 16 bytes {len, 0, 0, 0}.  With the checksum's four
 * bytes added at their place (crc_close_wave) it is the register after the whole header.  slice: the slicing tables. */
__device__ inline uint32_t crc_header_part_wave(const uint32_t *slice, uint32_t w, uint32_t h, uint32_t len, int lane) {
  constexpr uint32_t INIT8 = crc_mulmod(0xFFFFFFFFu, crc_pow(CRC_X8, 8u)); /* 0xFFFFFFFF clocked through 8 zero bytes */
  const int sh = 8 * (3 - (lane & 3));
  const uint32_t fb = ((lane < 4 ? w : h) >> sh) & 0xFFu; /* header byte `lane`, lane < 8: followed by 7 - lane bytes */
  const uint32_t st8 = INIT8 ^ wave_xor_all(lane < 8 ? slice[(7 - (lane & 7)) * 256 + fb] : 0u);
  /* the next 16 bytes: st8 goes into the first four (the length's), twelve zero bytes follow */
  const uint32_t lb = ((len >> sh) & 0xFFu) ^ ((st8 >> (8 * (lane & 3))) & 0xFFu);
  return wave_xor_all(lane < 4 ? slice[(15 - (lane & 3)) * 256 + lb] : 0u);
}

/* reg: the register after the frame's whole 16-byte groups (0xFFFFFFFF when it has none); ntail < 16 bytes follow, lane l
 * < ntail holding byte l in tail_byte.  Returns the register after the frame; pkt (when want_pkt) = the CRC of header ||
 * frame from hpart = crc_header_part_wave() and xl = x^(8 len).  powtab: CrcLds::o_powtab. */
struct CrcClose {
  uint32_t st, pkt;
};
__device__ inline CrcClose crc_close_wave(const uint32_t *slice, const uint32_t *powtab, uint32_t reg, uint32_t ntail,
                                          uint32_t tail_byte, bool want_pkt, uint32_t hpart, uint32_t xl, int lane,
                                          uint32_t xk) {
  CrcClose r{reg, 0u};
  if (ntail) { /* reg * x^(8 ntail) + raw(tail): byte l of the tail is followed by ntail - 1 - l bytes */
    const bool mine = (uint32_t)lane < ntail;
    const uint32_t c = mine ? slice[(mine ? ntail - 1u - (uint32_t)lane : 0u) * 256u + (tail_byte & 0xFFu)] : 0u;
    r.st = wave_mulmod_uniform(reg, powtab[ntail], lane, xk) ^ wave_xor_all(c);
  }
  if (want_pkt) {
    const uint32_t crc = ~r.st; /* bytes 16..19 of the header, big-endian; byte 16 + k is followed by 7 - k bytes */
    const uint32_t cb = (crc >> (8 * (3 - (lane & 3)))) & 0xFFu;
    const uint32_t st24 = hpart ^ wave_xor_all(lane < 4 ? slice[(7 - (lane & 3)) * 256 + cb] : 0u);
    /* clocking the frame in from register st24: st24 * x^(8 len) + raw(frame); r.st = 0xFFFFFFFF * x^(8 len) + raw(frame) */
    r.pkt = ~(wave_mulmod_uniform(st24 ^ 0xFFFFFFFFu, xl, lane, xk) ^ r.st);
  }
  return r;
}

/* the 24-byte ascii_frame_packet_t of frame i in network byte order (one thread) */
__device__ inline void crc_store_header(uint8_t *__restrict__ hdr_out, int i, uint32_t w, uint32_t h, uint32_t len, uint32_t crc) {
  uint32_t *hp = reinterpret_cast<uint32_t *>(hdr_out + (size_t)i * 24u); /* 8-byte aligned */
  hp[0] = bswap32(w); /* HOST_TO_NET_U32 */
  hp[1] = bswap32(h);
  hp[2] = bswap32(len);
  hp[3] = 0u;
  hp[4] = bswap32(crc);
  hp[5] = 0u;
}

/* The constant tables of a workgroup of BLOCK threads that checksums one frame (crc32c_frame_kernel; the PACK == 2
 * instantiations of the stream kernel, which checksum the frame's LDS image): crc_kernels.hpp's slicing tables, the Horner table for x^(128 * BLOCK)
 * and the two power tables, the first ACHIP_FRAME_CRC_TAB_BYTES of the CrcLds layout, written once per process into global memory; every launch copies the image into LDS. */
#define ACHIP_FRAME_CRC_TAB_BYTES (22 * 1024)
template <int BLOCK> __global__ void __launch_bounds__(256) crc_frame_tables_init_kernel(uint32_t *tab) {
  static_assert(CrcLds::o_slice == 0 && CrcLds::o_mulh == 16 * 1024 && CrcLds::o_powtab == 20 * 1024 &&
                    CrcLds::o_tree == ACHIP_FRAME_CRC_TAB_BYTES, "slicing tables, the Horner table, the power tables");
  const int tid = (int)threadIdx.x;
  uint32_t *slice = lds_ptr<uint32_t>(CrcLds::o_slice), *mulh = lds_ptr<uint32_t>(CrcLds::o_mulh);
  crc_build_tables<BLOCK>(slice, mulh, tid);
  lds_ptr<uint32_t>(CrcLds::o_powtab)[tid] = CRC_POW_TAB.t[0][tid];
  lds_ptr<uint32_t>(CrcLds::o_powtab)[256 + tid] = CRC_POW_TAB.t[1][tid];
  __syncthreads();
  for (int k = tid; k < ACHIP_FRAME_CRC_TAB_BYTES / 4; k += 256)
    tab[k] = lds_ptr<const uint32_t>(0)[k];
}


} // namespace achip
