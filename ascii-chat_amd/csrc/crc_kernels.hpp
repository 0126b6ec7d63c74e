/*
 * crc_kernels.hpp -- the wire stage right after render: CRC-32C of every rendered
 * frame and the ascii_frame_packet_t header that precedes it on the wire.
 *
 *   asciichat_crc32 (lib/network/crc32.c:95-190): CRC-32C (Castagnoli), reflected, polynomial 0x82F63B78,
 *       initial value 0xFFFFFFFF, final complement; hardware and software paths give the same value.
 *   acip_send_ascii_frame (lib/network/acip/server.c:186-214): header {width, height, original_size,
 *       compressed_size = 0, checksum = crc(frame), flags = 0}, every field in network byte order, then
 *       the frame bytes;  packet_send_via_transport (lib/network/acip/send.c:59-69) checksums header+frame
 *       once more for the outer packet header.
 *
 * A CRC is linear over GF(2): with raw(M) = register after M starting from 0,
 *       raw(A || B) = raw(A) * x^(8|B|)  xor  raw(B)           (mod P, reflected bit order)
 *       crc(M)      = ~( 0xFFFFFFFF * x^(8|M|)  xor  raw(M) )
 * so a frame is cut into 16-byte groups that are checksummed independently and combined with constant
 * multipliers.  Thread t of a workgroup owns groups t, t+256, t+512, ... of its span (coalesced 16-byte loads)
 * and folds them Horner-style with the constant x^(128*256); the 256 thread results are combined by a tree
 * whose level k multiplies by x^(128 * 2^k): all multipliers are compile-time constants.  Multiplication by
 * the Horner constant is 4 lookups in a 4 x 256 table; raw() of one group is 16 lookups in slicing tables --
 * 20 LDS lookups per 16 bytes, all tables (20 KB) built in LDS by the workgroup itself.
 *
 * Frames up to 128 KB (every rendered frame of the BASELINE configurations) are checksummed by one
 * 1024-thread workgroup (crc32c_frame_kernel): the group grid is padded with zero groups in FRONT (leading
 * zeros do not move a zero register), the initial value is folded into the data (complementing the first four
 * bytes), and the < 16 tail bytes are clocked in byte-wise -- no variable power of x is needed for the frame
 * CRC.  The packet CRC needs x^(8*len) once; a second wave computes it (and the CRC state of the header
 * fields that are known up front) while the others reduce the frame.  Larger buffers (ingest payloads) are cut
 * into 64 KB spans (crc32c_span_kernel) whose registers a small second kernel combines.
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
  static constexpr int o_tree = o_mulh + 4 * 1024;   /* uint32 [1024]                                     */
  static constexpr int o_pow = o_tree + 4096;        /* uint32 [64]: product trees x^(8*len), x^(-8*surplus) */
  static constexpr int bytes = o_pow + 256;
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

/*
 * One workgroup of BLOCK threads per frame of at most 128 KB.  len == NULL: every frame is fixed_len bytes.
 * Lengths >= 0xFFFFFFF0 are the render kernel's error codes: such a frame gets CRC 0 and a header with zero
 * dimensions.  The group grid of ceil(full_groups / BLOCK) rounds is padded with zero groups in front; loads are
 * requested four rounds ahead of the table lookups that consume them.
 */
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
    crc32c_frame_kernel(const uint8_t *__restrict__ base, uint64_t stride, const uint32_t *__restrict__ len,
                        uint32_t fixed_len, int n_frames, const uint32_t *__restrict__ dims,
                        uint32_t *__restrict__ crc_out, uint8_t *__restrict__ hdr_out, uint32_t *__restrict__ pkt_crc_out) {
  static_assert(BLOCK == 256 || BLOCK == 1024, "tree constants");
  uint32_t *slice = lds_ptr<uint32_t>(CrcLds::o_slice);
  uint32_t *mulh = lds_ptr<uint32_t>(CrcLds::o_mulh);
  uint32_t *tree = lds_ptr<uint32_t>(CrcLds::o_tree);
  uint32_t *pw = lds_ptr<uint32_t>(CrcLds::o_pow);
  const int tid = (int)threadIdx.x;
  const int i = (int)blockIdx.x;
  if (i >= n_frames)
    return;
  uint32_t L = len ? len[i] : fixed_len;
  const bool bad = L >= 0xFFFFFFF0u;
  if (bad)
    L = 0;
  const uint32_t w = dims && !bad ? dims[2 * i] : 0u, h = dims && !bad ? dims[2 * i + 1] : 0u;
  const uint8_t *src = base + (size_t)i * stride;
  const int full = (int)(L >> 4);                     /* whole 16-byte groups                    */
  const int rounds = (full + BLOCK - 1) / BLOCK;      /* 0 for a frame shorter than 16 bytes      */
  const int lead = rounds * BLOCK - full;             /* zero groups in front of the frame        */

  /* the two scalar jobs of the packet CRC run on waves that idle while waves 0..3 build the tables */
  if (hdr_out && pkt_crc_out) {
    if (tid == BLOCK - 64)
      pw[0] = crc_header_state16(w, h, L);
    else if (tid == BLOCK - 128)
      pw[1] = crc_x8_pow_len(L);
  }
  crc_build_tables<BLOCK>(slice, mulh, tid);
  __syncthreads();

  auto load_group = [&](int j) -> uint4 {
    const int g = j * BLOCK + tid - lead;
    uint4 d = make_uint4(0u, 0u, 0u, 0u);
    if (j < rounds && g >= 0) {
      d = *reinterpret_cast<const uint4 *>(src + (size_t)g * 16u);
      if (g == 0)
        d.x = ~d.x; /* initial value 0xFFFFFFFF = the first four message bytes complemented */
    }
    return d;
  };
  uint32_t s = 0;
  for (int j0 = 0; j0 < rounds; j0 += 4) {
    uint4 d[4];
#pragma unroll
    for (int u = 0; u < 4; u++)
      d[u] = load_group(j0 + u);
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (j0 + u < rounds)
        s = crc_mul_table(mulh, s) ^ crc_raw16(slice, d[u]);
  }
  crc_tree<BLOCK>(tree, s, tid);
  if (tid == 0) {
    uint32_t st = full > 0 ? tree[0] : 0xFFFFFFFFu;
    for (uint32_t k = (uint32_t)full * 16u; k < L; k++) /* < 16 tail bytes */
      st = (st >> 8) ^ slice[(st ^ src[k]) & 0xFFu];
    const uint32_t crc = bad ? 0u : ~st;
    crc_out[i] = crc;
    if (hdr_out)
      crc_emit_packet(pw[0], pw[1], st, crc, w, h, L, bad, i, slice, hdr_out, pkt_crc_out);
  }
}

/*
 * Large buffers: grid = n_frames * parts workgroups of 256 threads; workgroup (i, p) reduces the fixed span
 * [p*span, (p+1)*span) of frame i, span = rounds * 4 KB, parts*span >= every length (zeros behind the end of
 * the frame), and stores its raw register (initial value 0) in partial[i*parts + p].
 */
__global__ void __launch_bounds__(256)
    crc32c_span_kernel(const uint8_t *__restrict__ base, uint64_t stride, const uint32_t *__restrict__ len,
                       uint32_t fixed_len, int n_frames, int parts, int rounds, uint32_t *__restrict__ partial) {
  constexpr int BLOCK = 256;
  uint32_t *slice = lds_ptr<uint32_t>(CrcLds::o_slice);
  uint32_t *mulh = lds_ptr<uint32_t>(CrcLds::o_mulh);
  uint32_t *tree = lds_ptr<uint32_t>(CrcLds::o_tree);
  const int tid = (int)threadIdx.x;
  const int i = (int)blockIdx.x / parts, p = (int)blockIdx.x - i * parts;
  if (i >= n_frames)
    return;
  uint32_t L = len ? len[i] : fixed_len;
  if (L >= 0xFFFFFFF0u)
    L = 0;
  const uint64_t lo = (uint64_t)p * (uint64_t)rounds * (16u * BLOCK);
  if (lo >= L) { /* nothing but zeros: raw() of zeros from 0 is 0 */
    if (tid == 0)
      partial[(size_t)i * parts + p] = 0u;
    return;
  }
  const uint64_t avail = (uint64_t)L - lo; /* bytes of the frame from the start of this span (may exceed the span) */
  crc_build_tables<BLOCK>(slice, mulh, tid);
  __syncthreads();
  const uint8_t *src = base + (size_t)i * stride + lo;
  auto load_group = [&](int j) -> uint4 {
    const uint64_t off = ((uint64_t)j * BLOCK + (uint64_t)tid) * 16u;
    uint4 d = make_uint4(0u, 0u, 0u, 0u);
    if (j < rounds && off < avail) {
      const uint64_t left = avail - off;
      if (left >= 16u) {
        d = *reinterpret_cast<const uint4 *>(src + off);
      } else { /* the frame ends inside this group: byte loads, the bytes behind the end count as zeros */
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        for (uint32_t k = 0; k < (uint32_t)left; k++)
          w[k >> 2] |= (uint32_t)src[off + k] << (8u * (k & 3u));
        d = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    return d;
  };
  uint32_t s = 0;
  for (int j0 = 0; j0 < rounds; j0 += 4) {
    uint4 d[4];
#pragma unroll
    for (int u = 0; u < 4; u++)
      d[u] = load_group(j0 + u);
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (j0 + u < rounds)
        s = crc_mul_table(mulh, s) ^ crc_raw16(slice, d[u]);
  }
  crc_tree<BLOCK>(tree, s, tid);
  if (tid == 0)
    partial[(size_t)i * parts + p] = tree[0];
}

/* One 64-thread workgroup per frame combines the span registers.  Register q of the frame is followed by
 * parts-1-q spans: 64 at a time, tree-combined with powers of cspan = x^(8*span); the surplus zero bytes
 * parts*span - len are divided out (xinv_v = x^(-8*parts*span) from the host; x is invertible mod P). */
__global__ void __launch_bounds__(64)
    crc32c_finish_kernel(const uint32_t *__restrict__ partial, int parts, uint32_t cspan, uint32_t xinv_v,
                         const uint32_t *__restrict__ len, uint32_t fixed_len, int n_frames,
                         const uint32_t *__restrict__ dims, uint32_t *__restrict__ crc_out, uint8_t *__restrict__ hdr_out,
                         uint32_t *__restrict__ pkt_crc_out) {
  uint32_t *tree = lds_ptr<uint32_t>(0);
  const int i = (int)blockIdx.x, tid = (int)threadIdx.x;
  if (i >= n_frames)
    return;
  uint32_t L = len ? len[i] : fixed_len;
  const bool bad = L >= 0xFFFFFFF0u;
  if (bad)
    L = 0;
  uint32_t c64 = cspan; /* cspan^64 */
  for (int k = 0; k < 6; k++)
    c64 = crc_mulmod(c64, c64);
  uint32_t acc = 0;
  const int lead = (64 - parts % 64) % 64; /* zero registers in front keep every batch of 64 full */
  for (int q0 = -lead; q0 < parts; q0 += 64) {
    const int q = q0 + tid;
    tree[tid] = q >= 0 ? partial[(size_t)i * parts + q] : 0u;
    __syncthreads();
    uint32_t cx = cspan;
    for (int d = 1; d < 64; d <<= 1) {
      if ((tid & (2 * d - 1)) == 0)
        tree[tid] = crc_mulmod(tree[tid], cx) ^ tree[tid + d];
      cx = crc_mulmod(cx, cx);
      __syncthreads();
    }
    acc = crc_mulmod(acc, c64) ^ tree[0];
    __syncthreads();
  }
  if (tid == 0) {
    const uint32_t xl = crc_x8_pow_len(L);
    const uint32_t raw = crc_mulmod(acc, crc_mulmod(xinv_v, xl)); /* raw() of exactly len bytes */
    const uint32_t st = crc_mulmod(0xFFFFFFFFu, xl) ^ raw;         /* register after the frame from 0xFFFFFFFF */
    const uint32_t crc = bad ? 0u : ~st;
    crc_out[i] = crc;
    if (hdr_out) {
      const uint32_t w = dims && !bad ? dims[2 * i] : 0u, h = dims && !bad ? dims[2 * i + 1] : 0u;
      crc_emit_packet(crc_header_state16(w, h, L), xl, st, crc, w, h, L, bad, i, nullptr, hdr_out, pkt_crc_out);
    }
  }
}

} // namespace achip
