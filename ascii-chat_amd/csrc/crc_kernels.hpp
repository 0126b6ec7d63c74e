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

#include "crc_math.hpp"

namespace achip {

/*
 * COPY instantiations: the pass that checksums a slab also COMPACTS it (stream_kernels.hpp pack_frames_kernel's layout:
 * frame i at dst + off[i], off[i] = sum over j < i of round16(len[j]); dst may be mapped pinned host memory) -- every
 * 16-byte group is stored where it was loaded for the checksum, so "checksums + headers + exact-length frames on the host"
 * is ONE pass over the slab instead of two (a rows-kernel plan's tick: render -> this; profiles/r03_bench.json wire_stage).
 * The last group of a frame travels whole (up to 15 bytes of the slot behind the frame's end: the padding the layout allows).
 */
struct CrcPack {
  uint8_t *dst;
  uint64_t capacity;
  uint64_t *off_out; /* n + 1 offsets, or NULL */
  uint32_t *len_out; /* n lengths (error codes as they are), or NULL */
};
__device__ inline uint32_t crc_pack_len_ok(uint32_t l) { return l >= 0xFFFFFFF0u ? 0u : l; }
/* off[i] and the total, recomputed by every workgroup from len[] (as pack_frames_kernel does); called by all BLOCK threads
 * BEFORE the workgroup's first barrier, read back (crc_pack_offset_read) after it */
template <int BLOCK>
__device__ inline void crc_pack_offset_post(const uint32_t *len, uint32_t fixed_len, int n, int i, int tid) {
  uint32_t *wsum = lds_ptr<uint32_t>(CrcLds::o_pack);
  uint32_t below = 0, all = 0;
  for (int j = tid; j < n; j += BLOCK) {
    const uint32_t g = (crc_pack_len_ok(len ? len[j] : fixed_len) + 15u) >> 4;
    below += j < i ? g : 0u;
    all += g;
  }
  below = wave_read_lane(wave_inclusive_scan(below), 63);
  all = wave_read_lane(wave_inclusive_scan(all), 63);
  if ((tid & 63) == 0) {
    wsum[tid >> 6] = below;
    wsum[16 + (tid >> 6)] = all;
  }
}
template <int BLOCK> __device__ inline void crc_pack_offset_read(uint64_t &off, uint64_t &total) {
  const uint32_t *wsum = lds_ptr<const uint32_t>(CrcLds::o_pack);
  uint64_t a = 0, b = 0;
#pragma unroll
  for (int w = 0; w < BLOCK / 64; w++) {
    a += wsum[w];
    b += wsum[16 + w];
  }
  off = 16ull * a;
  total = 16ull * b;
}

/*
 * One workgroup of BLOCK threads per frame of at most 128 KB.  len == NULL: every frame is fixed_len bytes.
 * Lengths >= 0xFFFFFFF0 are the render kernel's error codes: such a frame gets CRC 0 and a header with zero
 * dimensions.  The group grid of ceil(full_groups / BLOCK) rounds is padded with zero groups in front; loads are
 * requested four rounds ahead of the table lookups that consume them.
 */
template <int BLOCK, bool COPY = false>
__global__ void __launch_bounds__(BLOCK)
    crc32c_frame_kernel(const uint8_t *__restrict__ base, uint64_t stride, const uint32_t *__restrict__ len,
                        uint32_t fixed_len, int n_frames, const uint32_t *__restrict__ dims,
                        uint32_t *__restrict__ crc_out, uint8_t *__restrict__ hdr_out, uint32_t *__restrict__ pkt_crc_out,
                        CrcPack pack, const uint4 *__restrict__ tab) {
  /* tab: the 22 KB image of crc_frame_tables_init_kernel<BLOCK> (slicing tables, Horner table, power tables), built once per process
   * and copied into LDS here -- building it in every workgroup cost ~2 us of a ~8 us launch (round 4; the stream kernel's
   * checksum went the same way in round 2) */
  static_assert(BLOCK == 256 || BLOCK == 1024, "Horner table of the prebuilt image");
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
  /* (pack.off_out without COPY: the frames lie packed at base + off[i] -- what the render's exact-length forms leave behind) */
  const uint8_t *src = !COPY && pack.off_out ? base + pack.off_out[i] : base + (size_t)i * stride;
  const int full = (int)(L >> 4);                     /* whole 16-byte groups                    */
  const int rounds = (full + BLOCK - 1) / BLOCK;      /* 0 for a frame shorter than 16 bytes      */
  const int lead = rounds * BLOCK - full;             /* zero groups in front of the frame        */

  if (COPY)
    crc_pack_offset_post<BLOCK>(len, fixed_len, n_frames, i, tid);
  const uint32_t lane_k = CRC_LANE_TAB.k[tid & 63], lane_xk = CRC_LANE_TAB.xk[tid & 63]; /* the final reduction's constants */
  for (int k = tid; k < ACHIP_FRAME_CRC_TAB_BYTES / 16; k += BLOCK)
    lds_ptr<uint4>(CrcLds::o_slice)[k] = tab[k];
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  const uint32_t ntail = L & 15u;
  /* the < 16 tail bytes (lane l of wave 0 takes byte l), requested with the first groups */
  const uint32_t tail_byte = (uint32_t)tid < ntail ? src[(size_t)full * 16u + (uint32_t)tid] : 0u;
  __syncthreads();
  /* the header's share of the packet CRC (x^(8 len); the register after {w, h, len, 0}), by the last two waves */
  const bool want_pkt = hdr_out && pkt_crc_out && !bad;
  if (want_pkt) {
    if (wave == BLOCK / 64 - 1) {
      const uint32_t xl = crc_x8_pow_len_wave(lds_ptr<const uint32_t>(CrcLds::o_powtab), L, lane, lane_xk);
      if (lane == 0)
        pw[1] = xl;
    } else if (wave == BLOCK / 64 - 2) {
      const uint32_t hp = crc_header_part_wave(slice, w, h, L, lane);
      if (lane == 0)
        pw[0] = hp;
    }
  }
  uint4 *dst4 = nullptr; /* COPY: where this frame's groups go; stays NULL for a frame that does not fit */
  if (COPY) {
    uint64_t off, total;
    crc_pack_offset_read<BLOCK>(off, total);
    if (tid == 0) {
      if (pack.off_out) {
        pack.off_out[i] = off;
        if (i == n_frames - 1)
          pack.off_out[n_frames] = total;
      }
      if (pack.len_out)
        pack.len_out[i] = len ? len[i] : fixed_len; /* error codes travel as they are */
    }
    if (off + 16ull * ((L + 15u) >> 4) <= pack.capacity) /* whole groups travel: the last one must fit too */
      dst4 = reinterpret_cast<uint4 *>(pack.dst + off);
    /* the frame's last, partial group travels whole, like every other one (the checksum takes its bytes one by one) */
    if (dst4 && (L & 15u) && tid == BLOCK - 1)
      dst4[full] = *reinterpret_cast<const uint4 *>(src + (size_t)full * 16u);
  }

  auto load_group = [&](int j) -> uint4 {
    const int g = j * BLOCK + tid - lead;
    uint4 d = make_uint4(0u, 0u, 0u, 0u);
    if (j < rounds && g >= 0) {
      d = *reinterpret_cast<const uint4 *>(src + (size_t)g * 16u);
      if (COPY && dst4)
        dst4[g] = d;
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
  /* (one multiplication by the lane's constant, one xor reduction per wave, one wave folding the wave registers: the
   * barrier-fenced tree of log2(BLOCK) levels this replaces was half of a small frame's time) */
  const uint32_t whole = crc_reduce_waves<BLOCK>(tree, s, tid, lane_k, lane_xk);
  if (wave == 0) { /* (pw[] was written in front of crc_reduce_waves' barrier) */
    const CrcClose c = crc_close_wave(slice, lds_ptr<const uint32_t>(CrcLds::o_powtab), full > 0 ? whole : 0xFFFFFFFFu, ntail,
                                      tail_byte, want_pkt, pw[0], pw[1], lane, lane_xk);
    if (lane == 0) {
      const uint32_t crc = bad ? 0u : ~c.st;
      crc_out[i] = crc;
      if (hdr_out) {
        crc_store_header(hdr_out, i, w, h, L, crc);
        if (pkt_crc_out)
          pkt_crc_out[i] = bad ? 0u : c.pkt;
      }
    }
  }
}

/* Launch constants of the span checksum's finish, computed on the host (crc_span_pows): c[6] = cspan^64 and lane[t] =
 * cspan^(63 - t), cspan = x^(8 * span bytes).  Squaring and multiplying them up on the device was a chain of bit-serial
 * multiplications in front of everything else the one wave does. */
struct CrcSpanPows {
  uint32_t c[7];     /* cspan^(2^k) */
  uint32_t lane[64]; /* cspan^(63 - t): what register t of a batch of 64 is followed by */
};
__host__ inline CrcSpanPows crc_span_pows(uint64_t span_bytes) {
  CrcSpanPows cp;
  cp.c[0] = crc_pow(CRC_X8, span_bytes);
  for (int k = 1; k < 7; k++)
    cp.c[k] = crc_mulmod(cp.c[k - 1], cp.c[k - 1]);
  cp.lane[63] = crc_mulmod(0x80000000u, 0x80000000u); /* 1 (reflected: bit 31 is x^0) */
  for (int t = 62; t >= 0; t--)
    cp.lane[t] = crc_mulmod(cp.lane[t + 1], cp.c[0]);
  return cp;
}
/* where a span launch leaves its frames' results when it finishes them itself (counters != NULL: one word per frame, zero
 * between launches) */
struct CrcFinish {
  uint32_t *counters;
  CrcSpanPows cp;
  uint32_t xinv_v;
  const uint32_t *dims;
  uint32_t *crc_out;
  uint8_t *hdr_out;
  uint32_t *pkt_crc_out;
};

/* ONE WAVE combines a frame's span registers (all 64 lanes active; slice / powtab: the prebuilt tables in LDS).  Register q of
 * the frame is followed by parts-1-q spans: it is multiplied by its lane's constant, one xor reduction per 64 registers, then
 * the closing multiplications, the header's share and the packet CRC wave-wide, as crc32c_frame_kernel closes a frame -- the
 * one-thread chain of bit-serial multiplications and byte loops this replaces (round 4, session 2) was most of the 26-31 us
 * the span path cost however little it checksummed.  The surplus zero bytes parts*span - len are divided out (xinv_v =
 * x^(-8*parts*span) from the host; x is invertible mod P).  AGENT: the registers were written by other workgroups of THIS
 * launch (loads past the reader's caches). */
template <bool AGENT>
__device__ inline void crc_finish_wave(const uint32_t *slice, const uint32_t *powtab, const uint32_t *__restrict__ partial, int parts,
                                       const CrcSpanPows &cp, uint32_t xinv_v, uint32_t L, bool bad, uint32_t w, uint32_t h, int i,
                                       uint32_t *__restrict__ crc_out, uint8_t *__restrict__ hdr_out,
                                       uint32_t *__restrict__ pkt_crc_out, int lane) {
  const uint32_t lane_xk = CRC_LANE_TAB.xk[lane], lane_pow = cp.lane[lane];
  uint32_t acc = 0;
  const int lead = (64 - parts % 64) % 64; /* zero registers in front keep every batch of 64 full */
  for (int q0 = -lead; q0 < parts; q0 += 64) {
    const int q = q0 + lane;
    uint32_t v = 0u;
    if (q >= 0)
      v = AGENT ? agent_load_u32(&partial[(size_t)i * parts + q]) : partial[(size_t)i * parts + q];
    const uint32_t batch = wave_xor_all(crc_mulmod(v, lane_pow));
    acc = (q0 > -lead ? wave_mulmod_uniform(acc, cp.c[6], lane, lane_xk) : 0u) ^ batch;
  }
  const uint32_t xl = crc_x8_pow_len_wave(powtab, L, lane, lane_xk);
  const uint32_t raw = wave_mulmod_uniform(acc, wave_mulmod_uniform(xinv_v, xl, lane, lane_xk), lane, lane_xk); /* raw() of exactly len bytes */
  const uint32_t st = wave_mulmod_uniform(0xFFFFFFFFu, xl, lane, lane_xk) ^ raw; /* register after the frame from 0xFFFFFFFF */
  const uint32_t crc = bad ? 0u : ~st;
  uint32_t pkt = 0u;
  if (hdr_out && pkt_crc_out) {
    const uint32_t hpart = crc_header_part_wave(slice, w, h, L, lane);
    pkt = crc_close_wave(slice, powtab, bad ? 0xFFFFFFFFu : st, 0u, 0u, true, hpart, xl, lane, lane_xk).pkt;
  }
  if (lane == 0) {
    crc_out[i] = crc;
    if (hdr_out) {
      crc_store_header(hdr_out, i, w, h, L, crc);
      if (pkt_crc_out)
        pkt_crc_out[i] = bad ? 0u : pkt;
    }
  }
}

/*
 * Large buffers: grid = n_frames * parts workgroups of 256 threads; workgroup (i, p) reduces the fixed span
 * [p*span, (p+1)*span) of frame i, span = rounds * 4 KB, parts*span >= every length (zeros behind the end of
 * the frame), and stores its raw register (initial value 0) in partial[i*parts + p].
 */
template <bool COPY = false>
__global__ void __launch_bounds__(256)
    crc32c_span_kernel(const uint8_t *__restrict__ base, uint64_t stride, const uint32_t *__restrict__ len,
                       uint32_t fixed_len, int n_frames, int parts, int rounds, uint32_t *__restrict__ partial,
                       const uint4 *__restrict__ tab, CrcFinish fin, CrcPack pack = CrcPack{nullptr, 0, nullptr, nullptr}) {
  /* fin.counters != NULL: ONE launch -- every workgroup reports its register and arrives at its frame's counter; the last one
   * to arrive finishes the frame (crc_finish_wave) and leaves the counter at zero for the next launch.  The second launch
   * (crc32c_finish_kernel) this saves is worth ~2 us of a small call (render + wire stage of a lone 320x90 truecolor frame: 18.6 -> 16.7 us);
   * the launcher asks for it only while the call has at most 128 spans -- an arrival is an L2 write-back + invalidate. */
  /* tab: the image of crc_frame_tables_init_kernel<256>; its first 20 KB (slicing tables + the Horner table of 256 threads)
   * are copied into LDS -- building them here was a chain of fifteen dependent LDS round trips and four bit-serial
   * multiplications in front of every workgroup's first load */
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
  const bool bad = (len ? len[i] : fixed_len) >= 0xFFFFFFF0u;
  const uint64_t lo = (uint64_t)p * (uint64_t)rounds * (16u * BLOCK);
  const bool empty = lo >= L && !(COPY && p == 0); /* nothing but zeros: raw() of zeros from 0 is 0 */
  if (empty && !fin.counters) {
    if (tid == 0)
      partial[(size_t)i * parts + p] = 0u;
    return;
  }
  /* (one-launch form: a span behind the frame's end still arrives -- it may be the one that finishes the frame -- and takes
   * the tables along for that) */
  const uint64_t avail = lo < L ? (uint64_t)L - lo : 0u; /* bytes of the frame from the start of this span (may exceed the span) */
  if (COPY && !empty)
    crc_pack_offset_post<BLOCK>(len, fixed_len, n_frames, i, tid);
  static_assert(CrcLds::o_slice == 0 && CrcLds::o_mulh == 16 * 1024 && CrcLds::o_powtab == 20 * 1024, "the image's first 20 (22) KB");
  (void)slice, (void)mulh;
  for (int k = tid; k < (fin.counters ? ACHIP_FRAME_CRC_TAB_BYTES : 20 * 1024) / 16; k += BLOCK) /* (the finish wants the power tables too) */
    lds_ptr<uint4>(CrcLds::o_slice)[k] = tab[k];
  __syncthreads();
  uint8_t *dstb = nullptr; /* COPY: where this span's groups go; stays NULL for a frame that does not fit */
  if (COPY && !empty) {
    uint64_t off, total;
    crc_pack_offset_read<BLOCK>(off, total);
    if (p == 0 && tid == 0) {
      if (pack.off_out) {
        pack.off_out[i] = off;
        if (i == n_frames - 1)
          pack.off_out[n_frames] = total;
      }
      if (pack.len_out)
        pack.len_out[i] = len ? len[i] : fixed_len;
    }
    if (off + 16ull * (((uint64_t)L + 15u) >> 4) <= pack.capacity) /* whole groups travel: the last one must fit too */
      dstb = pack.dst + off + lo;
  }
  const uint8_t *src = (!COPY && pack.off_out ? base + pack.off_out[i] : base + (size_t)i * stride) + lo;
  auto load_group = [&](int j) -> uint4 {
    const uint64_t off = ((uint64_t)j * BLOCK + (uint64_t)tid) * 16u;
    uint4 d = make_uint4(0u, 0u, 0u, 0u);
    if (j < rounds && off < avail) {
      const uint64_t left = avail - off;
      if (COPY && dstb) /* whole groups, the frame's last one included (padding the layout allows) */
        *reinterpret_cast<uint4 *>(dstb + off) = *reinterpret_cast<const uint4 *>(src + off);
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
  for (int j0 = 0; j0 < (empty ? 0 : rounds); j0 += 4) {
    uint4 d[4];
#pragma unroll
    for (int u = 0; u < 4; u++)
      d[u] = load_group(j0 + u);
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (j0 + u < rounds)
        s = crc_mul_table(mulh, s) ^ crc_raw16(slice, d[u]);
  }
  /* (one multiplication by the lane's constant, one xor reduction per wave, one wave folding the four wave registers, as the
   * frame kernel: the barrier-fenced tree of eight bit-serial levels this replaces was ~5 us of every span) */
  const uint32_t whole = empty ? 0u : crc_reduce_waves<BLOCK>(tree, s, tid, CRC_LANE_TAB.k[tid & 63], CRC_LANE_TAB.xk[tid & 63]);
  if (!fin.counters) {
    if (tid == 0)
      partial[(size_t)i * parts + p] = whole;
    return;
  }
  uint32_t *last_flag = tree + BLOCK / 64; /* (behind crc_reduce_waves' wave registers) */
  if (tid == 0) {
    agent_store_u32(&partial[(size_t)i * parts + p], whole);
    const bool last = agent_arrive_u32(&fin.counters[i]) == (uint32_t)parts - 1u; /* release + acquire, agent scope */
    if (last)
      agent_store_u32(&fin.counters[i], 0u); /* re-armed: the plan's next launch follows in stream order */
    *last_flag = last ? 1u : 0u;
  }
  __syncthreads();
  if (*last_flag != 0u && tid < 64) {
    const uint32_t w = fin.dims && !bad ? fin.dims[2 * i] : 0u, h = fin.dims && !bad ? fin.dims[2 * i + 1] : 0u;
    crc_finish_wave<true>(slice, lds_ptr<const uint32_t>(CrcLds::o_powtab), partial, parts, fin.cp, fin.xinv_v, L, bad, w, h, i, fin.crc_out,
                          fin.hdr_out, fin.pkt_crc_out, tid);
  }
}

/* One WAVE per frame combines the span registers of a two-launch call (crc_finish_wave).  tab: the image of
 * crc_frame_tables_init_kernel<256>. */
__global__ void __launch_bounds__(64)
    crc32c_finish_kernel(const uint32_t *__restrict__ partial, int parts, CrcSpanPows cp, uint32_t xinv_v,
                         const uint32_t *__restrict__ len, uint32_t fixed_len, int n_frames,
                         const uint32_t *__restrict__ dims, uint32_t *__restrict__ crc_out, uint8_t *__restrict__ hdr_out,
                         uint32_t *__restrict__ pkt_crc_out, const uint4 *__restrict__ tab) {
  const int i = (int)blockIdx.x, lane = (int)threadIdx.x;
  if (i >= n_frames)
    return;
  for (int k = lane; k < ACHIP_FRAME_CRC_TAB_BYTES / 16; k += 64)
    lds_ptr<uint4>(CrcLds::o_slice)[k] = tab[k];
  uint32_t L = len ? len[i] : fixed_len;
  const bool bad = L >= 0xFFFFFFF0u;
  if (bad)
    L = 0;
  const uint32_t w = dims && !bad ? dims[2 * i] : 0u, h = dims && !bad ? dims[2 * i + 1] : 0u;
  __syncthreads(); /* the tables */
  crc_finish_wave<false>(lds_ptr<const uint32_t>(CrcLds::o_slice), lds_ptr<const uint32_t>(CrcLds::o_powtab), partial, parts, cp, xinv_v, L, bad,
                         w, h, i, crc_out, hdr_out, pkt_crc_out, lane);
}


/* headers + packet CRCs from frame CRCs that are already known (the fused render): one thread per frame */
__global__ void __launch_bounds__(256)
    crc_packets_kernel(const uint32_t *__restrict__ len, const uint32_t *__restrict__ crc_in, const uint32_t *__restrict__ dims,
                       int n, uint8_t *__restrict__ hdr_out, uint32_t *__restrict__ pkt_crc_out) {
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  if (i >= n)
    return;
  uint32_t L = len[i];
  const bool bad = L >= 0xFFFFFFF0u;
  if (bad)
    L = 0;
  const uint32_t w = dims && !bad ? dims[2 * i] : 0u, h = dims && !bad ? dims[2 * i + 1] : 0u;
  const uint32_t crc = bad ? 0u : crc_in[i];
  /* ~crc is the CRC register after the frame (from 0xFFFFFFFF): what crc32c_frame_kernel hands to crc_emit_packet */
  crc_emit_packet(crc_header_state16(w, h, L), crc_x8_pow_len(L), ~crc, crc, w, h, L, bad, i, nullptr, hdr_out, pkt_crc_out);
}

} // namespace achip
