/*
 * gfx950_ops.hpp (emulator twin) -- the interface of ascii-chat_amd/csrc/gfx950_ops.hpp on top of hip_emu.h's fibers, so
 * that the kernel sources compile unchanged under g++ and run on the CPU test box.  TESTS ONLY: found ahead of the
 * product header because the emulator build lists this directory first on the include path.
 */
#pragma once
#include <string.h>

#include "hip_emu.h"

#define ACHIP_SMEM (hipemu::g_smem.data())
#define ACHIP_GLOBAL
#define ACHIP_EMULATED 1
#define ACHIP_DEVICE_ONLY(...)
#define ACHIP_WAVES_PER_EU(n)

namespace achip {

template <int OFF, bool HI> inline void ds_store_byte(uint32_t addr, uint32_t v) {
  ACHIP_SMEM[addr + OFF] = (unsigned char)(HI ? v >> 16 : v);
}
inline void ds_or_u32(uint32_t addr, uint32_t v) { *reinterpret_cast<uint32_t *>(ACHIP_SMEM + addr) |= v; }
template <int OFF> inline void ds_or_u32_at(uint32_t addr, uint32_t v) { *reinterpret_cast<uint32_t *>(ACHIP_SMEM + addr + OFF) |= v; }
inline uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31u)); }
inline uint32_t dot4_u8(uint32_t a, uint32_t b, uint32_t c) {
  return (a & 0xFFu) * (b & 0xFFu) + ((a >> 8) & 0xFFu) * ((b >> 8) & 0xFFu) + ((a >> 16) & 0xFFu) * ((b >> 16) & 0xFFu) +
         (a >> 24) * (b >> 24) + c;
}
inline uint32_t bfe_u32(uint32_t v, uint32_t off, uint32_t width) { /* v_bfe_u32: offset and width modulo 32 */
  off &= 31u;
  width &= 31u;
  return width ? (v >> off) & ((1u << width) - 1u) : 0u;
}
inline int32_t mad_i24(int32_t a, int32_t b, int32_t c) { /* operands sign-extended from 24 bits, low 32 bits of the product */
  const int64_t sa = ((int32_t)((uint32_t)a << 8)) >> 8, sb = ((int32_t)((uint32_t)b << 8)) >> 8;
  return (int32_t)((uint32_t)(sa * sb) + (uint32_t)c);
}
inline uint32_t mul_u24(uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu)); }
inline void keep_alive(uint32_t, uint32_t) {}
inline uint32_t lds_base_addr() { return 0u; }
/* a wave's lanes run in lockstep on the GPU: every lane's stores precede the reads behind the fence */
inline void lds_store_fence() { hipemu::wave_barrier(); }
inline void wave_lockstep() { hipemu::wave_barrier(); }
inline void wait_vmem_all() {}

/* fibers of one OS thread: no switch inside a plain read-modify-write */
inline void agent_store_u64(unsigned long long *p, unsigned long long w) { *reinterpret_cast<volatile unsigned long long *>(p) = w; }
inline unsigned long long agent_load_u64(const unsigned long long *p) {
  return *reinterpret_cast<const volatile unsigned long long *>(p);
}
inline unsigned long long agent_fetch_add_u64(unsigned long long *p, unsigned long long v) {
  const unsigned long long old = *reinterpret_cast<volatile unsigned long long *>(p);
  *reinterpret_cast<volatile unsigned long long *>(p) = old + v;
  return old;
}
inline void agent_store_u32(uint32_t *p, uint32_t w) { *reinterpret_cast<volatile uint32_t *>(p) = w; }
inline uint32_t agent_load_u32(const uint32_t *p) { return *reinterpret_cast<const volatile uint32_t *>(p); }
inline uint32_t agent_arrive_u32(uint32_t *p) {
  const uint32_t old = *reinterpret_cast<volatile uint32_t *>(p);
  *reinterpret_cast<volatile uint32_t *>(p) = old + 1u;
  return old;
}
template <int N> inline void spin_nap() {}
inline void wg_store_u32(uint32_t *p, uint32_t v) { *reinterpret_cast<volatile uint32_t *>(p) = v; }
inline uint32_t wg_load_u32(const uint32_t *p) { return *reinterpret_cast<const volatile uint32_t *>(p); }
inline uint32_t wg_fetch_add_u32(uint32_t *p, uint32_t v) {
  const uint32_t old = *reinterpret_cast<volatile uint32_t *>(p);
  *reinterpret_cast<volatile uint32_t *>(p) = old + v;
  return old;
}
inline void wg_xor_u32(uint32_t *p, uint32_t v) {
  *reinterpret_cast<volatile uint32_t *>(p) = *reinterpret_cast<volatile uint32_t *>(p) ^ v;
}

inline uint64_t wave_ballot(bool p) { return hipemu::ballot(p); }
inline bool lane_bit(uint64_t wave_uniform_mask) { return ((wave_uniform_mask >> hipemu::lane()) & 1ull) != 0ull; }
inline uint32_t wave_shfl_up(uint32_t v, int d) {
  return hipemu::shfl_from(v, hipemu::lane() - d); /* src < 0 -> own value, like __shfl_up */
}
inline uint32_t wave_read_lane(uint32_t v, int lane) { return hipemu::shfl_from(v, lane); }
inline uint32_t wave_shfl(uint32_t v, int src) { return hipemu::shfl_from(v, src & 63); }
inline int wave_uniform(int v) { return v; }
inline uint32_t wave_inclusive_scan(uint32_t v) {
  const int l = hipemu::lane();
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = hipemu::shfl_from(v, l - d);
    if (l >= d)
      v += t;
  }
  return v;
}
inline uint32_t wave_xor_to_last(uint32_t v) {
  const int l = hipemu::lane();
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = hipemu::shfl_from(v, l - d);
    if (l >= d)
      v ^= t;
  }
  return v;
}
inline uint32_t wave_shift_up1(uint32_t v, uint32_t first) {
  const int l = hipemu::lane();
  const uint32_t t = hipemu::shfl_from(v, l - 1);
  return l == 0 ? first : t;
}
inline uint32_t bitreverse32(uint32_t v) {
  uint32_t r = 0;
  for (int i = 0; i < 32; i++)
    r |= ((v >> i) & 1u) << (31 - i);
  return r;
}

struct __attribute__((packed)) unaligned_u32 {
  uint32_t v;
};
inline uint32_t load_u32_unaligned_nt(const uint8_t *p) { return reinterpret_cast<const unaligned_u32 *>(p)->v; }
inline uint32_t opaque(uint32_t v) { return v; }
inline void store_u4_nt(uint8_t *p, uint4 v) { *reinterpret_cast<uint4 *>(p) = v; }
inline void store_u4_unaligned(uint8_t *p, uint4 v) { memcpy(p, &v, 16); }
inline void store_u2_unaligned(uint8_t *p, uint32_t a, uint32_t b) {
  memcpy(p, &a, 4);
  memcpy(p + 4, &b, 4);
}
inline void store_u1_unaligned(uint8_t *p, uint32_t a) { memcpy(p, &a, 4); }
inline void store_u16_unaligned(uint8_t *p, uint16_t a) { memcpy(p, &a, 2); }

inline unsigned long long cycle_now() { return 0ull; }
inline unsigned long long wall_now() { return 0ull; }

} // namespace achip
