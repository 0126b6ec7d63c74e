/*
 * gfx950_ops.hpp -- every piece of the kernels that is written FOR THE MACHINE rather than in portable HIP C++: inline
 * DS instructions, DPP wave operations, scoped atomics, non-temporal memory operations, clocks, occupancy attributes.
 * The kernels (render_kernels.hpp, render_stream.hpp, ...) include it as <gfx950_ops.hpp> and contain no conditional
 * compilation of their own.  The CPU test suite puts tests/hipemu/ in front of this directory on the include path and
 * so compiles the same kernel sources against tests/hipemu/gfx950_ops.hpp, a fiber emulation of exactly this interface;
 * the product library is built by hipcc from this file only.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

/* The one dynamic-LDS block of a kernel.  Every LDS access is derived from this symbol (never from a pointer stored in
 * a struct), so the compiler keeps the accesses in the LDS address space: ds_read/ds_write, not flat_load/flat_store. */
extern __shared__ __attribute__((aligned(16))) unsigned char achip_smem[];
#define ACHIP_SMEM achip_smem

/* pointers read out of descriptors are generic; tell the compiler they are global memory so that it
 * emits global_load (vmcnt only) rather than flat_load (vmcnt + lgkmcnt, shared with the LDS queue) */
#define ACHIP_GLOBAL __attribute__((address_space(1)))

#define ACHIP_EMULATED 0                 /* 1 in the emulator's twin of this file: tests may widen a configuration */
#define ACHIP_DEVICE_ONLY(...) __VA_ARGS__ /* statements with no meaning off the device (register-class asm, waitcnt) */
#define ACHIP_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))

namespace achip {

/* one LDS byte store at (LDS byte address `addr`) + OFF; HI selects bits 23..16 of `v` instead of 7..0.
 * Written as asm so that neighbouring byte stores are never fused into a misaligned wide store. */
template <int OFF, bool HI> __device__ inline void ds_store_byte(uint32_t addr, uint32_t v) {
  if (HI)
    asm volatile("ds_write_b8_d16_hi %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
  else
    asm volatile("ds_write_b8 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
/* LDS atomic OR of an aligned dword, no return value */
__device__ inline void ds_or_u32(uint32_t addr, uint32_t v) {
  asm volatile("ds_or_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
/* ... at (addr) + OFF, the offset as the instruction's immediate */
template <int OFF> __device__ inline void ds_or_u32_at(uint32_t addr, uint32_t v) {
  asm volatile("ds_or_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
/* v_alignbit_b32: the low dword of {hi:lo} >> (sh & 31) */
__device__ inline uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
/* v_dot4_u32_u8: a.b0*b.b0 + a.b1*b.b1 + a.b2*b.b2 + a.b3*b.b3 + c -- the BT.601 luminance of a packed pixel is ONE
 * instruction (a zero coefficient also discards whatever byte 3 holds) */
__device__ inline uint32_t dot4_u8(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
/* v_bfe_u32: `width` bits of v from bit `off` (both per lane) */
__device__ inline uint32_t bfe_u32(uint32_t v, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(v, off, width); }
/* v_mad_i32_i24 / v_mul_u32_u24 / v_mad_u32_u24: full-rate multiplies of 24-bit operands (v_mul_lo_u32 runs at a quarter) */
__device__ inline int32_t mad_i24(int32_t a, int32_t b, int32_t c) { return __mul24(a, b) + c; }
__device__ inline uint32_t mul_u24(uint32_t a, uint32_t b) { return __umul24(a, b); }
/* keeps operands alive without issuing anything (ablation builds) */
__device__ inline void keep_alive(uint32_t a, uint32_t b) { asm volatile("" ::"v"(a), "v"(b)); }
/* LDS byte address of ACHIP_SMEM[0] (0 for a kernel without static LDS, but do not assume) */
__device__ inline uint32_t lds_base_addr() {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(achip_smem);
}
/* all DS operations issued by inline asm must have landed before they are read back */
__device__ inline void lds_store_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
/* The lanes of a wave execute in lockstep and a wave's DS operations complete in order: what every lane read from LDS
 * above this point was read before anything below it is stored.  Nothing to emit on the machine; the emulator's twin
 * lines its fibers up here. */
__device__ inline void wave_lockstep() {}
/* every outstanding vector-memory load of this wave has returned (diagnostics: a point in time for a stamp) */
__device__ inline void wait_vmem_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

/* ---- words shared between workgroups (agent scope: relaxed loads bypass the reader's L1) and between the waves of
 * one workgroup (LDS words, workgroup scope) ----------------------------------------------------------------------- */
__device__ inline void agent_store_u64(unsigned long long *p, unsigned long long w) {
  __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline unsigned long long agent_load_u64(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline unsigned long long agent_fetch_add_u64(unsigned long long *p, unsigned long long v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* a hand-off through memory between workgroups of one launch (the span checksum's last arriver): the data word goes out
 * relaxed, the arrival counter is bumped with release + acquire at agent scope -- every arrival's data is visible to
 * whoever sees its count -- and the collector reads the data words past its own L1 / L2 */
__device__ inline void agent_store_u32(uint32_t *p, uint32_t w) { __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint32_t agent_load_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint32_t agent_arrive_u32(uint32_t *p) { return __hip_atomic_fetch_add(p, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT); }
template <int N> __device__ inline void spin_nap() { __builtin_amdgcn_s_sleep(N); }
__device__ inline void wg_store_u32(uint32_t *p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ inline uint32_t wg_load_u32(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ inline uint32_t wg_fetch_add_u32(uint32_t *p, uint32_t v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ inline void wg_xor_u32(uint32_t *p, uint32_t v) {
  (void)__hip_atomic_fetch_xor(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

/* ---- wave64 ------------------------------------------------------------------------------------------------------ */
__device__ inline uint64_t wave_ballot(bool p) { return __ballot(p); }
/* the lane's own bit of a wave-uniform mask, as a predicate: the mask itself becomes the instruction's lane mask (no shift
 * by the lane number, no vector instruction at all) */
__device__ inline bool lane_bit(uint64_t wave_uniform_mask) { return __builtin_amdgcn_inverse_ballot_w64(wave_uniform_mask); }
__device__ inline uint32_t wave_shfl_up(uint32_t v, int d) { return __shfl_up(v, d, 64); }
/* lane `src`'s value (a per-lane index: ds_bpermute_b32, an LDS crossbar round trip) */
__device__ inline uint32_t wave_shfl(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
__device__ inline uint32_t wave_read_lane(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ inline int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
/* inclusive scan / xor reduction without LDS: DPP row shifts + row broadcasts (the ds_bpermute that __shfl_up compiles
 * to costs an LDS round trip per step).  Lanes whose DPP source is invalid (or whose row is masked off) take 0. */
template <int CTRL, int ROW_MASK> __device__ inline uint32_t dpp_add(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK> __device__ inline uint32_t dpp_xor(uint32_t v) {
  return v ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ inline uint32_t wave_inclusive_scan(uint32_t v) {
  v = dpp_add<0x111, 0xF>(v); /* row_shr:1  */
  v = dpp_add<0x112, 0xF>(v); /* row_shr:2  */
  v = dpp_add<0x114, 0xF>(v); /* row_shr:4  */
  v = dpp_add<0x118, 0xF>(v); /* row_shr:8  : every 16-lane row now holds its own inclusive scan */
  v = dpp_add<0x142, 0xA>(v); /* row_bcast:15 into rows 1 and 3 */
  v = dpp_add<0x143, 0xC>(v); /* row_bcast:31 into rows 2 and 3 */
  return v;
}
/* xor over the wave; the result is valid in lane 63 */
__device__ inline uint32_t wave_xor_to_last(uint32_t v) {
  v = dpp_xor<0x111, 0xF>(v);
  v = dpp_xor<0x112, 0xF>(v);
  v = dpp_xor<0x114, 0xF>(v);
  v = dpp_xor<0x118, 0xF>(v);
  v = dpp_xor<0x142, 0xA>(v);
  v = dpp_xor<0x143, 0xC>(v);
  return v;
}
/* lane l receives lane l-1's value; lane 0 receives `first` (one DPP move, no LDS round trip) */
__device__ inline uint32_t wave_shift_up1(uint32_t v, uint32_t first) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ inline uint32_t bitreverse32(uint32_t v) { return __builtin_bitreverse32(v); }

/* ---- memory operations with a cache policy ------------------------------------------------------------------------ */
struct __attribute__((packed)) unaligned_u32 {
  uint32_t v;
};
/* 4 bytes at any address, non-temporal: samples a cache line apart or more share no line with their neighbours and are
 * kept out of the L2.  `distinct` keeps the load from being merged with a cached twin (it would lose its hint). */
__device__ inline uint32_t load_u32_unaligned_nt(const uint8_t *p) {
  typedef uint32_t u32_unaligned __attribute__((aligned(1)));
  return __builtin_nontemporal_load((const ACHIP_GLOBAL u32_unaligned *)p);
}
__device__ inline uint32_t opaque(uint32_t v) {
  asm volatile("" : "+v"(v));
  return v;
}
/* 16 output bytes to HBM.  The stream is written once and never read back by the kernel: a non-temporal store lets the
 * lines leave the L2 during the kernel instead of in the write-back at its end. */
__device__ inline void store_u4_nt(uint8_t *__restrict__ p, uint4 v) {
#ifdef ACHIP_NO_NT_STORE /* diagnostics build */
  *reinterpret_cast<uint4 *>(p) = v;
#else
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(p));
#endif
}

/* 16 / 8 / 4 / 2 bytes at ANY byte address of global memory (the hardware takes unaligned vector accesses; a piece that
 * straddles a line becomes two requests).  Plain stores: the pieces of a line meet in the L2 and leave it as a line. */
__device__ inline void store_u4_unaligned(uint8_t *p, uint4 v) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  struct __attribute__((packed)) U {
    u32x4 v;
  };
  u32x4 w = {v.x, v.y, v.z, v.w};
  reinterpret_cast<U *>(p)->v = w;
}
__device__ inline void store_u2_unaligned(uint8_t *p, uint32_t a, uint32_t b) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  struct __attribute__((packed)) U {
    u32x2 v;
  };
  u32x2 w = {a, b};
  reinterpret_cast<U *>(p)->v = w;
}
__device__ inline void store_u1_unaligned(uint8_t *p, uint32_t a) {
  struct __attribute__((packed)) U {
    uint32_t v;
  };
  reinterpret_cast<U *>(p)->v = a;
}
__device__ inline void store_u16_unaligned(uint8_t *p, uint16_t a) {
  struct __attribute__((packed)) U {
    uint16_t v;
  };
  reinterpret_cast<U *>(p)->v = a;
}

/* ---- clocks (diagnostics) ------------------------------------------------------------------------------------------ */
__device__ inline unsigned long long cycle_now() { return (unsigned long long)clock64(); }
__device__ inline unsigned long long wall_now() { return (unsigned long long)wall_clock64(); } /* 100 MHz */

} // namespace achip
