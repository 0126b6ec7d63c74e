/*
 * hip_emu.h -- a tiny single-threaded fiber emulator of the HIP execution model, for TESTS ONLY.
 *
 * Purpose: the render kernels (ascii-chat_amd/csrc/render_kernels.hpp) are plain HIP C++.  Compiled
 * with -DACHIP_HIPEMU under g++ they run here, one workgroup at a time, every work-item a ucontext
 * fiber, so that block barriers, wave64 ballots/shuffles and LDS indexing are exercised bit-exactly
 * against the oracle in the CPU test suite -- GPU minutes are scarce, logic bugs should die here.
 *
 * This is not a fallback and is never linked into the product library: libasciichat_hip.so contains
 * only hipcc-compiled gfx950 code and fails loudly without a GPU.
 */
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 {
  uint32_t x, y, z, w;
};
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct uint2 {
  uint32_t x, y;
};
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }

namespace hipemu {

struct idx3 {
  unsigned x, y, z;
};
inline idx3 g_threadIdx, g_blockIdx;
inline dim3 g_blockDim, g_gridDim;
inline std::vector<unsigned char> g_smem;
inline size_t g_smem_limit = 160 * 1024;

enum { RUN = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
struct Fiber {
  ucontext_t ctx;
  void *stack = nullptr;
  int state = RUN;
};
inline std::vector<Fiber> g_fibers;
inline ucontext_t g_sched;
inline int g_cur = -1;
inline std::function<void()> g_body;
inline uint32_t g_wave_slots[64][64]; /* [wave][lane] exchange slots (block <= 4096 threads) */

inline void yield_as(int state) {
  g_fibers[g_cur].state = state;
  swapcontext(&g_fibers[g_cur].ctx, &g_sched);
}
inline void trampoline() {
  g_body();
  yield_as(DONE);
}
inline void block_barrier() { yield_as(WAIT_BLOCK); }
inline void wave_barrier() { yield_as(WAIT_WAVE); }

inline int lane() { return (int)(g_threadIdx.x & 63u); }
inline int wave() { return (int)(g_threadIdx.x >> 6); }

inline uint64_t ballot(int pred) {
  const int w = wave(), l = lane();
  g_wave_slots[w][l] = pred ? 1u : 0u;
  wave_barrier();
  uint64_t m = 0;
  const int nl = std::min<int>(64, (int)g_blockDim.x - w * 64);
  for (int i = 0; i < nl; i++)
    m |= (uint64_t)g_wave_slots[w][i] << i;
  wave_barrier();
  return m;
}
inline uint32_t shfl_from(uint32_t v, int src) {
  const int w = wave(), l = lane();
  g_wave_slots[w][l] = v;
  wave_barrier();
  uint32_t r = (src >= 0 && src < 64) ? g_wave_slots[w][src] : v;
  wave_barrier();
  return r;
}

template <class F> void launch(dim3 grid, dim3 block, size_t smem_bytes, F body) {
  if (smem_bytes > g_smem_limit) {
    fprintf(stderr, "hipemu: dynamic LDS request %zu exceeds %zu\n", smem_bytes, g_smem_limit);
    abort();
  }
  assert(block.y == 1 && block.z == 1 && block.x % 64 == 0 && block.x <= 4096);
  const size_t stack_bytes = 256 * 1024;
  g_blockDim = block;
  g_gridDim = grid;
  g_smem.assign(smem_bytes + 64, 0xCD); /* poison: kernels must initialise what they read */
  g_fibers.resize(block.x);
  for (auto &f : g_fibers)
    if (!f.stack)
      f.stack = malloc(stack_bytes);
  g_body = body;
  for (unsigned bb = 0; bb < grid.x * grid.y * grid.z; bb++) { /* workgroups one at a time, x fastest */
    const unsigned b = bb % grid.x;
    g_blockIdx = idx3{b, (bb / grid.x) % grid.y, bb / (grid.x * grid.y)};
    std::fill(g_smem.begin(), g_smem.end(), 0xCD);
    for (unsigned t = 0; t < block.x; t++) {
      Fiber &f = g_fibers[t];
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = f.stack;
      f.ctx.uc_stack.ss_size = stack_bytes;
      f.ctx.uc_link = nullptr;
      makecontext(&f.ctx, (void (*)())trampoline, 0);
      f.state = RUN;
    }
    for (;;) {
      bool ran = false;
      for (unsigned t = 0; t < block.x; t++) {
        if (g_fibers[t].state != RUN)
          continue;
        g_cur = (int)t;
        g_threadIdx = idx3{t, 0, 0};
        swapcontext(&g_sched, &g_fibers[t].ctx);
        ran = true;
      }
      /* release wave barriers */
      unsigned done = 0, at_block = 0;
      for (unsigned w0 = 0; w0 < block.x; w0 += 64) {
        unsigned nwait = 0, ndone = 0, n = std::min(64u, block.x - w0);
        for (unsigned t = w0; t < w0 + n; t++) {
          nwait += g_fibers[t].state == WAIT_WAVE;
          ndone += g_fibers[t].state == DONE;
        }
        if (nwait && nwait + ndone == n) {
          if (ndone) {
            fprintf(stderr, "hipemu: wave op with exited lanes (wave %u)\n", w0 / 64);
            abort();
          }
          for (unsigned t = w0; t < w0 + n; t++)
            g_fibers[t].state = RUN;
          ran = true;
        }
      }
      for (unsigned t = 0; t < block.x; t++) {
        done += g_fibers[t].state == DONE;
        at_block += g_fibers[t].state == WAIT_BLOCK;
      }
      if (done == block.x)
        break;
      if (at_block && at_block + done == block.x) {
        if (done) {
          fprintf(stderr, "hipemu: __syncthreads with exited threads\n");
          abort();
        }
        for (auto &f : g_fibers)
          f.state = RUN;
        ran = true;
      }
      if (!ran) {
        fprintf(stderr, "hipemu: deadlock (divergent barrier?) block %u\n", b);
        abort();
      }
    }
  }
}

} // namespace hipemu

#define threadIdx hipemu::g_threadIdx
#define blockIdx hipemu::g_blockIdx
#define blockDim hipemu::g_blockDim
#define gridDim hipemu::g_gridDim

static inline void __syncthreads() { hipemu::block_barrier(); }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline uint32_t __umul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); } /* low 32 bits */
static inline int __popcll(uint64_t v) { return __builtin_popcountll(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll(v) : 64; }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
using std::max;
using std::min;
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { /* fibers run one at a time */
  const uint32_t old = *p;
  *p = old + v;
  return old;
}
