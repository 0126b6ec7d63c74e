/*
 * render_stream_inst.hip -- instantiates the stream kernel (render_stream.hpp) for ONE geometry
 * (-DACHIP_SINST=<variant id>): four per-cell modes x {plain, composite sampler}.  One translation unit per
 * geometry so that the build runs in parallel.  Built only with hipcc --offload-arch=gfx950.
 */
#include <hip/hip_runtime.h>

#include <mutex>

#include "render_inst.h"
#define ACHIP_FRAME_KERNEL_ONLY
#include "render_stream.hpp"
#include "render_variants.h"

#ifndef ACHIP_SINST
#error "compile with -DACHIP_SINST=<stream variant id>"
#endif

namespace {

template <int ID> struct SGeometry;
#define X(id, W, C)                                                                                                    \
  template <> struct SGeometry<id> {                                                                                   \
    static constexpr int WAVES = W, CPL = C;                                                                           \
  };
ACHIP_STREAM_VARIANTS(X)
#undef X
using G = SGeometry<ACHIP_SINST>;

/* the frame CRC rides the drain (CRC = true) in the two geometries the policy picks by itself */
constexpr bool HAS_CRC = ACHIP_SINST == 16 || ACHIP_SINST == 17;
/* ... and they carry the multi-byte-palette form of truecolor foreground */
constexpr bool HAS_U8 = ACHIP_SINST == 16 || ACHIP_SINST == 17 || ACHIP_SINST == 20;

/* the constant tables of <MODE>'s CRC instantiation: built on the device once per process, then read-only */
template <int MODE> hipError_t crc_tables(const uint4 **out) {
  using L = achip::SLds<MODE, G::WAVES, G::CPL, true>;
  constexpr int MAX_DEVICES = 16;
  static std::mutex mu;
  static uint32_t *tab[MAX_DEVICES] = {}; /* one image per device of the process */
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess)
    return e;
  if (dev < 0 || dev >= MAX_DEVICES)
    return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(mu);
  if (!tab[dev]) {
    uint32_t *t = nullptr;
    e = hipMalloc(reinterpret_cast<void **>(&t), (size_t)L::TAB_BYTES);
    if (e != hipSuccess)
      return e;
    hipLaunchKernelGGL((achip::crc_tables_init_kernel<L>), dim3(1), dim3(256), 0, nullptr, t);
    e = hipGetLastError();
    if (e == hipSuccess)
      e = hipDeviceSynchronize(); /* launches on every stream may read it from here on */
    if (e != hipSuccess) {
      (void)hipFree(t);
      return e;
    }
    tab[dev] = t;
  }
  *out = reinterpret_cast<const uint4 *>(tab[dev]);
  return hipSuccess;
}

template <int MODE, bool COMP, bool CRC>
hipError_t launch_one(const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out, uint64_t stride,
                      uint32_t *len, const achip_uniform_t &uni, unsigned long long *prof, const achip_wire_t &wire,
                      hipStream_t stream) {
  using L = achip::SLds<MODE, G::WAVES, G::CPL, CRC>;
  auto kern = achip::render_stream_kernel<MODE, G::WAVES, G::CPL, COMP, CRC>;
  static bool attr_set = false; /* one flag per instantiation; benign race (idempotent call) */
  if (!attr_set) {
    if (L::bytes > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, L::bytes);
      if (e != hipSuccess)
        return e;
    }
    attr_set = true;
  }
  const uint4 *tab = nullptr;
  if constexpr (CRC) {
    hipError_t e = crc_tables<MODE>(&tab);
    if (e != hipSuccess)
      return e;
  }
  /* the per-block words are sized by the launch's largest frame when the host states it: a small footprint lets
   * workgroups of launches in flight on other streams share a CU */
  const size_t lds = (size_t)((L::bytes_for(achip::stream_maxblk(uni.flags, L::EFF)) + 15) & ~15);
  hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(G::WAVES * 64), lds, stream, frames, lut, out, stride, len, n, uni,
                     prof, wire, tab, achip_packdev_t{}, achip_partsdev_t{});
  return hipGetLastError();
}

/* PARTS instantiations (a frame's blocks shared out over several workgroups: small launches; geometry 18 -- four waves,
 * one per SIMD -- only) */
constexpr bool HAS_PARTS = ACHIP_SINST == 18;
template <int MODE, bool COMP>
hipError_t launch_parts(const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out, uint64_t stride, uint32_t *len,
                        const achip_uniform_t &uni, unsigned long long *prof, const achip_partsdev_t &ps, hipStream_t stream) {
  if constexpr (HAS_PARTS) {
    using L = achip::SLds<MODE, G::WAVES, G::CPL, false>;
    auto kern = achip::render_stream_kernel<MODE, G::WAVES, G::CPL, COMP, false, 0, true>;
    static bool attr_set = false;
    if (!attr_set) {
      if (L::bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L::bytes);
        if (e != hipSuccess)
          return e;
      }
      attr_set = true;
    }
    const size_t lds = (size_t)((L::bytes_for(achip::stream_maxblk(uni.flags, L::EFF)) + 15) & ~15);
    hipLaunchKernelGGL(kern, dim3((unsigned)n * (unsigned)ps.parts), dim3(G::WAVES * 64), lds, stream, frames, lut, out, stride, len, n,
                       uni, prof, achip_wire_t{}, static_cast<const uint4 *>(nullptr), achip_packdev_t{}, ps);
    return hipGetLastError();
  } else {
    (void)frames, (void)n, (void)lut, (void)out, (void)stride, (void)len, (void)uni, (void)prof, (void)ps, (void)stream;
    return hipErrorInvalidValue;
  }
}

/* PACK instantiations (exact-length frames straight from the render; geometries 16 and 17): frames only, or with the
 * frame checksummed from its LDS image (wire stage) */
template <int BLOCK> hipError_t frame_crc_tables(const uint4 **out) {
  constexpr int MAX_DEVICES = 16;
  static std::mutex mu;
  static uint32_t *tab[MAX_DEVICES] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess)
    return e;
  if (dev < 0 || dev >= MAX_DEVICES)
    return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(mu);
  if (!tab[dev]) {
    uint32_t *t = nullptr;
    e = hipMalloc(reinterpret_cast<void **>(&t), (size_t)ACHIP_FRAME_CRC_TAB_BYTES);
    if (e != hipSuccess)
      return e;
    hipLaunchKernelGGL((achip::crc_frame_tables_init_kernel<BLOCK>), dim3(1), dim3(256), ACHIP_FRAME_CRC_TAB_BYTES, nullptr, t);
    e = hipGetLastError();
    if (e == hipSuccess)
      e = hipDeviceSynchronize();
    if (e != hipSuccess) {
      (void)hipFree(t);
      return e;
    }
    tab[dev] = t;
  }
  *out = reinterpret_cast<const uint4 *>(tab[dev]);
  return hipSuccess;
}

template <int MODE, bool WIRE>
hipError_t launch_pack(const achip_frame_t *frames, int n, const achip_lut_t *lut, uint64_t stride, uint32_t *len,
                       const achip_uniform_t &uni, const achip_wire_t &wire, const achip_packdev_t &pack, hipStream_t stream) {
  if constexpr ((ACHIP_SINST == 16 || ACHIP_SINST == 17) && MODE != ACHIP_MODE_TRUE_BG) {
    constexpr int PACK = WIRE ? 2 : 1;
    using L = achip::SLds<MODE, G::WAVES, G::CPL, false, PACK>;
    auto kern = achip::render_stream_kernel<MODE, G::WAVES, G::CPL, false, false, PACK>;
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L::bytes);
      if (e != hipSuccess)
        return e;
      attr_set = true;
    }
    const uint4 *tab = nullptr;
    if constexpr (WIRE) {
      hipError_t e = frame_crc_tables<64 * achip::pack_crc_waves(G::WAVES)>(&tab);
      if (e != hipSuccess)
        return e;
    }
    /* LDS: the tables, the per-block words of the launch's largest frame, then the frame's image -- as long as the
     * launch's frames can be (`stride`, the plan's bound): two 8-wave workgroups of 1080p -> 80x24 frames share a CU */
    const size_t lds = (size_t)((L::bytes_for_pack(achip::stream_maxblk(uni.flags, L::EFF), (int)stride) + 15) & ~15);
    hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(G::WAVES * 64), lds, stream, frames, lut, static_cast<uint8_t *>(nullptr),
                       stride, len, n, uni, static_cast<unsigned long long *>(nullptr), wire, tab, pack, achip_partsdev_t{});
    return hipGetLastError();
  } else {
    (void)frames, (void)n, (void)lut, (void)stride, (void)len, (void)uni, (void)wire, (void)pack, (void)stream;
    return hipErrorInvalidValue;
  }
}

} // namespace

#define ACHIP_CAT2(a, b) a##b
#define ACHIP_CAT(a, b) ACHIP_CAT2(a, b)

extern "C" int ACHIP_CAT(achipk_render_sinst_launch_, ACHIP_SINST)(int mode, int comp, const achip_frame_t *frames, int n,
                                                                  const achip_lut_t *lut, uint8_t *out, uint64_t stride,
                                                                  uint32_t *len, const achip_uniform_t *uniform,
                                                                  unsigned long long *prof, const achip_wire_t *wire,
                                                                  void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  achip_uniform_t uni = {};
  if (uniform && uniform->enabled) /* (composite batches too: achip_frames_uniform) */
    uni = *uniform;
  if (uniform)
    uni.flags = uniform->flags; /* launch-wide facts travel even when the descriptors come from the device array */
  /* truecolor foreground with a palette that holds multi-byte glyphs: an instantiation of its own (render_stream.hpp);
   * whole frames of single sources without the fused checksum (the host plans the rest elsewhere) */
  if (mode == ACHIP_MODE_TRUE_FG && !(uni.flags & ACHIP_UNIFORM_PALETTE_ASCII)) {
    if constexpr (HAS_U8) {
      if (comp || wire)
        return (int)hipErrorInvalidValue;
      return (int)launch_one<ACHIP_STREAM_MODE_TRUE_FG_U8, false, false>(frames, n, lut, out, stride, len, uni, prof, achip_wire_t{}, s);
    } else {
      return (int)hipErrorInvalidValue;
    }
  }
  switch (mode) {
#define M(m)                                                                                                           \
  case m:                                                                                                              \
    if (wire) {                                                                                                        \
      if constexpr (HAS_CRC)                                                                                           \
        return (int)(!wire->crc ? hipErrorInvalidValue                                                                 \
                     : comp     ? launch_one<m, true, true>(frames, n, lut, out, stride, len, uni, prof, *wire, s)     \
                                : launch_one<m, false, true>(frames, n, lut, out, stride, len, uni, prof, *wire, s));  \
      else                                                                                                             \
        return (int)hipErrorInvalidValue;                                                                              \
    }                                                                                                                  \
    return (int)(comp ? launch_one<m, true, false>(frames, n, lut, out, stride, len, uni, prof, achip_wire_t{}, s)     \
                      : launch_one<m, false, false>(frames, n, lut, out, stride, len, uni, prof, achip_wire_t{}, s));
    M(ACHIP_MODE_TRUE_FG)
    M(ACHIP_MODE_256_FG)
    M(ACHIP_MODE_16_FG)
    M(ACHIP_MODE_TRUE_BG)
#undef M
  }
  return (int)hipErrorInvalidValue;
}

#if ACHIP_SINST == 16 || ACHIP_SINST == 17
extern "C" int ACHIP_CAT(achipk_render_sinst_pack_launch_, ACHIP_SINST)(int mode, const achip_frame_t *frames, int n, const achip_lut_t *lut,
                                              uint64_t stride, uint32_t *len, const achip_uniform_t *uniform,
                                              const achip_wire_t *wire, const achip_packdev_t *pack, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  achip_uniform_t uni = {};
  if (uniform && uniform->enabled)
    uni = *uniform;
  if (uniform)
    uni.flags = uniform->flags;
  if (!pack || !pack->dst || !pack->cursor || (wire && !wire->crc))
    return (int)hipErrorInvalidValue;
  switch (mode) {
#define M(m)                                                                                                           \
  case m:                                                                                                              \
    return (int)(wire ? launch_pack<m, true>(frames, n, lut, stride, len, uni, *wire, *pack, s)                        \
                      : launch_pack<m, false>(frames, n, lut, stride, len, uni, achip_wire_t{}, *pack, s));
    M(ACHIP_MODE_TRUE_FG)
    M(ACHIP_MODE_256_FG)
    M(ACHIP_MODE_16_FG)
#undef M
  }
  return (int)hipErrorInvalidValue;
}
#endif

#if ACHIP_SINST == 16 || ACHIP_SINST == 17
/* LENGTH-FIRST: exact-length truecolor frames of any size in ONE launch, the lean loop run twice (render_stream.hpp LF);
 * `stride` only bounds a frame's length */
extern "C" int ACHIP_CAT(achipk_render_sinst_lenfirst_launch_, ACHIP_SINST)(const achip_frame_t *frames, int n, const achip_lut_t *lut,
                                                                            uint64_t stride, uint32_t *len, const achip_uniform_t *uniform,
                                                                            const achip_packdev_t *pack, void *stream) {
  achip_uniform_t uni = {};
  if (uniform && uniform->enabled)
    uni = *uniform;
  if (uniform)
    uni.flags = uniform->flags;
  if (!pack || !pack->dst || !pack->cursor || !(uni.flags & ACHIP_UNIFORM_PALETTE_ASCII))
    return (int)hipErrorInvalidValue;
  using L = achip::SLds<ACHIP_MODE_TRUE_FG, G::WAVES, G::CPL, false>;
  auto kern = achip::render_stream_kernel<ACHIP_MODE_TRUE_FG, G::WAVES, G::CPL, false, false, 0, false, true>;
  static bool attr_set = false;
  if (!attr_set) {
    if (L::bytes > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L::bytes);
      if (e != hipSuccess)
        return (int)e;
    }
    attr_set = true;
  }
  const size_t lds = (size_t)((L::bytes_for(achip::stream_maxblk(uni.flags, L::EFF)) + 15) & ~15);
  hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(G::WAVES * 64), lds, static_cast<hipStream_t>(stream), frames, lut,
                     static_cast<uint8_t *>(nullptr), stride, len, n, uni, static_cast<unsigned long long *>(nullptr), achip_wire_t{},
                     static_cast<const uint4 *>(nullptr), *pack, achip_partsdev_t{});
  return (int)hipGetLastError();
}
#endif

#if ACHIP_SINST == 18
extern "C" int achipk_render_sinst_parts_launch_18(int mode, int comp, const achip_frame_t *frames, int n, const achip_lut_t *lut,
                                                  uint8_t *out, uint64_t stride, uint32_t *len, const achip_uniform_t *uniform,
                                                  unsigned long long *prof, const achip_partsdev_t *ps, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  achip_uniform_t uni = {};
  if (uniform && uniform->enabled) /* (composite batches too: achip_frames_uniform) */
    uni = *uniform;
  if (uniform)
    uni.flags = uniform->flags;
  if (!ps || ps->parts < 2 || ps->parts > 64 || !ps->sync || ps->epoch == 0u)
    return (int)hipErrorInvalidValue;
  switch (mode) {
#define M(m)                                                                                                           \
  case m:                                                                                                              \
    return (int)(comp ? launch_parts<m, true>(frames, n, lut, out, stride, len, uni, prof, *ps, s)                     \
                      : launch_parts<m, false>(frames, n, lut, out, stride, len, uni, prof, *ps, s));
    M(ACHIP_MODE_TRUE_FG)
    M(ACHIP_MODE_256_FG)
    M(ACHIP_MODE_16_FG)
    M(ACHIP_MODE_TRUE_BG)
#undef M
  }
  return (int)hipErrorInvalidValue;
}
#endif

extern "C" int ACHIP_CAT(achipk_render_sinst_lds_, ACHIP_SINST)(int mode) {
  switch (mode) {
#define M(m)                                                                                                           \
  case m:                                                                                                              \
    return achip::SLds<m, G::WAVES, G::CPL>::bytes;
    M(ACHIP_MODE_TRUE_FG)
    M(ACHIP_MODE_256_FG)
    M(ACHIP_MODE_16_FG)
    M(ACHIP_MODE_TRUE_BG)
#undef M
  }
  return -1;
}
