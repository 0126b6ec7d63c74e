/*
 * render_rows_inst.hip -- instantiates the rows kernel (render_rows.hpp) for ONE geometry and ONE mode
 * (-DACHIP_RINST=<variant id> -DACHIP_RMODE=<mode id>): {plain, composite sampler} x {plain, frame CRC riding the drain}.
 * One translation unit per (geometry, mode) so that the build runs in parallel (the seven-slot geometry's five modes in
 * one unit were the build's critical path: 55 s); hip_launch.hip dispatches on the mode.  Built only with hipcc
 * --offload-arch=gfx950.
 */
#include <hip/hip_runtime.h>

#include <mutex>

#include "render_inst.h"
#define ACHIP_FRAME_KERNEL_ONLY
#include "render_rows.hpp"
#include "render_variants.h"

#if !defined(ACHIP_RINST) || !defined(ACHIP_RMODE)
#error "compile with -DACHIP_RINST=<rows variant id> -DACHIP_RMODE=<mode id>"
#endif

namespace {

template <int ID> struct RGeometry;
#define X(id, W, C)                                                                                                    \
  template <> struct RGeometry<id> {                                                                                   \
    static constexpr int WAVES = W, CPL = C;                                                                           \
    static constexpr bool WIDE = ACHIP_ROWS_VARIANT_WIDE(id), PARTS = ACHIP_ROWS_VARIANT_PARTS(id);                    \
  };
ACHIP_ROWS_VARIANTS(X)
#undef X
using G = RGeometry<ACHIP_RINST>;

/* the frame CRC riding the rows kernel's drain costs more than the stand-alone pass (hip_launch.hip:
 * achip_variant_crc_pays), so no plan takes it by itself: those instantiations exist in -DACHIP_ALL_GEOMETRIES builds only */
#ifdef ACHIP_ALL_GEOMETRIES
constexpr bool HAS_CRC = ACHIP_RINST != 26 && !G::WIDE && !G::PARTS;
#else
constexpr bool HAS_CRC = false;
#endif
constexpr bool HAS_COMP = ACHIP_RINST != 26 && !G::WIDE && !G::PARTS; /* (the sixteen-wave geometry carries the fast sampler only: achip_choose_geometry never takes it for composites / 1x1 sources) */

/* the constant tables of <MODE>'s CRC instantiation: built on the device once per process, then read-only */
template <int MODE> hipError_t crc_tables(const uint4 **out) {
  using L = achip::RLds<MODE, G::WAVES, true>;
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
    e = hipMalloc(reinterpret_cast<void **>(&t), (size_t)L::TAB_BYTES);
    if (e != hipSuccess)
      return e;
    hipLaunchKernelGGL((achip::crc_tables_init_kernel<L>), dim3(1), dim3(256), 0, nullptr, t);
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

template <int MODE, bool COMP, bool CRC>
hipError_t launch_one(const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out, uint64_t stride,
                      uint32_t *len, const achip_uniform_t &uni, const achip_wire_t &wire, const achip_partsdev_t &ps, hipStream_t stream) {
  using L = achip::RLds<MODE, G::WAVES, CRC, G::WIDE>;
  auto kern = achip::render_rows_kernel<MODE, G::WAVES, G::CPL, COMP, CRC, G::WIDE, G::PARTS>;
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
  /* uni.flags carries the blocks of the launch's largest frame (achip_rows_max_blocks): the per-block words */
  const size_t lds = (size_t)((L::bytes_for(achip::stream_maxblk(uni.flags, 1)) + 15) & ~15);
  /* (PARTS: workgroup f * parts + p renders the p-th run of frame f's blocks) */
  hipLaunchKernelGGL(kern, dim3((unsigned)n * (unsigned)(G::PARTS ? ps.parts : 1)), dim3(G::WAVES * 64), lds, stream, frames, lut, out,
                     stride, len, n, uni, wire, tab, ps);
  return hipGetLastError();
}

} // namespace

#define ACHIP_CAT2(a, b) a##b
#define ACHIP_CAT(a, b) ACHIP_CAT2(a, b)

extern "C" int ACHIP_CAT(ACHIP_CAT(ACHIP_CAT(achipk_render_rinst_launch_, ACHIP_RINST), _m), ACHIP_RMODE)(int mode, int comp, const achip_frame_t *frames, int n,
                                                                  const achip_lut_t *lut, uint8_t *out, uint64_t stride,
                                                                  uint32_t *len, const achip_uniform_t *uniform,
                                                                  const achip_wire_t *wire, const achip_partsdev_t *parts, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  achip_partsdev_t ps = {1, 1u, nullptr};
  if (parts)
    ps = *parts;
  if (ps.parts < 1 || ps.parts > 64 || (ps.parts > 1 && (!G::PARTS || !ps.sync || ps.epoch == 0u)))
    return (int)hipErrorInvalidValue;
  achip_uniform_t uni = {};
  if (uniform && uniform->enabled) /* (composite batches too: achip_frames_uniform) */
    uni = *uniform;
  if (uniform)
    uni.flags = uniform->flags;
  switch (mode) {
#define M(m)                                                                                                           \
  case m:                                                                                                              \
    if (wire) {                                                                                                        \
      if constexpr (HAS_CRC)                                                                                           \
        return (int)(!wire->crc ? hipErrorInvalidValue                                                                 \
                     : comp     ? launch_one<m, true, true>(frames, n, lut, out, stride, len, uni, *wire, ps, s)       \
                                : launch_one<m, false, true>(frames, n, lut, out, stride, len, uni, *wire, ps, s));    \
      else                                                                                                             \
        return (int)hipErrorInvalidValue;                                                                              \
    }                                                                                                                  \
    if (comp) {                                                                                                        \
      if constexpr (HAS_COMP)                                                                                          \
        return (int)launch_one<m, true, false>(frames, n, lut, out, stride, len, uni, achip_wire_t{}, ps, s);          \
      else                                                                                                             \
        return (int)hipErrorInvalidValue;                                                                              \
    }                                                                                                                  \
    return (int)launch_one<m, false, false>(frames, n, lut, out, stride, len, uni, achip_wire_t{}, ps, s);
    M(ACHIP_RMODE)
#undef M
  }
  return (int)hipErrorInvalidValue;
}

extern "C" int ACHIP_CAT(ACHIP_CAT(ACHIP_CAT(achipk_render_rinst_lds_, ACHIP_RINST), _m), ACHIP_RMODE)(int mode) {
  switch (mode) {
#define M(m)                                                                                                           \
  case m:                                                                                                              \
    return achip::RLds<m, G::WAVES, false, G::WIDE>::bytes;
    M(ACHIP_RMODE)
#undef M
  }
  return -1;
}
