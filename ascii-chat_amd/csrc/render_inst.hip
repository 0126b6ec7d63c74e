/*
 * render_inst.hip -- instantiates the frame kernel for ONE geometry (-DACHIP_INST=<variant id>): ten modes x
 * {plain, composite sampler} x {whole-frame, row-band} launches.  One translation unit per geometry so that
 * the build runs in parallel (make -j).  Built only with hipcc --offload-arch=gfx950.
 */
#include <hip/hip_runtime.h>

#include "render_inst.h"
#define ACHIP_FRAME_KERNEL_ONLY
#include "render_kernels.hpp"
#include "render_variants.h"

#if !defined(ACHIP_INST) || !defined(ACHIP_PART)
#error "compile with -DACHIP_INST=<variant id> -DACHIP_PART=<0: modes 0..2 | 1: modes 3, 4 | 2: modes 5..7 | 3: modes 8, 9> (render_inst.h: ACHIP_INST_PART_OF)"
#endif
#define ACHIP_IN_PART(m) (ACHIP_INST_PART_OF(m) == ACHIP_PART)

namespace {

#define X(id, B, C, R)                                                                                                 \
  template <> struct Geometry<id> {                                                                                    \
    static constexpr int BLOCK = B, CAP = C, RING = R;                                                                 \
  };
template <int ID> struct Geometry;
ACHIP_VARIANTS(X)
#undef X
using G = Geometry<ACHIP_INST>;

/* row bands are only ever launched with the geometries the host policy picks for them (achip_choose_geometry) */
constexpr bool HAS_SPLIT = ACHIP_INST == 1 || ACHIP_INST == 2 || ACHIP_INST == 4;
/* the half-block modes never run in the 512- / 256-thread geometries (they need more than the 128 VGPRs that make those
 * geometries worthwhile; the host policy sends them to the 1024-thread one, the wide one or the rows kernel): those
 * instantiations do not exist */
template <int MODE> constexpr bool has_mode() {
  return ACHIP_IN_PART(MODE) && !(achip::mode_is_halfblock(MODE) && (ACHIP_INST == 1 || ACHIP_INST == 2));
}

template <int MODE, bool COMP, bool SPLIT>
hipError_t launch_one(const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out, uint64_t stride,
                      uint32_t *len, unsigned long long *prof, int parts, int rows_per_part, unsigned long long *part_sync,
                      uint32_t epoch, const achip_uniform_t &uni, hipStream_t stream) {
  using L = achip::Lds<MODE, G::BLOCK, G::CAP, G::RING>;
  auto kern = achip::render_frames_kernel<MODE, G::BLOCK, G::CAP, G::RING, COMP, SPLIT>;
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
  hipLaunchKernelGGL(kern, dim3((unsigned)n * (unsigned)parts), dim3(G::BLOCK), (size_t)L::bytes, stream, frames, lut,
                     out, stride, len, n, prof, parts, rows_per_part, part_sync, epoch, uni);
  return hipGetLastError();
}

template <int MODE>
hipError_t launch_mode(bool comp, const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out,
                       uint64_t stride, uint32_t *len, unsigned long long *prof, int parts, int rows_per_part,
                       unsigned long long *part_sync, uint32_t epoch, const achip_uniform_t &uni, hipStream_t stream) {
  if (parts > 1) {
    if constexpr (HAS_SPLIT) {
      return comp ? launch_one<MODE, true, true>(frames, n, lut, out, stride, len, prof, parts, rows_per_part, part_sync,
                                                 epoch, uni, stream)
                  : launch_one<MODE, false, true>(frames, n, lut, out, stride, len, prof, parts, rows_per_part,
                                                  part_sync, epoch, uni, stream);
    } else {
      return hipErrorInvalidValue;
    }
  }
  return comp ? launch_one<MODE, true, false>(frames, n, lut, out, stride, len, prof, 1, rows_per_part, nullptr, epoch,
                                              uni, stream)
              : launch_one<MODE, false, false>(frames, n, lut, out, stride, len, prof, 1, rows_per_part, nullptr, epoch,
                                               uni, stream);
}

} // namespace

#define ACHIP_CAT2(a, b) a##b
#define ACHIP_CAT(a, b) ACHIP_CAT2(a, b)

extern "C" int ACHIP_CAT(ACHIP_CAT(ACHIP_CAT(achipk_render_inst_launch_, ACHIP_INST), _p), ACHIP_PART)(int mode, int comp, const achip_frame_t *frames, int n,
                                                                const achip_lut_t *lut, uint8_t *out, uint64_t stride,
                                                                uint32_t *len, unsigned long long *prof, int parts,
                                                                int rows_per_part, unsigned long long *part_sync,
                                                                uint32_t epoch, const achip_uniform_t *uniform,
                                                                void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  achip_uniform_t uni = {};
  if (uniform && uniform->enabled) /* (composite batches too: achip_frames_uniform) */
    uni = *uniform;
  switch (mode) {
#define M(m)                                                                                                           \
  case m:                                                                                                              \
    if constexpr (has_mode<m>())                                                                                       \
      return (int)launch_mode<m>(comp != 0, frames, n, lut, out, stride, len, prof, parts, rows_per_part, part_sync,   \
                                 epoch, uni, s);                                                                       \
    else                                                                                                               \
      return (int)hipErrorInvalidValue;
    M(ACHIP_MODE_MONO)
    M(ACHIP_MODE_TRUE_FG)
    M(ACHIP_MODE_256_FG)
    M(ACHIP_MODE_16_FG)
    M(ACHIP_MODE_TRUE_BG)
    M(ACHIP_MODE_HB_TRUE)
    M(ACHIP_MODE_HB_256)
    M(ACHIP_MODE_HB_16)
    M(ACHIP_MODE_HB_MONO)
    M(ACHIP_MODE_16_DITHER_BG)
#undef M
  }
  return (int)hipErrorInvalidValue;
}

extern "C" int ACHIP_CAT(ACHIP_CAT(ACHIP_CAT(achipk_render_inst_lds_, ACHIP_INST), _p), ACHIP_PART)(int mode) {
  switch (mode) {
#define M(m)                                                                                                           \
  case m:                                                                                                              \
    return ACHIP_IN_PART(m) ? achip::Lds<m, G::BLOCK, G::CAP, G::RING>::bytes : -1;
    M(ACHIP_MODE_MONO)
    M(ACHIP_MODE_TRUE_FG)
    M(ACHIP_MODE_256_FG)
    M(ACHIP_MODE_16_FG)
    M(ACHIP_MODE_TRUE_BG)
    M(ACHIP_MODE_HB_TRUE)
    M(ACHIP_MODE_HB_256)
    M(ACHIP_MODE_HB_16)
    M(ACHIP_MODE_HB_MONO)
    M(ACHIP_MODE_16_DITHER_BG)
#undef M
  }
  return -1;
}
