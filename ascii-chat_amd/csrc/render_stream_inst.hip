/*
 * render_stream_inst.hip -- instantiates the stream kernel (render_stream.hpp) for ONE geometry
 * (-DACHIP_SINST=<variant id>): four per-cell modes x {plain, composite sampler}.  One translation unit per
 * geometry so that the build runs in parallel.  Built only with hipcc --offload-arch=gfx950.
 */
#include <hip/hip_runtime.h>

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

template <int MODE, bool COMP>
hipError_t launch_one(const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out, uint64_t stride,
                      uint32_t *len, const achip_uniform_t &uni, unsigned long long *prof, hipStream_t stream) {
  using L = achip::SLds<MODE, G::WAVES, G::CPL>;
  auto kern = achip::render_stream_kernel<MODE, G::WAVES, G::CPL, COMP>;
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
  hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(G::WAVES * 64), (size_t)L::bytes, stream, frames, lut, out, stride,
                     len, n, uni, prof);
  return hipGetLastError();
}

} // namespace

#define ACHIP_CAT2(a, b) a##b
#define ACHIP_CAT(a, b) ACHIP_CAT2(a, b)

extern "C" int ACHIP_CAT(achip_render_sinst_launch_, ACHIP_SINST)(int mode, int comp, const achip_frame_t *frames, int n,
                                                                  const achip_lut_t *lut, uint8_t *out, uint64_t stride,
                                                                  uint32_t *len, const achip_uniform_t *uniform,
                                                                  unsigned long long *prof, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  achip_uniform_t uni = {};
  if (uniform && uniform->enabled && !comp)
    uni = *uniform;
  if (uniform)
    uni.flags = uniform->flags; /* launch-wide facts travel even when the descriptors come from the device array */
  switch (mode) {
#define M(m)                                                                                                           \
  case m:                                                                                                              \
    return (int)(comp ? launch_one<m, true>(frames, n, lut, out, stride, len, uni, prof, s)                                  \
                      : launch_one<m, false>(frames, n, lut, out, stride, len, uni, prof, s));
    M(ACHIP_MODE_TRUE_FG)
    M(ACHIP_MODE_256_FG)
    M(ACHIP_MODE_16_FG)
    M(ACHIP_MODE_TRUE_BG)
#undef M
  }
  return (int)hipErrorInvalidValue;
}

extern "C" int ACHIP_CAT(achip_render_sinst_lds_, ACHIP_SINST)(int mode) {
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
