/*
 * hip_launch.hip -- instantiates the gfx950 kernels and exposes plain-C launchers.
 * Built only with hipcc --offload-arch=gfx950; there is no host/CPU variant of these entry points.
 */
#include "hip_launch.h"

#include <hip/hip_runtime.h>

#include "render_kernels.hpp"
#include "render_variants.h"

namespace {

template <int MODE, int BLOCK, int CAP, int RING, bool COMP>
hipError_t launch_one(const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out, uint64_t stride,
                      uint32_t *len, unsigned long long *prof, int parts, int rows_per_part, unsigned long long *part_sync,
                      uint32_t epoch, hipStream_t stream) {
  using L = achip::Lds<MODE, BLOCK, CAP, RING>;
  auto kern = achip::render_frames_kernel<MODE, BLOCK, CAP, RING, COMP>;
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
  hipLaunchKernelGGL(kern, dim3((unsigned)n * (unsigned)parts), dim3(BLOCK), (size_t)L::bytes, stream, frames, lut, out,
                     stride, len, n, prof, parts, rows_per_part, part_sync, epoch);
  return hipGetLastError();
}

template <int BLOCK, int CAP, int RING>
hipError_t launch_mode(int mode, bool comp, const achip_frame_t *frames, int n, const achip_lut_t *lut, uint8_t *out,
                       uint64_t stride, uint32_t *len, unsigned long long *prof, int parts, int rows_per_part,
                       unsigned long long *part_sync, uint32_t epoch, hipStream_t stream) {
  switch (mode) {
#define M(m)                                                                                                           \
  case m:                                                                                                              \
    return comp ? launch_one<m, BLOCK, CAP, RING, true>(frames, n, lut, out, stride, len, prof, parts, rows_per_part,  \
                                                        part_sync, epoch, stream)                                      \
                : launch_one<m, BLOCK, CAP, RING, false>(frames, n, lut, out, stride, len, prof, parts, rows_per_part, \
                                                         part_sync, epoch, stream);
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
  return hipErrorInvalidValue;
}

template <int BLOCK, int CAP, int RING> int lds_for_mode(int mode) {
  switch (mode) {
#define M(m)                                                                                                           \
  case m:                                                                                                              \
    return achip::Lds<m, BLOCK, CAP, RING>::bytes;
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

} // namespace

extern "C" int achip_launch_render(int mode, int variant, int has_composite, const achip_frame_t *frames_dev, int n_frames,
                                   const achip_lut_t *lut_dev, uint8_t *out, uint64_t out_stride, uint32_t *out_len,
                                   unsigned long long *prof, int parts, int rows_per_part,
                                   unsigned long long *part_sync, uint32_t epoch, void *stream) {
  if (n_frames <= 0)
    return (int)hipSuccess;
  if (parts < 1 || (parts > 1 && (!part_sync || rows_per_part < 1)))
    return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (variant) {
#define X(id, B, C, R)                                                                                                 \
  case id:                                                                                                             \
    return (int)launch_mode<B, C, R>(mode, has_composite != 0, frames_dev, n_frames, lut_dev, out, out_stride, out_len, prof,    \
                                     parts, rows_per_part, part_sync, epoch, s);
    ACHIP_VARIANTS(X)
#undef X
  }
  return (int)hipErrorInvalidValue;
}

extern "C" int achip_variant_block(int variant) {
  switch (variant) {
#define X(id, B, C, R)                                                                                                 \
  case id:                                                                                                             \
    return B;
    ACHIP_VARIANTS(X)
#undef X
  }
  return -1;
}

extern "C" int achip_variant_cap(int variant) {
  switch (variant) {
#define X(id, B, C, R)                                                                                                 \
  case id:                                                                                                             \
    return C;
    ACHIP_VARIANTS(X)
#undef X
  }
  return -1;
}

extern "C" int achip_variant_lds_bytes(int mode, int variant) {
  switch (variant) {
#define X(id, B, C, R)                                                                                                 \
  case id:                                                                                                             \
    return lds_for_mode<B, C, R>(mode);
    ACHIP_VARIANTS(X)
#undef X
  }
  return -1;
}

extern "C" int achip_launch_resize(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh, void *stream) {
  const uint32_t xr = (uint32_t)((((uint64_t)sw << 16) / (uint64_t)dw) + 1u);
  const uint32_t yr = (uint32_t)((((uint64_t)sh << 16) / (uint64_t)dh) + 1u);
  const uint64_t total = (uint64_t)dw * (uint64_t)dh;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 2048u)
    blocks = 2048u;
  hipLaunchKernelGGL(achip::resize_nn_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, sw, sh,
                     3 * sw, dst, dw, dh, xr, yr);
  return (int)hipGetLastError();
}

extern "C" int achip_launch_composite(const achip_composite_t *comp_dev, int canvas_w, int canvas_h, uint8_t *dst,
                                      void *stream) {
  const uint64_t total = (uint64_t)canvas_w * (uint64_t)canvas_h;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 2048u)
    blocks = 2048u;
  hipLaunchKernelGGL(achip::composite_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), comp_dev,
                     dst);
  return (int)hipGetLastError();
}

/* ---- display-path streaming passes (stream_kernels.hpp) -------------------------------------------- */
#include "stream_kernels.hpp"

static unsigned stream_blocks(uint64_t items) {
  uint64_t b = (items + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 256u * 16u ? 256u * 16u : b)); /* <= 16 workgroups per CU, grid-stride beyond */
}

extern "C" int achip_launch_tint(uint8_t *px, int w, int h, int stride, uint32_t ops, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (stride == 3 * w && (reinterpret_cast<uintptr_t>(px) & 15u) == 0) {
    const uint64_t nbytes = (uint64_t)w * (uint64_t)h * 3u;
    hipLaunchKernelGGL(achip::tint_stream_kernel, dim3(stream_blocks(nbytes / 48u + 1)), dim3(256), 0, s, px, nbytes,
                       ops);
  } else {
    hipLaunchKernelGGL(achip::tint_pixels_kernel, dim3(stream_blocks((uint64_t)w * (uint64_t)h)), dim3(256), 0, s, px,
                       w, h, stride, ops);
  }
  return (int)hipGetLastError();
}

extern "C" int achip_launch_flip(const uint8_t *src, uint8_t *dst, int w, int h, int src_stride, int dst_stride,
                                 uint32_t ops, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = w % 16 == 0 && src_stride == 3 * w && dst_stride == 3 * w &&
                   ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
  if (vec)
    hipLaunchKernelGGL(achip::flip_stream_kernel, dim3(stream_blocks((uint64_t)(w / 16) * (uint64_t)h)), dim3(256), 0,
                       s, src, dst, w, h, ops);
  else
    hipLaunchKernelGGL(achip::flip_pixels_kernel, dim3(stream_blocks((uint64_t)w * (uint64_t)h)), dim3(256), 0, s, src,
                       dst, w, h, src_stride, dst_stride, ops);
  return (int)hipGetLastError();
}
