/*
 * render_variants.h -- the (BLOCK, CAP, RING) geometries the frame kernel is instantiated for.
 *   BLOCK  threads per workgroup (one workgroup renders one frame)
 *   CAP    cells per chunk; a padded text row (pad_left + out_w) must fit in one chunk
 *   RING   bytes of the LDS output ring
 * X(id, BLOCK, CAP, RING)
 */
#ifndef ACHIP_RENDER_VARIANTS_H
#define ACHIP_RENDER_VARIANTS_H

#define ACHIP_VARIANTS(X)                                                                                         \
  X(0, 1024, 4096, 65536) /* wide: any row up to 4096 cells, 1 workgroup per CU                                */ \
  X(1, 512, 2048, 32768)  /* narrow: rows up to 2048 cells, 2-3 workgroups per CU                               */ \
  X(2, 256, 1024, 16384)  /* small grids (<= 1024-cell rows): 4+ workgroups per CU                              */ \
  X(3, 64, 256, 256)      /* test geometry: forces multi-chunk frames and ring wrap-around on tiny inputs      */ \
  X(4, 1024, 2048, 65536) /* rows up to 2048 cells, 2 cells per thread: lowest register pressure                */

#define ACHIP_VARIANT_COUNT 5

#endif
