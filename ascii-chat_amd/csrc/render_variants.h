/*
 * render_variants.h -- the (BLOCK, CAP, RING) geometries the frame kernel is instantiated for.
 *   BLOCK  threads per workgroup (one workgroup renders one frame)
 *   CAP    cells per chunk; a padded text row (pad_left + out_w) must fit in one chunk
 *   RING   bytes of the LDS output staging buffer (multiple of 16)
 * X(id, BLOCK, CAP, RING)
 */
#ifndef ACHIP_RENDER_VARIANTS_H
#define ACHIP_RENDER_VARIANTS_H

#ifdef ACHIP_TEST_GEOMETRY /* emulator builds only (tests/hipemu): forces multi-chunk frames and window cuts on tiny inputs */
#define ACHIP_TEST_VARIANT(X) X(3, 64, 256, 256)
#else
#define ACHIP_TEST_VARIANT(X)
#endif
/* Geometries the automatic choice (achip_choose_geometry) takes for no plan of the round-4 audit grid (420 plans,
 * profiles/r04_policy_audit.txt) nor for any BASELINE config are built only with -DACHIP_ALL_GEOMETRIES (make
 * EXTRA=-DACHIP_ALL_GEOMETRIES; the emulator and the mock runtime of the CPU suite always are): frame geometry 2, stream
 * geometry 19, the rows kernel's fused-CRC instantiations.  They stay selectable by hand (plan_set_variant /
 * plan_set_fused_crc) in such a build; in the default build those calls say so. */
#ifdef ACHIP_ALL_GEOMETRIES
#define ACHIP_EXTRA_VARIANT(X) X(2, 256, 1024, 16384) /* small grids (<= 1024-cell rows): 4+ workgroups per CU */
#define ACHIP_STREAM_EXTRA_VARIANT(X) X(19, 16, 1)    /* 1024 threads, one cell per lane: fewest registers     */
#else
#define ACHIP_EXTRA_VARIANT(X)
#define ACHIP_STREAM_EXTRA_VARIANT(X)
#endif
#define ACHIP_VARIANTS(X)                                                                                         \
  X(0, 512, 4096, 65536)  /* wide: any row up to 4096 cells; eight cells per thread in 512-thread workgroups, which may
                             use 256 VGPRs (a 1024-thread workgroup is capped at 128: its four cells per thread spilled) */ \
  X(1, 512, 2048, 32768)  /* narrow: rows up to 2048 cells, 2-3 workgroups per CU                               */ \
  ACHIP_EXTRA_VARIANT(X)  /* id 2                                                                               */ \
  ACHIP_TEST_VARIANT(X)   /* id 3 does not exist in the product library                                         */ \
  X(4, 1024, 2048, 114688) /* rows up to 2048 cells, 2 cells per thread (no spills); 112 KB staging: one window per
                              chunk even for 41-byte half-block tokens; with the other tables ~132-156 KB of the 160 KB LDS */

#define ACHIP_VARIANT_COUNT 5 /* ids 0..4 (3 = the emulator's test geometry, absent from the product) */

/* Geometries of the wave-autonomous stream kernel (render_stream.hpp; per-cell modes, whole-frame launches):
 *   WAVES  waves per workgroup (one workgroup renders one frame)
 *   CPL    cells per lane per block (a block = 64 * CPL consecutive cells, taken through the path by one wave)
 * X(id, WAVES, CPL); ids continue behind a gap so that the two families cannot be confused. */
#define ACHIP_STREAM_VARIANT_FIRST 16
#ifdef ACHIP_TEST_GEOMETRY /* emulator builds only: many tiny blocks, look-back windows beyond 64 predecessors */
#define ACHIP_STREAM_TEST_VARIANT(X) X(20, 2, 1)
#else
#define ACHIP_STREAM_TEST_VARIANT(X)
#endif
#ifndef ACHIP_STREAM17_CPL
#define ACHIP_STREAM17_CPL 2 /* (A/B builds: more cells per lane and turn) */
#endif
#define ACHIP_STREAM_VARIANTS(X)                                                                                  \
  X(16, 16, 2) /* 1024 threads: a 1080p -> 80x24 frame is one block per wave                                   */ \
  X(17, 8, ACHIP_STREAM17_CPL)  /* 512 threads: two to four workgroups per CU                                  */ \
  X(18, 4, 2)  /* 256 threads                                                                                   */ \
  ACHIP_STREAM_EXTRA_VARIANT(X) /* id 19                                                                        */ \
  ACHIP_STREAM_TEST_VARIANT(X)
#define ACHIP_ROWS_VARIANT_FIRST 24
#define ACHIP_IS_STREAM_VARIANT(v) ((v) >= ACHIP_STREAM_VARIANT_FIRST && (v) < ACHIP_ROWS_VARIANT_FIRST)

/* Geometries of the wave-autonomous kernel of the run-structured modes (render_rows.hpp; mono, half blocks; whole-frame
 * launches): a block is a whole number of text rows taken through the path by ONE wave, so a padded row must fit
 * 64 * CPL cells.
 *   WAVES  waves per workgroup (one workgroup renders one frame)
 *   CPL    cells per lane per block
 * X(id, WAVES, CPL) */
#ifdef ACHIP_TEST_GEOMETRY /* emulator builds only: two-row blocks of tiny frames, many blocks per wave; rows of up to four
                              64-cell segments */
#define ACHIP_ROWS_TEST_VARIANT(X) X(28, 2, 2) X(30, 4, 1) X(33, 2, 1) X(34, 2, 1)
#else
#define ACHIP_ROWS_TEST_VARIANT(X)
#endif
#ifndef ACHIP_ROWS24_WAVES
#define ACHIP_ROWS24_WAVES 8 /* (A/B builds: other workgroup sizes of the seven-slot geometry) */
#endif
#ifndef ACHIP_ROWS_WIDE_CPL
#define ACHIP_ROWS_WIDE_CPL 5 /* cell slots of a segment (achip_host.c restates it).  Measured with 5 / 6 / 7 (A/B builds,
                                 profiles/r06_wide_rows.txt): 640-cell rows = two segments of 320 cells, sampled 640x360 half
                                 blocks 346 / 361 / 377 us -- the slots a segment leaves empty cost a quarter of what they
                                 hold, so ONE width serves: 320 cells, which cuts 640 / 960 / 1280 / 1920 exactly */
#endif
#ifndef ACHIP_ROWS_PARTS_CPL
#define ACHIP_ROWS_PARTS_CPL 2 /* rows of up to 128 cells (achip_host.c restates it).  Measured with 2 and 4 (A/B builds,
                                  profiles/r06_small_rows_parts.txt): with four slots -- three 80-cell rows per block, two
                                  workgroups per frame -- nothing is gained over one eight-wave workgroup (a lone mono frame
                                  8.1-8.4 us against 7.6); with two -- ONE row per block, six workgroups -- 6.5 us, a lone
                                  half-block truecolor frame 8.3 against the row bands' 9.4: what shortens the frame's
                                  latency chain is the shorter block, not the idle SIMDs */
#endif
#define ACHIP_ROWS_VARIANTS(X)                                                                                    \
  X(24, ACHIP_ROWS24_WAVES, 7) /* rows up to 448 cells: 4K -> 400x120 half blocks is one row per block (89 % of the slots)        */ \
  X(25, 8, 4) /* rows up to 256 cells: 200x60, 160x48 one row per block; three 80-cell rows per block             */ \
  X(26, 16, 7) /* geometry 24 as ONE sixteen-wave workgroup per frame (round 5): whole-frame launches of at most a frame
                  per CU of the plan's share (achip_choose_geometry); fast sampler only, no fused CRC              */ \
  X(27, 16, ACHIP_ROWS_WIDE_CPL) /* WIDE (round 6): rows beyond 448 cells cut into at most sixteen segments of <= 320 cells, a
                  segment per block, its two ghost cells in one more slot (render_rows.hpp); rows up to 4096 cells */ \
  X(29, 8, ACHIP_ROWS_WIDE_CPL)  /* WIDE, two eight-wave workgroups per CU: rows of at most eight segments (2560 cells) */ \
  X(31, 4, ACHIP_ROWS_PARTS_CPL) /* PARTS (round 6): small launches of rows up to 128 cells, a frame's blocks (a text row each at
                  80 columns) shared out over four-wave workgroups, a block per wave; fast sampler only, no fused CRC */ \
  X(32, 4, 2)  /* WIDE + PARTS: small launches of rows of 129-512 cells -- a row cut into at most four segments of <= 128
                  cells, whole rows per four-wave workgroup                                                          */ \
  ACHIP_ROWS_TEST_VARIANT(X)
#define ACHIP_IS_ROWS_VARIANT(v) ((v) >= ACHIP_ROWS_VARIANT_FIRST)
/* the geometries whose blocks are SEGMENTS of a row (render_rows.hpp WIDE): fast sampler only, no fused CRC */
#define ACHIP_ROWS_VARIANT_WIDE(v) ((v) == 27 || (v) == 29 || (v) == 30 || (v) == 32 || (v) == 34)
#define ACHIP_ROWS_WIDE_MAX_ROW 4096
/* the geometries that share a frame's blocks out over several workgroups (render_rows.hpp PARTS; 33: the emulator's) */
#define ACHIP_ROWS_VARIANT_PARTS(v) ((v) == 31 || (v) == 32 || (v) == 33 || (v) == 34)

#endif
