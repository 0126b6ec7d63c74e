"""Multi-workgroup frames: a frame cut into row bands rendered by separate workgroups that learn their output
offset from each other (part_publish / part_wait).  Emulated here; the -m gpu tests run it on the MI355X."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import emu  # noqa: E402
import orc  # noqa: E402
from achip_ctypes import ALL_MODES, MODE_16_DITHER_BG, MODE_CAPS, MODE_NAMES, MODE_TRUE_BG, Frame  # noqa: E402

SPLITTABLE = [m for m in ALL_MODES if m != MODE_16_DITHER_BG]


def oracle(img, mode, W, H, pad, aspect, palette=orc.PALETTE_STANDARD):
    if mode == MODE_TRUE_BG:
        return orc.print_truecolor_bg(orc.resize_nn(img, W, H), palette)
    cl, rm = MODE_CAPS[mode]
    return orc.convert_with_caps(img, W, H, cl, rm, pad, aspect, False, palette)


@pytest.mark.parametrize("mode", SPLITTABLE, ids=[MODE_NAMES[m] for m in SPLITTABLE])
def test_split_frames_emulated(mode):
    img = orc.frame_torture()
    rm = MODE_CAPS.get(mode, (3, 0))[1]
    for (W, H, rpp, variant, pad) in [(80, 24, 6, 2, False), (80, 24, 1, 2, True), (97, 31, 4, 2, True), (40, 50, 3, 3, False),
                                      (200, 60, 5, 2, False)]:
        aspect = pad and mode != MODE_TRUE_BG
        f = emu.frame_for_convert(img, W, H, rm, pad, aspect)
        got = emu.render_frames(mode, [f], orc.PALETTE_STANDARD, variant, rows_per_part=rpp)[0]
        assert got == oracle(img, mode, W, H, pad, aspect), (MODE_NAMES[mode], W, H, rpp)


def test_split_ragged_batch_and_policy():
    # frames with different row counts in one launch: later parts of short frames simply do not exist
    imgs = [orc.frame_hash_noise(120, 90, i) for i in range(4)]
    dims = [(80, 24), (60, 7), (33, 40), (80, 1)]
    frames = [emu.frame_for_convert(im, w, h, 0) for im, (w, h) in zip(imgs, dims)]
    for mode in (1, 2, 0):
        got = emu.render_frames(mode, frames, orc.PALETTE_STANDARD, 2, rows_per_part=5)
        for k, (im, (w, h)) in enumerate(zip(imgs, dims)):
            assert got[k] == oracle(im, mode, w, h, False, False), (mode, k)
    # host policy (achip_choose_geometry)
    L = emu.lib()
    caps = (C.c_int * 5)(4096, 2048, 1024, 256, 2048)
    v, p, r = C.c_int(), C.c_int(), C.c_int()

    def choose(mode, fr, ascii_only=True, req=0, cus=256, forced=-1):
        arr = (Frame * len(fr))(*fr)
        assert L.achip_choose_geometry(mode, arr, len(fr), ascii_only, caps, cus, req, forced, C.byref(v), C.byref(p),
                                       C.byref(r)) == 0
        return v.value, p.value, r.value

    one = [emu.frame_for_convert(imgs[0], 80, 24, 0)]
    # small launches of the per-cell modes share a frame's blocks out over four-wave workgroups of the stream kernel (PARTS,
    # geometry 18; round 4, profiles/r04_small_batch_parts.txt) while every workgroup has a CU and every wave one block;
    # beyond that, frames of at most one block per wave go whole (a lone 80x24 frame: 5.9 us in four parts, 6.3 us whole on
    # the stream kernel, 8.4 us as 24 bands of the phase kernel; a lone mono frame 7.0 against 8.1 us)
    assert choose(1, one) == (18, 16, 1)                      # sixteen blocks: a workgroup each while the CUs allow
    assert choose(1, one * 64) == (18, 4, 1) and choose(1, one * 65) == (16, 1, 24) and choose(1, one * 150) == (16, 1, 24)
    assert choose(1, one * 16, cus=64) == (18, 4, 1) and choose(1, one * 17, cus=64) == (16, 1, 24)  # a share of the CUs
    # round 6 (profiles/r06_small_rows_parts.txt): small launches of the run-structured modes the same way on the rows kernel
    # (geometry 31: a text row per block, a block per wave of six four-wave workgroups; a lone mono frame 6.5 us against 7.6
    # whole, a lone half-block truecolor frame 8.3 against 9.0-9.4 as bands) while the workgroups have a CU each ...
    assert choose(0, one) == (31, 6, 1) and choose(0, one * 42) == (31, 6, 1)
    assert choose(0, one * 64) == (25, 1, 24)   # ... then mono whole: eight blocks of three rows, one per wave
    # ... the coloured half-block modes only from a frame per four CUs on: with fewer, row bands of the phase kernel (a thread
    # per cell) beat one wave taking 256 cells of long tokens through the path (profiles/r04_small_run_modes.txt: a lone 80x24
    # half-block truecolor frame 9.0 us as bands against 12.6 whole; 64 frames in four bands each 10.2 against 13.3)
    hb = [emu.frame_for_convert(imgs[0], 80, 24, 2)]
    assert choose(5, hb) == (31, 6, 1) and choose(6, hb * 8) == (31, 6, 1) and choose(7, hb * 64) == (4, 4, 6)
    assert choose(5, hb, req=1) == (4, 24, 1)  # (row bands when asked for rows)
    assert choose(5, hb * 65) == (25, 1, 24) and choose(8, hb) == (31, 6, 1) and choose(8, hb * 60) == (25, 1, 24)  # (half-block mono: short tokens, whole frames)
    assert choose(5, hb * 48) == (4, 5, 5)   # never more bands than CUs: five per frame at most (240 workgroups)
    # (round 6's last audits: at no more than a frame per CU the 256- / 16-colour and mono half blocks of one block per wave leave
    # full-frame sources to the phase kernel -- 8-10 % ahead there -- and keep the rows kernel for dense sources; truecolor half blocks keep it)
    assert choose(6, hb * 200)[:2] == (4, 1) and choose(7, hb * 256)[:2] == (4, 1) and choose(8, hb * 200)[:2] == (4, 1) and choose(5, hb * 200) == (25, 1, 24)
    dense_hb = [emu.frame_for_convert(np.ascontiguousarray(imgs[0][:48, :80]), 80, 24, 2)]
    assert choose(6, dense_hb * 200) == (25, 1, 24) and choose(6, hb * 300) == (25, 1, 24)   # (dense sources / more than a frame per CU: the rows kernel)
    mid = [emu.frame_for_convert(imgs[0], 160, 48, 0)]        # 7 680 cells = 61 blocks: sixteen parts (the grid's nine targets)
    assert choose(1, mid) == (18, 16, 1) and choose(1, mid * 9) == (18, 16, 1)
    # run-structured rows of 129-512 cells: cut into segments of at most 128 cells, whole rows per four-wave workgroup, a segment
    # per wave (geometry 32 = the rows kernel's WIDE + PARTS; one row per 256-slot block was measured before and lost to the row
    # bands, profiles/r06_small_rows_parts.txt) -- while every workgroup has a CU and every wave one block
    hbmid = [emu.frame_for_convert(imgs[0], 160, 48, 2)]
    assert choose(0, mid) == (32, 24, 1) and choose(5, hbmid) == (32, 24, 1) and choose(0, mid * 10) == (32, 24, 1) and choose(8, hbmid) == (32, 24, 1)
    assert choose(6, hbmid)[0] < 16 and choose(7, hbmid) == choose(6, hbmid) and choose(6, hbmid)[1] > 1  # 256 / 16 colours: 6-20 % behind their row bands (visit Q), which they keep
    assert choose(0, mid * 11)[0] < 16 and choose(0, mid * 11)[1] > 1   # 264 workgroups: row bands of the phase kernel as before
    wide3 = [emu.frame_for_convert(imgs[0], 300, 20, 0)]  # three segments of 100 cells: a row per workgroup
    assert choose(0, wide3) == (32, 20, 1) and choose(0, [emu.frame_for_convert(imgs[0], 512, 9, 0)]) == (32, 9, 1)
    assert choose(0, [emu.frame_for_convert(imgs[0], 513, 9, 0)])[0] < 16   # five segments: no four-wave workgroup holds the row
    assert choose(1, mid * 17)[0] < 16 and choose(1, mid * 17)[1] > 1  # sixteen parts no longer fit: row bands of the phase kernel as before
    # round 6, after the stream kernel's lean loop (profiles/r06_whole_from_three_eighths.txt): truecolor foreground from single
    # sources goes whole from 3/8 frame per CU on whatever the frame's size (96 frames of 200x60: 19.9 us against 25.7 as bands;
    # 160 frames 20.8 against 31.7), the other per-cell modes with frames of at most four blocks per wave (160x45)
    big = [emu.frame_for_convert(imgs[0], 200, 60, 0)]
    assert choose(1, big * 96)[:2] == (16, 1) and choose(1, big * 160)[:2] == (16, 1)
    assert choose(1, big * 95)[0] < 16 and choose(1, big * 95)[1] > 1       # below 3/8: row bands of the phase kernel (80 frames: 16.8 against 19.8 whole)
    assert choose(1, big * 24, cus=64)[:2] == (16, 1)                        # (a share of the CUs: the same fraction)
    assert choose(1, big * 257)[:2] == (17, 1) and choose(1, big * 512)[:2] == (17, 1) and choose(1, big * 256)[:2] == (16, 1)   # (the GPU to itself, one to two frames per CU, frames of more than one block per wave: 512-thread workgroups, 36-41 us against 41-50)
    assert choose(1, big * 128, cus=64)[:2] == (16, 1) and choose(1, one * 512)[:2] == (16, 1)                                # (a share of the CUs / small frames: as before)
    mid45 = [emu.frame_for_convert(imgs[0], 160, 45, 0)]
    assert choose(1, mid45 * 80)[:2] == (16, 1) and choose(1, mid45 * 79)[1] > 1 and choose(1, big * 80)[1] > 1   # (frames of at most four blocks per wave from 5/16)
    assert choose(2, mid45 * 96)[:2] == (16, 1) and choose(3, mid45 * 128)[:2] == (16, 1)
    assert choose(2, big * 128)[0] < 16 and choose(2, big * 128)[1] > 1    # 256 colours at 200x60: bands up to 3/4 (18.8 against 22.6 whole)
    # whole-frame launches of the per-cell modes take the stream kernel (render_stream.hpp): 1024 threads while every
    # frame has a CU to itself, 512-thread workgroups beyond that
    assert choose(1, one * 256) == (16, 1, 24)                # BASELINE batch = one frame per CU: whole frames
    assert choose(1, one * 600) == (17, 1, 24)                # many frames: whole frames, 512-thread workgroups
    assert choose(1, one * 256, cus=256 // 3) == (17, 1, 24)  # ... or several launches in flight sharing the CUs
    assert choose(1, one * 256, cus=256 // 4) == (17, 1, 24)
    # round 4 audit (profiles/r04_policy_audit.txt): what decides is the frames per CU of the plan's share, not the blocks per
    # wave -- sixteen-wave workgroups up to TWO frames per CU (128 frames at a share of 64: 23.5 vs 29.6 us at 200x60)
    assert choose(1, one * 128, cus=64) == (16, 1, 24) and choose(1, one * 129, cus=64) == (17, 1, 24)
    assert choose(2, one * 256) == (16, 1, 24) and choose(3, one * 256) == (16, 1, 24) and choose(4, one * 300) == (16, 1, 24) and choose(4, one * 513) == (17, 1, 24)
    # whole-frame launches of the run-structured modes take the rows kernel (render_rows.hpp) when the padded row fits a
    # block: the geometry whose blocks waste fewer lane slots (three 80-cell rows fill 240 of 256 slots, five fill 400 of 448)
    assert choose(0, one * 256) == (25, 1, 24)
    assert choose(0, one * 600) == (25, 1, 24)
    assert choose(0, one * 256, forced=4) == (4, 1, 24) and choose(0, one * 256, forced=24) == (24, 1, 24)
    with pytest.raises(AssertionError):
        choose(1, one * 256, forced=24)                       # a per-cell mode has no rows kernel
    assert choose(1, one * 256, ascii_only=False) == (16, 1, 24)  # truecolor-fg with multi-byte glyphs: the stream kernel too (round 6)
    assert choose(1, one * 256, forced=4) == (4, 1, 24) and choose(1, one * 256, forced=19) == (19, 1, 24)
    assert choose(5, [emu.frame_for_convert(imgs[0], 80, 24, 2)] * 600) == (25, 1, 24)  # half-block whole frames: rows kernel
    assert choose(9, one) == (4, 1, 24)                       # serial dither: never split
    assert choose(1, one, ascii_only=False) == (16, 1, 24)    # truecolor-fg with multi-byte glyphs: never split nor shared out (its RLE state crosses blocks through LDS words)
    assert choose(2, one, ascii_only=False) == (18, 15, 1)    # (the other per-cell modes carry multi-byte glyphs on the stream kernel)
    assert choose(1, one, req=-1) == (16, 1, 24)             # never split: one whole frame -> stream kernel
    assert choose(1, one, req=12) == (4, 2, 12)
    assert choose(1, one, req=3, forced=2) == (2, 8, 3)
    assert choose(1, one, forced=1, req=-1) == (1, 1, 24)
    k3w = [emu.frame_for_convert(imgs[0], 200, 60, 0)]
    assert choose(1, k3w * 256) == (16, 1, 60)                # several blocks per wave, a frame per CU: sixteen waves (34.1 vs 43.4 us)
    assert choose(1, k3w * 256, cus=64) == (17, 1, 60)        # BASELINE configs[2] with four launches in flight: 512-thread workgroups
    assert choose(1, k3w * 128, cus=64) == (16, 1, 60)
    m40 = [emu.frame_for_convert(imgs[0], 120, 40, 0)]        # 4 800 cells = 38 blocks: at most three per wave of sixteen
    assert choose(1, m40 * 128) == (16, 1, 40)                # ... whole from half a frame per CU on (12.9 us against 16.8 as bands)
    assert choose(1, m40 * 64)[1] > 1                         # below that: row bands (10.6 against 12.1)
    hb40 = [emu.frame_for_convert(imgs[0], 120, 40, 2)]       # half blocks (40 text rows): twenty blocks a frame on the four-slot rows geometry
    assert choose(5, hb40 * 256) == (26, 1, 40)               # a frame per CU: ONE sixteen-wave workgroup of the rows kernel (round 6 audit: level with the phase kernel at 256 frames, 5-8 % ahead at 128-192)
    assert choose(8, hb40 * 256) == (4, 1, 40)                # ... the mono half-block mode too (26.1 vs 28.9)
    assert choose(5, hb40 * 256, cus=64) == (25, 1, 40)       # above a frame per CU: the rows kernel
    assert choose(5, hb40 * 128) == (26, 1, 40)               # half blocks are not cut into bands from half a frame per CU on (28.1 on geometry 26 vs 30.2 on the phase kernel, 32.1 as bands)
    assert choose(5, hb40 * 64)[1] > 1                        # (64 frames: bands, 14.0 vs 28.4)
    m90 = [emu.frame_for_convert(imgs[0], 320, 90, 0)]
    assert choose(0, m90 * 256) == (26, 1, 90)                # wide mono rows: seven slots, sixteen waves (round 5: 59.5 us against 81.9 on eight)
    assert choose(0, m90 * 300) == (24, 1, 90)                # ... more than a frame per CU: two eight-wave workgroups per CU
    # ... and what the audit with launches IN FLIGHT (bench.py's schedule, a share of 64 CUs per plan) added: on a shared GPU mono
    # frames of several blocks per wave and the half-block modes' one-block frames take the phase kernel up to a frame per CU of
    # the share; above two frames per CU mono rows of at most 256 cells take the four-slot rows geometry whatever it wastes
    m70 = [emu.frame_for_convert(imgs[0], 238, 70, 0)]
    assert choose(0, m70 * 64, cus=64) == (26, 1, 70) and choose(0, m70 * 256) == (26, 1, 70)  # (round 5: 11.2 vs 13.8 us shared; 43.2 vs 48.4 alone)
    assert choose(0, m70 * 128, cus=64) == (25, 1, 70)
    assert choose(5, hb * 64, cus=64) == (4, 1, 24) and choose(5, hb * 256) == (25, 1, 24)     # 5.9 vs 6.9 shared; 21.2 vs 23.1 alone
    m45 = [emu.frame_for_convert(imgs[0], 160, 45, 0)]
    assert choose(0, m45 * 256, cus=64) == (25, 1, 45) and choose(0, m45 * 128, cus=64) == (24, 1, 45)  # 14.3 vs 17.3; 9.0 vs 10.6
    k3 = [emu.frame_for_convert(imgs[0], 200, 60, 0)]
    assert choose(1, k3 * 64) == (1, 6, 10)                   # bands limited by the 2048-cell chunk
    k5 = [emu.frame_for_convert(imgs[0], 400, 120, 2)]        # half-block: 120 text rows of 400 cells
    assert choose(5, k5 * 32) == (4, 24, 5)                   # few frames: row bands on the phase kernel
    # BASELINE configs[4]: one 400-cell row per 448-slot block when launches share the GPU (or frames outnumber CUs); ONE
    # launch of a frame per CU stays on the phase kernel (sixteen waves per CU against the 7-slot geometry's eight)
    assert choose(5, k5 * 300) == (24, 1, 120) and choose(5, k5 * 256, cus=64) == (24, 1, 120)
    assert choose(5, k5 * 256) == (26, 1, 120)               # (round 5: from sources up to 1080p one sixteen-wave workgroup per frame)
    k5_4k = [emu.frame_for_convert(np.zeros((2160, 3840, 3), np.uint8), 400, 120, 2)]
    assert choose(5, k5_4k * 256) == (4, 1, 120)             # BASELINE configs[4] itself, a frame per CU from 4K sources: the phase kernel
    assert choose(5, k5_4k * 192) == (26, 1, 120)            # ... up to three quarters of a frame per CU: 219 against 245 us
    wide = [emu.frame_for_convert(imgs[0], 449, 20, 2)]
    assert choose(5, wide * 256) == (27, 1, 20) and choose(5, wide * 257) == (29, 1, 20)  # a row wider than a block: cut into segments (round 6)
    assert choose(5, wide * 256, forced=4) == (4, 1, 20)
    wide = [emu.frame_for_convert(imgs[0], 3000, 4, 0)]
    assert choose(0, wide) == (0, 1, 4)                       # rows wider than the band geometries: no split
    assert L.achip_palette_ascii_only(orc.PALETTE_STANDARD.encode()) and not L.achip_palette_ascii_only(orc.PALETTE_COOL.encode())
