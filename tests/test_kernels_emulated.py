"""Runs the unmodified HIP kernel source under the CPU fiber emulator and checks it byte-for-byte
against the oracle.  (The -m gpu tests repeat these cases on the real MI355X through the C-ABI.)"""
import numpy as np
import pytest

import emu
import orc
from achip_ctypes import (ALL_MODES, MODE_16_FG, MODE_256_FG, MODE_CAPS, MODE_HB_TRUE, MODE_MONO, MODE_NAMES,
                          MODE_TRUE_BG, MODE_TRUE_FG)


def oracle_convert(img, mode, W, H, palette, wants_padding=False, use_aspect=False):
    if mode == MODE_TRUE_BG:
        assert not use_aspect
        rows = H
        rs = orc.resize_nn(img, W, rows)
        return orc.print_truecolor_bg(rs, palette)
    cl, rm = MODE_CAPS[mode]
    return orc.convert_with_caps(img, W, H, cl, rm, wants_padding, use_aspect, False, palette)


def emu_convert(img, mode, W, H, palette, variant, wants_padding=False, use_aspect=False):
    rm = MODE_CAPS.get(mode, (3, 0))[1]
    f = emu.frame_for_convert(img, W, H, rm, wants_padding, use_aspect)
    return emu.render_frames(mode, [f], palette, variant)[0]


def emu_convert_parts(img, mode, W, H, palette, variant, parts, wants_padding=False, use_aspect=False):
    rm = MODE_CAPS.get(mode, (3, 0))[1]
    f = emu.frame_for_convert(img, W, H, rm, wants_padding, use_aspect)
    return emu.render_frames(mode, [f], palette, variant, parts=parts)[0]


TORTURE = orc.frame_torture()


@pytest.mark.parametrize("mode", ALL_MODES, ids=MODE_NAMES)
@pytest.mark.parametrize("variant", [3, 2])
def test_torture_all_modes(mode, variant):
    # Appendix-B torture image: transparent runs, REP runs, gradient, noise
    for (W, H) in [(80, 24), (97, 31)]:
        exp = oracle_convert(TORTURE, mode, W, H, orc.PALETTE_STANDARD)
        got = emu_convert(TORTURE, mode, W, H, orc.PALETTE_STANDARD, variant)
        assert got == exp, (MODE_NAMES[mode], W, H)


@pytest.mark.parametrize("mode", [MODE_MONO, MODE_TRUE_FG, MODE_HB_TRUE], ids=["mono", "true_fg", "hb_true"])
def test_aspect_and_padding(mode):
    for (W, H) in [(80, 24), (97, 31), (60, 40)]:
        exp = oracle_convert(TORTURE, mode, W, H, orc.PALETTE_STANDARD, True, True)
        got = emu_convert(TORTURE, mode, W, H, orc.PALETTE_STANDARD, 3, True, True)
        assert got == exp, (MODE_NAMES[mode], W, H)


@pytest.mark.parametrize("palette", [orc.PALETTE_BLOCKS, orc.PALETTE_COOL, orc.PALETTE_DIGITAL, orc.PALETTE_MINIMAL,
                                     "ab", "x", "é漢😀 ."], ids=["blocks", "cool", "digital", "minimal", "ab", "x", "mixed"])
@pytest.mark.parametrize("mode", [MODE_MONO, MODE_TRUE_FG, MODE_256_FG, MODE_16_FG, MODE_TRUE_BG],
                         ids=["mono", "true_fg", "256_fg", "16_fg", "true_bg"])
def test_palettes(palette, mode):
    exp = oracle_convert(TORTURE, mode, 61, 17, palette)
    got = emu_convert(TORTURE, mode, 61, 17, palette, 3)
    assert got == exp


def test_wide_variant_big_frame():
    img = orc.frame_hash_noise(320, 200, 7)
    for mode in (MODE_TRUE_FG, MODE_HB_TRUE, MODE_MONO):
        exp = oracle_convert(img, mode, 200, 60, orc.PALETTE_STANDARD)
        got = emu_convert(img, mode, 200, 60, orc.PALETTE_STANDARD, 0)
        assert got == exp, MODE_NAMES[mode]


def test_512_thread_geometry_multi_chunk_frames():
    """Geometry 1 (512 threads x 4 cells) is what plans pick with several launches in flight; it requests the first chunk's
    samples in the prologue and every later chunk's at the top of the chunk loop (no request-ahead): frames of 1, 2 and
    6 chunks, padded and not."""
    img = orc.frame_hash_noise(320, 200, 17)
    for mode in (MODE_TRUE_FG, MODE_256_FG, MODE_HB_TRUE, MODE_MONO):
        for (W, H, pad) in ((80, 24, False), (97, 31, True), (200, 60, False)):
            exp = oracle_convert(img, mode, W, H, orc.PALETTE_STANDARD, pad, pad)
            got = emu_convert(img, mode, W, H, orc.PALETTE_STANDARD, 1, pad, pad)
            assert got == exp, (MODE_NAMES[mode], W, H)


def test_dither_multi_sweep_and_carry():
    """Floyd-Steinberg mode: > 64 rows per chunk (several 64-row sweeps chained through the LDS carry),
    chunk-to-chunk carry, 1-pixel-wide and 1-row images, padding."""
    from achip_ctypes import MODE_16_DITHER_BG
    img = orc.frame_hash_noise(97, 211, 21)
    img[50:120, 20:60] = 37  # flat area: error accumulates without clamping
    for (W, H, variant) in [(10, 150, 4), (7, 200, 0), (1, 130, 4), (130, 1, 2), (64, 65, 4), (300, 20, 2), (80, 24, 1)]:
        exp = oracle_convert(img, MODE_16_DITHER_BG, W, H, orc.PALETTE_STANDARD)
        got = emu_convert(img, MODE_16_DITHER_BG, W, H, orc.PALETTE_STANDARD, variant)
        assert got == exp, (W, H, variant)
    exp = oracle_convert(img, MODE_16_DITHER_BG, 60, 40, orc.PALETTE_COOL, True, True)
    got = emu_convert(img, MODE_16_DITHER_BG, 60, 40, orc.PALETTE_COOL, 4, True, True)
    assert got == exp


def test_dither_foreground_only_forms():
    """image_print_16color_dithered_with_background(.., false, ..) and image_print_16color_dithered (exported,
    image.h:462,488; no dispatcher reaches them): one fg SGR per cell, glyph cache[Y] / cache[ramp[Y>>2]]."""
    import ctypes as C

    from achip_ctypes import MODE_16_DITHER_BG
    for (W, H, variant, pal) in [(61, 23, 4, orc.PALETTE_STANDARD), (10, 150, 0, orc.PALETTE_BLOCKS),
                                 (130, 3, 2, "é漢😀 .")]:
        img = orc.resize_nn(orc.frame_hash_noise(97, 211, 5), W, H)
        for bgm, ramp in ((True, False), (False, False), (False, True)):
            f = emu.frame_for_convert(img, W, H, 0)
            assert emu.lib().achip_frame_set_dither_style(C.byref(f), bgm, ramp) == 0
            assert emu.lib().achip_frame_set_display_ops(C.byref(f), False, False, 0) == 0  # keeps the style bits
            got = emu.render_frames(MODE_16_DITHER_BG, [f], pal, variant)[0]
            assert got == orc.print_16_dithered(img, bgm, pal, ramp_glyph=ramp), (W, H, bgm, ramp)
    f = emu.frame_for_convert(img, W, H, 0)
    assert emu.lib().achip_frame_set_dither_style(C.byref(f), True, True) == -1


def test_rainbow_filter_folded_into_emission():
    """COLOR_FILTER_RAINBOW (display.c:639-650): rainbow_replace_ansi_colors over the finished frame == the frame
    rendered with every ESC[38;2;..m carrying the colour of the moment."""
    import ctypes as C

    from achip_ctypes import MODE_16_DITHER_BG, MODE_HB_256
    img = orc.frame_torture()
    for t in (0.0, 0.4, 1.3, 2.05, 3.49, 1234.567):
        for mode, pal in ((MODE_TRUE_FG, orc.PALETTE_STANDARD), (MODE_TRUE_FG, "é漢😀 .m"), (MODE_HB_TRUE, orc.PALETTE_STANDARD),
                          (MODE_TRUE_BG, orc.PALETTE_STANDARD), (MODE_256_FG, orc.PALETTE_STANDARD),
                          (MODE_HB_256, orc.PALETTE_STANDARD), (MODE_MONO, orc.PALETTE_STANDARD)):
            rm = MODE_CAPS.get(mode, (3, 0))[1]
            pad = mode != MODE_TRUE_BG
            f = emu.frame_for_convert(img, 97, 31, rm, pad, pad)
            assert emu.lib().achip_frame_set_display_ops(C.byref(f), True, False, 3) == 0  # the tint is dropped, the flip kept
            assert emu.lib().achip_frame_set_rainbow(C.byref(f), t) == 0
            got = emu.render_frames(mode, [f], pal, 2)[0]
            plain = oracle_convert(np.ascontiguousarray(img[:, ::-1]), mode, 97, 31, pal, pad, pad)
            assert got == orc.rainbow_replace(plain, t), (t, MODE_NAMES[mode])
    # an all-black half-block frame holds no foreground SGR: nothing to replace
    black = np.zeros((40, 40, 3), np.uint8)
    f = emu.frame_for_convert(black, 20, 10, 2)
    emu.lib().achip_frame_set_rainbow(C.byref(f), 1.0)
    assert emu.render_frames(MODE_HB_TRUE, [f], orc.PALETTE_STANDARD, 2)[0] == oracle_convert(black, MODE_HB_TRUE, 20, 10,
                                                                                             orc.PALETTE_STANDARD)


def test_uniform_batch_descriptor_by_value():
    """Batches whose descriptors differ only by a constant source pitch (and single frames) get the common descriptor
    with the kernel arguments (achip_frames_uniform): same bytes as through the descriptor array, in every mode,
    whole-frame and row-band launches; anything else falls back to the array."""
    import ctypes as C

    from achip_ctypes import Uniform
    L = emu.lib()
    slab = np.ascontiguousarray(np.stack([orc.frame_hash_noise(96, 64, 30 + k) for k in range(3)]))
    frames = [emu.frame_for_convert(slab[k], 40, 12, 0) for k in range(3)]
    u = Uniform()
    assert L.achip_frames_uniform((emu.Frame * 3)(*frames), 3, C.byref(u)) == 1 and u.src_pitch == 96 * 64 * 3
    assert L.achip_frames_uniform((emu.Frame * 1)(frames[2]), 1, C.byref(u)) == 1 and u.src_pitch == 0
    odd = [frames[0], frames[2], frames[1]]                       # not an arithmetic progression of sources
    assert L.achip_frames_uniform((emu.Frame * 3)(*odd), 3, C.byref(u)) == 0 and u.enabled == 0
    other = emu.frame_for_convert(slab[1], 41, 12, 0)             # a differing field
    assert L.achip_frames_uniform((emu.Frame * 2)(frames[0], other), 2, C.byref(u)) == 0
    back = [frames[2], frames[1], frames[0]]                      # negative pitch is a pitch too
    assert L.achip_frames_uniform((emu.Frame * 3)(*back), 3, C.byref(u)) == 1 and u.src_pitch == -96 * 64 * 3
    for mode in ALL_MODES:
        rm = MODE_CAPS.get(mode, (3, 0))[1]
        fs = [emu.frame_for_convert(slab[k], 40, 12, rm) for k in range(3)]
        exp = [oracle_convert(slab[k], mode, 40, 12, orc.PALETTE_STANDARD) for k in range(3)]
        assert emu.render_frames(mode, fs, orc.PALETTE_STANDARD, 2, uniform=True) == exp, MODE_NAMES[mode]
        assert emu.render_frames(mode, fs[::-1], orc.PALETTE_STANDARD, 2, uniform=True) == exp[::-1]
        assert emu.render_frames(mode, [fs[0], fs[2], fs[1]], orc.PALETTE_STANDARD, 2, uniform=True) == [exp[0], exp[2], exp[1]]
    fs = [emu.frame_for_convert(slab[k], 40, 12, 2, True, True) for k in range(3)]
    exp = [oracle_convert(slab[k], MODE_HB_TRUE, 40, 12, orc.PALETTE_STANDARD, True, True) for k in range(3)]
    assert emu.render_frames(MODE_HB_TRUE, fs, orc.PALETTE_STANDARD, 2, rows_per_part=2, uniform=True) == exp


def test_display_prepasses_folded_into_sampler():
    """flip_x / flip_y / colour filters of the client display path (display.c:546-623) as sampler maps."""
    import ctypes as C

    import numpy as np
    img = orc.frame_torture()
    # the oracle's flips agree with plain array reversal; odd sizes keep the middle column/row
    assert np.array_equal(orc.flip(img, True, False), img[:, ::-1])
    assert np.array_equal(orc.flip(img, False, True), img[::-1])
    assert np.array_equal(orc.flip(img[:1], True, True), img[:1])  # h == 1: flips are skipped (display.c:549)
    # reference filter semantics on known points: grey = (77R+150G+29B)>>8, channel = tint*grey/255
    px = np.array([[[255, 255, 255], [0, 0, 0], [10, 200, 30]]], dtype=np.uint8)
    assert orc.color_filter(px, 3).tolist() == [[[0, 255, 65], [0, 0, 0], [0, 123, 31]]]      # green (0,255,65)
    assert orc.color_filter(px, 1).tolist() == [[[255, 255, 255], [0, 0, 0], [123, 123, 123]]]  # black-on-white
    for mode, (cl, rm) in ((1, (3, 0)), (5, (3, 2)), (0, (0, 0)), (2, (2, 0))):
        for fx, fy, flt in ((True, False, 0), (False, True, 0), (True, True, 3), (False, False, 1), (True, False, 9),
                            (False, True, 11)):
            f = emu.frame_for_convert(img, 97, 31, rm, True, True)
            assert emu.lib().achip_frame_set_display_ops(C.byref(f), fx, fy, flt) == 0
            got = emu.render_frames(mode, [f], orc.PALETTE_STANDARD, 2)[0]
            exp = orc.display_convert(img, 97, 31, cl, rm, True, True, fx, fy, flt)
            assert got == exp, (mode, fx, fy, flt)
    f = emu.frame_for_convert(img, 97, 31, 0)
    assert emu.lib().achip_frame_set_display_ops(C.byref(f), False, False, 12) == -1  # rainbow: not a pixel pre-pass


def test_tall_padding_crosses_threads_groups():
    """pad_top in the thousands (a terminal far taller than the fitted frame): the newline fill of one thread lands in
    staging groups that other threads zero for the OR-filled half-block buffer -- ordered by a barrier."""
    img = orc.frame_hash_noise(7, 2, 3)
    for (W, H, mode) in ((1, 10001, 5), (80, 4000, 1), (3, 9000, 0), (40, 2500, 6)):
        cl, rm = MODE_CAPS[mode]
        exp = orc.convert_with_caps(img, W, H, cl, rm, True, True, False, "@")
        f = emu.frame_for_convert(img, W, H, rm, True, True)
        for variant in (4, 3):
            assert emu.render_frames(mode, [f], "@", variant)[0] == exp, (W, H, mode, variant)


# --------------------------------------------------------------------------------------------------------------- #
# the wave-autonomous stream kernel (render_stream.hpp): per-cell modes, whole frames.  Geometry 20 (2 waves x 1    #
# cell per lane, emulator builds only) cuts every frame into many 64-cell blocks, so look-back windows, the         #
# request-ahead loop and blocks that straddle rows all run on small inputs; 16 / 17 / 19 are product geometries.    #
# --------------------------------------------------------------------------------------------------------------- #
STREAM_MODES = [MODE_TRUE_FG, MODE_256_FG, MODE_16_FG, MODE_TRUE_BG]


@pytest.mark.parametrize("mode", STREAM_MODES, ids=["true_fg", "256_fg", "16_fg", "true_bg"])
@pytest.mark.parametrize("variant", [20, 16, 17, 19])
def test_stream_kernel_torture(mode, variant):
    for (W, H) in [(80, 24), (97, 31), (3, 2), (1, 1), (64, 1), (65, 3)]:
        exp = oracle_convert(TORTURE, mode, W, H, orc.PALETTE_STANDARD)
        got = emu_convert(TORTURE, mode, W, H, orc.PALETTE_STANDARD, variant)
        assert got == exp, (MODE_NAMES[mode], W, H, variant)


@pytest.mark.parametrize("mode", [MODE_TRUE_FG, MODE_256_FG, MODE_16_FG], ids=["true_fg", "256_fg", "16_fg"])
def test_stream_kernel_aspect_padding_and_many_blocks(mode):
    # left / top padding (the raster predecessor of a row's first pixel sits pad_left cells back) and a frame of
    # 200x60 = 188 blocks in geometry 20: look-back windows beyond 64 predecessors
    for (W, H, variant) in [(80, 24, 20), (97, 31, 17), (60, 40, 20), (200, 60, 20), (200, 60, 16)]:
        exp = oracle_convert(TORTURE, mode, W, H, orc.PALETTE_STANDARD, True, True)
        got = emu_convert(TORTURE, mode, W, H, orc.PALETTE_STANDARD, variant, True, True)
        assert got == exp, (MODE_NAMES[mode], W, H, variant)


@pytest.mark.parametrize("palette", [orc.PALETTE_BLOCKS, orc.PALETTE_COOL, "ab", "x", "é漢😀 ."],
                         ids=["blocks", "cool", "ab", "x", "mixed"])
def test_stream_kernel_palettes(palette):
    # multi-byte glyphs in every per-cell mode (truecolor-fg with such a palette: the instantiation of its own, round 6)
    for mode in (MODE_256_FG, MODE_16_FG, MODE_TRUE_BG, MODE_TRUE_FG):
        exp = oracle_convert(TORTURE, mode, 61, 17, palette)
        assert emu_convert(TORTURE, mode, 61, 17, palette, 20) == exp, (MODE_NAMES[mode], palette)


@pytest.mark.parametrize("palette", [orc.PALETTE_BLOCKS, orc.PALETTE_COOL, orc.PALETTE_DIGITAL, "é漢😀 .m", "a█", "██ "],
                         ids=["blocks", "cool", "digital", "mixed", "two", "mostly_multibyte"])
@pytest.mark.parametrize("variant", [20, 16, 17])
def test_stream_kernel_truecolor_with_multibyte_palettes(palette, variant):
    """foreground.c:281-296: a cell whose glyph is not one ASCII byte always carries its SGR and leaves the RLE state alone,
    so an ASCII cell is compared with the nearest EARLIER ASCII cell -- across rows, blocks (the second look-back of
    ACHIP_STREAM_MODE_TRUE_FG_U8) and long stretches without one.  Both cell orders (sources a cache line apart / closer),
    padding, flips + the rainbow override, frames of many blocks, flat areas whose colour returns after a multi-byte stretch."""
    import ctypes as C
    img = orc.frame_torture()
    wide = orc.frame_hash_noise(1920, 54, 11)
    wide[:, 600:1400] = (3, 3, 3)            # dark flat band: ASCII cells (the palettes' leading spaces) of one colour ...
    wide[10:30, 800:1000] = (250, 250, 250)  # ... interrupted by a bright (multi-byte) island: the colour comes back behind it
    flat = np.zeros((40, 97, 3), np.uint8)
    flat[:, :] = (200, 220, 240)             # bright: no ASCII cell at all in most palettes
    flat[20:, 50:] = (1, 2, 3)
    far = np.full((50, 120, 3), 240, np.uint8)  # 95 blocks of geometry 20 between two dark corners: the look-back walks more
    far[0, 0:3] = (2, 2, 2)                     # than 64 words back (and the other way round in the inverted palettes)
    far[-1, -5:] = (2, 2, 2)
    cases = [(img, 80, 24), (img, 97, 31), (wide, 80, 24), (wide, 200, 12), (flat, 97, 40), (img, 200, 60), (img, 3, 2), (img, 1, 1),
             (far, 120, 50), (255 - far, 120, 50)]
    for im, W, H in cases:
        if variant != 20 and W * H > 6000:
            continue
        exp = oracle_convert(im, MODE_TRUE_FG, W, H, palette)
        assert emu_convert(im, MODE_TRUE_FG, W, H, palette, variant) == exp, (palette, W, H, variant)
    for im, W, H in [(img, 80, 24), (wide, 61, 17)]:
        exp = oracle_convert(im, MODE_TRUE_FG, W, H, palette, True, True)
        assert emu_convert(im, MODE_TRUE_FG, W, H, palette, variant, True, True) == exp, (palette, W, H, variant, "padded")
    f = emu.frame_for_convert(wide, 80, 24, 0)
    assert emu.lib().achip_frame_set_display_ops(C.byref(f), True, True, 0) == 0
    assert emu.lib().achip_frame_set_rainbow(C.byref(f), 1.3) == 0
    plain = oracle_convert(np.ascontiguousarray(wide[::-1, ::-1]), MODE_TRUE_FG, 80, 24, palette)
    assert emu.render_frames(MODE_TRUE_FG, [f], palette, variant)[0] == orc.rainbow_replace(plain, 1.3), (palette, variant)
    # a ragged batch, and the batch's descriptor by value
    frames = [emu.frame_for_convert(im, W, H, 0) for im, W, H in cases[:4]]
    got = emu.render_frames(MODE_TRUE_FG, frames, palette, variant)
    for k, (im, W, H) in enumerate(cases[:4]):
        assert got[k] == oracle_convert(im, MODE_TRUE_FG, W, H, palette), (palette, variant, k)
    uni = emu.render_frames(MODE_TRUE_FG, [frames[2]] * 3, palette, variant, uniform=True)
    assert uni[0] == uni[1] == uni[2] == got[2]


def test_stream_kernel_ragged_batch_flips_tint_and_overflow():
    import ctypes as C
    imgs = [orc.frame_hash_noise(120, 90, i) for i in range(5)] + [orc.frame_bars(64, 48, 3), orc.frame_smooth(33, 17)]
    dims = [(80, 24), (60, 7), (33, 40), (80, 1), (1, 50), (17, 9), (128, 2)]
    frames = [emu.frame_for_convert(im, w, h, 0) for im, (w, h) in zip(imgs, dims)]
    for mode in (MODE_TRUE_FG, MODE_256_FG):
        for variant in (20, 17):
            got = emu.render_frames(mode, frames, orc.PALETTE_STANDARD, variant)
            for k, (im, (w, h)) in enumerate(zip(imgs, dims)):
                assert got[k] == oracle_convert(im, mode, w, h, orc.PALETTE_STANDARD), (mode, variant, k)
            uni = emu.render_frames(mode, [frames[0]] * 3, orc.PALETTE_STANDARD, variant, uniform=True)
            assert uni[0] == uni[1] == uni[2] == got[0]
    # the display path's flips and colour filter folded into the sampler, checked against the oracle's
    # full-frame passes
    img = imgs[0]
    for fx, fy, flt in [(True, False, 0), (False, True, 3), (True, True, 7)]:
        f = emu.frame_for_convert(img, 80, 24, 0)
        assert emu.lib().achip_frame_set_display_ops(C.byref(f), fx, fy, flt) == 0
        exp = orc.display_convert(img, 80, 24, 3, 0, False, False, fx, fy, flt)
        assert emu.render_frames(MODE_TRUE_FG, [f], orc.PALETTE_STANDARD, 20)[0] == exp, (fx, fy, flt)
    # a slot that is too small reports ACHIP_LEN_OVERFLOW (and nothing is written past it)
    f = emu.frame_for_convert(img, 80, 24, 0)
    got = emu.render_frames(MODE_TRUE_FG, [f], orc.PALETTE_STANDARD, 20, stride=1024)
    assert got[0] == 0xFFFFFFFF


@pytest.mark.parametrize("variant", [20, 16, 17, 19])
def test_stream_kernel_lean_loop_orders_and_samplers(variant):
    """The lean loop of the truecolor-foreground instantiations (render_stream.hpp, round 6) has two cell orders -- lane-major
    for sources whose samples share cache lines, slot-major for samples a line apart or more (the non-temporal copy) -- and
    two samplers -- ratio 1.0 (a sampled image: no multiplications) and the split 16.16 multiply.  Every combination, with
    flips, the colour filter, the rainbow override, padding, runs of equal pixels (no SGR), and frames whose last block is
    nearly empty, against the oracle."""
    import ctypes as C
    rng = np.random.default_rng(7)
    wide = orc.frame_hash_noise(1920, 54, 11)                    # 1920 -> 80 columns: samples 72 bytes apart (slot-major)
    wide[:, 700:1300] = wide[20, 700]                            # a flat band: cells without an SGR
    dense = orc.frame_hash_noise(97, 31, 5)                      # rendered at its own size: ratio 1.0
    dense[7:9, :] = (9, 200, 31)
    dense[8, 40:] = (255, 255, 255)
    cases = [(wide, 80, 24), (wide, 61, 17), (wide, 3, 2), (dense, 97, 31), (orc.frame_hash_noise(80, 24, 3), 80, 24),
             (TORTURE, 200, 60), (orc.frame_hash_noise(4000, 8, 2), 130, 8)]
    for img, W, H in cases:
        f = emu.frame_for_convert(img, W, H, 0)
        exp = oracle_convert(img, MODE_TRUE_FG, W, H, orc.PALETTE_STANDARD)
        assert emu.render_frames(MODE_TRUE_FG, [f], orc.PALETTE_STANDARD, variant)[0] == exp, (W, H, variant)
        assert emu.render_frames(MODE_TRUE_FG, [f, f], orc.PALETTE_STANDARD, variant, uniform=True)[1] == exp, (W, H, variant, "uniform")
    for img, W, H in [(wide, 80, 24), (dense, 97, 31)]:  # aspect + padding
        exp = oracle_convert(img, MODE_TRUE_FG, W, H, orc.PALETTE_STANDARD, True, True)
        assert emu_convert(img, MODE_TRUE_FG, W, H, orc.PALETTE_STANDARD, variant, True, True) == exp, (W, H, variant, "padded")
    for img, W, H in [(wide, 80, 24), (dense, 97, 31)]:  # display ops folded into the sampler / the emission
        for fx, fy, flt in [(True, False, 0), (False, True, 3), (True, True, 7)]:
            f = emu.frame_for_convert(img, W, H, 0)
            assert emu.lib().achip_frame_set_display_ops(C.byref(f), fx, fy, flt) == 0
            exp = orc.display_convert(img, W, H, 3, 0, False, False, fx, fy, flt)
            assert emu.render_frames(MODE_TRUE_FG, [f], orc.PALETTE_STANDARD, variant)[0] == exp, (W, H, variant, fx, fy, flt)
        f = emu.frame_for_convert(img, W, H, 0)  # the rainbow override: every SGR carries the colour of the moment
        assert emu.lib().achip_frame_set_display_ops(C.byref(f), False, True, 0) == 0
        assert emu.lib().achip_frame_set_rainbow(C.byref(f), 2.05) == 0
        plain = oracle_convert(np.ascontiguousarray(img[::-1]), MODE_TRUE_FG, W, H, orc.PALETTE_STANDARD)
        assert emu.render_frames(MODE_TRUE_FG, [f], orc.PALETTE_STANDARD, variant)[0] == orc.rainbow_replace(plain, 2.05), (W, H, variant)
    if variant in (20, 17):  # ... and shared out over workgroups (the ghost of a part's first block is another part's cell)
        for img, W, H, parts in [(wide, 80, 24, 4), (dense, 97, 31, 7), (wide, 61, 17, 16)]:
            exp = oracle_convert(img, MODE_TRUE_FG, W, H, orc.PALETTE_STANDARD)
            assert emu_convert_parts(img, MODE_TRUE_FG, W, H, orc.PALETTE_STANDARD, variant if variant == 20 else 18, parts) == exp, (W, H, parts)
    del rng


@pytest.mark.parametrize("variant", [20, 16, 17])
def test_stream_kernel_length_first_exact_length_frames(variant):
    """LENGTH-FIRST (render_stream.hpp LF; lib/network/acip/server.c:190-222 ships exactly frame_size bytes): frames of any
    size leave ONE launch at their exact lengths -- the lean loop runs twice, lengths first.  Every frame against the oracle;
    frames tile the destination in 16-byte-rounded pieces (completion order), the padding bytes are zero, nothing is written
    behind the total; both cell orders, padding, a ragged batch, frames far beyond the 48 KB of the LDS-image form; a frame
    that does not fit its bound and a destination that is too small are reported; the cursor re-arms."""
    wide = orc.frame_hash_noise(1920, 54, 11)
    dense = orc.frame_hash_noise(200, 60, 5)
    imgs = [TORTURE, wide, dense, orc.frame_bars(64, 48, 3), orc.frame_smooth(33, 17), orc.frame_hash_noise(80, 24, 9)]
    dims = [(97, 31), (80, 24), (200, 60), (17, 9), (33, 40), (80, 24)]
    if variant == 20:
        imgs, dims = imgs + [TORTURE], dims + [(200, 60)]
    for pad in (False, True):
        frames = [emu.frame_for_convert(im, w, h, 0, pad, pad) for im, (w, h) in zip(imgs, dims)]
        want = [oracle_convert(im, MODE_TRUE_FG, w, h, orc.PALETTE_STANDARD, pad, pad) for im, (w, h) in zip(imgs, dims)]
        cur = np.zeros(2, dtype=np.uint64)
        for rep in range(2):  # the second launch runs on the cursor words the first one left
            r = emu.render_frames_length_first(frames, orc.PALETTE_STANDARD, variant, cursor=cur)
            n = len(frames)
            spans = sorted((int(r["off"][i]), int(r["off"][i]) + (len(want[i]) + 15) // 16 * 16) for i in range(n))
            assert spans[0][0] == 0 and all(spans[i][1] == spans[i + 1][0] for i in range(n - 1)) and spans[-1][1] == int(r["off"][n])
            for i in range(n):
                o = int(r["off"][i])
                assert int(r["lens"][i]) == int(r["plen"][i]) == len(want[i]), (variant, pad, i)
                assert r["dst"][o:o + len(want[i])].tobytes() == want[i], (variant, pad, i)
                assert not r["dst"][o + len(want[i]):o + (len(want[i]) + 15) // 16 * 16].any()
            assert (r["dst"][int(r["off"][n]):int(r["off"][n]) + 16] == 0xEE).all()
            assert int(cur[0]) == 0 and int(cur[1]) == 0
    # the batch's descriptor by value
    f = emu.frame_for_convert(dense, 200, 60, 0)
    r = emu.render_frames_length_first([f] * 3, orc.PALETTE_STANDARD, variant, uniform=True)
    exp = oracle_convert(dense, MODE_TRUE_FG, 200, 60, orc.PALETTE_STANDARD)
    for i in range(3):
        o = int(r["off"][i])
        assert r["dst"][o:o + len(exp)].tobytes() == exp
    # a frame beyond its bound, and a destination that cannot hold every frame
    r = emu.render_frames_length_first([f, emu.frame_for_convert(TORTURE, 17, 9, 0)], orc.PALETTE_STANDARD, variant, stride=4096)
    assert int(r["lens"][0]) == 0xFFFFFFFF and int(r["lens"][1]) == len(oracle_convert(TORTURE, MODE_TRUE_FG, 17, 9, orc.PALETTE_STANDARD))
    r = emu.render_frames_length_first([f, f], orc.PALETTE_STANDARD, variant, capacity=(len(exp) + 15) // 16 * 16 + 64)
    assert sorted(int(x) for x in r["lens"]) == [len(exp), 0xFFFFFFFF]


@pytest.mark.parametrize("mode", STREAM_MODES, ids=["true_fg", "256_fg", "16_fg", "true_bg"])
def test_stream_kernel_frames_shared_out_over_workgroups(mode):
    """PARTS instantiations (render_stream.hpp): a frame's blocks shared out over several workgroups that hand their
    byte counts to each other through global words -- what small launches (a lone frame, the grid's nine targets) use in
    geometry 18.  Part counts that divide the blocks, that do not, and that exceed them (parts without a block only report
    in); padding; a ragged batch; an overflowing slot; the same words launched on again (a new epoch each time)."""
    import ctypes as C
    for (W, H, variant, parts) in [(80, 24, 18, 4), (80, 24, 18, 3), (80, 24, 20, 7), (97, 31, 20, 16), (3, 2, 18, 2), (1, 1, 20, 5),
                                   (160, 48, 18, 16), (200, 60, 20, 64)]:
        exp = oracle_convert(TORTURE, mode, W, H, orc.PALETTE_STANDARD)
        f = emu.frame_for_convert(TORTURE, W, H, 0)
        got = emu.render_frames(mode, [f], orc.PALETTE_STANDARD, variant, parts=parts)[0]
        assert got == exp, (MODE_NAMES[mode], W, H, variant, parts)
    # aspect + padding (pad_top newlines come from part 0 alone; the ghost of a part's first block is another part's cell)
    for (W, H, variant, parts) in [(80, 24, 18, 4), (60, 40, 20, 9)] if mode != MODE_TRUE_BG else []:
        exp = oracle_convert(TORTURE, mode, W, H, orc.PALETTE_STANDARD, True, True)
        got = emu_convert_parts(TORTURE, mode, W, H, orc.PALETTE_STANDARD, variant, parts, True, True)
        assert got == exp, (MODE_NAMES[mode], W, H, variant, parts, "padded")
    # a ragged batch on ONE set of words, launched three times (epochs), uniform and not
    imgs = [orc.frame_hash_noise(120, 90, i) for i in range(4)] + [orc.frame_bars(64, 48, 3)]
    dims = [(80, 24), (60, 7), (33, 40), (80, 1), (1, 50)]
    frames = [emu.frame_for_convert(im, w, h, 0) for im, (w, h) in zip(imgs, dims)]
    sync = np.zeros(len(frames) * 6, dtype=np.uint64)
    for _ in range(3):
        got = emu.render_frames(mode, frames, orc.PALETTE_STANDARD, 20, parts=6, sync=sync)
        for k, (im, (w, h)) in enumerate(zip(imgs, dims)):
            assert got[k] == oracle_convert(im, mode, w, h, orc.PALETTE_STANDARD), (mode, k)
    uni = emu.render_frames(mode, [frames[0]] * 3, orc.PALETTE_STANDARD, 18, uniform=True, parts=4)
    assert uni[0] == uni[1] == uni[2] == oracle_convert(imgs[0], mode, 80, 24, orc.PALETTE_STANDARD)
    # a slot that is too small: the overflow is found by whichever part holds the block that crosses the bound
    f = emu.frame_for_convert(imgs[0], 80, 24, 0)
    for stride in (1024, 4096, 16384):
        full = oracle_convert(imgs[0], mode, 80, 24, orc.PALETTE_STANDARD)
        got = emu.render_frames(mode, [f], orc.PALETTE_STANDARD, 18, stride=stride, parts=4)[0]
        assert got == (full if len(full) <= stride else 0xFFFFFFFF), (mode, stride)


@pytest.mark.parametrize("mode", STREAM_MODES, ids=["true_fg", "256_fg", "16_fg", "true_bg"])
def test_stream_kernel_fused_frame_crc(mode):
    """asciichat_crc32 of the frame (lib/network/crc32.c:95-190) computed inside the render kernel while the frame's
    bytes sit in LDS on their way out: bytes AND checksum must match the oracle -- blocks that straddle rows, top
    padding (newlines nobody stages), one-block frames, a ragged batch, and a slot that is too small (CRC 0)."""
    cases = [(80, 24, False, False, 20), (97, 31, False, False, 17), (3, 2, False, False, 20), (1, 1, False, False, 20),
             (200, 60, True, True, 20), (60, 40, True, True, 20), (64, 1, False, False, 17), (5, 300, True, True, 20)]
    if mode in (MODE_TRUE_FG, MODE_256_FG):
        cases.append((80, 24, False, False, 16))
    for (W, H, asp, pad, variant) in cases:
        if mode == MODE_TRUE_BG and asp:
            continue  # the background renderer is not reached through the aspect / padding front end
        exp = oracle_convert(TORTURE, mode, W, H, orc.PALETTE_STANDARD, pad, asp)
        f = emu.frame_for_convert(TORTURE, W, H, MODE_CAPS.get(mode, (3, 0))[1], pad, asp)
        got, crc = emu.render_frames_crc(mode, [f], orc.PALETTE_STANDARD, variant)
        assert got[0] == exp, (MODE_NAMES[mode], W, H, variant)
        assert crc[0] == orc.crc32c(exp), (MODE_NAMES[mode], W, H, variant, hex(crc[0]))
    imgs = [orc.frame_hash_noise(120, 90, i) for i in range(4)]
    dims = [(80, 24), (60, 7), (33, 40), (1, 50)]
    frames = [emu.frame_for_convert(im, w, h, 0) for im, (w, h) in zip(imgs, dims)]
    got, crc = emu.render_frames_crc(mode, frames, orc.PALETTE_STANDARD, 20)
    for k, (im, (w, h)) in enumerate(zip(imgs, dims)):
        exp = oracle_convert(im, mode, w, h, orc.PALETTE_STANDARD)
        assert got[k] == exp and crc[k] == orc.crc32c(exp), (mode, k)
    # the wave that finishes a frame also writes its ascii_frame_packet_t header and the CRC of header || frame
    # (acip_send_ascii_frame, lib/network/acip/server.c:186-214; packet_send_via_transport, send.c:59-69)
    for variant in (20, 17):
        got, crc, hdr, pkt = emu.render_frames_crc(mode, frames, orc.PALETTE_STANDARD, variant, dims=dims)
        for k, (w, h) in enumerate(dims):
            eh, ep = orc.ascii_frame_packet(got[k], w, h)
            assert crc[k] == orc.crc32c(got[k]) and hdr[k] == eh and pkt[k] == ep, (mode, variant, k)
    got, crc, hdr, pkt = emu.render_frames_crc(mode, [frames[0]], orc.PALETTE_STANDARD, 20, stride=1024, dims=[(80, 24)])
    eh, ep = orc.ascii_frame_packet(b"", 0, 0)  # a frame that did not fit: zeros in the header, CRC 0 inside it
    assert got[0] == 0xFFFFFFFF and crc[0] == 0 and hdr[0] == eh and pkt[0] == ep
    if mode != MODE_TRUE_FG:  # multi-byte glyphs: token lengths vary inside a block
        f = emu.frame_for_convert(TORTURE, 61, 17, MODE_CAPS.get(mode, (3, 0))[1])
        got, crc = emu.render_frames_crc(mode, [f], "é漢😀 .", 20)
        assert got[0] == oracle_convert(TORTURE, mode, 61, 17, "é漢😀 .") and crc[0] == orc.crc32c(got[0])
    got, crc = emu.render_frames_crc(mode, [frames[0]], orc.PALETTE_STANDARD, 20, stride=1024)
    assert got[0] == 0xFFFFFFFF and crc[0] == 0


def check_packed(res, expected, what):
    """frames at their exact lengths, 16-byte aligned starts, tiling [0, total) in SOME order; error frames take no room"""
    n = len(expected)
    spans = []
    for k, exp in enumerate(expected):
        o = int(res["off"][k])
        if isinstance(exp, int):  # a render error code
            assert int(res["plen"][k]) == exp and int(res["lens"][k]) == exp, (what, k)
            continue
        assert int(res["plen"][k]) == len(exp) == int(res["lens"][k]), (what, k, int(res["plen"][k]), len(exp))
        assert o % 16 == 0 and res["dst"][o:o + len(exp)].tobytes() == exp, (what, k)
        room = (len(exp) + 15) // 16 * 16
        assert not res["dst"][o + len(exp):o + room].any(), (what, k)  # the padding leaves as zeros
        spans.append((o, o + room))
    spans.sort()
    at = 0
    for a, b in spans:
        assert a == at, (what, spans)
        at = b
    assert int(res["off"][n]) == at, (what, int(res["off"][n]), at)
    assert not res["cursor"].any(), what  # re-armed for the plan's next launch


@pytest.mark.parametrize("mode", [MODE_TRUE_FG, MODE_256_FG, 3], ids=["true_fg", "256_fg", "16_fg"])
def test_stream_kernel_exact_length_frames(mode):
    """PACK instantiations of the stream kernel (acip_send_ascii_frame ships frame_size bytes,
    lib/network/acip/server.c:190-222): the frame is staged whole in LDS, claims its place with one atomic add and leaves at
    its exact length -- bytes, lengths, the tiling of the destination, fused CRCs / headers, the cursor re-armed, in the
    product geometry and in the two-wave test geometry (many blocks per wave), with top / left padding, a ragged batch, a
    frame that does not fit, a destination that is too small."""
    imgs = [orc.frame_hash_noise(120, 90, 40 + i) for i in range(5)] + [TORTURE]
    dims = [(80, 24), (60, 7), (33, 40), (1, 50), (3, 2), (80, 24)]
    frames = [emu.frame_for_convert(im, w, h, 0) for im, (w, h) in zip(imgs, dims)]
    expected = [oracle_convert(im, mode, w, h, orc.PALETTE_STANDARD) for im, (w, h) in zip(imgs, dims)]
    for variant in (20, 16, 17):
        for want_crc in (True, False):
            res = emu.render_frames_packed(mode, frames, orc.PALETTE_STANDARD, variant, dims=dims, want_crc=want_crc)
            check_packed(res, expected, (mode, variant, want_crc))
            if want_crc:
                for k, (w, h) in enumerate(dims):
                    eh, ep = orc.ascii_frame_packet(expected[k], w, h)
                    assert int(res["crc"][k]) == orc.crc32c(expected[k]), (mode, variant, k)
                    assert res["hdr"][24 * k:24 * k + 24].tobytes() == eh and int(res["pkt"][k]) == ep, (mode, variant, k)
            # again on the same cursor words: the last workgroup cleared them
            res2 = emu.render_frames_packed(mode, frames, orc.PALETTE_STANDARD, variant, dims=dims, want_crc=want_crc,
                                            cursor=res["cursor"])
            check_packed(res2, expected, (mode, variant, want_crc, "again"))
    # aspect + padding (newlines in front that no block stages, pad cells), the uniform-descriptor path
    padded = [emu.frame_for_convert(TORTURE, W, H, 0, True, True) for (W, H) in ((80, 24), (200, 20), (31, 60))]
    exp_p = [oracle_convert(TORTURE, mode, W, H, orc.PALETTE_STANDARD, True, True) for (W, H) in ((80, 24), (200, 20), (31, 60))]
    for variant in (20, 16, 17):
        check_packed(emu.render_frames_packed(mode, padded, orc.PALETTE_STANDARD, variant), exp_p, (mode, variant, "padded"))
    # a frame that does not fit its bound takes no room and reports the overflow code; the others are unaffected
    small = emu.render_frames_packed(mode, frames, orc.PALETTE_STANDARD, 20, stride=2048)
    exp_s = [e if len(e) <= 2048 else 0xFFFFFFFF for e in expected]
    assert any(isinstance(e, int) for e in exp_s) and any(not isinstance(e, int) for e in exp_s)
    check_packed(small, exp_s, (mode, "overflow"))
    assert int(small["crc"][0]) == 0
    # a destination too small for everything: frames whose place lies within it arrive, the total still tells
    total = sum((len(e) + 15) // 16 * 16 for e in expected)
    cut = emu.render_frames_packed(mode, frames, orc.PALETTE_STANDARD, 20, capacity=total // 2 // 16 * 16)
    assert int(cut["off"][len(frames)]) == total
    arrived = 0
    for k, e in enumerate(expected):
        o = int(cut["off"][k])
        if o + (len(e) + 15) // 16 * 16 <= total // 2 // 16 * 16:
            assert cut["dst"][o:o + len(e)].tobytes() == e
            arrived += 1
    assert 0 < arrived < len(frames)
    assert (cut["dst"][total // 2 // 16 * 16:total] == 0xEE).all()  # nothing stored at or behind the capacity


# --------------------------------------------------------------------------------------------------------------- #
# the rows kernel (render_rows.hpp): the run-structured modes, a block = whole text rows owned by ONE wave.          #
# Geometry 28 (2 waves x 2 cells per lane = 128-cell blocks, emulator builds only) gives tiny frames many blocks per #
# wave and several rows per block; 24 / 25 are the product geometries.                                               #
# --------------------------------------------------------------------------------------------------------------- #
from achip_ctypes import MODE_HB_256, MODE_HB_16, MODE_HB_MONO, MODE_MONO  # noqa: E402

ROWS_MODES = [MODE_MONO, MODE_HB_TRUE, MODE_HB_256, MODE_HB_16, MODE_HB_MONO]
ROWS_IDS = ["mono", "hb_true", "hb_256", "hb_16", "hb_mono"]


def run_frames(w, h, kind):
    """inputs with real run structure: long runs, runs that end at row ends, transparent (black) stretches, single cells"""
    img = np.zeros((h, w, 3), np.uint8)
    if kind == "blocks":      # 7-pixel-wide colour blocks, 5 rows tall, every fourth one black (transparent half-block runs)
        for y in range(h):
            for x in range(w):
                b = (x // 7 + 3 * (y // 5)) % 8
                img[y, x] = (0, 0, 0) if b % 4 == 0 else (30 * b, 255 - 30 * b, (b * 77) % 256)
    elif kind == "flat":      # one colour: one run per row, REP with three- and four-digit counts on wide grids
        img[:] = (200, 120, 40)
    elif kind == "black":     # fully transparent in the half-block colour modes
        pass
    elif kind == "stripes":   # alternating single columns: no run longer than one cell
        img[:, ::2] = (255, 255, 255)
        img[:, 1::2] = (10, 200, 90)
    return img


@pytest.mark.parametrize("mode", ROWS_MODES, ids=ROWS_IDS)
@pytest.mark.parametrize("variant", [28, 25, 24, 26])
def test_rows_kernel_torture(mode, variant):
    cap = {28: 128, 25: 256, 24: 448, 26: 448}[variant]
    for (W, H) in [(80, 24), (97, 31), (3, 2), (1, 1), (64, 1), (65, 3), (128, 5), (200, 7), (448, 3)]:
        if W > cap:
            continue
        exp = oracle_convert(TORTURE, mode, W, H, orc.PALETTE_STANDARD)
        got = emu_convert(TORTURE, mode, W, H, orc.PALETTE_STANDARD, variant)
        assert got == exp, (MODE_NAMES[mode], W, H, variant)


def digit_length_image(w, h, seed):
    """every pixel's channels drawn from values of one, two and three decimal digits (the boundaries included): the SGRs of
    neighbouring cells differ in length in every combination, so word-built SGRs (render_kernels.hpp word_sgr) meet every
    (field lengths) x (byte alignment of the token) case"""
    vals = np.array([0, 5, 9, 10, 55, 99, 100, 200, 255], np.uint8)
    rng = np.random.default_rng(seed)
    return vals[rng.integers(0, len(vals), (h, w, 3))]


@pytest.mark.parametrize("variant", [24, 25, 26])
def test_rows_kernel_word_built_sgrs_at_every_field_length_and_alignment(variant):
    # 27 x 27 digit-length combinations per pixel pair, every cell a run head (the all-heads path), odd widths so that
    # tokens start at every byte alignment; then the same image with runs in it (doubled columns: the general token path
    # next to word-built SGRs, lone half blocks, repeat counts)
    for (w, h, seed) in [(97, 14, 1), (200, 6, 2), (61, 10, 3)]:
        img = digit_length_image(w, h, seed)
        for src in (img, np.ascontiguousarray(np.repeat(img, 2, axis=1)[:, :w]), np.ascontiguousarray(np.repeat(img, 5, axis=1)[:, :w])):
            exp = oracle_convert(src, MODE_HB_TRUE, w, h // 2, orc.PALETTE_STANDARD)
            assert emu_convert(src, MODE_HB_TRUE, w, h // 2, orc.PALETTE_STANDARD, variant) == exp, (variant, w, h, seed)
    img = digit_length_image(120, 8, 4)  # with padding: pad cells in front of every row's first SGR
    exp = oracle_convert(img, MODE_HB_TRUE, 120, 4, orc.PALETTE_STANDARD, True, True)
    assert emu_convert(img, MODE_HB_TRUE, 120, 4, orc.PALETTE_STANDARD, variant, True, True) == exp


@pytest.mark.parametrize("mode", ROWS_MODES, ids=ROWS_IDS)
def test_rows_kernel_a_word_of_heads_whose_last_run_goes_on(mode):
    """the all-heads path (render_rows.hpp make_tok_heads): 64 cells that each start a run, the last of them a run that goes
    on in the next word -- its repeat count comes from the words above.  (Found by scripts/gpu_soak.py: a mono row of 63 pad
    cells and a flat image lost its ESC[6b.)"""
    rng = np.random.default_rng(3)
    for (heads, flat, rows) in ((64, 36, 3), (128, 40, 2), (64, 200, 2), (192, 7, 4), (63, 37, 3)):
        w = heads + flat
        img = np.zeros((2 * rows, w, 3), np.uint8)
        img[:, :heads] = rng.integers(1, 256, (2 * rows, heads, 3))
        img[:, :heads:2, 0] = 255  # (neighbours differ whatever the quantiser: bright red against dark)
        img[:, 1:heads:2] //= 8
        img[:, heads:] = img[:, heads - 1:heads]  # the last head's run goes on to the end of the row
        rm_rows = rows if MODE_CAPS[mode][1] == 2 else 2 * rows
        exp = oracle_convert(img, mode, w, rm_rows, orc.PALETTE_STANDARD)
        for variant in (24, 25, 26):
            if w > {24: 448, 25: 256, 26: 448}[variant]:
                continue
            assert emu_convert(img, mode, w, rm_rows, orc.PALETTE_STANDARD, variant) == exp, (MODE_NAMES[mode], variant, heads, flat)
    # ... and by padding: 63 pad cells in front of a flat row (the soak's case)
    flat_img = np.full((40, 9, 3), 7, np.uint8)
    f = emu.frame_for_convert(flat_img, 134, 12, MODE_CAPS[mode][1], True, True)
    exp = oracle_convert(flat_img, mode, 134, 12, orc.PALETTE_STANDARD, True, True)
    for variant in (24, 25, 26):
        assert emu.render_frames(mode, [f], orc.PALETTE_STANDARD, variant)[0] == exp, (MODE_NAMES[mode], variant, f.pad_left)


@pytest.mark.parametrize("seed", range(6))
def test_rows_kernel_random_run_structures_across_word_boundaries(seed):
    """random rows built from segments -- noise (every cell a run head), flat colour (one long run), black (a transparent
    run), two-colour stripes -- whose lengths cluster around the 64-cell words the rows kernel's head masks are cut into, at
    random widths, with and without aspect + padding: every run-structured mode on every rows geometry against the oracle"""
    rng = np.random.default_rng(1000 + seed)
    for case in range(10):
        w = int(rng.choice([64, 65, 100, 128, 129, 191, 192, 200, 256, 300, 384, 447, 448])) if case % 3 else int(rng.integers(1, 449))
        rows = int(rng.integers(1, 5))
        img = np.zeros((2 * rows, w, 3), np.uint8)
        for y in range(2 * rows):
            x = 0
            while x < w:
                n = int(rng.choice([1, 2, 3, 5, 62, 63, 64, 65, 66, 127, 128, 129])) if rng.integers(0, 3) else int(rng.integers(1, 90))
                n = min(n, w - x)
                kind = int(rng.integers(0, 4))
                if kind == 0:
                    img[y, x:x + n] = rng.integers(0, 256, (n, 3))
                elif kind == 1:
                    img[y, x:x + n] = rng.integers(0, 256, 3)
                elif kind == 2:
                    img[y, x:x + n] = 0
                else:
                    img[y, x:x + n:2] = rng.integers(0, 256, 3)
                    img[y, x + 1:x + n:2] = rng.integers(0, 256, 3)
                x += n
            if y % 2 and rng.integers(0, 2):  # half the half-block rows: bottom = top (runs decided by the top alone)
                img[y] = img[y - 1]
        pad = bool(rng.integers(0, 2))
        for mode in ROWS_MODES:
            rm = MODE_CAPS[mode][1]
            H = rows if rm == 2 else 2 * rows
            W = w if not pad else min(448, w + int(rng.integers(0, 130)))
            f = emu.frame_for_convert(img, W, H, rm, pad, pad)
            if f is None or f.pad_left + f.out_w > 448:
                continue
            exp = oracle_convert(img, mode, W, H, orc.PALETTE_STANDARD, pad, pad)
            for variant in (24, 25, 26):
                if f.pad_left + f.out_w > {24: 448, 25: 256, 26: 448}[variant]:
                    continue
                assert emu.render_frames(mode, [f], orc.PALETTE_STANDARD, variant)[0] == exp, (seed, case, MODE_NAMES[mode], variant, w, W, H, pad)


# ---- rows cut into segments (render_rows.hpp WIDE; round 6): geometries 27 (16 waves x 320-cell segments), 29 (8 waves) and
# the emulator's 30 (4 waves x 64-cell segments: rows of up to 256 cells put tiny frames through every segment case)
WIDE_CAP = {30: 256, 27: 4096, 29: 2560}
WIDE_SLOTS = {30: 64, 27: 320, 29: 320}


@pytest.mark.parametrize("mode", ROWS_MODES, ids=ROWS_IDS)
def test_rows_kernel_wide_rows_torture_and_run_structures(mode):
    """rows of two to four segments (and of one: the same kernel must take narrow rows), runs that cross none, one and every
    segment boundary of a row (flat: one run per row, its repeat count summed over the segments; black: a transparent run;
    blocks: 7-cell runs that straddle the boundaries; stripes: no run at all), against the oracle"""
    for (W, H) in [(65, 3), (100, 5), (128, 4), (129, 6), (200, 7), (256, 3), (64, 2), (30, 4), (1, 1), (255, 5), (193, 9)]:
        for src in (TORTURE, run_frames(W, 2 * H, "blocks"), run_frames(W, 2 * H, "flat"), run_frames(W, 2 * H, "black"),
                    run_frames(W, 2 * H, "stripes")):
            exp = oracle_convert(src, mode, W, H, orc.PALETTE_STANDARD)
            assert emu_convert(src, mode, W, H, orc.PALETTE_STANDARD, 30) == exp, (MODE_NAMES[mode], W, H, src.shape)
    # the product's segment geometries: 449 cells = two segments of 225 / 224; 1000 = four of 250; one row of 3840 cells (the
    # widest the reference resizes to, image.c: twelve segments, a four-digit repeat count) on 27, 2560 (eight) on 29
    for (W, H, variant) in [(449, 2, 27), (449, 3, 29), (1000, 2, 27), (640, 3, 29), (3840, 1, 27), (2560, 1, 29)]:
        for kind in ("blocks", "flat", "black"):
            src = run_frames(W, 2 * H, kind)
            exp = oracle_convert(src, mode, W, H, orc.PALETTE_STANDARD)
            assert emu_convert(src, mode, W, H, orc.PALETTE_STANDARD, variant) == exp, (MODE_NAMES[mode], W, H, variant, kind)


@pytest.mark.parametrize("seed", range(6))
def test_rows_kernel_wide_rows_random_run_structures_across_segment_boundaries(seed):
    """random rows built from pieces -- noise, flat colour, black, stripes, NEAR-black (raw rgb 0..2: equal keys in the 256- /
    16-colour modes, where the run HEAD's raw rgb decides transparency, halfblock.c:357,476 -- a head in one segment, its
    run's cells in the next) -- whose lengths cluster around the segment widths, with and without aspect + padding (pad cells
    may fill whole segments): every run-structured mode on every segment geometry against the oracle"""
    rng = np.random.default_rng(6000 + seed)
    for case in range(9):
        variant = (30, 30, 27, 30, 29, 30)[case % 6]
        cap = WIDE_CAP[variant]
        if variant == 30:
            w = int(rng.choice([65, 100, 127, 128, 129, 191, 192, 193, 255, 256])) if case % 2 else int(rng.integers(60, 257))
        else:
            w = int(rng.choice([449, 450, 511, 512, 640, 767, 768, 769, 1000, 1153])) if case % 2 else int(rng.integers(449, 1300))
        n0 = -(-w // WIDE_SLOTS[variant])
        segw = -(-w // n0)
        rows = int(rng.integers(1, 4))
        img = np.zeros((2 * rows, w, 3), np.uint8)
        for y in range(2 * rows):
            x = 0
            while x < w:
                if rng.integers(0, 3):
                    n = int(rng.choice([1, 2, segw - 1, segw, segw + 1, 2 * segw - 1, 2 * segw, 2 * segw + 1, 3 * segw]))
                else:
                    n = int(rng.integers(1, 2 * segw))
                if rng.integers(0, 4) == 0:  # end exactly at a segment boundary, or one cell off it
                    n = max(1, (x // segw + 1) * segw - x + int(rng.integers(-1, 2)))
                n = min(n, w - x)
                kind = int(rng.integers(0, 5))
                if kind == 0:
                    img[y, x:x + n] = rng.integers(0, 256, (n, 3))
                elif kind == 1:
                    img[y, x:x + n] = rng.integers(0, 256, 3)
                elif kind == 2:
                    img[y, x:x + n] = 0
                elif kind == 3:
                    img[y, x:x + n:2] = rng.integers(0, 256, 3)
                    img[y, x + 1:x + n:2] = rng.integers(0, 256, 3)
                else:
                    img[y, x:x + n] = rng.integers(0, 3, (n, 3))
                x += n
            if y % 2 and rng.integers(0, 2):
                img[y] = img[y - 1]
        pad = bool(rng.integers(0, 2))
        for mode in ROWS_MODES:
            rm = MODE_CAPS[mode][1]
            H = rows if rm == 2 else 2 * rows
            W = w if not pad else min(cap, w + int(rng.integers(0, 2 * segw)))
            f = emu.frame_for_convert(img, W, H, rm, pad, pad)
            if f is None or f.pad_left + f.out_w > cap:
                continue
            exp = oracle_convert(img, mode, W, H, orc.PALETTE_STANDARD, pad, pad)
            assert emu.render_frames(mode, [f], orc.PALETTE_STANDARD, variant)[0] == exp, (seed, case, MODE_NAMES[mode], variant, w, W, H, pad)


@pytest.mark.parametrize("mode", [MODE_HB_256, MODE_HB_16, MODE_HB_TRUE], ids=["hb_256", "hb_16", "hb_true"])
def test_rows_kernel_wide_rows_open_run_transparency_across_segments(mode):
    """the state a transparent run leaves (no SGR state: the next head carries both SGRs, halfblock.c:357,476) when the run's
    HEAD lies one or two segments in front of the head that needs it: raw black heads followed by cells of equal keys but
    other raw rgb, and the other way round, at every offset around the boundaries of 64-cell segments"""
    for head_at in (0, 40, 63, 64, 65, 127, 128):
        for run in (1, 24, 64, 65, 130):
            for head_rgb, tail_rgb in (((0, 0, 0), (1, 1, 1)), ((1, 1, 1), (0, 0, 0)), ((0, 0, 0), (0, 0, 0)), ((2, 1, 0), (1, 2, 2))):
                w = 230
                img = np.zeros((2, w, 3), np.uint8)
                img[:, :] = (200, 10, 90)
                img[:, :head_at] = np.random.default_rng(head_at).integers(3, 256, (2, head_at, 3)) if head_at else 0
                end = min(w - 3, head_at + run)
                img[:, head_at:end] = tail_rgb
                img[:, head_at] = head_rgb
                img[:, end:end + 2] = (9, 200, 33)  # the head that reads the state
                exp = oracle_convert(img, mode, w, 1, orc.PALETTE_STANDARD)
                assert emu_convert(img, mode, w, 1, orc.PALETTE_STANDARD, 30) == exp, (MODE_NAMES[mode], head_at, run, head_rgb)
    # ... and in a workgroup that is not the frame's first (WIDE + PARTS, the emulator's 34: two 64-cell segments, a row per part)
    for head_at in (0, 40, 63, 64, 65):
        for run in (1, 24, 60):
            for head_rgb, tail_rgb in (((0, 0, 0), (1, 1, 1)), ((1, 1, 1), (0, 0, 0)), ((0, 0, 0), (0, 0, 0))):
                w = 126
                img = np.zeros((6, w, 3), np.uint8)
                img[:, :] = (200, 10, 90)
                img[:, :head_at] = np.random.default_rng(head_at).integers(3, 256, (6, head_at, 3)) if head_at else 0
                end = min(w - 3, head_at + run)
                img[:, head_at:end] = tail_rgb
                img[:, head_at] = head_rgb
                img[:, end:end + 2] = (9, 200, 33)
                exp = oracle_convert(img, mode, w, 3, orc.PALETTE_STANDARD)
                assert emu_convert_parts(img, mode, w, 3, orc.PALETTE_STANDARD, 34, 3) == exp, (MODE_NAMES[mode], head_at, run, head_rgb)


@pytest.mark.parametrize("mode", ROWS_MODES, ids=ROWS_IDS)
def test_rows_kernel_wide_rows_with_padding_flips_tint_and_batches(mode):
    """segments with everything the cell records fold in (left padding from the aspect fit -- pad cells filling the first
    segment --, top padding, flips, the colour filter, odd half-block heights, the buffer's first pixel), then a ragged batch
    (frames of one, two and four segments in ONE launch, by descriptor table and by value)"""
    import ctypes as C
    rm, cl = MODE_CAPS[mode][1], MODE_CAPS[mode][0]
    imgs = [orc.frame_hash_noise(120, 90, 3), run_frames(37, 29, "blocks"), orc.frame_smooth(300, 7)]
    padded = 0
    for img in imgs:
        for (W, H, variant) in [(250, 21, 30), (200, 9, 30), (129, 5, 30), (700, 7, 27), (500, 5, 29)]:
            for fx, fy, flt in [(False, False, 0), (True, True, 3), (False, True, 7)]:
                f = emu.frame_for_convert(img, W, H, rm, True, True)
                padded += f.pad_left >= 64
                assert emu.lib().achip_frame_set_display_ops(C.byref(f), fx, fy, flt) == 0
                exp = orc.display_convert(img, W, H, cl, rm, True, True, fx, fy, flt)
                assert emu.render_frames(mode, [f], orc.PALETTE_STANDARD, variant)[0] == exp, (MODE_NAMES[mode], img.shape, W, H, variant, fx, fy, flt)
    assert padded >= 3, padded
    dims = [(256, 6), (100, 3), (60, 4), (129, 7), (255, 2), (64, 5), (200, 1)]
    srcs = [run_frames(w, 2 * h, kind) for (w, h), kind in zip(dims, ("blocks", "flat", "stripes", "black", "blocks", "flat", "blocks"))]
    frames = [emu.frame_for_convert(s_, w, h, rm, False, False) for s_, (w, h) in zip(srcs, dims)]
    exp = [oracle_convert(s_, mode, w, h, orc.PALETTE_STANDARD) for s_, (w, h) in zip(srcs, dims)]
    assert emu.render_frames(mode, frames, orc.PALETTE_STANDARD, 30) == exp
    same = [emu.frame_for_convert(srcs[0], 256, 6, rm, False, False) for _ in range(3)]
    assert emu.render_frames(mode, same, orc.PALETTE_STANDARD, 30, uniform=True) == [exp[0]] * 3


def test_rows_kernel_wide_rows_limits_and_choice():
    """a row of more segments than the workgroup has waves is refused by the kernel (a segment may wait for every other
    segment of its row: all of them must be in flight) and never sent there; composites and 1x1 sources neither; whole-frame
    launches of rows beyond 448 cells take the segment geometries by themselves"""
    import ctypes as C
    img = run_frames(300, 4, "blocks")
    f = emu.frame_for_convert(img, 257, 2, 0, False, False)
    assert emu.render_frames(MODE_MONO, [f], orc.PALETTE_STANDARD, 30)[0] == 0xFFFFFFFE  # ACHIP_LEN_BADDESC: five segments, four waves
    L = emu.lib()
    caps = (C.c_int * 5)(4096, 2048, 1024, 256, 2048)

    def choice(src, W, H, mode, n, n_cus, forced=-1):
        rm = MODE_CAPS[mode][1]
        fr = emu.frame_for_convert(src, W, H, rm, False, False)
        arr = (emu.Frame * n)(*([fr] * n))
        v, parts, rpp = C.c_int(-1), C.c_int(0), C.c_int(0)
        rc = L.achip_choose_geometry(mode, arr, n, True, caps, n_cus, 0, forced, C.byref(v), C.byref(parts), C.byref(rpp))
        return v.value if rc == 0 else None

    uhd = np.zeros((2160, 3840, 3), np.uint8)
    assert choice(uhd, 640, 180, MODE_HB_TRUE, 256, 256) == 27
    assert choice(uhd, 640, 180, MODE_HB_TRUE, 512, 256) == 29
    assert choice(uhd, 1000, 40, MODE_MONO, 256, 256) == 27
    assert choice(uhd, 3500, 40, MODE_MONO, 512, 256) == 27      # eleven segments: beyond the eight-wave geometry
    assert choice(uhd, 640, 180, MODE_HB_TRUE, 256, 256, forced=29) == 29 and choice(uhd, 3500, 40, MODE_MONO, 256, 256, forced=29) is None
    assert choice(uhd, 448, 120, MODE_HB_TRUE, 512, 256) == 24    # one block still holds the row
    one = np.zeros((1, 1, 3), np.uint8)
    assert choice(one, 640, 20, MODE_MONO, 256, 256) in (0, 4)    # 1x1 source: the general sampler, which the segment geometries lack


# ---- a frame's blocks shared out over workgroups of the rows kernel (render_rows.hpp PARTS; round 6): geometry 31 (four waves x
# two slots: rows of up to 128 cells) and the emulator's 33 (two waves x one slot: 64-cell blocks, many of them for tiny frames)
@pytest.mark.parametrize("mode", ROWS_MODES, ids=ROWS_IDS)
def test_rows_kernel_shared_out_over_workgroups(mode):
    """every part count from one block per workgroup to more workgroups than blocks (the parts behind the frame's last block
    only report in), runs that end and start at part boundaries, padding, ragged batches, three launches on the same hand-off
    words (a new epoch each)"""
    for (W, H, variant, parts) in [(80, 24, 31, 2), (80, 24, 31, 6), (80, 24, 31, 24), (120, 9, 31, 4), (60, 7, 33, 4), (60, 7, 33, 7),
                                   (64, 9, 33, 5), (30, 10, 33, 5), (30, 10, 33, 16), (128, 5, 31, 2), (1, 1, 33, 2), (97, 31, 31, 8), (40, 30, 31, 3)]:
        for src in (TORTURE, run_frames(W, 2 * H, "blocks"), run_frames(W, 2 * H, "flat"), run_frames(W, 2 * H, "black")):
            exp = oracle_convert(src, mode, W, H, orc.PALETTE_STANDARD)
            assert emu_convert_parts(src, mode, W, H, orc.PALETTE_STANDARD, variant, parts) == exp, (MODE_NAMES[mode], W, H, variant, parts)
    rm = MODE_CAPS[mode][1]
    f = emu.frame_for_convert(TORTURE, 61, 19, rm, True, True)  # aspect fit + padding: pad cells, pad_top newlines (part 0 writes them)
    exp = oracle_convert(TORTURE, mode, 61, 19, orc.PALETTE_STANDARD, True, True)
    for parts in (2, 5):
        assert emu.render_frames(mode, [f], orc.PALETTE_STANDARD, 33, parts=parts)[0] == exp, (MODE_NAMES[mode], parts)
    dims = [(64, 6), (50, 3), (60, 4), (33, 7), (64, 2), (1, 5), (20, 1)]
    srcs = [run_frames(w, 2 * h, kind) for (w, h), kind in zip(dims, ("blocks", "flat", "stripes", "black", "blocks", "flat", "blocks"))]
    frames = [emu.frame_for_convert(s_, w, h, rm, False, False) for s_, (w, h) in zip(srcs, dims)]
    exp = [oracle_convert(s_, mode, w, h, orc.PALETTE_STANDARD) for s_, (w, h) in zip(srcs, dims)]
    sync = np.zeros(len(frames) * 3, dtype=np.uint64)
    for _ in range(3):
        assert emu.render_frames(mode, frames, orc.PALETTE_STANDARD, 33, parts=3, sync=sync) == exp


@pytest.mark.parametrize("mode", ROWS_MODES, ids=ROWS_IDS)
def test_rows_kernel_segments_shared_out_over_workgroups(mode):
    """WIDE + PARTS (geometry 32: rows of 129-512 cells in segments of at most 128, whole rows per four-wave workgroup) through
    the emulator's 34 (two waves x 64-cell segments: rows of up to 128 cells): runs across the segment boundary inside a
    workgroup, rows that end a part, more parts than rows, one-segment rows, padding, ragged batches on the same hand-off words"""
    for (W, H, parts) in [(100, 6, 2), (128, 7, 3), (65, 5, 5), (128, 4, 4), (90, 9, 9), (127, 3, 7), (64, 4, 2), (30, 5, 3), (128, 1, 2)]:
        for src in (TORTURE, run_frames(W, 2 * H, "blocks"), run_frames(W, 2 * H, "flat"), run_frames(W, 2 * H, "black"), run_frames(W, 2 * H, "stripes")):
            exp = oracle_convert(src, mode, W, H, orc.PALETTE_STANDARD)
            assert emu_convert_parts(src, mode, W, H, orc.PALETTE_STANDARD, 34, parts) == exp, (MODE_NAMES[mode], W, H, parts)
    rm = MODE_CAPS[mode][1]
    f = emu.frame_for_convert(TORTURE, 120, 19, rm, True, True)  # aspect fit + padding: pad cells in the first segment, pad_top newlines
    exp = oracle_convert(TORTURE, mode, 120, 19, orc.PALETTE_STANDARD, True, True)
    for parts in (2, 5, 19):
        assert emu.render_frames(mode, [f], orc.PALETTE_STANDARD, 34, parts=parts)[0] == exp, (MODE_NAMES[mode], parts)
    dims = [(128, 6), (100, 3), (65, 4), (33, 7), (128, 2), (1, 5), (90, 1)]
    srcs = [run_frames(w, 2 * h, kind) for (w, h), kind in zip(dims, ("blocks", "flat", "stripes", "black", "blocks", "flat", "blocks"))]
    frames = [emu.frame_for_convert(s_, w, h, rm, False, False) for s_, (w, h) in zip(srcs, dims)]
    exp = [oracle_convert(s_, mode, w, h, orc.PALETTE_STANDARD) for s_, (w, h) in zip(srcs, dims)]
    sync = np.zeros(len(frames) * 3, dtype=np.uint64)
    for _ in range(3):
        assert emu.render_frames(mode, frames, orc.PALETTE_STANDARD, 34, parts=3, sync=sync) == exp


def test_quant16_without_the_table_walk_for_every_colour():
    """render_kernels.hpp quant16 / ansi16_rgb decide the nearest of the 16 ANSI colours from the table's structure (two cubes and
    colour 7 competing as distance << 4 | index); against the reference's table walk (ansi.c:437-477, first minimum) for all
    2^24 colours, and against the reference's own known answers (ansi_test.c via reference_kats.json) through the walk"""
    import ctypes as C
    first = C.c_uint32(0)
    assert emu.lib().emu_quant16_check(C.byref(first)) == 0, hex(first.value)


def test_rep_rule_as_one_comparison():
    """render_kernels.hpp rep_profitable is `run >= 6`; the rule as the reference writes it (output_buffer.c:148-155, pinned by its
    own known answers in test_reference_kats.py) over the first 2^22 runs and around every power of ten / two"""
    assert emu.lib().emu_rep_rule_check() == 0


@pytest.mark.parametrize("variant", [16, 17, 20])
def test_stream_kernel_word_built_sgrs_at_every_field_length_and_alignment(variant):
    for (w, h, seed) in [(97, 7, 5), (200, 3, 6), (61, 5, 7)]:
        img = digit_length_image(w, h, seed)
        for src in (img, np.ascontiguousarray(np.repeat(img, 2, axis=1)[:, :w])):  # (doubled columns: cells without an SGR between the others)
            exp = oracle_convert(src, MODE_TRUE_FG, w, h, orc.PALETTE_STANDARD)
            assert emu_convert(src, MODE_TRUE_FG, w, h, orc.PALETTE_STANDARD, variant) == exp, (variant, w, h, seed)


@pytest.mark.parametrize("mode", ROWS_MODES, ids=ROWS_IDS)
def test_rows_kernel_run_structure(mode):
    # runs, REP counts, transparent runs and the state they reset, runs cut by row ends -- at widths that put one, two and
    # several rows into a block, and with aspect + padding (pad cells are run heads of their own)
    for kind in ("blocks", "flat", "black", "stripes"):
        img = run_frames(160, 90, kind)
        for (W, H, variant, asp) in [(100, 9, 28, False), (37, 11, 28, False), (128, 4, 28, False), (250, 6, 25, False),
                                     (440, 4, 24, False), (61, 19, 28, True), (80, 24, 25, True), (300, 40, 26, False), (90, 33, 26, True)]:
            exp = oracle_convert(img, mode, W, H, orc.PALETTE_STANDARD, asp, asp)
            got = emu_convert(img, mode, W, H, orc.PALETTE_STANDARD, variant, asp, asp)
            assert got == exp, (MODE_NAMES[mode], kind, W, H, variant, asp)


@pytest.mark.parametrize("mode", ROWS_MODES, ids=ROWS_IDS)
def test_rows_kernel_one_row_blocks_with_padding_flips_and_tint(mode):
    """Round 5: a block that is ONE text row takes its source rows from scalar registers and its samples' byte offsets from
    per-frame cell records (render_rows.hpp).  That path with everything the records fold in: left padding (aspect fit), top
    padding, flips in both axes, the colour filter, odd half-block heights, sources narrower than the grid (samples repeat),
    the buffer's first pixel (requested at offset 0, finished by masking) in the first AND -- flipped -- in the last row."""
    import ctypes as C
    rm = MODE_CAPS[mode][1]
    cl = MODE_CAPS[mode][0]
    imgs = [orc.frame_hash_noise(120, 90, 3), run_frames(37, 29, "blocks"), orc.frame_smooth(300, 7)]
    one_row = padded = 0
    for img in imgs:
        for (W, H, variant) in [(120, 21, 28), (100, 9, 28), (127, 5, 28), (250, 7, 25), (440, 3, 24), (300, 37, 26)]:
            for fx, fy, flt in [(False, False, 0), (True, True, 3), (False, True, 7)]:
                f = emu.frame_for_convert(img, W, H, rm, True, True)  # use_aspect_ratio + wants_padding
                slots = 64 * {28: 2, 25: 4, 24: 7, 26: 7}[variant]
                one_row += slots // 2 < f.pad_left + f.out_w <= slots
                padded += f.pad_left > 0 and slots // 2 < f.pad_left + f.out_w
                assert emu.lib().achip_frame_set_display_ops(C.byref(f), fx, fy, flt) == 0
                exp = orc.display_convert(img, W, H, cl, rm, True, True, fx, fy, flt)
                got = emu.render_frames(mode, [f], orc.PALETTE_STANDARD, variant)[0]
                assert got == exp, (MODE_NAMES[mode], img.shape, W, H, variant, fx, fy, flt)
    assert one_row >= 15 and padded >= 6, (one_row, padded)  # (the rest of the cases put several rows into a block)


def test_rows_kernel_leaves_very_wide_sources_to_the_phase_kernel():
    """a cell record keeps the sample's byte offset in 16 bits (3 * (src_w - 1) < 65 536): hand-built descriptors of sources
    wider than 21 845 pixels (the reference's entry points stop at 10 000, ascii.c:204) are refused by the rows kernel itself
    and never sent there by achip_choose_geometry -- the phase kernel renders them"""
    import ctypes as C
    w = 22000
    img = np.zeros((2, w, 3), np.uint8)
    img[:, ::7] = (200, 50, 10)
    img[:, -40:] = (9, 250, 77)  # the columns a 16-bit offset would alias
    L = emu.lib()
    L.achip_nn_ratio.restype = C.c_uint32
    L.achip_nn_ratio.argtypes = [C.c_int, C.c_int]

    def wide_frame(im):  # (achip_frame_setup refuses sources beyond the reference's 10 000: the descriptor is filled by hand)
        f = emu.Frame()
        f.src, f.src_w, f.src_h, f.out_w, f.out_h = im.ctypes.data, im.shape[1], im.shape[0], 80, 2
        f.x_ratio, f.y_ratio = L.achip_nn_ratio(im.shape[1], 80), L.achip_nn_ratio(im.shape[0], 2)
        return f

    f = wide_frame(img)
    assert emu.render_frames(MODE_MONO, [f], orc.PALETTE_STANDARD, 25)[0] == 0xFFFFFFFE  # ACHIP_LEN_BADDESC
    xs = np.minimum((np.arange(80, dtype=np.int64) * int(f.x_ratio)) >> 16, w - 1)
    small = np.ascontiguousarray(img[:, xs])  # the resize the reference would have made (image.c:293-325), then scale 1
    exp = oracle_convert(small, MODE_MONO, 80, 2, orc.PALETTE_STANDARD)
    assert emu.render_frames(MODE_MONO, [f], orc.PALETTE_STANDARD, 4)[0] == exp
    caps = (C.c_int * 5)(4096, 2048, 1024, 0, 2048)
    v, parts, rpp = C.c_int(), C.c_int(), C.c_int()
    assert L.achip_choose_geometry(MODE_MONO, (emu.Frame * 300)(*([f] * 300)), 300, True, caps, 256, 0, -1, C.byref(v), C.byref(parts), C.byref(rpp)) == 0
    assert v.value < 16, v.value  # a phase-kernel geometry, not rows 24 / 25
    narrow = np.ascontiguousarray(img[:, :21845])
    f2 = wide_frame(narrow)
    assert emu.render_frames(MODE_MONO, [f2], orc.PALETTE_STANDARD, 25)[0] != 0xFFFFFFFE
    assert L.achip_choose_geometry(MODE_MONO, (emu.Frame * 300)(*([f2] * 300)), 300, True, caps, 256, 0, -1, C.byref(v), C.byref(parts), C.byref(rpp)) == 0
    assert v.value in (24, 25), v.value


def test_policy_takes_the_sixteen_wave_rows_geometry_for_one_launch_of_dense_half_block_frames():
    """geometry 26 (round 5, re-audited with the word-built SGRs: profiles/r05_policy_audit_hb.txt): whole coloured
    half-block frames at no more than a frame per CU of the plan's share, big enough for a block per wave -- from 120
    columns for truecolor from dense sources, 160 from sources up to 1080p, rows beyond the four-slot geometry from larger
    ones (and there only up to three quarters of a frame per CU, or in flight); 256 / 16 colours: those rows, or from 160
    columns when the sources are dense or the launch at most three quarters of a frame per CU -- and nothing else"""
    import ctypes as C
    L = emu.lib()
    caps = (C.c_int * 5)(4096, 2048, 1024, 0, 2048)

    def choice(src_wh, W, H, mode, n, cus):
        f = emu.frame_for_convert(np.zeros((src_wh[1], src_wh[0], 3), np.uint8), W, H, 2)
        v, parts, rpp = C.c_int(), C.c_int(), C.c_int()
        assert L.achip_choose_geometry(mode, (emu.Frame * n)(*([f] * n)), n, True, caps, cus, 0, -1, C.byref(v), C.byref(parts), C.byref(rpp)) == 0
        return v.value

    dense = lambda W, H: (W, 2 * H)  # noqa: E731  (the source IS the image a W x H half-block target samples)
    hd, uhd = (1920, 1080), (3840, 2160)
    # truecolor
    assert choice(dense(400, 120), 400, 120, MODE_HB_TRUE, 256, 256) == 26
    assert choice(dense(120, 40), 120, 40, MODE_HB_TRUE, 256, 256) == 26
    assert choice(dense(120, 40), 120, 40, MODE_HB_TRUE, 64, 64) == 26        # a share of the GPU, a frame per CU of it
    assert choice(dense(80, 24), 80, 24, MODE_HB_TRUE, 256, 256) == 25        # too small for sixteen waves
    assert choice(hd, 200, 60, MODE_HB_TRUE, 256, 256) == 26
    assert choice(hd, 200, 60, MODE_HB_TRUE, 64, 64) == 26
    assert choice(hd, 120, 40, MODE_HB_TRUE, 256, 256) == 26                  # (round 6 audit: level with the phase kernel at 256 frames, 5-8 % ahead at 128-192)
    assert choice(uhd, 400, 120, MODE_HB_TRUE, 192, 256) == 26
    assert choice(uhd, 400, 120, MODE_HB_TRUE, 256, 256) != 26                # BASELINE configs[4], one launch at a time: the phase kernel
    assert choice(uhd, 400, 120, MODE_MONO, 256, 256) == 26 and choice(uhd, 320, 90, MODE_MONO, 256, 256) == 26   # (mono there: 104 against 137-145 us, round 6's last audit)
    assert choice(uhd, 200, 60, MODE_MONO, 256, 256) != 26                     # (rows the four-slot geometry holds: the phase kernel, 40.5 against 45.6)
    assert choice(uhd, 200, 60, MODE_HB_256, 192, 256) != 26 and choice(uhd, 200, 60, MODE_HB_16, 256, 256) != 26 and choice(uhd, 200, 60, MODE_HB_256, 128, 256) == 26   # (256 / 16 colours from 4K sources: the phase kernel from 3/4 frame per CU on, 65.5 against 70.8 us)
    assert choice(uhd, 400, 120, MODE_HB_TRUE, 64, 64) == 26
    assert choice(uhd, 200, 60, MODE_HB_TRUE, 128, 256) != 26
    assert choice(dense(400, 120), 400, 120, MODE_HB_TRUE, 257, 256) == 24    # more than a frame per CU
    assert choice(hd, 200, 60, MODE_HB_TRUE, 256, 64) == 25                   # ... in flight from full frames: the four-slot geometry
    assert choice(dense(200, 60), 200, 60, MODE_HB_TRUE, 256, 64) == 24       # ... from dense sources: the slot rule
    # 256 / 16 colours
    assert choice(dense(400, 120), 400, 120, MODE_HB_256, 200, 256) == 26
    assert choice(hd, 320, 90, MODE_HB_16, 256, 256) == 26
    assert choice(dense(200, 60), 200, 60, MODE_HB_256, 256, 256) == 26
    assert choice(hd, 200, 60, MODE_HB_256, 128, 256) == 26 and choice(hd, 200, 60, MODE_HB_256, 192, 256) == 26
    assert choice(hd, 200, 60, MODE_HB_256, 256, 256) == 26 and choice(hd, 160, 45, MODE_HB_256, 256, 256) != 26  # (round 6 audit: from 200 columns at a full frame per CU too -- 58.1 against 60.5 us, 238x70 70.5 against 81.6; 160x45 level)
    # (dense sources from 120 columns on since quant16's diet -- profiles/r06_policy_audit_hb16.txt: 128-256 frames of 120x40
    # 22.9-25.4 us against 26.5-27.7; full-frame sources keep 160 / 200 columns)
    assert choice(dense(120, 40), 120, 40, MODE_HB_16, 256, 256) == 26 and choice(dense(120, 40), 120, 40, MODE_HB_256, 64, 64) == 26
    assert choice(hd, 120, 40, MODE_HB_16, 256, 256) != 26 and choice(dense(100, 40), 100, 40, MODE_HB_256, 256, 256) != 26
    # the short-token modes (profiles/r05_policy_audit_mono.txt): mono as truecolor half blocks; mono half blocks from dense
    # sources, from full frames only rows beyond the four-slot geometry with the GPU to itself
    assert choice(dense(400, 120), 400, 120, MODE_HB_MONO, 256, 256) == 26
    assert choice(dense(120, 40), 120, 40, MODE_HB_MONO, 64, 64) == 26
    assert choice(hd, 320, 90, MODE_HB_MONO, 256, 256) == 26 and choice(hd, 320, 90, MODE_HB_MONO, 64, 64) == 26   # (shared GPU too since round 6's last audit: 17.6 against 22.2 us)
    assert choice(hd, 238, 70, MODE_HB_MONO, 64, 64) != 26                                                          # (rows the four-slot geometry holds: as before)
    assert choice(hd, 200, 60, MODE_HB_MONO, 256, 256) != 26


def test_rows_kernel_refuses_rows_wider_than_a_block():
    f = emu.frame_for_convert(TORTURE, 129, 3, 0)
    assert emu.render_frames(MODE_MONO, [f], orc.PALETTE_STANDARD, 28)[0] == 0xFFFFFFFE  # ACHIP_LEN_BADDESC


@pytest.mark.parametrize("palette", [orc.PALETTE_BLOCKS, "ab", "x", "é漢😀 ."], ids=["blocks", "ab", "x", "mixed"])
def test_rows_kernel_mono_palettes(palette):
    for (W, H) in [(61, 17), (100, 5)]:
        assert emu_convert(TORTURE, MODE_MONO, W, H, palette, 28) == oracle_convert(TORTURE, MODE_MONO, W, H, palette)


def test_rows_kernel_ragged_batch_odd_heights_flips_and_overflow():
    import ctypes as C
    imgs = [orc.frame_hash_noise(120, 90, i) for i in range(3)] + [run_frames(64, 48, "blocks"), orc.frame_smooth(33, 17)]
    dims = [(80, 24), (60, 7), (33, 41), (100, 1), (17, 9)]
    for mode in (MODE_HB_TRUE, MODE_MONO, MODE_HB_16):
        rm = MODE_CAPS[mode][1]
        frames = [emu.frame_for_convert(im, w, h, rm) for im, (w, h) in zip(imgs, dims)]
        for variant in (28, 25):
            got = emu.render_frames(mode, frames, orc.PALETTE_STANDARD, variant)
            for k, (im, (w, h)) in enumerate(zip(imgs, dims)):
                assert got[k] == oracle_convert(im, mode, w, h, orc.PALETTE_STANDARD), (mode, variant, k)
            uni = emu.render_frames(mode, [frames[0]] * 3, orc.PALETTE_STANDARD, variant, uniform=True)
            assert uni[0] == uni[1] == uni[2] == got[0]
    img = imgs[0]
    for fx, fy, flt in [(True, False, 0), (False, True, 3), (True, True, 7)]:
        f = emu.frame_for_convert(img, 80, 24, 2)
        assert emu.lib().achip_frame_set_display_ops(C.byref(f), fx, fy, flt) == 0
        exp = orc.display_convert(img, 80, 24, 3, 2, False, False, fx, fy, flt)
        assert emu.render_frames(MODE_HB_TRUE, [f], orc.PALETTE_STANDARD, 28)[0] == exp, (fx, fy, flt)
    f = emu.frame_for_convert(img, 80, 24, 2)
    assert emu.render_frames(MODE_HB_TRUE, [f], orc.PALETTE_STANDARD, 28, stride=1024)[0] == 0xFFFFFFFF


@pytest.mark.parametrize("mode", ROWS_MODES, ids=ROWS_IDS)
def test_rows_kernel_fused_frame_crc(mode):
    """the frame CRC rides the rows kernel's drain too (VERDICT r2 gap 3): slices are checksummed in the staging area and
    chained per block, blocks are placed as in the stream kernel; headers and packet CRCs from the finishing wave"""
    rm = MODE_CAPS[mode][1]
    cases = [(80, 24, False, 28), (97, 31, False, 28), (3, 2, False, 28), (1, 1, False, 28), (200, 9, False, 25),
             (60, 40, True, 28), (440, 3, False, 24), (5, 300, True, 28)]
    for (W, H, asp, variant) in cases:
        for img in (TORTURE, run_frames(160, 90, "blocks")):
            exp = oracle_convert(img, mode, W, H, orc.PALETTE_STANDARD, asp, asp)
            f = emu.frame_for_convert(img, W, H, rm, asp, asp)
            got, crc = emu.render_frames_crc(mode, [f], orc.PALETTE_STANDARD, variant)
            assert got[0] == exp, (MODE_NAMES[mode], W, H, variant)
            assert crc[0] == orc.crc32c(exp), (MODE_NAMES[mode], W, H, variant, hex(crc[0]))
    imgs = [orc.frame_hash_noise(120, 90, i) for i in range(3)] + [run_frames(64, 48, "flat")]
    dims = [(80, 24), (60, 7), (33, 40), (100, 50)]
    frames = [emu.frame_for_convert(im, w, h, rm) for im, (w, h) in zip(imgs, dims)]
    for variant in (28, 25):
        got, crc, hdr, pkt = emu.render_frames_crc(mode, frames, orc.PALETTE_STANDARD, variant, dims=dims)
        for k, (im, (w, h)) in enumerate(zip(imgs, dims)):
            assert got[k] == oracle_convert(im, mode, w, h, orc.PALETTE_STANDARD), (mode, variant, k)
            eh, ep = orc.ascii_frame_packet(got[k], w, h)
            assert crc[k] == orc.crc32c(got[k]) and hdr[k] == eh and pkt[k] == ep, (mode, variant, k)
    got, crc, hdr, pkt = emu.render_frames_crc(mode, [frames[0]], orc.PALETTE_STANDARD, 28, stride=256, dims=[(80, 24)])
    eh, ep = orc.ascii_frame_packet(b"", 0, 0)
    assert got[0] == 0xFFFFFFFF and crc[0] == 0 and hdr[0] == eh and pkt[0] == ep


@pytest.mark.parametrize("phase", range(8))
def test_drains_follow_the_line_boundaries_of_the_address(phase):
    """Round 4: every kernel's drain maps lanes to 16-byte groups from a 128-byte LINE boundary of the slot's ADDRESS (whole
    lines per store instruction), the rows kernel carries the bytes behind a slice's last whole line to the block's next
    slice, and the phase kernel's carry is moved by whichever thread drained group 0.  Slabs that start at each of the
    eight 16-byte phases of a line, with a stride that walks the slots through other phases: the oracle's bytes, and not
    one byte outside the frames."""
    imgs = [orc.frame_hash_noise(120, 90, i) for i in range(2)] + [run_frames(160, 90, "blocks"), TORTURE]
    for (mode, variant, dims) in [
            (MODE_TRUE_FG, 17, [(80, 24), (97, 31), (3, 2), (200, 60)]), (MODE_256_FG, 16, [(80, 24), (61, 7), (1, 1), (130, 9)]),
            (MODE_HB_TRUE, 25, [(80, 24), (60, 7), (33, 40), (100, 50)]), (MODE_HB_TRUE, 24, [(440, 3), (97, 31), (5, 60), (300, 9)]),
            (MODE_MONO, 28, [(100, 9), (37, 11), (128, 4), (80, 24)]), (MODE_HB_256, 24, [(400, 12), (80, 24), (7, 3), (250, 6)]),
            (MODE_HB_TRUE, 4, [(80, 24), (97, 31), (460, 5), (10, 150)]), (MODE_MONO, 3, [(80, 24), (97, 31), (120, 20), (1, 130)]),
            (MODE_TRUE_FG, 2, [(80, 24), (130, 1), (64, 65), (200, 60)])]:
        rm = MODE_CAPS.get(mode, (3, 0))[1]
        frames = [emu.frame_for_convert(im, w, h, rm) for im, (w, h) in zip(imgs, dims)]
        bound = max(len(oracle_convert(im, mode, w, h, orc.PALETTE_STANDARD)) for im, (w, h) in zip(imgs, dims))
        for extra in ((16, 48, 112, 80)[phase % 4],):  # the stride's own phase: slots 1.. start at other phases than slot 0
            stride = (bound + 1 + 15) // 16 * 16 + 2048 + extra
            got = emu.render_frames(mode, frames, orc.PALETTE_STANDARD, variant, stride=stride, line_phase=phase)
            for k, (im, (w, h)) in enumerate(zip(imgs, dims)):
                assert got[k] == oracle_convert(im, mode, w, h, orc.PALETTE_STANDARD), (mode, variant, k, phase, extra)
