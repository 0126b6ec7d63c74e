"""Pins the CPU oracle (oracle/asciichat_oracle.c).

Two kinds of pins, both traceable to the reference (zfogg/ascii-chat, paths relative to its tree):
 (1) every known-answer value the reference's own unit tests hold for this path;
 (2) whole-frame length + FNV-1a-32 anchors of the reference's output, recorded in SURVEY.md
     section 8(c) and Appendix B (produced by the survey stage from the reference's unmodified sources).
"""
import ctypes as C
import os

import numpy as np
import pytest

import orc


def _sgr(fn, *args):
    buf = C.create_string_buffer(64)
    n = fn(buf, *args)
    return buf.raw[:n]


# ---- (1) reference unit-test known answers --------------------------------------------------
def test_truecolor_sgr_known_answers():
    L = orc.lib()
    # tests/unit/util/ansi_fast_test.c:51-64 (+18 bytes), :73-131 (7 fg + 7 bg edge strings)
    s = _sgr(L.orc_sgr_truecolor, 0, 255, 128, 64)
    assert s == b"\033[38;2;255;128;64m" and len(s) == 18
    s = _sgr(L.orc_sgr_truecolor, 1, 100, 200, 50)
    assert s == b"\033[48;2;100;200;50m" and len(s) == 18
    for r, g, b in [(0, 0, 0), (1, 1, 1), (255, 255, 255), (255, 0, 0), (0, 255, 0), (0, 0, 255), (128, 128, 128)]:
        assert _sgr(L.orc_sgr_truecolor, 0, r, g, b) == b"\033[38;2;%d;%d;%dm" % (r, g, b)
        assert _sgr(L.orc_sgr_truecolor, 1, r, g, b) == b"\033[48;2;%d;%d;%dm" % (r, g, b)


def test_16color_known_answers():
    L = orc.lib()
    # tests/unit/util/ansi_fast_test.c:458-493
    for rgb, idx in [((255, 0, 0), 9), ((0, 255, 0), 10), ((0, 0, 255), 12), ((0, 0, 0), 0), ((255, 255, 255), 15),
                     ((128, 0, 0), 1), ((0, 128, 0), 2), ((0, 0, 128), 4), ((192, 192, 192), 7)]:
        assert L.orc_rgb_to_16(*rgb) == idx
    # (the append_16color_* / append_256color_* strings of the same file are extracted by script:
    #  tests/golden/reference_kats.json "indexed_sgr", checked in test_reference_kats.py)


def test_256color_ranges():
    L = orc.lib()
    # tests/unit/util/ansi_fast_test.c:319-370 pins ranges only
    for v in range(0, 256, 5):
        assert 232 <= L.orc_rgb_to_256(v, v, v) <= 255
    assert 16 <= L.orc_rgb_to_256(255, 0, 0) <= 231
    assert _sgr(L.orc_sgr_256, 0, 123) == b"\033[38;5;123m"
    assert _sgr(L.orc_sgr_256, 1, 7) == b"\033[48;5;7m"


def test_rep_is_profitable_thresholds():
    L = orc.lib()
    # tests/unit/util/output_buffer_test.c:295-305
    for n in range(0, 6):
        assert not L.orc_rep_is_profitable(n)
    for n in (6, 10, 100, 3840):
        assert L.orc_rep_is_profitable(n)
    assert [L.orc_digits_u32(v) for v in (0, 9, 10, 99, 100, 12345, 1000000000)] == [1, 1, 2, 2, 3, 5, 10]


def test_aspect_ratio_stretch_identity_and_bounds():
    # tests/unit/util/aspect_ratio_test.c:32-40 (stretch returns exactly (W,H)); others bound-checked
    assert orc.aspect_ratio(1920, 1080, 80, 24, stretch=True) == (80, 24)
    for iw, ih, w, h in [(1920, 1080, 80, 24), (100, 1000, 80, 24), (1000, 100, 80, 24), (1, 1, 200, 60)]:
        ow, oh = orc.aspect_ratio(iw, ih, w, h)
        assert 0 < ow <= w and 0 < oh <= h


def test_line_count_equals_height_mono(oracle):
    # tests/unit/video/ascii_test.c:864-908
    for n in (1, 2, 5, 16, 40):
        img = orc.frame_gray(n * 3, n * 2)
        out = orc.convert(img, n, n, False, False, False)
        assert out.count(b"\n") + 1 == n


def test_pad_zero_is_identity():
    # tests/unit/video/ascii_test.c:467-476, :513-522
    L = orc.lib()
    s = b"ab\ncd"
    assert orc._take(L.orc_pad_width(s, 0)) == s
    assert orc._take(L.orc_pad_height(s, 0)) == s
    assert orc._take(L.orc_pad_width(s, 2)) == b"  ab\n  cd"
    assert orc._take(L.orc_pad_height(s, 2)) == b"\n\nab\ncd"


def test_grid_null_and_empty():
    # tests/unit/video/ascii_test.c:584-637: bad args -> NULL; two empty sources at 2x1 -> out_size 0
    assert orc.create_grid([], 80, 24) is None
    assert orc.create_grid([b"x"], 0, 24) is None
    assert orc.create_grid([b"", b""], 2, 1) == b""


# ---- (2) SURVEY.md 8(c) whole-frame anchors of the reference's output ---------------------
ANCHORS = [
    ("ascii_convert mono stretch", lambda g: orc.convert(g, 80, 24, False, False, False), 1635, 0xCEFD0A18),
    ("E2 NONE/FG aspect+pad", lambda g: orc.convert_with_caps(g, 80, 24, orc.COLOR_NONE, orc.RENDER_FG, True, True),
     1721, 0x7D62F78F),
    ("E2 256/FG", lambda g: orc.convert_with_caps(g, 80, 24, orc.COLOR_256, orc.RENDER_FG), 22255, 0xBE60A438),
    ("E2 TRUECOLOR/FG", lambda g: orc.convert_with_caps(g, 80, 24, orc.COLOR_TRUECOLOR, orc.RENDER_FG), 35852,
     0x885DA51D),
    ("E2 TRUECOLOR/HALF_BLOCK", lambda g: orc.convert_with_caps(g, 80, 24, orc.COLOR_TRUECOLOR, orc.RENDER_HALF_BLOCK),
     73802, 0x362719AD),
]


@pytest.mark.parametrize("name,fn,length,fnv", ANCHORS, ids=[a[0] for a in ANCHORS])
def test_survey_whole_frame_anchor(name, fn, length, fnv):
    out = fn(orc.frame_anchor_gradient())
    assert len(out) == length
    assert orc.fnv1a32(out) == fnv


def test_survey_anchor_prefixes():
    g = orc.frame_anchor_gradient()
    assert orc.convert_with_caps(g, 80, 24, 2, 0).startswith(b"\033[38;5;232m \033[38;5;232m ")
    assert orc.convert_with_caps(g, 80, 24, 3, 0).startswith(b"\033[38;2;0;0;0m \033[38;2;3;0;8m ")
    assert orc.convert_with_caps(g, 80, 24, 3, 2).startswith(b"\033[38;2;0;0;0m\033[48;2;0;5;10m\xe2\x96\x80")
    # the 640x480 gradient in mono only ever produces these glyphs
    mono = orc.convert(g, 80, 24, False, False, False)
    import re
    assert set(re.sub(rb"\033\[\d+b", b"", mono).replace(b"\n", b"")) <= set(b" .',")


def test_survey_aspect_ratio_values():
    # SURVEY 8(a) row A1 (verified against the reference by the survey)
    assert orc.aspect_ratio(1920, 1080, 80, 24) == (80, 23)
    assert orc.aspect_ratio(3840, 2160, 200, 60) == (200, 56)
    assert orc.aspect_ratio(3840, 2160, 400, 120) == (400, 113)
    assert orc.aspect_ratio(640, 480, 80, 24) == (64, 24)
    assert orc.aspect_ratio(160, 96, 160, 48) == (160, 48)


MODES = dict(mono=(0, 0), c16=(1, 0), c256=(2, 0), true=(3, 0), hb_true=(3, 2), hb256=(2, 2), hb16=(1, 2),
             hb_mono=(0, 2))


def test_survey_appendix_b_torture_lengths():
    # byte-identical to the reference on the 333x201 torture image; recorded lengths at 80x24
    t = orc.frame_torture()
    expect = dict(mono=1159, c16=11639, c256=22429, true=21664, hb_true=43121, hb256=15725, hb16=7697, hb_mono=3921)
    for k, (cl, rm) in MODES.items():
        assert len(orc.convert_with_caps(t, 80, 24, cl, rm)) == expect[k], k


def test_survey_appendix_b_multibyte_palettes():
    t = orc.frame_torture()
    for pal in (orc.PALETTE_BLOCKS, orc.PALETTE_COOL):
        got = [len(orc.convert_with_caps(t, 97, 31, cl, 0, palette=pal)) for cl in (0, 3, 2, 1)]
        assert got == [1274, 48984, 39632, 18196]
    # 16-colour fg with a built-in palette only ever emits spaces (+SGRs)
    out = orc.convert_with_caps(t, 97, 31, 1, 0, palette=orc.PALETTE_BLOCKS)
    import re
    assert set(re.sub(rb"\033\[\d+m", b"", out)) <= set(b" \n")


def test_survey_appendix_b_truecolor_background_is_dithered16():
    t = orc.frame_torture()
    assert len(orc.convert_with_caps(t, 97, 31, 3, 1)) == 34602


def test_survey_f5_text_grid_layout():
    # ascii_create_grid(.., 9, 160, 48) picks 4x3 cells of 39x15, output 7 728 bytes
    img = orc.frame_anchor_gradient()
    frames = [orc.convert(img, 39 + i, 15, False, False, False) for i in range(9)]
    out = orc.create_grid(frames, 160, 48)
    assert out.count(b"\n") == 48 and len(out) == 7728
    rows = out.split(b"\n")
    # separators: '|' after columns 39, 79, 119 and '_' rows at 15, 31
    assert rows[0][39:40] == b"|" or b"\033" in rows[0][:40]
    assert set(rows[15][:39]) == {ord("_")} and rows[15][39:40] == b"+"
    assert set(rows[31][:39]) == {ord("_")}


def test_survey_server_grid_layout_3x3():
    # nine 16:9 sources at 160x48 -> 3x3; composite cell 53x32 px, tiles 53x30 centred
    assert orc.grid_layout([(1920, 1080)] * 9, 160, 48) == (3, 3)
    srcs = [np.full((108, 192, 3), 10 + 20 * i, dtype=np.uint8) for i in range(9)]
    comp = orc.composite(srcs, 160, 48)
    assert comp.shape == (96, 160, 3)
    for i in range(9):
        r, c = divmod(i, 3)
        cell = comp[r * 32:(r + 1) * 32, c * 53:(c + 1) * 53]
        assert (cell[0] == 0).all() and (cell[31] == 0).all()
        assert (cell[1:31] == 10 + 20 * i).all()
    assert (comp[:, 159] == 0).all()


def test_reference_color_filter_known_answers():
    """tests/unit/video/color_filter_test.c: grayscale of the primaries (:19-48), the tint of every filter (:197-218; a
    white pixel takes exactly the filter's colour), white-on-colour CYAN and black-on-white BLACK on black / white /
    grey pixels (:95-152), NONE leaves the pixels alone (:157-166), invalid parameters return -1 (:170-192)."""
    px = lambda *rgb: np.array([[list(rgb)]], dtype=np.uint8)
    # WHITE (255,255,255) in white-on-colour mode is the grayscale itself: channel = 255 * gray / 255
    for rgb, lo, hi in (((255, 0, 0), 75, 79), ((0, 255, 0), 148, 152), ((0, 0, 255), 27, 31), ((255, 255, 255), 255, 255),
                        ((0, 0, 0), 0, 0), ((128, 128, 128), 126, 130)):
        g = orc.color_filter(px(*rgb), 2)[0, 0]
        assert g[0] == g[1] == g[2] and lo <= g[0] <= hi, (rgb, g)
    import json
    kats = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json"), encoding="utf-8"))
    tints = {flt: (r, g, b) for flt, (name, r, g, b) in enumerate(kats["color_filter_tints"], start=1) if name != "BLACK"}
    assert len(tints) == 10
    for flt, rgb in tints.items():
        assert tuple(orc.color_filter(px(255, 255, 255), flt)[0, 0]) == rgb, flt
        assert tuple(orc.color_filter(px(0, 0, 0), flt)[0, 0]) == (0, 0, 0), flt
    img = np.array([[[0, 0, 0], [255, 255, 255]], [[128, 128, 128], [64, 64, 64]]], dtype=np.uint8)
    cy = orc.color_filter(img, 8)
    assert cy[0, 0].tolist() == [0, 0, 0] and cy[0, 1].tolist() == [0, 255, 255]
    assert cy[1, 0, 0] == 0 and cy[1, 1, 0] == 0 and cy[1, 0, 1] > cy[1, 1, 1] > 0        # grey levels scale the tint
    bw = orc.color_filter(img[:1], 1)                                                     # BLACK: black on white
    assert (bw[0, 0] < 50).all() and bw[0, 1].tolist() == [255, 255, 255]
    assert np.array_equal(orc.color_filter(img, 0), img)
    L = orc.lib()
    buf = (C.c_uint8 * 3)(255, 255, 255)
    assert L.orc_color_filter(None, 1, 1, 3, 3) == -1
    assert L.orc_color_filter(buf, 0, 1, 3, 3) == -1 and L.orc_color_filter(buf, 1, 0, 3, 3) == -1
    assert L.orc_color_filter(buf, 1, 1, 0, 3) == -1 and L.orc_color_filter(buf, 1, 1, 3, 999) == -1
