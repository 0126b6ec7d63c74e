"""Every known-answer value the reference's own unit tests hold for this path (tests/golden/reference_kats.json,
extracted from the reference tree by tests/golden/make_reference_kats.py) checked against BOTH the CPU oracle and the
product library's exported host functions -- plus the behavioural assertions of the RLE-context tests
(tests/unit/util/ansi_fast_test.c:175-257 in the reference), replayed on the product's ansi_rle_* and, as two-pixel
frames, on the oracle's truecolor renderer.  No GPU needed: these entry points are host C."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import orc

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KATS = json.load(open(os.path.join(HERE, "golden", "reference_kats.json"), encoding="utf-8"))


class OutBuf(C.Structure):  # outbuf_t
    _fields_ = [("buf", C.c_void_p), ("len", C.c_size_t), ("cap", C.c_size_t)]


class RleCtx(C.Structure):  # ansi_rle_context_t
    _fields_ = [("buffer", C.c_void_p), ("capacity", C.c_size_t), ("length", C.c_size_t), ("mode", C.c_int),
                ("first_pixel", C.c_bool), ("last_r", C.c_uint8), ("last_g", C.c_uint8), ("last_b", C.c_uint8)]


@pytest.fixture(scope="module")
def prod():
    L = C.CDLL(os.path.join(ROOT, "ascii-chat_amd", "libasciichat_hip.so"))
    L.rep_is_profitable.restype = C.c_bool
    L.rep_is_profitable.argtypes = [C.c_uint32]
    L.rgb_to_16color.restype = C.c_uint8
    L.rgb_to_16color.argtypes = [C.c_uint8] * 3
    for f in (L.append_truecolor_fg, L.append_truecolor_bg):
        f.restype = C.c_void_p
        f.argtypes = [C.c_void_p, C.c_uint8, C.c_uint8, C.c_uint8]
    L.ob_u8.argtypes = [C.POINTER(OutBuf), C.c_uint8]
    L.ob_u32.argtypes = [C.POINTER(OutBuf), C.c_uint32]
    L.emit_rep.argtypes = [C.POINTER(OutBuf), C.c_uint32]
    L.ob_term.argtypes = [C.POINTER(OutBuf)]
    L.ansi_rle_init.argtypes = [C.POINTER(RleCtx), C.c_void_p, C.c_size_t, C.c_int]
    L.ansi_rle_add_pixel.argtypes = [C.POINTER(RleCtx), C.c_uint8, C.c_uint8, C.c_uint8, C.c_char]
    L.ansi_rle_finish.argtypes = [C.POINTER(RleCtx)]
    L.achip_palette_ascii_only.restype = C.c_bool
    L.achip_palette_ascii_only.argtypes = [C.c_char_p]
    for f in (L.append_16color_fg, L.append_16color_bg, L.append_256color_fg, L.append_256color_bg):
        f.restype = C.c_void_p
        f.argtypes = [C.c_void_p, C.c_uint8]
    L.get_16color_rgb.restype = None
    L.get_16color_rgb.argtypes = [C.c_uint8] + [C.POINTER(C.c_uint8)] * 3
    L.rgb_to_16color_dithered.restype = C.c_uint8
    L.rgb_to_16color_dithered.argtypes = [C.c_int] * 7 + [C.c_void_p]
    return L


def _ob_string(L, fn, value):
    ob = OutBuf()
    fn(C.byref(ob), value)
    L.ob_term(C.byref(ob))
    s = C.string_at(ob.buf, ob.len)
    assert C.string_at(ob.buf + ob.len, 1) == b"\0"  # ob_term: NUL written, not counted (output_buffer.c:80-84)
    C.CDLL(None).free(C.c_void_p(ob.buf))
    return s


def test_decimal_writers(prod):
    # output_buffer_test.c:143-201
    for v, s in KATS["ob_u8"]:
        assert _ob_string(prod, prod.ob_u8, v) == s.encode(), v
    for v, s in KATS["ob_u32"]:
        assert _ob_string(prod, prod.ob_u32, v) == s.encode(), v
    assert len(KATS["ob_u8"]) == 8 and len(KATS["ob_u32"]) == 11


def test_digits_and_rep_rule(prod):
    # output_buffer_test.c:337-351 (digits_u32), :295-305 (rep_is_profitable), :314-327 (emit_rep contains N)
    O = orc.lib()
    for v, d in KATS["digits_u32"]:
        assert O.orc_digits_u32(v) == d, v
        assert len(_ob_string(prod, prod.ob_u32, v)) == d, v  # the product has no digits_u32 export: via ob_u32
    for n, want in KATS["rep_is_profitable"]:
        assert bool(O.orc_rep_is_profitable(n)) == want and bool(prod.rep_is_profitable(n)) == want, n
    for n in KATS["emit_rep_contains"]:
        s = _ob_string(prod, prod.emit_rep, n)
        assert s == b"\033[%db" % n and str(n).encode() in s
    assert len(KATS["digits_u32"]) == 15 and len(KATS["rep_is_profitable"]) == 9


def test_colour_helpers(prod):
    # ansi_fast_test.c:458-493 (rgb_to_16color), :51-131 (truecolor SGR strings)
    O = orc.lib()
    for r, g, b, idx in KATS["rgb_to_16color"]:
        assert O.orc_rgb_to_16(r, g, b) == idx and prod.rgb_to_16color(r, g, b) == idx, (r, g, b)
    buf = C.create_string_buffer(64)
    for r, g, b, s in KATS["truecolor_sgr"]:
        bg = s.startswith("\x1b[48")
        end = (prod.append_truecolor_bg if bg else prod.append_truecolor_fg)(buf, r, g, b)
        n = end - C.addressof(buf)
        assert buf.raw[:n] == s.encode() and n == len(s), s
        ob = C.create_string_buffer(64)
        k = O.orc_sgr_truecolor(ob, 1 if bg else 0, r, g, b)
        assert ob.raw[:k] == s.encode()
    assert len(KATS["rgb_to_16color"]) == 9 and len(KATS["truecolor_sgr"]) == 14


def test_indexed_colour_helpers_and_dither_pixel(prod):
    """ansi_fast_test.c:289-456 (append_256color_* / append_16color_* strings incl. the defaults an invalid index takes),
    :495-530 (get_16color_rgb, invalid index -> light grey), :550-572 (rgb_to_16color_dithered: NULL buffer, right edge,
    bottom edge, corner).  Checked on the oracle and on the product's exported functions."""
    O = orc.lib()
    buf = C.create_string_buffer(64)
    for kind, where, n, st in KATS["indexed_sgr"]:
        fn = getattr(prod, "append_%scolor_%s" % (kind, where))
        end = fn(buf, n)
        assert buf.raw[:end - C.addressof(buf)] == st.encode(), (kind, where, n)
        ob = C.create_string_buffer(64)
        k = (O.orc_sgr_16 if kind == "16" else O.orc_sgr_256)(ob, 1 if where == "bg" else 0, n)
        assert ob.raw[:k] == st.encode(), (kind, where, n)
    for i, r, g, b in KATS["get_16color_rgb"]:
        got = [C.c_uint8() for _ in range(3)]
        prod.get_16color_rgb(i, *[C.byref(v) for v in got])
        assert [v.value for v in got] == [r, g, b], i
    for r, g, b, x, y, w, h, with_buffer, idx in KATS["rgb_to_16color_dithered"]:
        err = (C.c_int * (3 * w * h))() if with_buffer else None
        assert prod.rgb_to_16color_dithered(r, g, b, x, y, w, h, err) == idx, (x, y, with_buffer)
        if with_buffer:  # nothing leaves the 10 x 10 buffer at an edge, and what (255,0,0) -> bright red leaves is zero
            assert not any(err)
    # the exported per-pixel step chained over a whole image equals the oracle's dithered renderer (which the GPU pass is
    # tested against): same colour index for every pixel of a 23 x 9 noise image
    img = orc.frame_hash_noise(23, 9, 4)
    h, w, _ = img.shape
    err = (C.c_int * (3 * w * h))()
    idx = [[prod.rgb_to_16color_dithered(int(img[y, x, 0]), int(img[y, x, 1]), int(img[y, x, 2]), x, y, w, h, err)
            for x in range(w)] for y in range(h)]
    ref = orc.print_16_dithered(img, True)  # ESC[4x/10xm ESC[97|30m glyph per cell: the background code is the index
    import re as _re
    codes = [int(m) for m in _re.findall(rb"\x1b\[(4[0-7]|10[0-7])m", ref)]
    flat = [c - 40 if c < 100 else c - 92 for c in codes]
    assert flat == [v for row in idx for v in row]
    assert len(KATS["indexed_sgr"]) == 15 and len(KATS["get_16color_rgb"]) == 5 and len(KATS["rgb_to_16color_dithered"]) == 4


def test_colour_filter_tints_from_the_reference_table():
    """color_filter_test.c:197-218: the tint of each filter, in color_filter_t order (BLACK = 1 ... YELLOW = 11): a white
    pixel takes exactly the filter's colour (BLACK is black-on-white: white stays white)."""
    px = np.array([[[255, 255, 255]]], dtype=np.uint8)
    assert [t[0] for t in KATS["color_filter_tints"]] == ["BLACK", "WHITE", "GREEN", "MAGENTA", "FUCHSIA", "ORANGE", "TEAL",
                                                         "CYAN", "PINK", "RED", "YELLOW"]
    for flt, (name, r, g, b) in enumerate(KATS["color_filter_tints"], start=1):
        got = tuple(int(v) for v in orc.color_filter(px, flt)[0, 0])
        assert got == ((255, 255, 255) if name == "BLACK" else (r, g, b)), name


def test_builtin_palettes(prod):
    # palette.h:161-197 (strings), palette_test.c:24-28 / :70-74 (which palettes require UTF-8)
    mine = {"STANDARD": orc.PALETTE_STANDARD, "BLOCKS": orc.PALETTE_BLOCKS, "DIGITAL": orc.PALETTE_DIGITAL,
            "MINIMAL": orc.PALETTE_MINIMAL, "COOL": orc.PALETTE_COOL}
    for name, chars in KATS["palette_chars"].items():
        assert mine[name] == chars, name
    hdr = open(os.path.join(ROOT, "include", "asciichat_render.h"), encoding="utf-8").read()
    for name, chars in KATS["palette_chars"].items():  # the public header carries the same constants
        assert ('PALETTE_CHARS_%s "%s"' % (name, chars)) in hdr, name
    for name, _, utf8 in KATS["builtin_palettes"]:
        assert prod.achip_palette_ascii_only(KATS["palette_chars"][name].encode()) == (not utf8), name
    for name, utf8 in KATS["palette_requires_utf8"]:
        assert any(ord(c) > 127 for c in KATS["palette_chars"][name]) == utf8, name
    assert {n: len(c) for n, c in KATS["palette_chars"].items()} == dict(STANDARD=23, BLOCKS=11, DIGITAL=10, MINIMAL=8, COOL=11)


def test_crc32c_known_answers():
    # crc32_hw_test.c:19-20 (empty -> 0), :33-49 ("Hello, World!" -> 0x4d551068, CRC-32C)
    for s, v in KATS["crc32c"]:
        assert orc.crc32c(s.encode()) == v, s


def test_rle_context_behaviour(prod):
    # ansi_fast_test.c:175-257, replayed on the product's exported ansi_rle_* (hostutil.c)
    buf = C.create_string_buffer(256)
    ctx = RleCtx()
    prod.ansi_rle_init(C.byref(ctx), buf, 256, 0)
    assert (ctx.capacity, ctx.length, ctx.mode, ctx.first_pixel) == (256, 0, 0, True)
    assert (ctx.last_r, ctx.last_g, ctx.last_b) == (0xFF, 0xFF, 0xFF)
    prod.ansi_rle_add_pixel(C.byref(ctx), 255, 128, 64, b"A")
    assert ctx.length > 0 and (ctx.last_r, ctx.last_g, ctx.last_b, ctx.first_pixel) == (255, 128, 64, False)
    assert buf.raw[ctx.length - 1:ctx.length] == b"A"
    first = ctx.length
    assert buf.raw[:first] == b"\033[38;2;255;128;64mA"
    prod.ansi_rle_add_pixel(C.byref(ctx), 255, 128, 64, b"B")          # same colour: exactly one more byte
    assert ctx.length == first + 1 and buf.raw[ctx.length - 1:ctx.length] == b"B"
    prod.ansi_rle_add_pixel(C.byref(ctx), 100, 200, 50, b"C")          # different colour: SGR + glyph
    assert ctx.length > first + 2 and (ctx.last_r, ctx.last_g, ctx.last_b) == (100, 200, 50)
    before = ctx.length
    prod.ansi_rle_finish(C.byref(ctx))
    assert ctx.length > before and buf.raw[ctx.length:ctx.length + 1] == b"\0" and b"\033[0m" in buf.raw[:ctx.length]
    assert buf.raw[:ctx.length] == b"\033[38;2;255;128;64mAB\033[38;2;100;200;50mC\033[0m"


def test_rle_context_behaviour_as_frames_through_the_oracle():
    """The same three assertions on whole frames: image_print_color drives ansi_rle_* per pixel (foreground.c:268-303),
    so a 2x1 image of equal pixels is one byte longer than a 1x1 image, a 2x1 image of different pixels is longer by an
    SGR, and every truecolor frame ends with the single ESC[0m of ansi_rle_finish."""
    a, b = (255, 128, 64), (100, 200, 50)
    one = orc.print_with_caps(np.array([[a]], dtype=np.uint8), 3, 0)
    same = orc.print_with_caps(np.array([[a, a]], dtype=np.uint8), 3, 0)
    diff = orc.print_with_caps(np.array([[a, b]], dtype=np.uint8), 3, 0)
    assert one.startswith(b"\033[38;2;255;128;64m") and one.endswith(b"\033[0m") and one.count(b"\033[0m") == 1
    assert len(same) == len(one) + 1 and same.count(b"\033[38;2;") == 1
    assert len(diff) == len(one) + 1 + len(b"\033[38;2;100;200;50m") and diff.endswith(b"\033[0m")
    two_rows = orc.print_with_caps(np.array([[a], [a]], dtype=np.uint8), 3, 0)    # the state survives the row end
    assert two_rows.count(b"\033[38;2;") == 1 and two_rows.count(b"\n") == 1
