"""CPU-side checks of the product library: it loads, exports every symbol the public headers declare, and
its host-side (non-GPU) logic agrees with the oracle.  No GPU compute is invoked here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import emu  # noqa: E402
import orc  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


@pytest.fixture(scope="module")
def pkg():
    p = load_package()
    p.build()
    p.lib()
    return p


def declared_functions(header):
    text = open(header, encoding="utf-8").read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"^\s*#.*?$", "", text, flags=re.M)
    text = re.sub(r"typedef\s+(struct|enum)\s*\w*\s*\{.*?\}\s*[^;]*;", "", text, flags=re.S)
    names = []
    for stmt in text.split(";"):
        s = " ".join(stmt.split())
        if not s or s.startswith("typedef") or "(" not in s or s.startswith("extern \"C\""):
            s = s.replace('extern "C" {', "").strip()
            if not s or s.startswith("typedef") or "(" not in s:
                continue
        m = re.match(r"^[\w\s\*]+?\b(\w+)\s*\(", s)
        if m and m.group(1) not in ("defined", "__attribute__"):
            names.append(m.group(1))
    return names


def test_library_exports_every_declared_symbol(pkg):
    L = C.CDLL(pkg.LIB_PATH)
    missing = []
    total = 0
    for h in ("asciichat_hip.h", "asciichat_render.h", "achip_host.h"):
        names = declared_functions(os.path.join(ROOT, "include", h))
        assert len(names) >= 8, (h, names)
        for n in names:
            total += 1
            if not hasattr(L, n):
                missing.append((h, n))
    assert not missing, missing
    assert total >= 85
    nm = subprocess.run(["nm", "-D", "--defined-only", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert " g_default_luminance_palette" in nm


def test_library_exports_only_the_abi(pkg):
    """ascii-chat_amd/exports.map: the defined dynamic symbols are the drop-in surface the public headers declare, the
    batch API (asciichat_hip_*) and the host helpers (achip_*) -- no kernel launch stubs, no cross-file helpers."""
    nm = subprocess.run(["nm", "-D", "--defined-only", pkg.LIB_PATH], capture_output=True, text=True, check=True).stdout
    syms = [l.split()[-1] for l in nm.splitlines() if l.strip()]
    declared = set()
    for h in ("asciichat_hip.h", "asciichat_render.h", "achip_host.h"):
        declared.update(declared_functions(os.path.join(ROOT, "include", h)))
    declared.add("g_default_luminance_palette")
    stray = [s for s in syms if not s.startswith(("asciichat_hip_", "achip_")) and s not in declared]
    assert not stray, stray
    assert not [s for s in syms if s.startswith("_Z")], "C++ symbols leave the library"
    assert len(syms) < 260, len(syms)


def test_no_gpu_means_loud_failure_not_fallback(pkg):
    L = pkg.lib()
    if L.asciichat_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    f = pkg.frame_setup(0x1000, 640, 480, 80, 24, 0)
    with pytest.raises(RuntimeError, match="no HIP device"):
        pkg.Plan(1, orc.PALETTE_STANDARD, [f])
    arr = np.zeros((48, 64, 3), np.uint8)
    im = pkg.Image(64, 48, arr.ctypes.data, 0)
    caps = pkg.TermCaps()
    assert not L.ascii_convert_with_capabilities(C.byref(im), 80, 24, C.byref(caps), False, False, b"ab")
    assert b"no HIP device" in L.asciichat_hip_last_error()
    assert not L.image_print(C.byref(im), b"ab")


def test_product_and_oracle_do_not_share_code(pkg):
    """The product library must not link, load or reference anything under oracle/."""
    deps = subprocess.run(["readelf", "-d", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in deps
    nm = subprocess.run(["nm", "-D", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert " orc_" not in nm
    for root, _, files in os.walk(os.path.join(ROOT, "ascii-chat_amd")):
        for fn in files:
            if fn.endswith((".c", ".h", ".hpp", ".hip", ".py")) or fn == "Makefile":
                src = open(os.path.join(root, fn), encoding="utf-8", errors="replace").read()
                assert "oracle/" not in src and "orc_" not in src, fn
                # the CPU fiber emulator is a test-only build of the same kernel sources: nothing of it in the product tree
                assert "ACHIP_HIPEMU" not in src and "hip_emu.h" not in src and "hipemu::" not in src, fn
    assert "hipemu" not in subprocess.run(["nm", "-C", pkg.LIB_PATH], capture_output=True, text=True).stdout


def test_aspect_ratio_matches_oracle(pkg):
    L = pkg.lib()
    rng = np.random.default_rng(3)
    cases = [(1920, 1080, 80, 24), (3840, 2160, 200, 60), (3840, 2160, 400, 120), (640, 480, 80, 24), (160, 96, 160, 48),
             (1, 1, 1, 1), (10000, 1, 80, 24), (1, 10000, 80, 24), (0, 5, 80, 24)]
    cases += [tuple(int(v) for v in rng.integers(1, 4000, 4)) for _ in range(500)]
    for iw, ih, w, h in cases:
        for stretch in (False, True):
            ow, oh = C.c_ssize_t(), C.c_ssize_t()
            L.aspect_ratio(iw, ih, w, h, stretch, C.byref(ow), C.byref(oh))
            assert (ow.value, oh.value) == orc.aspect_ratio(iw, ih, w, h, stretch), (iw, ih, w, h, stretch)


def test_frame_setup_matches_reference_null_conditions(pkg):
    # ascii.c:204 (source dims in [1,10000]), :256-265 (positive target), image_new() limits 3840x2160
    assert pkg.frame_setup(1, 0, 10, 80, 24, 0) is None
    assert pkg.frame_setup(1, 10001, 10, 80, 24, 0) is None
    assert pkg.frame_setup(1, 100, 100, 0, 24, 0) is None
    assert pkg.frame_setup(1, 100, 100, 80, -1, 0) is None
    assert pkg.frame_setup(1, 100, 100, 3841, 10, 0) is None
    assert pkg.frame_setup(1, 100, 100, 80, 1081, 2) is None  # doubled height 2162 > 2160
    f = pkg.frame_setup(1, 1920, 1080, 80, 24, 2, True, True, False)
    assert (f.out_w, f.out_h, f.pad_left, f.pad_top) == (80, 46, 0, 0)
    f = pkg.frame_setup(1, 640, 480, 80, 24, 0, True, True, False)
    assert (f.out_w, f.out_h, f.pad_left, f.pad_top) == (64, 24, 8, 0)
    f = pkg.frame_setup(1, 640, 480, 80, 24, 0, False, True, False)  # wants_padding off -> no padding
    assert (f.pad_left, f.pad_top) == (0, 0)
    assert pkg.lib().achip_mode_from_caps(3, 1) == 9  # TRUECOLOR+BACKGROUND = Floyd-Steinberg 16-colour (sgr.c:429)
    assert [pkg.lib().achip_mode_from_caps(c, 0) for c in (-1, 0, 1, 2, 3)] == [0, 0, 3, 2, 1]
    assert [pkg.lib().achip_mode_from_caps(c, 2) for c in (-1, 0, 1, 2, 3)] == [8, 8, 7, 6, 5]


def test_out_bound_is_an_upper_bound(pkg):
    """achip_out_bound() must dominate the true output length for adversarial inputs in every mode."""
    from achip_ctypes import ALL_MODES, MODE_CAPS, MODE_TRUE_BG
    imgs = [orc.frame_noise(64, 48, 1), np.full((48, 64, 3), 255, np.uint8), orc.frame_bars(64, 48, 2),
            (orc.frame_noise(64, 48, 9) | 0x80)]
    for mode in ALL_MODES:
        for pal in (orc.PALETTE_STANDARD, "😀😀😀😀"):
            for im in imgs:
                for (W, H) in ((64, 48), (7, 3), (640, 5)):
                    rm = MODE_CAPS.get(mode, (3, 0))[1]
                    f = emu.frame_for_convert(im, W, H, rm, True, True)
                    if f is None:
                        continue
                    bound = emu.lib().achip_out_bound(mode, C.byref(f))
                    got = emu.render_frames(mode, [f], pal, 0)[0]
                    assert isinstance(got, bytes) and len(got) <= bound, (mode, W, H)


def test_host_utilities_match_oracle(pkg):
    L = pkg.lib()
    # scalar colour helpers over the full cube (subsampled) -- these are per-value host functions
    OL = orc.lib()
    for r in range(0, 256, 5):
        for g in range(0, 256, 7):
            for b in range(0, 256, 11):
                assert L.rgb_to_256color(r, g, b) == OL.orc_rgb_to_256(r, g, b)
                assert L.rgb_to_16color(r, g, b) == OL.orc_rgb_to_16(r, g, b)
    for n in list(range(0, 40)) + [99, 100, 101, 1000, 3840]:
        assert L.rep_is_profitable(n) == OL.orc_rep_is_profitable(n)
    buf = C.create_string_buffer(64)
    end = L.append_truecolor_fg(buf, 255, 128, 64)
    assert buf.raw[:end - C.addressof(buf)] == b"\033[38;2;255;128;64m"
    # text grid + padding against the oracle (SURVEY App. B list)
    img = orc.frame_anchor_gradient()
    frames = [orc.convert(img, 39 + i, 15, False, False, False) for i in range(9)]
    frames[3] = orc.convert_with_caps(img, 30, 10, 3, 0)  # escape-laden source
    for n, (w, h) in ((9, (160, 48)), (4, (160, 48)), (2, (80, 24)), (3, (120, 40)), (5, (200, 60)), (1, (80, 24)),
                      (2, (60, 40)), (9, (80, 24)), (9, (20, 6))):
        arr = (pkg.FrameSource * n)()
        for i in range(n):
            arr[i].frame_data = frames[i]
            arr[i].frame_size = len(frames[i])
        sz = C.c_size_t()
        p = L.ascii_create_grid(arr, n, w, h, C.byref(sz))
        got = C.string_at(p, sz.value)
        L.free(p)
        assert got == orc.create_grid(frames[:n], w, h), (n, w, h)
    # escape-laden lines in the last grid cell: pasting them raw would run past the canvas; the reference's
    # SAFE_MEMCPY refuses such a copy as a whole (lib/platform/posix/system.c:653-666) and so do product and oracle
    heavy = [orc.convert_with_caps(img, 40, 12, 3, 0) for _ in range(4)]
    for (w, h) in ((41, 13), (60, 9), (25, 7), (100, 30)):
        arr = (pkg.FrameSource * 4)()
        for i in range(4):
            arr[i].frame_data = heavy[i]
            arr[i].frame_size = len(heavy[i])
        p = L.ascii_create_grid(arr, 4, w, h, C.byref(sz))
        got = C.string_at(p, sz.value)
        L.free(p)
        assert got == orc.create_grid(heavy, w, h), (w, h)
        assert len(got) <= w * h + h
    assert not L.ascii_create_grid(None, 1, 80, 24, C.byref(sz))


def test_palette_cache_matches_oracle_and_is_shared(pkg):
    """get_utf8_palette_cache (common.c:270-377): borrowed pointer, same tables as the oracle's L2/L3 restatement."""
    L = C.CDLL(pkg.LIB_PATH)
    OL = orc.lib()

    class U8Char(C.Structure):
        _fields_ = [("width", C.c_uint8), ("bytes", C.c_uint8 * 4), ("len", C.c_uint8), ("pad", C.c_uint8)]

    class Cache(C.Structure):
        _fields_ = [("cache", U8Char * 256), ("cache64", U8Char * 64), ("ramp", C.c_uint8 * 256)]

    class OGlyph(C.Structure):
        _fields_ = [("bytes", C.c_uint8 * 4), ("len", C.c_uint8)]

    class OPal(C.Structure):
        _fields_ = [("cache", OGlyph * 256), ("cache64", OGlyph * 64), ("ramp", C.c_uint8 * 64), ("count", C.c_int)]

    L.get_utf8_palette_cache.restype = C.POINTER(Cache)
    L.get_utf8_palette_cache.argtypes = [C.c_char_p]
    OL.orc_palette_build.argtypes = [C.c_char_p, C.POINTER(OPal)]
    seen = {}
    for pal in (orc.PALETTE_STANDARD, orc.PALETTE_BLOCKS, orc.PALETTE_COOL, orc.PALETTE_DIGITAL, orc.PALETTE_MINIMAL, "ab"):
        c = L.get_utf8_palette_cache(pal.encode())
        assert c, pal
        o = OPal()
        OL.orc_palette_build(pal.encode(), C.byref(o))
        for i in range(256):
            g, e = c.contents.cache[i], o.cache[i]
            assert g.len == e.len and bytes(g.bytes)[:g.len] == bytes(e.bytes)[:e.len], (pal, i)
        for i in range(64):
            g, e = c.contents.cache64[i], o.cache64[i]
            assert g.len == e.len and bytes(g.bytes)[:g.len] == bytes(e.bytes)[:e.len], (pal, i)
            assert c.contents.ramp[i] == o.ramp[i]
        seen[pal] = C.addressof(c.contents)
        assert C.addressof(L.get_utf8_palette_cache(pal.encode()).contents) == seen[pal]  # cached, not rebuilt
    assert len(set(seen.values())) == len(seen)
    assert not L.get_utf8_palette_cache(None) and not L.get_utf8_palette_cache(b"")


def test_row_divisor_magic_is_exact():
    """The frame kernel divides cell indices by the padded row width with umulhi(i, m), m = uint32(2^32 / wp in
    float32) + 65 (render_kernels.hpp): exact for every row width and cell index a chunk can hold."""
    for wp in range(2, 4097):
        m = int(np.uint32(np.float32(4294967296.0) / np.float32(wp))) + 65
        i = np.arange(0, 4097, dtype=np.uint64)
        assert (((i * np.uint64(m)) >> np.uint64(32)) == i // np.uint64(wp)).all(), wp


def test_rle_and_frame_validator_utilities(pkg):
    """SURVEY 8f.4 leftovers (rle.c, frame_validator.c): product vs the oracle's independent restatement, plus the
    properties that make them useful as checkers: expanding a REP-compressed mono frame gives W characters per
    line, and compressing that expansion again never grows it."""
    L = C.CDLL(pkg.LIB_PATH)
    OL = orc.lib()
    for f in (L.ansi_expand_rle, L.ansi_compress_rle):
        f.restype = C.c_void_p
        f.argtypes = [C.c_char_p, C.c_size_t]
    L.frame_validate_integrity.restype = C.c_bool
    L.frame_validate_integrity.argtypes = [C.c_char_p, C.c_size_t]
    L.frame_get_valid_end.restype = C.c_size_t
    L.frame_get_valid_end.argtypes = [C.c_char_p, C.c_size_t]
    L.free.argtypes = [C.c_void_p]

    def prod(fn, b):
        p = fn(b, len(b))
        if not p:
            return None
        out = C.string_at(p)
        L.free(p)
        return out

    img = orc.frame_bars(400, 300, 2)
    mono = orc.convert(img, 120, 30, False, False, False)
    tc = orc.convert_with_caps(orc.frame_torture(), 80, 24, 3, 0)
    hb = orc.convert_with_caps(orc.frame_torture(), 80, 24, 3, 2)
    rng = np.random.default_rng(3)
    junk = [bytes(rng.integers(0, 256, 300, dtype=np.uint8)).replace(b"\0", b"\x01") for _ in range(4)]
    cases = [mono, tc, hb, b"a\033[5b", b"\033[3bX", b"ab\033[0b", b"x\033[2;3bz", b"\033[", b"ab\033[12", b"aaaaaaaaaa", b"aaaaa",
             "▀▀▀▀▀▀▀▀".encode(), b"\t\t\t\t\t\t\t", b"~\x7f\x7f\x7f\x7f\x7f\x7f\x7f", b"z" * 5000] + junk
    for b in cases:
        assert prod(L.ansi_expand_rle, b) == orc.expand_rle(b), b[:40]
        assert prod(L.ansi_compress_rle, b) == orc.compress_rle(b), b[:40]
        assert L.frame_validate_integrity(b, len(b)) == bool(OL.orc_frame_validate_integrity(b, len(b)))
        assert L.frame_get_valid_end(b, len(b)) == OL.orc_frame_get_valid_end(b, len(b))
    assert prod(L.ansi_expand_rle, b"") is None and prod(L.ansi_compress_rle, b"") is None
    # properties on real frames
    lines = orc.expand_rle(mono).split(b"\n")
    assert len(lines) == 30 and all(len(l) == 120 for l in lines)
    again = orc.compress_rle(orc.expand_rle(mono))
    assert orc.expand_rle(again) == orc.expand_rle(mono) and len(again) <= len(orc.expand_rle(mono))
    assert L.frame_validate_integrity(tc, len(tc)) and L.frame_validate_integrity(hb, len(hb))
    assert not L.frame_validate_integrity(mono, len(mono))              # mono frames carry no reset at all
    assert not L.frame_validate_integrity(tc + b"xx", len(tc) + 2) and L.frame_get_valid_end(tc + b"xx", len(tc) + 2) == len(tc)


def test_rainbow_colour_and_host_string_pass(pkg):
    """COLOR_FILTER_RAINBOW (color_filter.c:169-243, 348-408): the colour of the moment (float HSV walk + luminance floor)
    and the host-string replacement, product vs oracle, plus known points of the colour wheel."""
    L = pkg.lib()
    r, g, b = C.c_uint8(), C.c_uint8(), C.c_uint8()

    def colour(t, fn=L.achip_rainbow_color):
        fn(t, C.byref(r), C.byref(g), C.byref(b))
        return (r.value, g.value, b.value)

    assert colour(0.0)[0] == 255 and colour(0.0)[1] == colour(0.0)[2]  # t = 0: red, lifted towards white by the floor
    assert colour(3.5 / 3)[1] == 255                                   # a third of the period: green
    lum = lambda c: 0.2126 * c[0] + 0.7152 * c[1] + 0.0722 * c[2]
    ts = [k * 0.0137 for k in range(600)] + [-1.0, -0.2, 1e6, 123456.78, 3.5, 7.0, 3.4999]
    for t in ts:
        c = colour(t)
        assert c == orc.rainbow_color(t), t
        assert c == colour(t, L.color_filter_calculate_rainbow)
        # the boost truncates each channel, so the floor is approached from below by at most one step per channel
        assert lum(c) >= 119.0 or 255 in c, (t, c)
    tc = orc.convert_with_caps(orc.frame_torture(), 80, 24, 3, 0)
    hb = orc.convert_with_caps(orc.frame_torture(), 80, 24, 3, 2)
    mono = orc.convert(orc.frame_torture(), 80, 24, False, False, False)
    for frame in (tc, hb, b"\033[38;2;1;2;3mA", b"xy\033[38;2;1;2;3", b"A\033[38;2;9;9;9mB\033[38;2;7;7", b"\033[38;2;m"):
        for t in (0.0, 0.9, 2.2):
            p = L.rainbow_replace_ansi_colors(frame, t)
            assert p
            got = pkg.take_string(p)
            assert got == orc.rainbow_replace(frame, t), (frame[:20], t)
            code = b"\033[38;2;%d;%d;%dm" % colour(t)
            if frame in (tc, hb):  # well-formed: every foreground SGR now carries the one colour, nothing else moved
                assert got.count(code) == frame.count(b"\033[38;2;") == got.count(b"\033[38;2;")
                assert got.count(b"\033[48;2;") == frame.count(b"\033[48;2;")
    assert not L.rainbow_replace_ansi_colors(mono, 1.0)       # nothing to recolour: NULL, the caller keeps its string
    assert not L.rainbow_replace_ansi_colors(None, 1.0)
    assert orc.rainbow_replace(mono, 1.0) == mono


def test_frame_blob_validation_matches_reference_rules(pkg):
    """achip_frame_blob_parse vs the oracle's restatement of stream.c:330-372 / protocol.c:784-815."""
    import struct
    L = emu.lib()

    def parse(blob, exact):
        w, h, px = C.c_uint32(), C.c_uint32(), C.c_void_p()
        rc = L.achip_frame_blob_parse(blob, len(blob), exact, C.byref(w), C.byref(h), C.byref(px))
        return (rc, (w.value, h.value) if rc == 0 else None)

    def blob(w, h, extra=0):
        n = max(0, w * h * 3 + extra) if 0 < w <= 4096 and 0 < h <= 4096 else max(0, 3 + extra)
        return struct.pack(">II", w & 0xFFFFFFFF, h & 0xFFFFFFFF) + bytes(n)

    cases = [blob(4, 3), blob(4, 3, 5), blob(4, 3, -1), blob(1, 1), blob(1, 1, -1), blob(0, 5), blob(5, 0), blob(3840, 1),
             blob(3841, 1), blob(4096, 1), blob(4097, 1), blob(1, 2160), blob(1, 2161), blob(640, 480), blob(640, 480, 1),
             b"", b"\0" * 7, b"\0" * 10, struct.pack(">II", 0xBEBEBEBE, 0xBEBEBEBE) + bytes(16),
             struct.pack(">II", 0xFFFFFFFF, 0xFFFFFFFF) + bytes(16)]
    for b in cases:
        for exact in (False, True):
            rc, dims = parse(b, exact)
            want = orc.frame_blob_accept(b, exact) if len(b) >= 8 else None
            assert dims == want, (len(b), b[:8], exact, rc)
            assert (rc == 0) == (want is not None)
    assert parse(b"\0" * 10, False)[0] == -1 and parse(blob(0, 5), False)[0] == -2 and parse(blob(4, 3, -1), False)[0] == -3


def test_composite_geometry_and_kernel_emulated(pkg):
    """achip_composite_setup + the composite/fused sampler (run under the emulator) vs the oracle's C1+C2."""
    EL = emu.lib()
    for n, (tw, th), dims in ((9, (160, 48), [(192, 108)] * 9), (4, (160, 48), [(192, 108)] * 4),
                              (2, (120, 40), [(64, 64), (200, 50)]), (5, (200, 60), [(96, 54), (54, 96)] * 2 + [(33, 77)]),
                              (3, (80, 24), [(80, 60)] * 3)):
        imgs = [orc.frame_hash_noise(w, h, 50 + i) for i, (w, h) in enumerate(dims)]
        ptrs = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
        ws = (C.c_int * n)(*[d[0] for d in dims])
        hs = (C.c_int * n)(*[d[1] for d in dims])
        comp = emu.Composite()
        EL.achip_composite_setup(C.byref(comp), ptrs, ws, hs, n, tw, th)
        assert (comp.cols, comp.rows) == orc.grid_layout(dims, tw, th)
        ref = orc.composite(imgs, tw, th)
        out = np.zeros((2 * th, tw, 3), np.uint8)
        EL.emu_composite(C.byref(comp), out.ctypes.data)
        assert np.array_equal(out, ref), (n, tw, th)
        # fused: render from the virtual canvas without materialising it
        for mode, (cl, rm) in ((1, (3, 0)), (5, (3, 2))):
            h = 2 * th if rm == 2 else th
            f = emu.Frame()
            assert EL.achip_frame_setup(C.byref(f), None, tw, 2 * th, tw, h, rm, True, True, False) == 0
            f.comp = C.addressof(comp)
            exp = orc.convert_with_caps(ref, tw, h, cl, rm, True, True, False)
            # the phase kernel (whole frame, and cut into row bands) and the stream kernel's composite instantiations:
            # all sample through the LDS copy of the descriptor (comp_stage / sample_composite_lds)
            for variant, rpp in ((2, 0), (2, 5)) + (((20, 0), (17, 0)) if mode == 1 else ()):
                got = emu.render_frames(mode, [f, f], orc.PALETTE_STANDARD, variant, rows_per_part=rpp)
                assert got[0] == exp and got[1] == exp, (n, mode, variant, rpp)


def test_composite_skips_clients_without_video(pkg):
    """a client without video takes no grid cell and does not count in the layout
    (calculate_optimal_grid_layout gets sources_with_video, src/server/stream.c:523-558, 671, 690-692).
    Anchor derived by hand from those lines: five clients, the third without video, 16:9 sources at 160x48 -> four
    cells: cols = rows = 2 (utilisation 22*80/(80*24) = 0.917 beats 3x2's 0.583), cell 80x48 px on the 160x96 canvas,
    every tile 80 x (int)(80 / (16/9) + 0.5) = 80x45 centred with a 1-px top margin.  (Counting all five -- the bug --
    gives 2 columns x 3 rows of 80x32 px cells.)"""
    EL = emu.lib()
    imgs = [orc.frame_hash_noise(192, 108, 70 + i) for i in range(5)]
    imgs[2] = None
    ptrs = (C.c_void_p * 5)(*[None if i is None else i.ctypes.data for i in imgs])
    ws, hs = (C.c_int * 5)(*[192] * 5), (C.c_int * 5)(*[108] * 5)
    comp = emu.Composite()
    EL.achip_composite_setup(C.byref(comp), ptrs, ws, hs, 5, 160, 48)
    assert (comp.cols, comp.rows, comp.cell_w, comp.cell_h, comp.n_src) == (2, 2, 80, 48, 4)
    for k, src_i in enumerate((0, 1, 3, 4)):
        s = comp.s[k]
        assert (s.tile_w, s.tile_h, s.org_x, s.org_y) == (80, 45, (k % 2) * 80, (k // 2) * 48 + 1), k
        assert s.src == imgs[src_i].ctypes.data
    ref = orc.composite(imgs, 160, 48)
    out = np.zeros((96, 160, 3), np.uint8)
    EL.emu_composite(C.byref(comp), out.ctypes.data)
    assert np.array_equal(out, ref)
    assert np.array_equal(ref[1:46, 0:80], orc.resize_nn(imgs[0], 80, 45)) and not ref[0].any() and not ref[46:48, 0:80].any()
    assert np.array_equal(ref[49:94, 80:160], orc.resize_nn(imgs[4], 80, 45))
    # nobody has video: an empty (black) canvas; one client with video: one cell
    none = (C.c_void_p * 2)(None, None)
    EL.achip_composite_setup(C.byref(comp), none, ws, hs, 2, 160, 48)
    assert (comp.cols, comp.rows, comp.n_src) == (0, 0, 0)
    one = (C.c_void_p * 3)(None, imgs[1].ctypes.data, None)
    EL.achip_composite_setup(C.byref(comp), one, ws, hs, 3, 160, 48)
    assert (comp.cols, comp.rows, comp.n_src) == (1, 1, 1) and comp.s[0].src == imgs[1].ctypes.data
    assert np.array_equal(orc.composite([None, imgs[1], None], 160, 48)[3:93, :], orc.resize_nn(imgs[1], 160, 90))


def test_shard_partition_is_balanced(pkg):
    """achip_shard_bounds / _owner / _slots (comm.c): contiguous and balanced -- nine sources over eight GPUs leave
    no rank idle (the block partition left ranks 5-7 without work)."""
    L = pkg.lib()
    f, c = C.c_int(), C.c_int()

    def bounds(n, w):
        out = []
        for r in range(w):
            L.achip_shard_bounds(n, w, r, C.byref(f), C.byref(c))
            out.append((f.value, c.value))
        return out

    assert bounds(9, 8) == [(0, 2), (2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (7, 1), (8, 1)]
    assert bounds(256, 8) == [(32 * r, 32) for r in range(8)]
    assert bounds(5, 2) == [(0, 3), (3, 2)] and bounds(3, 4) == [(0, 1), (1, 1), (2, 1), (3, 0)] and bounds(0, 3) == [(0, 0)] * 3
    for n in range(0, 40):
        for w in range(1, 10):
            b = bounds(n, w)
            assert sum(c for _, c in b) == n and max(c for _, c in b) - min(c for _, c in b) <= 1
            assert L.achip_shard_slots(n, w) == max(c for _, c in b)
            for r, (first, count) in enumerate(b):
                for i in range(first, first + count):
                    assert L.achip_shard_owner(n, w, i) == r
            assert L.achip_shard_owner(n, w, n) == -1 and L.achip_shard_owner(n, w, -1) == -1


def test_resize_kernel_emulated():
    img = orc.frame_hash_noise(193, 109, 8)
    for (dw, dh) in ((80, 24), (1, 1), (193, 109), (400, 300)):
        out = np.zeros((dh, dw, 3), np.uint8)
        emu.lib().emu_resize_nn(img.ctypes.data, 193, 109, out.ctypes.data, dw, dh,
                                emu.lib().achip_nn_ratio(193, dw), emu.lib().achip_nn_ratio(109, dh))
        assert np.array_equal(out, orc.resize_nn(img, dw, dh))
    # several resizes in one launch (what the grid path issues per tick): blockIdx.y selects the image
    import ctypes as C

    class Item(C.Structure):
        _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("sw", C.c_int32), ("sh", C.c_int32), ("dw", C.c_int32),
                    ("dh", C.c_int32), ("xr", C.c_uint32), ("yr", C.c_uint32)]

    class Batch(C.Structure):
        _fields_ = [("item", Item * 16), ("n", C.c_int32), ("_pad", C.c_int32)]

    imgs = [orc.frame_hash_noise(50 + 17 * k, 40 + 9 * k, k) for k in range(5)]
    dims = [(53, 30), (1, 1), (7, 90), (64, 2), (33, 33)]
    outs = [np.zeros((h, w, 3), np.uint8) for (w, h) in dims]
    b = Batch()
    b.n = len(imgs)
    for k, (im, (w, h)) in enumerate(zip(imgs, dims)):
        b.item[k] = Item(im.ctypes.data, outs[k].ctypes.data, im.shape[1], im.shape[0], w, h,
                         emu.lib().achip_nn_ratio(im.shape[1], w), emu.lib().achip_nn_ratio(im.shape[0], h))
    emu.lib().emu_resize_batch(C.byref(b))
    for k, (im, (w, h)) in enumerate(zip(imgs, dims)):
        assert np.array_equal(outs[k], orc.resize_nn(im, w, h)), k


def test_descriptors_the_32_bit_source_offsets_cannot_address_are_refused():
    """The kernels address a source with 32-bit byte offsets and multiply the row stride as a 24-bit value: a descriptor
    whose explicit stride puts the last pixel 4 GiB or more behind the first is refused on the host."""
    import ctypes as C
    L = emu.lib()
    L.achip_frame_extent_ok.restype = C.c_bool
    L.achip_frame_extent_ok.argtypes = [C.POINTER(emu.Frame)]
    img = orc.frame_hash_noise(64, 48, 1)
    f = emu.frame_for_convert(img, 20, 10, 0)
    assert L.achip_frame_extent_ok(C.byref(f))
    f.src_stride = 3 * 64 + 5  # a padded row: fine
    assert L.achip_frame_extent_ok(C.byref(f))
    f.src_w, f.src_h, f.src_stride = 10000, 10000, 0  # the largest image ascii_convert accepts: 300 MB
    assert L.achip_frame_extent_ok(C.byref(f))
    f.src_stride = 1 << 24  # 16 MiB rows
    assert not L.achip_frame_extent_ok(C.byref(f))
    f.src_stride = 500000   # 10 000 rows x 500 KB = 5 GB
    assert not L.achip_frame_extent_ok(C.byref(f))
    f.src_stride = -192
    assert not L.achip_frame_extent_ok(C.byref(f))


def test_checksum_pass_picks_its_kernel_by_cost(pkg):
    """achip_crc_parts (hip_launch.hip): one workgroup per buffer (4.5 + 58 us per MB of the longest buffer) against 64 KB
    spans + the finish kernel (17 us + 0.25 us per MB of ALL buffers) -- the rule round 4's send-side audit put in place of
    "every buffer above 128 KB goes to the spans" (profiles/r04_wire_audit.txt).  Host arithmetic: no GPU needed."""
    if os.environ.get("ASCIICHAT_HIP_CRC_FRAME_MAX") or os.environ.get("ASCIICHAT_HIP_CRC_SMALL_SPANS"):
        pytest.skip("the diagnostic override is set")
    L = pkg.lib()
    L.achip_crc_parts.restype = C.c_int
    L.achip_crc_parts.argtypes = [C.c_uint32, C.c_int]
    spans = lambda n: (n + 65535) // 65536  # noqa: E731
    for n in (1, 16, 64, 256, 1000):
        assert L.achip_crc_parts(36 * 1024, n) == 1 and L.achip_crc_parts(128 * 1024, n) == 1  # as ever: small frames
    assert L.achip_crc_parts(277000, 16) == (277000 + 16383) // 16384  # sixteen 200x60 truecolor frames (stride 277 KB): 15.5 us of (16 KB) spans against 20.2
    assert L.achip_crc_parts(277000, 128) == 1           # ... 128 of them: 23 against 27
    assert L.achip_crc_parts(166400, 64) == 1            # sixty-four 160x45 frames
    assert L.achip_crc_parts(540000, 256) == 1           # a workgroup per CU: 44 us against 56
    assert L.achip_crc_parts(1845408, 256) == 1          # the configs[4] shape: 136 against 138 (a tie)
    assert L.achip_crc_parts(663936, 1) == (663936 + 16383) // 16384  # a lone 320x90 truecolor frame: spans (against 42 us), of 16 KB while
    assert L.achip_crc_parts(1 << 20, 3) == 64                        # the call has at most 512 of the 64 KB ones (a GPU that is not full)
    assert L.achip_crc_parts(1 << 20, 64) == spans(1 << 20)  # 36 against 72
    assert L.achip_crc_parts(6220808, 64) == spans(6220808)  # 1080p ingest payloads
