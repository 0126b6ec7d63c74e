#!/usr/bin/env python3
"""Randomised calls of the drop-in entry points on the GPU box: odd sizes, zero / negative / huge terminal sizes,
malformed palettes, every capability combination.  Valid calls must equal the oracle byte for byte, invalid ones
must return NULL exactly where the oracle (= the reference's argument checks) does; nothing may crash."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from test_random_differential import random_image  # noqa: E402


def main():
    import torch  # noqa: F401

    from __graft_entry__ import load_package

    pkg = load_package()
    L = pkg.lib()
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    pals = [orc.PALETTE_STANDARD.encode(), orc.PALETTE_BLOCKS.encode(), orc.PALETTE_COOL.encode(), b"@", b" .:-=+*#%@",
            "a█b".encode(), b"\xff\xfe", b"\xe2\x96", b"x" * 200]
    n_ok = n_null = 0
    for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3000):
        sw, sh = int(rng.choice([1, 2, 3, 7, 64, 333, 640, 1920])), int(rng.choice([1, 2, 5, 48, 201, 480, 1080]))
        img = random_image(rng, sw, sh)
        arr = np.ascontiguousarray(img)
        im = pkg.Image(sw, sh, arr.ctypes.data, 0)
        W = int(rng.choice([-5, 0, 1, 2, 17, 80, 97, 200, 511, 3000, 10001]))
        H = int(rng.choice([-1, 0, 1, 3, 24, 31, 60, 140, 2000, 10001]))
        cl, rm = int(rng.choice([-1, 0, 1, 2, 3])), int(rng.choice([0, 1, 2]))
        pad, aspect, stretch = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        pal = pals[int(rng.integers(0, len(pals)))]
        caps = pkg.TermCaps()
        caps.color_level, caps.render_mode, caps.wants_padding, caps.utf8_support = cl, rm, pad, True
        palb = pal
        p = L.ascii_convert_with_capabilities(C.byref(im), W, H, C.byref(caps), aspect, stretch, palb)
        got = pkg.take_string(p)
        exp = orc.convert_with_caps(img, W, H, cl, rm, pad, aspect, stretch, palb)
        assert got == exp, (it, sw, sh, W, H, cl, rm, pad, aspect, stretch, palb[:12], None if got is None else len(got), None if exp is None else len(exp))
        if got is None:
            n_null += 1
        else:
            n_ok += 1
    print(f"drop-in fuzz OK: {n_ok} renders byte-identical, {n_null} NULL returns matching the oracle")
    # the other entry points: ascii_convert (mode from the global option), image_print_* on images as they are,
    # half-block functions with explicit strides
    lum = C.create_string_buffer(b"x" * 255, 256)
    n2 = 0
    for it in range((int(sys.argv[2]) if len(sys.argv) > 2 else 3000) // 2):
        sw, sh = int(rng.choice([1, 2, 3, 7, 61, 200, 333, 640])), int(rng.choice([1, 2, 5, 23, 48, 201, 480]))
        img = random_image(rng, sw, sh)
        arr = np.ascontiguousarray(img)
        im = pkg.Image(sw, sh, arr.ctypes.data, 0)
        pal = pals[int(rng.integers(0, 6))]
        which = int(rng.integers(0, 5))
        if which == 0:
            W, H = int(rng.choice([0, 1, 17, 80, 200, 4000])), int(rng.choice([0, 1, 24, 60, 3000]))
            color, opt = bool(rng.integers(0, 2)), int(rng.integers(0, 3))
            aspect, stretch = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            L.asciichat_hip_set_option_render_mode(opt)
            got = pkg.take_string(L.ascii_convert(C.byref(im), W, H, color, aspect, stretch, pal, lum))
            exp = orc.convert(img, W, H, color, aspect, stretch, pal, opt)
            L.asciichat_hip_set_option_render_mode(0)
            ctx = ("ascii_convert", sw, sh, W, H, color, opt, aspect, stretch)
        elif which == 1:
            cl, rm = int(rng.choice([-1, 0, 1, 2, 3])), int(rng.choice([0, 1, 2]))
            caps = pkg.TermCaps()
            caps.color_level, caps.render_mode, caps.utf8_support = cl, rm, True
            got = pkg.take_string(L.image_print_with_capabilities(C.byref(im), C.byref(caps), pal))
            exp = orc.print_with_caps(img, cl, rm, pal)
            ctx = ("image_print_with_capabilities", sw, sh, cl, rm)
        elif which == 2:
            fn, (cl, rm) = [("image_print", (0, 0)), ("image_print_color", (3, 0)), ("image_print_256color", (2, 0)),
                            ("image_print_16color", (1, 0))][int(rng.integers(0, 4))]
            got = pkg.take_string(getattr(L, fn)(C.byref(im), pal))
            exp = orc.print_with_caps(img, cl, rm, pal)
            ctx = (fn, sw, sh)
        elif which == 4:  # the three exported forms of the Floyd-Steinberg renderer
            style = int(rng.integers(0, 3))
            if style == 2:
                got = pkg.take_string(L.image_print_16color_dithered(C.byref(im), pal))
                exp = orc.print_16_dithered(img, False, pal, ramp_glyph=True)
            else:
                got = pkg.take_string(L.image_print_16color_dithered_with_background(C.byref(im), style == 0, pal))
                exp = orc.print_16_dithered(img, style == 0, pal)
            ctx = ("dithered", style, sw, sh)
        else:
            fn, cl = [("rgb_to_truecolor_halfblocks_scalar", 3), ("rgb_to_256color_halfblocks_scalar", 2),
                      ("rgb_to_16color_halfblocks_scalar", 1), ("rgb_to_halfblocks_scalar", 0)][int(rng.integers(0, 4))]
            sub = int(rng.integers(1, sw + 1))  # the left `sub` columns through an explicit stride
            args = (im.pixels, sub, sh, sw * 3) if cl == 3 else (im.pixels, sub, sh, sw * 3, pal)  # palette unused by them
            got = pkg.take_string(getattr(L, fn)(*args))
            exp = orc.print_with_caps(np.ascontiguousarray(img[:, :sub]), cl, 2)
            ctx = (fn, sw, sh, sub)
        assert got == exp, (it, ctx, pal[:8], None if got is None else len(got), None if exp is None else len(exp))
        n2 += 1
    print(f"drop-in fuzz OK: {n2} calls of the other entry points match the oracle")


if __name__ == "__main__":
    main()
