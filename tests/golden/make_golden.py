#!/usr/bin/env python3
"""Regenerates tests/golden/*.json.

Two files, both DATA only:

reference_anchors.json  -- values of the REFERENCE's own output (zfogg/ascii-chat @ 2026-07-23) as recorded by the
    survey stage from the reference's unmodified sources, plus the
    known-answer values the reference's unit tests hold.  These are transcribed, not computed: this script only
    rewrites the file from the table below and checks that the oracle still reproduces every entry.

oracle_vectors.json     -- (length, FNV-1a-32, CRC-32C) of the CPU oracle's output for a matrix of procedural inputs
    (tests/orc.py generators, no image files) x modes x sizes x palettes.  They freeze the oracle: a change of the
    oracle's bytes shows up as a diff of this file, and the GPU / emulated kernels are checked against the same
    numbers without running the oracle (tests/test_golden.py).

Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402

REFERENCE_ANCHORS = {
    "_provenance": "SURVEY.md 8(c) / Appendix B: produced by the survey from the reference's unmodified C sources; "
                   "input 'anchor_gradient' = px(x,y)=(x*255/639, y*255/479, (x+y)&255) 640x480, 'torture' = the 333x201 "
                   "procedural image of Appendix B (tests/orc.py: frame_anchor_gradient / frame_torture)",
    "whole_frame": [
        # input, call, width, height, color_level, render_mode, wants_padding, use_aspect, stretch, palette, length, fnv1a32
        ["anchor_gradient", "ascii_convert", 80, 24, 0, 0, False, False, False, "STANDARD", 1635, "cefd0a18"],
        ["anchor_gradient", "ascii_convert_with_capabilities", 80, 24, 0, 0, True, True, False, "STANDARD", 1721, "7d62f78f"],
        ["anchor_gradient", "ascii_convert_with_capabilities", 80, 24, 2, 0, False, False, False, "STANDARD", 22255, "be60a438"],
        ["anchor_gradient", "ascii_convert_with_capabilities", 80, 24, 3, 0, False, False, False, "STANDARD", 35852, "885da51d"],
        ["anchor_gradient", "ascii_convert_with_capabilities", 80, 24, 3, 2, False, False, False, "STANDARD", 73802, "362719ad"],
    ],
    "lengths": [
        # input, width, height, color_level, render_mode, palette, length   (Appendix B)
        ["torture", 80, 24, 0, 0, "STANDARD", 1159], ["torture", 80, 24, 1, 0, "STANDARD", 11639],
        ["torture", 80, 24, 2, 0, "STANDARD", 22429], ["torture", 80, 24, 3, 0, "STANDARD", 21664],
        ["torture", 80, 24, 3, 2, "STANDARD", 43121], ["torture", 80, 24, 2, 2, "STANDARD", 15725],
        ["torture", 80, 24, 1, 2, "STANDARD", 7697], ["torture", 80, 24, 0, 2, "STANDARD", 3921],
        ["torture", 97, 31, 0, 0, "BLOCKS", 1274], ["torture", 97, 31, 3, 0, "BLOCKS", 48984],
        ["torture", 97, 31, 2, 0, "BLOCKS", 39632], ["torture", 97, 31, 1, 0, "BLOCKS", 18196],
        ["torture", 97, 31, 0, 0, "COOL", 1274], ["torture", 97, 31, 3, 0, "COOL", 48984],
        ["torture", 97, 31, 2, 0, "COOL", 39632], ["torture", 97, 31, 1, 0, "COOL", 18196],
        ["torture", 97, 31, 3, 1, "STANDARD", 34602],
    ],
    "aspect_ratio": [[1920, 1080, 80, 24, 80, 23], [3840, 2160, 200, 60, 200, 56], [3840, 2160, 400, 120, 400, 113],
                     [640, 480, 80, 24, 64, 24], [160, 96, 160, 48, 160, 48]],
    "crc32c": [["", "00000000"], ["Hello, World!", "4d551068"]],  # tests/unit/network/crc32_hw_test.c:14-50
}

PALETTES = {"STANDARD": orc.PALETTE_STANDARD, "BLOCKS": orc.PALETTE_BLOCKS, "COOL": orc.PALETTE_COOL,
            "DIGITAL": orc.PALETTE_DIGITAL, "MINIMAL": orc.PALETTE_MINIMAL}
INPUTS = {
    "anchor_gradient": lambda: orc.frame_anchor_gradient(),
    "torture": lambda: orc.frame_torture(),
    "noise_320x240_s7": lambda: orc.frame_noise(320, 240, 7),
    "smooth_640x360": lambda: orc.frame_smooth(640, 360),
    "bars_400x300_f3": lambda: orc.frame_bars(400, 300, 3),
    "hash_noise_1920x1080_s11": lambda: orc.frame_hash_noise(1920, 1080, 11),
}
MODES = [(0, 0), (1, 0), (2, 0), (3, 0), (3, 1), (3, 2), (2, 2), (1, 2), (0, 2)]  # (color_level, render_mode)


def vector_matrix():
    """(input, width, height, color_level, render_mode, wants_padding, use_aspect, palette)"""
    m = []
    for inp in INPUTS:
        for (cl, rm) in MODES:
            m.append((inp, 80, 24, cl, rm, False, False, "STANDARD"))
    for (cl, rm) in MODES:
        m.append(("torture", 97, 31, cl, rm, True, rm != 1, "STANDARD"))
        m.append(("torture", 200, 60, cl, rm, False, False, "STANDARD"))
        m.append(("noise_320x240_s7", 33, 7, cl, rm, True, True, "STANDARD"))
    for pal in ("BLOCKS", "COOL", "DIGITAL", "MINIMAL"):
        for cl in (0, 1, 2, 3):
            m.append(("torture", 97, 31, cl, 0, False, False, pal))
    m.append(("hash_noise_1920x1080_s11", 400, 120, 3, 2, False, False, "STANDARD"))
    m.append(("hash_noise_1920x1080_s11", 1, 1, 3, 0, False, False, "STANDARD"))
    return m


def ops_matrix():
    """Cases that go through achip_frame_t.ops: (input, width, height, kind, arg, palette) with kind =
    'dither_bg' | 'dither_fg' | 'dither_fg_ramp' (the three exported forms of the Floyd-Steinberg renderer, on the
    nearest-neighbour resize of the input) or 'rainbow_<color_level>_<render_mode>' (rainbow_replace_ansi_colors at
    time `arg` over the plain frame)."""
    m = []
    for pal in ("STANDARD", "BLOCKS"):
        for kind in ("dither_bg", "dither_fg", "dither_fg_ramp"):
            m.append(("torture", 61, 23, kind, 0.0, pal))
    m.append(("noise_320x240_s7", 10, 150, "dither_fg", 0.0, "STANDARD"))
    for kind in ("rainbow_3_0", "rainbow_3_2", "rainbow_2_0"):
        for t in (0.4, 2.05):
            m.append(("anchor_gradient", 80, 24, kind, t, "STANDARD"))
    m.append(("torture", 97, 31, "rainbow_3_0", 1234.5, "COOL"))
    return m


def render_ops(entry, cache={}):
    inp, w, h, kind, arg, pal = entry
    if inp not in cache:
        cache[inp] = INPUTS[inp]()
    img = cache[inp]
    if kind.startswith("dither"):
        small = orc.resize_nn(img, w, h)
        return orc.print_16_dithered(small, kind == "dither_bg", PALETTES[pal], ramp_glyph=kind == "dither_fg_ramp")
    _, cl, rm = kind.split("_")
    return orc.rainbow_replace(orc.convert_with_caps(img, w, h, int(cl), int(rm), False, False, False, PALETTES[pal]), arg)


def render(entry, cache={}):
    inp, w, h, cl, rm, pad, aspect, pal = entry
    if inp not in cache:
        cache[inp] = INPUTS[inp]()
    return orc.convert_with_caps(cache[inp], w, h, cl, rm, pad, aspect, False, PALETTES[pal])


def main():
    # 1. the transcribed reference values: verify, then write
    g = {"anchor_gradient": orc.frame_anchor_gradient(), "torture": orc.frame_torture()}
    for inp, call, w, h, cl, rm, pad, aspect, stretch, pal, length, fnv in REFERENCE_ANCHORS["whole_frame"]:
        out = (orc.convert(g[inp], w, h, False, aspect, stretch, PALETTES[pal]) if call == "ascii_convert" else
               orc.convert_with_caps(g[inp], w, h, cl, rm, pad, aspect, stretch, PALETTES[pal]))
        assert (len(out), "%08x" % orc.fnv1a32(out)) == (length, fnv), (inp, call, cl, rm)
    for inp, w, h, cl, rm, pal, length in REFERENCE_ANCHORS["lengths"]:
        assert len(orc.convert_with_caps(g[inp], w, h, cl, rm, False, False, False, PALETTES[pal])) == length
    for iw, ih, w, h, ow, oh in REFERENCE_ANCHORS["aspect_ratio"]:
        assert orc.aspect_ratio(iw, ih, w, h) == (ow, oh)
    for text, crc in REFERENCE_ANCHORS["crc32c"]:
        assert "%08x" % orc.crc32c(text.encode()) == crc
    with open(os.path.join(HERE, "reference_anchors.json"), "w") as f:
        json.dump(REFERENCE_ANCHORS, f, indent=1)
    # 2. the oracle's own vectors
    vecs = []
    for e in vector_matrix():
        out = render(e)
        vecs.append(list(e) + [len(out), "%08x" % orc.fnv1a32(out), "%08x" % orc.crc32c(out)])
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as f:
        f.write('{"_columns": ["input", "width", "height", "color_level", "render_mode", "wants_padding", "use_aspect", '
                '"palette", "length", "fnv1a32", "crc32c"],\n')
        f.write(' "_generator": "tests/golden/make_golden.py (oracle/asciichat_oracle.c through tests/orc.py)",\n')
        f.write(' "vectors": [\n' + ",\n".join("  " + json.dumps(v) for v in vecs) + "\n ],\n")
        ops = []
        for e in ops_matrix():
            out = render_ops(e)
            ops.append(list(e) + [len(out), "%08x" % orc.fnv1a32(out)])
        f.write(' "_ops_columns": ["input", "width", "height", "kind", "arg", "palette", "length", "fnv1a32"],\n')
        f.write(' "ops_vectors": [\n' + ",\n".join("  " + json.dumps(v) for v in ops) + "\n ]}\n")
    print(f"wrote {len(vecs)} oracle vectors and {len(REFERENCE_ANCHORS['whole_frame']) + len(REFERENCE_ANCHORS['lengths'])} "
          "reference anchors")


if __name__ == "__main__":
    main()
