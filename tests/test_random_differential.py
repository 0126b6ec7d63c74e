"""Randomised differential tests: random frame sizes, grid sizes, paddings, modes, palettes and pixel
statistics, HIP kernel source (under the CPU fiber emulator; on the GPU in the -m gpu variant) vs the oracle."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import emu  # noqa: E402
import orc  # noqa: E402
from achip_ctypes import ALL_MODES, MODE_CAPS, MODE_NAMES, MODE_TRUE_BG  # noqa: E402

PALETTES = [orc.PALETTE_STANDARD, orc.PALETTE_BLOCKS, orc.PALETTE_MINIMAL, orc.PALETTE_COOL, "@", " .:-=+*#%@", "a█b"]


def random_image(rng, w, h):
    kind = rng.integers(0, 6)
    if kind == 0:
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if kind == 1:  # few colours, long runs
        pal = rng.integers(0, 256, (4, 3), dtype=np.uint8)
        pal[0] = 0
        idx = np.repeat(rng.integers(0, 4, (h, (w + 6) // 7)), 7, axis=1)[:, :w]
        return pal[idx]
    if kind == 2:  # dark noise around the transparency / grey thresholds
        return rng.integers(0, 3, (h, w, 3), dtype=np.uint8)
    if kind == 3:
        return orc.frame_smooth(w, h)
    if kind == 4:
        return orc.frame_bars(w, h, int(rng.integers(0, 9)))
    g = rng.integers(0, 256, (h, w, 1), dtype=np.uint8)
    return np.repeat(g, 3, axis=2) ^ rng.integers(0, 2, (h, w, 3), dtype=np.uint8)  # near-grey


def random_case(rng):
    sw, sh = int(rng.integers(1, 200)), int(rng.integers(1, 120))
    W, H = int(rng.integers(1, 140)), int(rng.integers(1, 50))
    mode = int(rng.choice(ALL_MODES))
    aspect = bool(rng.integers(0, 2)) and mode != MODE_TRUE_BG
    pad = bool(rng.integers(0, 2))
    return sw, sh, W, H, mode, aspect, pad, PALETTES[int(rng.integers(0, len(PALETTES)))]


def oracle_case(img, W, H, mode, aspect, pad, palette):
    if mode == MODE_TRUE_BG:
        return orc.print_truecolor_bg(orc.resize_nn(img, W, H), palette)
    cl, rm = MODE_CAPS[mode]
    return orc.convert_with_caps(img, W, H, cl, rm, pad, aspect, False, palette)


@pytest.mark.parametrize("seed", range(6))
def test_random_cases_emulated(seed):
    rng = np.random.default_rng(1000 + seed)
    for _ in range(25):
        sw, sh, W, H, mode, aspect, pad, palette = random_case(rng)
        img = random_image(rng, sw, sh)
        rm = MODE_CAPS.get(mode, (3, 0))[1]
        f = emu.frame_for_convert(img, W, H, rm, pad, aspect)
        exp = oracle_case(img, W, H, mode, aspect, pad, palette)
        if f is None:
            assert exp is None
            continue
        variant = int(rng.choice([3, 3, 2, 4]))
        # (round 4: the slab at a random 16-byte phase of a 128-byte line -- the drains follow the address -- and nothing
        # written outside the frame)
        got = emu.render_frames(mode, [f], palette, variant, line_phase=int(rng.integers(0, 8)))[0]
        assert got == exp, (seed, sw, sh, W, H, MODE_NAMES[mode], aspect, pad, palette, variant)


@pytest.mark.parametrize("seed", range(4))
def test_random_display_ops_emulated(seed):
    """The descriptor's ops field under random combinations: flips, the eleven tints, the rainbow override, the three
    dithered styles -- descriptor by value or from the array, whole frames or row bands, all geometries incl. the
    512-thread one."""
    from achip_ctypes import MODE_16_DITHER_BG, MODE_TRUE_FG
    rng = np.random.default_rng(7000 + seed)
    L = emu.lib()
    for it in range(22):
        sw, sh, W, H, mode, aspect, pad, palette = random_case(rng)
        if mode == MODE_TRUE_BG:
            continue
        img = random_image(rng, sw, sh)
        cl, rm = MODE_CAPS[mode]
        fx, fy = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        kind = int(rng.integers(0, 3))  # 0: tint, 1: rainbow, 2: dithered style (mode 9 only) / plain flips
        style = None
        if mode == MODE_16_DITHER_BG:
            aspect = pad = False
            style = [(True, False), (False, False), (False, True)][int(rng.integers(0, 3))]
            kind = 2
        f = emu.frame_for_convert(img, W, H, rm, pad, aspect)
        if f is None:
            continue
        flt = int(rng.integers(1, 12)) if kind == 0 else 0
        assert L.achip_frame_set_display_ops(C.byref(f), fx, fy, flt) == 0
        t = float(rng.integers(0, 4000)) / 100.0
        if kind == 1:
            assert L.achip_frame_set_rainbow(C.byref(f), t) == 0
        if style is not None:
            assert L.achip_frame_set_dither_style(C.byref(f), style[0], style[1]) == 0
            exp = orc.print_16_dithered(orc.resize_nn(orc.flip(img, fx, fy), W, H), style[0], palette, ramp_glyph=style[1])
        else:
            exp = orc.display_convert(img, W, H, cl, rm, pad, aspect, fx, fy, flt, palette)
            if kind == 1:
                exp = orc.rainbow_replace(exp, t)
        variant = int(rng.choice([3, 2, 4, 1]))
        rows = (f.out_h + 1) // 2 if rm == 2 else f.out_h
        wp = f.pad_left + f.out_w
        cap = {1: 2048, 2: 1024, 3: 256, 4: 2048}[variant]
        ascii_only = all(ord(c) < 128 for c in palette)
        bands = 0
        if (variant != 3 and mode != MODE_16_DITHER_BG and rows > 1 and wp <= cap and (mode != MODE_TRUE_FG or ascii_only)
                and rng.integers(0, 2)):
            bands = max(1, min(cap // wp, int(rng.integers(1, rows))))
        got = emu.render_frames(mode, [f], palette, variant, rows_per_part=bands, uniform=bool(rng.integers(0, 2)))[0]
        assert got == exp, (seed, it, sw, sh, W, H, MODE_NAMES[mode], aspect, pad, palette, variant, bands, fx, fy, flt, kind, style)


@pytest.mark.gpu
def test_random_cases_gpu():
    import torch

    from __graft_entry__ import load_package

    pkg = load_package()
    torch.cuda.set_device(0)
    rng = np.random.default_rng(4242)
    for _round in range(12):
        mode = int(rng.choice(ALL_MODES))
        palette = PALETTES[int(rng.integers(0, len(PALETTES)))]
        aspect = bool(rng.integers(0, 2)) and mode != MODE_TRUE_BG
        pad = bool(rng.integers(0, 2))
        rm = MODE_CAPS.get(mode, (3, 0))[1]
        cases, frames, keep = [], [], []
        for _ in range(40):
            sw, sh = int(rng.integers(1, 700)), int(rng.integers(1, 400))
            W, H = int(rng.integers(1, 500)), int(rng.integers(1, 130))
            img = random_image(rng, sw, sh)
            dev = torch.from_numpy(np.ascontiguousarray(img)).cuda()
            f = pkg.frame_setup(dev.data_ptr(), sw, sh, W, H, rm, pad, aspect, False)
            if f is None:
                continue
            keep.append(dev)
            frames.append(f)
            cases.append((img, W, H))
        plan = pkg.Plan(mode, palette, frames)
        out = torch.zeros(len(frames) * plan.stride, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(len(frames), dtype=torch.int32, device="cuda")
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        host = out.cpu().numpy()
        lens = ln.cpu().numpy().astype(np.uint32)
        for k, (img, W, H) in enumerate(cases):
            assert lens[k] < 0xFFFFFFF0
            got = host[k * plan.stride:k * plan.stride + int(lens[k])].tobytes()
            assert got == oracle_case(img, W, H, mode, aspect, pad, palette), (MODE_NAMES[mode], k, img.shape, W, H)
        plan.close()
