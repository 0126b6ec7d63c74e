"""The display path's full-frame passes as stand-alone kernels: colour filter and flips."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import emu  # noqa: E402
import orc  # noqa: E402


def ops_for(flip_x, flip_y, flt, w=10, h=10):
    f = emu.Frame()
    f.src_w, f.src_h = w, h
    assert emu.lib().achip_frame_set_display_ops(C.byref(f), flip_x, flip_y, flt) == 0
    return f.ops


def test_tint_and_flip_kernels_emulated():
    for (w, h) in ((64, 9), (33, 7), (16, 1), (1, 5), (160, 3)):
        img = orc.frame_hash_noise(w, h, w * 31 + h)
        for flt in range(1, 12):
            for vec in (0, 1):
                buf = img.copy()
                emu.lib().emu_tint(buf.ctypes.data, w, h, 3 * w, ops_for(False, False, flt), vec)
                assert np.array_equal(buf, orc.color_filter(img, flt)), (w, h, flt, vec)
        for fx, fy in ((True, False), (False, True), (True, True)):
            for vec in ((0, 1) if w % 16 == 0 else (0,)):
                dst = np.zeros_like(img)
                emu.lib().emu_flip(img.ctypes.data, dst.ctypes.data, w, h, ops_for(fx, fy, 0, w, h), vec)
                assert np.array_equal(dst, orc.flip(img, fx, fy)), (w, h, fx, fy, vec)


@pytest.mark.gpu
def test_tint_and_flip_kernels_gpu():
    import torch

    from __graft_entry__ import load_package

    pkg = load_package()
    L = pkg.lib()
    torch.cuda.set_device(0)
    for (w, h) in ((1920, 1080), (333, 201), (640, 480), (17, 3)):
        img = orc.frame_hash_noise(w, h, 5)
        for flt in (1, 3, 9, 11):
            dev = torch.from_numpy(img.copy()).cuda()
            assert L.asciichat_hip_apply_color_filter(dev.data_ptr(), w, h, 3 * w, flt, None) == 0
            torch.cuda.synchronize()
            assert np.array_equal(dev.cpu().numpy(), orc.color_filter(img, flt)), (w, h, flt)
        assert L.asciichat_hip_apply_color_filter(dev.data_ptr(), w, h, 3 * w, 12, None) != 0  # rainbow: rejected
        src = torch.from_numpy(img).cuda()
        for fx, fy in ((1, 0), (0, 1), (1, 1)):
            dst = torch.zeros_like(src)
            assert L.asciichat_hip_image_flip(src.data_ptr(), dst.data_ptr(), w, h, fx, fy, None) == 0
            torch.cuda.synchronize()
            assert np.array_equal(dst.cpu().numpy(), orc.flip(img, bool(fx), bool(fy))), (w, h, fx, fy)
    # folded into the render sampler: same bytes as transforming the image first (display.c:546-632)
    img = orc.frame_hash_noise(1280, 720, 9)
    dev = torch.from_numpy(img).cuda()
    for mode, (cl, rm) in ((1, (3, 0)), (5, (3, 2))):
        f = pkg.frame_setup(dev.data_ptr(), 1280, 720, 120, 40, rm, True, True, False)
        assert L.achip_frame_set_display_ops(C.byref(f), True, True, 7) == 0
        plan = pkg.Plan(mode, orc.PALETTE_STANDARD, [f])
        out = torch.zeros(plan.stride, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(1, dtype=torch.int32, device="cuda")
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr())
        torch.cuda.synchronize()
        got = out[:int(ln[0].item())].cpu().numpy().tobytes()
        assert got == orc.display_convert(img, 120, 40, cl, rm, True, True, True, True, 7)
        plan.close()
    # COLOR_FILTER_RAINBOW folded into the emission (display.c:639-650): same bytes as rainbow_replace_ansi_colors over
    # the finished frame; 256-colour frames hold nothing to recolour
    for mode, (cl, rm) in ((1, (3, 0)), (5, (3, 2)), (2, (2, 0))):
        for t in (0.25, 1.9, 3.3):
            f = pkg.frame_setup(dev.data_ptr(), 1280, 720, 120, 40, rm, True, True, False)
            assert L.achip_frame_set_display_ops(C.byref(f), False, True, 0) == 0
            assert L.achip_frame_set_rainbow(C.byref(f), t) == 0
            plan = pkg.Plan(mode, orc.PALETTE_STANDARD, [f])
            out = torch.zeros(plan.stride, dtype=torch.uint8, device="cuda")
            ln = torch.zeros(1, dtype=torch.int32, device="cuda")
            plan.render(out.data_ptr(), plan.stride, ln.data_ptr())
            torch.cuda.synchronize()
            got = out[:int(ln[0].item())].cpu().numpy().tobytes()
            exp = orc.rainbow_replace(orc.display_convert(img, 120, 40, cl, rm, True, True, False, True, 0), t)
            assert got == exp, (mode, t)
            plan.close()
