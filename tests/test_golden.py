"""Committed golden fixtures (tests/golden/*.json, generator: tests/golden/make_golden.py).

reference_anchors.json holds values of the REFERENCE's output recorded by the survey;
oracle_vectors.json freezes the oracle's bytes as (length, FNV-1a-32, CRC-32C) for 99 procedural cases.
CPU: the oracle and the emulated kernels reproduce them.  GPU (-m gpu): the C-ABI reproduces them WITHOUT the
oracle in the loop, and the device CRC kernel reproduces the golden CRCs of the frames it renders."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "golden"))

import emu  # noqa: E402
import make_golden as mg  # noqa: E402
import orc  # noqa: E402
from achip_ctypes import MODE_TRUE_BG  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, "golden", "oracle_vectors.json")))["vectors"]
GOLD_OPS = json.load(open(os.path.join(HERE, "golden", "oracle_vectors.json")))["ops_vectors"]
REF = json.load(open(os.path.join(HERE, "golden", "reference_anchors.json")))
_inputs = {}


def image(name):
    if name not in _inputs:
        _inputs[name] = mg.INPUTS[name]()
    return _inputs[name]


def test_fixture_files_match_the_generator_tables():
    assert REF["whole_frame"] == [list(x) for x in mg.REFERENCE_ANCHORS["whole_frame"]]
    assert [tuple(v[:8]) for v in GOLD] == [tuple(e) for e in mg.vector_matrix()]
    assert [tuple(v[:6]) for v in GOLD_OPS] == [tuple(e) for e in mg.ops_matrix()]


def test_oracle_reproduces_reference_anchors():
    for inp, call, w, h, cl, rm, pad, aspect, stretch, pal, length, fnv in REF["whole_frame"]:
        out = (orc.convert(image(inp), w, h, False, aspect, stretch, mg.PALETTES[pal]) if call == "ascii_convert" else
               orc.convert_with_caps(image(inp), w, h, cl, rm, pad, aspect, stretch, mg.PALETTES[pal]))
        assert (len(out), "%08x" % orc.fnv1a32(out)) == (length, fnv), (inp, call, cl, rm)
    for inp, w, h, cl, rm, pal, length in REF["lengths"]:
        assert len(orc.convert_with_caps(image(inp), w, h, cl, rm, False, False, False, mg.PALETTES[pal])) == length
    for iw, ih, w, h, ow, oh in REF["aspect_ratio"]:
        assert orc.aspect_ratio(iw, ih, w, h) == (ow, oh)
    for text, crc in REF["crc32c"]:
        assert "%08x" % orc.crc32c(text.encode()) == crc


def test_oracle_reproduces_its_golden_vectors():
    for v in GOLD:
        out = mg.render(tuple(v[:8]))
        assert [len(out), "%08x" % orc.fnv1a32(out), "%08x" % orc.crc32c(out)] == v[8:], v[:8]
    for v in GOLD_OPS:
        out = mg.render_ops(tuple(v[:6]))
        assert [len(out), "%08x" % orc.fnv1a32(out)] == v[6:], v[:6]


def _mode(cl, rm):
    return emu.lib().achip_mode_from_caps(cl, rm)


def _ops_frame(L, setup, src_ptr, img, v):
    """Descriptor + mode of an ops vector: the dithered forms render the nearest-neighbour resize of the input in mode 9
    with a style, the rainbow cases render the plain frame with the foreground override of the moment."""
    inp, w, h, kind, arg, pal = v[:6]
    if kind.startswith("dither"):
        f = setup(src_ptr, img.shape[1], img.shape[0], w, h, 0, False, False, False)
        assert L.achip_frame_set_dither_style(C.byref(f), kind == "dither_bg", kind == "dither_fg_ramp") == 0
        return f, 9
    _, cl, rm = kind.split("_")
    f = setup(src_ptr, img.shape[1], img.shape[0], w, h, int(rm), False, False, False)
    assert L.achip_frame_set_rainbow(C.byref(f), arg) == 0
    return f, L.achip_mode_from_caps(int(cl), int(rm))


def test_emulated_kernels_reproduce_golden_vectors():
    """The HIP kernel source under the CPU emulator against the committed hashes (cases up to 200x60 cells)."""
    for v in GOLD:
        inp, w, h, cl, rm, pad, aspect, pal, length, fnv, crc = v
        if w * h > 12000:
            continue
        f = emu.frame_for_convert(image(inp), w, h, rm, pad, aspect)
        got = emu.render_frames(_mode(cl, rm), [f], mg.PALETTES[pal], 0 if w > 1000 else 2)[0]
        assert [len(got), "%08x" % orc.fnv1a32(got)] == [length, fnv], v[:8]


def test_emulated_kernels_reproduce_ops_vectors():
    L = emu.lib()

    def setup(ptr, sw, sh, w, h, rm, pad, aspect, stretch):
        f = emu.Frame()
        assert L.achip_frame_setup(C.byref(f), ptr, sw, sh, w, h, rm, pad, aspect, stretch) == 0
        return f

    for v in GOLD_OPS:
        img = image(v[0])
        f, mode = _ops_frame(L, setup, img.ctypes.data, img, v)
        got = emu.render_frames(mode, [f], mg.PALETTES[v[5]], 2, uniform=True)[0]
        assert [len(got), "%08x" % orc.fnv1a32(got)] == v[6:], v[:6]


@pytest.mark.gpu
def test_gpu_reproduces_golden_vectors_and_crcs():
    import torch

    from __graft_entry__ import load_package

    pkg = load_package()
    assert torch.cuda.is_available()
    L = pkg.lib()
    stream = torch.cuda.current_stream().cuda_stream
    dev = {}
    by_mode = {}
    for v in GOLD:
        by_mode.setdefault((v[3], v[4], v[7]), []).append(v)
    for (cl, rm, pal), vs in by_mode.items():
        frames = []
        for v in vs:
            inp, w, h = v[0], v[1], v[2]
            if inp not in dev:
                dev[inp] = torch.from_numpy(np.ascontiguousarray(image(inp))).cuda()
            img = image(inp)
            f = pkg.frame_setup(dev[inp].data_ptr(), img.shape[1], img.shape[0], w, h, rm, v[5], v[6], False)
            assert f is not None
            frames.append(f)
        plan = pkg.Plan(L.achip_mode_from_caps(cl, rm), mg.PALETTES[pal], frames)
        n = len(frames)
        out = torch.zeros(n * plan.stride, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(n, dtype=torch.int32, device="cuda")
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), stream)
        crc = torch.zeros(n, dtype=torch.int32, device="cuda")
        assert L.asciichat_hip_crc32c(out.data_ptr(), plan.stride, ln.data_ptr(), 0, plan.stride, n, crc.data_ptr(), stream) == 0
        torch.cuda.synchronize()
        host, lens = out.cpu().numpy(), ln.cpu().numpy().astype(np.uint32)
        crcs = crc.cpu().numpy().astype(np.uint32)
        for k, v in enumerate(vs):
            got = host[k * plan.stride:k * plan.stride + int(lens[k])].tobytes()
            assert [len(got), "%08x" % orc.fnv1a32(got), "%08x" % int(crcs[k])] == v[8:], v[:8]
        plan.close()
    # the cases that go through achip_frame_t.ops (dithered styles, rainbow override), one plan each
    for v in GOLD_OPS:
        img = image(v[0])
        if v[0] not in dev:
            dev[v[0]] = torch.from_numpy(np.ascontiguousarray(img)).cuda()
        f, mode = _ops_frame(L, pkg.frame_setup, dev[v[0]].data_ptr(), img, v)
        plan = pkg.Plan(mode, mg.PALETTES[v[5]], [f])
        out = torch.zeros(plan.stride, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(1, dtype=torch.int32, device="cuda")
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), stream)
        torch.cuda.synchronize()
        got = out[:int(ln[0].item())].cpu().numpy().tobytes()
        assert [len(got), "%08x" % orc.fnv1a32(got)] == v[6:], v[:6]
        plan.close()
