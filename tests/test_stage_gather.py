"""Host staging of the drop-in layer (achip_stage_extent / achip_stage_gather, achip_host.c): the compacted image plus
the rewritten descriptor must give the sampler exactly the pixels the original image would -- for every (x, y) of the
resized image: sx = min((x * x_ratio) >> 16, src_w - 1), flips applied afterwards (image.c:282-312;
render_kernels.hpp sample_frame_raw).  CPU only: no kernel runs here."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


@pytest.fixture(scope="module")
def pkg():
    p = load_package()
    p.build()
    p.lib()
    return p


def sample_all(img, f):
    """what the device sampler reads for every cell of the out_w x out_h image"""
    h, w, _ = img.shape
    assert (w, h) == (f.src_w, f.src_h)
    xs = np.minimum((np.arange(f.out_w, dtype=np.uint64) * f.x_ratio) >> 16, w - 1).astype(np.int64)
    ys = np.minimum((np.arange(f.out_h, dtype=np.uint64) * f.y_ratio) >> 16, h - 1).astype(np.int64)
    if f.ops & 1:
        xs = w - 1 - xs
    if f.ops & 2:
        ys = h - 1 - ys
    return img[ys][:, xs]


CASES = [  # src_w, src_h, term_w, term_h, render_mode, ops
    (1920, 1080, 80, 24, 0, 0), (1920, 1080, 80, 24, 2, 0), (1920, 1080, 80, 24, 0, 1), (1920, 1080, 80, 24, 0, 2),
    (1920, 1080, 80, 24, 0, 3), (333, 201, 97, 31, 0, 0), (333, 201, 200, 60, 0, 1), (640, 480, 400, 120, 2, 3),
    (100, 50, 80, 24, 0, 0), (40, 30, 80, 24, 0, 3), (3840, 2160, 400, 120, 2, 0), (161, 3, 80, 24, 0, 1),
    (2, 2, 1, 1, 0, 0), (9999, 7, 3840, 3, 0, 1),
]


@pytest.mark.parametrize("stretch", [False, True])
@pytest.mark.parametrize("case", CASES)
def test_gather_preserves_every_sample(pkg, case, stretch):
    lib = pkg.lib()
    sw, sh, tw, th, rmode, ops = case
    lib.achip_stage_extent.restype = C.c_size_t
    lib.achip_stage_extent.argtypes = [C.POINTER(pkg.Frame), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.achip_stage_gather.restype = None
    lib.achip_stage_gather.argtypes = [C.POINTER(pkg.Frame), C.c_void_p, C.c_void_p, C.POINTER(pkg.Frame)]
    rng = np.random.default_rng(sw * 31 + sh)
    img = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    f = pkg.Frame()
    assert lib.achip_frame_setup(C.byref(f), None, sw, sh, tw, th, rmode, True, not stretch, stretch) == 0
    f.ops = ops
    w, h = C.c_int(), C.c_int()
    nbytes = lib.achip_stage_extent(C.byref(f), C.byref(w), C.byref(h))
    want = sample_all(img, f)
    if nbytes == 0:
        assert f.out_w * 2 > sw and f.out_h >= sh
        return
    assert nbytes == w.value * h.value * 3 and nbytes < img.nbytes
    assert w.value in (sw, f.out_w) and h.value in (sh, f.out_h)
    dst = np.full(nbytes + 8, 0xEE, dtype=np.uint8)
    d = pkg.Frame()
    C.memmove(C.byref(d), C.byref(f), C.sizeof(f))
    lib.achip_stage_gather(C.byref(f), img.ctypes.data, dst.ctypes.data, C.byref(d))
    assert (dst[nbytes:] == 0xEE).all()
    assert (d.src_w, d.src_h, d.src_stride) == (w.value, h.value, w.value * 3)
    assert (d.out_w, d.out_h, d.pad_left, d.pad_top) == (f.out_w, f.out_h, f.pad_left, f.pad_top)
    got = sample_all(dst[:nbytes].reshape(h.value, w.value, 3), d)
    assert np.array_equal(got, want)


def test_rows_only_switch(pkg, monkeypatch):
    """2 * out_w > src_w: rows are compacted, columns are not (a per-pixel gather would cost more than it saves)"""
    lib = pkg.lib()
    lib.achip_stage_extent.restype = C.c_size_t
    f = pkg.Frame()
    assert lib.achip_frame_setup(C.byref(f), None, 120, 1080, 80, 24, 0, False, False, True) == 0
    w, h = C.c_int(), C.c_int()
    assert lib.achip_stage_extent(C.byref(f), C.byref(w), C.byref(h)) == 120 * 24 * 3
    assert (w.value, h.value) == (120, 24)
