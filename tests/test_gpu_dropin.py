"""-m gpu: the reference's own entry points (asciichat_render.h) exported by libasciichat_hip.so."""
import ctypes as C
import os
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import orc  # noqa: E402


@pytest.fixture(scope="module")
def api():
    import torch  # noqa: F401  (loads the HIP runtime first)

    from __graft_entry__ import load_package

    pkg = load_package()
    assert pkg.lib().asciichat_hip_device_count() > 0
    return pkg


def as_image(pkg, arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    im = pkg.Image(arr.shape[1], arr.shape[0], arr.ctypes.data, 0)
    im._keep = arr
    return im


def caps(pkg, color_level, render_mode, wants_padding=False):
    c = pkg.TermCaps()
    c.color_level = color_level
    c.render_mode = render_mode
    c.wants_padding = wants_padding
    c.utf8_support = True
    return c


TORTURE = orc.frame_torture()
PAL = orc.PALETTE_STANDARD.encode()


def test_ascii_convert_with_capabilities_matrix(api):
    L = api.lib()
    im = as_image(api, TORTURE)
    for cl in (0, 1, 2, 3):
        for rm in (0, 2):
            for (aspect, pad) in ((False, False), (True, True), (True, False)):
                c = caps(api, cl, rm, pad)
                got = api.take_string(L.ascii_convert_with_capabilities(C.byref(im), 97, 31, C.byref(c), aspect, False, PAL))
                exp = orc.convert_with_caps(TORTURE, 97, 31, cl, rm, pad, aspect, False)
                assert got == exp, (cl, rm, aspect, pad)


def test_config1_ascii_convert_mono_640x480(api):
    """BASELINE configs[0]: single 640x480 frame -> 80x24 monochrome via ascii_convert (host.c:696 call shape)."""
    L = api.lib()
    g = orc.frame_anchor_gradient()
    im = as_image(api, g)
    lum = C.create_string_buffer(b"x" * 255, 256)
    got = api.take_string(L.ascii_convert(C.byref(im), 80, 24, False, False, False, PAL, lum))
    assert len(got) == 1635 and orc.fnv1a32(got) == 0xCEFD0A18  # SURVEY 8(c) anchor of the reference's output
    for color, mode_opt in ((True, 0), (True, 2), (True, 1), (False, 0)):
        L.asciichat_hip_set_option_render_mode(mode_opt)
        for aspect in (False, True):
            got = api.take_string(L.ascii_convert(C.byref(im), 80, 24, color, aspect, False, PAL, lum))
            assert got == orc.convert(g, 80, 24, color, aspect, False, orc.PALETTE_STANDARD, mode_opt), (color, mode_opt, aspect)
    L.asciichat_hip_set_option_render_mode(0)


def test_null_and_error_conditions(api):
    L = api.lib()
    im = as_image(api, TORTURE)
    c = caps(api, 3, 0)
    lum = C.create_string_buffer(b"x" * 255, 256)
    # ascii.c:75-84, 198-212; tests/unit/video/ascii_test.c:118-156, 741-765
    assert not L.ascii_convert(None, 80, 24, False, False, False, PAL, lum)
    assert not L.ascii_convert(C.byref(im), 80, 24, False, False, False, None, lum)
    assert not L.ascii_convert(C.byref(im), 80, 24, False, False, False, b"", lum)
    assert not L.ascii_convert(C.byref(im), 0, 24, False, False, False, PAL, lum)
    assert not L.ascii_convert_with_capabilities(None, 80, 24, C.byref(c), False, False, PAL)
    assert not L.ascii_convert_with_capabilities(C.byref(im), 80, 24, None, False, False, PAL)
    assert not L.ascii_convert_with_capabilities(C.byref(im), 0, 0, C.byref(c), False, False, PAL)
    assert not L.ascii_convert_with_capabilities(C.byref(im), 5000, 24, C.byref(c), False, False, PAL)  # > 3840 wide
    bad = api.Image(0, 10, im.pixels, 0)
    assert not L.ascii_convert_with_capabilities(C.byref(bad), 80, 24, C.byref(c), False, False, PAL)
    assert not L.image_print(None, PAL)
    assert not L.image_print_with_capabilities(C.byref(im), None, PAL)
    # TRUECOLOR + BACKGROUND dispatches to the Floyd-Steinberg 16-colour renderer (sgr.c:429-430)
    cb = caps(api, 3, 1)
    got = api.take_string(L.ascii_convert_with_capabilities(C.byref(im), 97, 31, C.byref(cb), False, False, PAL))
    assert got == orc.convert_with_caps(TORTURE, 97, 31, 3, 1) and len(got) == 34602  # SURVEY App. B anchor
    got = api.take_string(L.image_print_color_simd(C.byref(im), True, False, PAL))
    assert got == orc.print_with_caps(TORTURE, 3, 1)
    # halfblock.c:50-51: non-positive dims -> empty string, not NULL
    assert api.take_string(L.rgb_to_truecolor_halfblocks_scalar(im.pixels, 0, 5, 0)) == b""


def test_image_print_family_and_halfblocks(api):
    L = api.lib()
    small = orc.resize_nn(TORTURE, 61, 23)  # odd height: last half-block row duplicates the top row
    im = as_image(api, small)
    for fn, (cl, rm) in (("image_print", (0, 0)), ("image_print_color", (3, 0)), ("image_print_256color", (2, 0)),
                         ("image_print_16color", (1, 0))):
        got = api.take_string(getattr(L, fn)(C.byref(im), PAL))
        assert got == orc.print_with_caps(small, cl, rm), fn
    assert api.take_string(L.image_print_color_background(C.byref(im), PAL)) == orc.print_truecolor_bg(small)
    # the three exported forms of the Floyd-Steinberg renderer (image.h:462,488)
    for pal in (PAL, orc.PALETTE_BLOCKS.encode()):
        for bgm in (True, False):
            got = api.take_string(L.image_print_16color_dithered_with_background(C.byref(im), bgm, pal))
            assert got == orc.print_16_dithered(small, bgm, pal), bgm
        got = api.take_string(L.image_print_16color_dithered(C.byref(im), pal))
        assert got == orc.print_16_dithered(small, False, pal, ramp_glyph=True)
    assert not L.image_print_16color_dithered(None, PAL)
    assert not L.image_print_16color_dithered_with_background(C.byref(im), False, None)
    for cl in (0, 1, 2, 3):
        c = caps(api, cl, 2)
        got = api.take_string(L.image_print_with_capabilities(C.byref(im), C.byref(c), PAL))
        assert got == orc.print_with_caps(small, cl, 2), cl
    # explicit row stride: render the left 40 columns of the 61-wide image
    got = api.take_string(L.rgb_to_truecolor_halfblocks_scalar(im.pixels, 40, 23, 61 * 3))
    assert got == orc.print_with_caps(np.ascontiguousarray(small[:, :40]), 3, 2)


def test_pool_images_are_read_in_place(api):
    """image_new_from_pool() frames (> 4 MiB) come from the pinned, device-mapped class: zero-copy source."""
    L = api.lib()
    img = orc.frame_hash_noise(1920, 1080, 5)
    p = L.image_new_from_pool(1920, 1080)
    assert p and p.contents.alloc_method == 1
    assert L.buffer_pool_is_pinned(p.contents.pixels)
    assert L.buffer_pool_pinned_blocks(None) >= 1
    C.memmove(p.contents.pixels, img.ctypes.data, img.nbytes)
    for cl, rm in ((3, 0), (2, 0), (3, 2)):
        c = caps(api, cl, rm, True)
        got = api.take_string(L.ascii_convert_with_capabilities(p, 80, 24, C.byref(c), True, False, PAL))
        assert got == orc.convert_with_caps(img, 80, 24, cl, rm, True, True, False)
    # image_resize: pool -> pool and pool -> malloc'd image
    d1 = L.image_new_from_pool(1600, 1000)  # > 4 MiB -> pinned destination
    d2 = L.image_new(53, 30)
    L.image_resize(p, d1)
    L.image_resize(p, d2)
    a1 = np.frombuffer(C.string_at(d1.contents.pixels, 1600 * 1000 * 3), np.uint8).reshape(1000, 1600, 3)
    a2 = np.frombuffer(C.string_at(d2.contents.pixels, 53 * 30 * 3), np.uint8).reshape(30, 53, 3)
    assert np.array_equal(a1, orc.resize_nn(img, 1600, 1000))
    assert np.array_equal(a2, orc.resize_nn(img, 53, 30))
    L.image_destroy_to_pool(d1)
    L.image_destroy(d2)
    L.image_destroy(p)  # pool-allocated images may also be released through image_destroy (image.c:86-125)
    # the block is recycled, not freed
    q = L.image_new_from_pool(1920, 1080)
    assert L.buffer_pool_is_pinned(q.contents.pixels)
    L.image_destroy_to_pool(q)
    small = L.buffer_pool_alloc(None, 1000)
    assert small and not L.buffer_pool_is_pinned(small)
    L.buffer_pool_free(None, small, 1000)


def test_concurrent_render_threads(api):
    """One render thread per client (src/server/render.c:1233): the drop-in layer is re-entrant."""
    L = api.lib()
    imgs = [orc.frame_hash_noise(640, 360, 100 + i) for i in range(6)]
    exp = [orc.convert_with_caps(im, 80, 24, 3, 0) for im in imgs]
    errors = []

    def worker(k):
        im = as_image(api, imgs[k])
        c = caps(api, 3, 0)
        for _ in range(20):
            got = api.take_string(L.ascii_convert_with_capabilities(C.byref(im), 80, 24, C.byref(c), False, False, PAL))
            if got != exp[k]:
                errors.append(k)
                return

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors


def test_into_caller_buffer_from_many_threads(api):
    """ascii_convert_with_capabilities_into (additive): each render thread keeps ONE buffer and gets every frame in it --
    direct path (few threads) and through the combiner (many), a buffer too small reports ERROR_BUFFER and the size needed,
    and plain calls interleaved on the same threads still hand out malloc blocks."""
    L = api.lib()
    L.asciichat_hip_set_coalesce_min_callers.restype = C.c_int
    L.asciichat_hip_set_coalesce_min_callers.argtypes = [C.c_int]
    imgs = [orc.frame_hash_noise(640, 360, 300 + i) if i % 3 else orc.frame_bars(640, 360, i) for i in range(24)]
    modes = [(3, 0), (2, 0), (3, 2), (0, 0)]
    exp = [orc.convert_with_caps(im, 80, 24, *modes[k % 4]) for k, im in enumerate(imgs)]
    for nthreads, setting in ((3, 0), (24, 1), (24, 6)):
        L.asciichat_hip_set_coalesce_min_callers(setting)
        errors = []

        def worker(k):
            im = as_image(api, imgs[k])
            c = caps(api, *modes[k % 4])
            buf = C.create_string_buffer(len(exp[k]) + 1)
            n = C.c_size_t(0)
            for it in range(12):
                rc = L.ascii_convert_with_capabilities_into(C.byref(im), 80, 24, C.byref(c), False, False, PAL, buf, len(buf), C.byref(n))
                if rc != 0 or n.value != len(exp[k]) or buf.raw[:n.value] != exp[k] or buf.raw[n.value] != 0:
                    errors.append(("into", k, it, rc, n.value))
                    return
                if it % 4 == 3:
                    if api.take_string(L.ascii_convert_with_capabilities(C.byref(im), 80, 24, C.byref(c), False, False, PAL)) != exp[k]:
                        errors.append(("plain", k, it))
                        return
            small = C.create_string_buffer(len(exp[k]) // 2)
            rc = L.ascii_convert_with_capabilities_into(C.byref(im), 80, 24, C.byref(c), False, False, PAL, small, len(small), C.byref(n))
            if rc != 81 or n.value != len(exp[k]):
                errors.append(("small", k, rc, n.value))

        ts = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errors, (nthreads, setting, errors[:3])
    L.asciichat_hip_set_coalesce_min_callers(6)


def test_coalesced_calls_from_many_threads(api):
    """combine.c: concurrent drop-in calls share launches.  32 threads with different images, sizes, colour levels, render
    modes and palettes (so that one generation holds several (mode, palette) groups, whole-frame and row-band launches,
    pool-pinned and pageable sources) with coalescing forced, then with the default threshold; every result compared
    with the oracle."""
    L = api.lib()
    L.asciichat_hip_set_coalesce_min_callers.restype = C.c_int
    L.asciichat_hip_set_coalesce_min_callers.argtypes = [C.c_int]
    rng = np.random.default_rng(5)
    jobs = []
    for k in range(32):
        w, h = [(640, 360), (320, 200), (1920, 1080), (97, 61)][k % 4]
        im = orc.frame_hash_noise(w, h, 700 + k) if k % 3 else orc.frame_bars(w, h, k)
        cl, rm = [(3, 0), (2, 0), (3, 2), (0, 0), (1, 0), (2, 2)][k % 6]
        W, H = [(80, 24), (120, 40), (33, 17)][k % 3]
        pal = [PAL, orc.PALETTE_BLOCKS.encode(), b"ab"][k % 3] if cl else PAL
        asp = bool(k & 1)
        jobs.append((im, cl, rm, W, H, pal, asp, orc.convert_with_caps(im, W, H, cl, rm, asp, asp, False, pal.decode())))
    for setting in (1, 24):
        before = L.asciichat_hip_set_coalesce_min_callers(setting)
        errors = []

        def worker(k):
            im_np, cl, rm, W, H, pal, asp, exp = jobs[k]
            pooled = None
            if k % 4 == 2:  # the 1080p frames of every other such thread: pool-pinned, read in place
                pooled = L.image_new_from_pool(im_np.shape[1], im_np.shape[0])
                C.memmove(pooled.contents.pixels, np.ascontiguousarray(im_np).ctypes.data, im_np.size)
            im = pooled.contents if pooled else as_image(api, im_np)
            c = caps(api, cl, rm)
            c.wants_padding = asp
            for it in range(15):
                got = api.take_string(L.ascii_convert_with_capabilities(C.byref(im), W, H, C.byref(c), asp, False, pal))
                if got != exp:
                    errors.append((k, it, None if got is None else len(got), len(exp)))
                    break
            if pooled:
                L.image_destroy_to_pool(pooled)

        ts = [threading.Thread(target=worker, args=(k,)) for k in range(32)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        L.asciichat_hip_set_coalesce_min_callers(before)
        assert not errors, (setting, errors[:4])


def test_grid_and_padding_host_utilities(api):
    L = api.lib()
    img = orc.frame_anchor_gradient()
    frames = [orc.convert(img, 39 + i, 15, False, False, False) for i in range(9)]
    for n, (w, h) in ((9, (160, 48)), (4, (160, 48)), (2, (80, 24)), (3, (120, 40)), (5, (200, 60)), (1, (80, 24)),
                      (2, (60, 40)), (9, (80, 24))):
        arr = (api.FrameSource * n)()
        for i in range(n):
            arr[i].frame_data = frames[i]
            arr[i].frame_size = len(frames[i])
        sz = C.c_size_t()
        p = L.ascii_create_grid(arr, n, w, h, C.byref(sz))
        got = C.string_at(p, sz.value)
        L.free(p)
        assert got == orc.create_grid(frames[:n], w, h), (n, w, h)
    assert api.take_string(L.ascii_pad_frame_width(b"ab\ncd", 3)) == b"   ab\n   cd"
    assert api.take_string(L.ascii_pad_frame_height(b"ab\ncd", 2)) == b"\n\nab\ncd"
    assert L.rgb_to_16color(255, 0, 0) == 9 and L.rgb_to_256color(10, 10, 10) == 232
    assert L.rep_is_profitable(6) and not L.rep_is_profitable(5)


def test_palette_cache_churn(api):
    """More distinct palettes than the device glyph-table cache holds (2048, the reference's own cap): unpinned
    entries are recycled, a table a plan holds is never evicted, and every render stays byte-exact."""
    import torch

    L = api.lib()
    img = orc.frame_hash_noise(64, 48, 9)
    im = as_image(api, img)
    c = caps(api, 2, 0)
    dev = torch.from_numpy(np.ascontiguousarray(TORTURE)).cuda()
    held_pal = "  .oO@"
    plan = api.Plan(1, held_pal, [api.frame_setup(dev.data_ptr(), TORTURE.shape[1], TORTURE.shape[0], 40, 12, 0)])
    alphabet = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"
    for k in range(2100):
        pal = "  " + alphabet[k % 62] + alphabet[(k // 62) % 62] + alphabet[(k * 7) % 62] + "#"
        got = api.take_string(L.ascii_convert_with_capabilities(C.byref(im), 16, 6, C.byref(c), False, False, pal.encode()))
        if k % 150 == 0 or k > 2090:
            assert got == orc.convert_with_caps(img, 16, 6, 2, 0, False, False, False, pal), pal
    out = torch.zeros(plan.stride, dtype=torch.uint8, device="cuda")
    ln = torch.zeros(1, dtype=torch.int32, device="cuda")
    plan.render(out.data_ptr(), plan.stride, ln.data_ptr())
    torch.cuda.synchronize()
    assert out[:int(ln[0].item())].cpu().numpy().tobytes() == orc.convert_with_caps(TORTURE, 40, 12, 3, 0, False, False, False, held_pal)
    plan.close()
