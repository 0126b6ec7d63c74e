"""The product's host C on a mock HIP runtime (tests/mockgpu.py, tests/mockhip): device memory is host memory and every
launch runs the product kernels under the CPU fiber emulator.  CPU only.  Covers what otherwise needs a GPU box: the drop-in
entry points end to end (staging of the sampled pixels, direct path, the flat-combining layer with callers that poll and
callers that sleep), plans, and the frame table's publish forms -- all against the oracle, byte for byte."""
import ctypes as C
import os
import struct
import subprocess
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import mockgpu  # noqa: E402
import orc  # noqa: E402

PAL = orc.PALETTE_STANDARD.encode()


@pytest.fixture(scope="module")
def mock():
    pkg = mockgpu.package()
    assert pkg.lib().asciichat_hip_device_count() == 1
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


def test_dropin_direct_path_stages_sampled_pixels(mock):
    """one caller at a time: every colour level x render mode, downscales (sampled rows AND columns staged), a mild
    downscale (rows only), an upscale (the whole image), aspect + padding"""
    L = mock.lib()
    for (w, h, W, H) in ((333, 201, 40, 12), (333, 201, 200, 40), (40, 30, 64, 20), (160, 120, 40, 12)):
        img = orc.frame_hash_noise(w, h, w + H)
        im = as_image(mock, img)
        for cl, rm in ((0, 0), (1, 0), (2, 0), (3, 0), (3, 2), (2, 2), (3, 1)):
            for aspect in (False, True):
                c = caps(mock, cl, rm, aspect)
                got = mock.take_string(L.ascii_convert_with_capabilities(C.byref(im), W, H, C.byref(c), aspect, False, PAL))
                assert got == orc.convert_with_caps(img, W, H, cl, rm, aspect, aspect, False), (w, h, W, H, cl, rm, aspect)


def test_dropin_geometry_is_chosen_for_the_image_the_kernel_will_see(mock):
    """a source ONE pixel wide sampled by a wide one-row target is staged as a 1 x 1 image, which needs the kernels' general
    sampler -- and rows geometry 26, which the policy takes for wide mono / half-block rows, does not carry it (the mock's
    launcher refuses the combination as the product's does).  Found by scripts/gpu_dropin_fuzz.py on the GPU (round 5): the
    direct path chose its geometry for the caller's descriptor, not for the staged one, and returned NULL."""
    L = mock.lib()
    for (w, h) in ((1, 1080), (1, 1), (2, 1080)):
        img = orc.frame_hash_noise(w, h, 7 + w + h)
        im = as_image(mock, img)
        for (W, H, cl, rm, pal) in ((200, 1, 0, 0, b"@"), (200, 1, 0, 0, PAL), (160, 1, 3, 2, PAL), (320, 3, 0, 2, PAL), (200, 3, 3, 2, PAL)):
            c = caps(mock, cl, rm, True)
            got = mock.take_string(L.ascii_convert_with_capabilities(C.byref(im), W, H, C.byref(c), False, False, pal))
            assert got is not None and got == orc.convert_with_caps(img, W, H, cl, rm, True, False, False, pal), (w, h, W, H, cl, rm)


@pytest.mark.parametrize("threads", [4, 12, 24])
def test_dropin_calls_through_the_combiner(mock, threads):
    """combine.c on the CPU: `threads` concurrent callers with different images, sizes, modes and palettes, coalescing
    forced -- fewer callers than CPUs (they poll), then more (they sleep on futexes and are woken as a tree; the launcher
    takes the blocking wait) -- every result against the oracle, over several rounds of generations"""
    L = mock.lib()
    L.asciichat_hip_set_coalesce_min_callers.restype = C.c_int
    L.asciichat_hip_set_coalesce_min_callers.argtypes = [C.c_int]
    jobs = []
    for k in range(threads):
        w, h = [(160, 120), (97, 61), (320, 200), (64, 48)][k % 4]
        img = orc.frame_hash_noise(w, h, 500 + k) if k % 3 else orc.frame_bars(w, h, k)
        cl, rm = [(3, 0), (2, 0), (3, 2), (0, 0), (1, 0), (2, 2)][k % 6]
        W, H = [(40, 12), (33, 17), (20, 10)][k % 3]
        pal = [PAL, b"ab", orc.PALETTE_BLOCKS.encode()][k % 3] if cl else PAL
        asp = bool(k & 1)
        jobs.append((img, cl, rm, W, H, pal, asp, orc.convert_with_caps(img, W, H, cl, rm, asp, asp, False, pal.decode())))
    before = L.asciichat_hip_set_coalesce_min_callers(1)
    errors = []

    def worker(k):
        img, cl, rm, W, H, pal, asp, exp = jobs[k]
        im = as_image(mock, img)
        c = caps(mock, cl, rm, asp)
        for it in range(6):
            got = mock.take_string(L.ascii_convert_with_capabilities(C.byref(im), W, H, C.byref(c), asp, False, pal))
            if got != exp:
                errors.append((k, it, None if got is None else len(got), len(exp)))
                return

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    L.asciichat_hip_set_coalesce_min_callers(before)
    assert not errors, errors[:4]


def test_dropin_into_caller_buffer_on_the_mock(mock):
    """ascii_convert_with_capabilities_into (additive): the frame in the caller's buffer, no malloc -- the direct path and
    through the combiner, the padded form that goes through intermediate strings, a buffer that is too small (ERROR_BUFFER and
    the size it needs), NULL conditions, and the plain entry point right behind it (the thread-local target is gone again)."""
    L = mock.lib()
    img = orc.frame_hash_noise(160, 120, 5)
    im = as_image(mock, img)
    for (W, H, cl, rm, pad) in ((40, 12, 3, 0, False), (33, 17, 2, 0, True), (40, 12, 3, 2, True), (20, 9, 0, 0, False)):
        cp = caps(mock, cl, rm, pad)
        exp = orc.convert_with_caps(img, W, H, cl, rm, pad, True, False)
        for coalesce in (0, 1):
            L.asciichat_hip_set_coalesce_min_callers(coalesce)
            buf = C.create_string_buffer(len(exp) + 1)
            n = C.c_size_t(0)
            rc = L.ascii_convert_with_capabilities_into(C.byref(im), W, H, C.byref(cp), True, False, PAL, buf, len(buf), C.byref(n))
            assert rc == 0 and n.value == len(exp) and buf.raw[:n.value] == exp and buf.raw[n.value] == 0, (W, H, cl, rm, coalesce)
            small = C.create_string_buffer(len(exp))  # one byte short: the NUL does not fit
            rc = L.ascii_convert_with_capabilities_into(C.byref(im), W, H, C.byref(cp), True, False, PAL, small, len(small), C.byref(n))
            assert rc == 81 and n.value == len(exp), (rc, n.value, len(exp))
            # ... and the plain entry point right behind it hands out a malloc block again
            assert mock.take_string(L.ascii_convert_with_capabilities(C.byref(im), W, H, C.byref(cp), True, False, PAL)) == exp
    L.asciichat_hip_set_coalesce_min_callers(6)
    cp = caps(mock, 3, 0)
    buf = C.create_string_buffer(64)
    assert L.ascii_convert_with_capabilities_into(None, 40, 12, C.byref(cp), True, False, PAL, buf, 64, None) == 86
    assert L.ascii_convert_with_capabilities_into(C.byref(im), 40, 12, C.byref(cp), True, False, PAL, None, 64, None) == 86
    assert L.ascii_convert_with_capabilities_into(C.byref(im), 40, 12, C.byref(cp), True, False, PAL, buf, 0, None) == 86
    # a terminal wider than the kernel's row: the frame is padded on the host through intermediate strings, then handed over
    cpw = caps(mock, 0, 0, True)
    exp = orc.convert_with_caps(img, 5000, 6, 0, 0, True, True, False)
    big = C.create_string_buffer(len(exp) + 1)
    n = C.c_size_t(0)
    assert L.ascii_convert_with_capabilities_into(C.byref(im), 5000, 6, C.byref(cpw), True, False, PAL, big, len(big), C.byref(n)) == 0
    assert big.raw[:n.value] == exp


def test_dropin_fuzz_on_the_mock(mock):
    """random sizes, colour levels, render modes, stretch / aspect / padding through the drop-in entry point on the mock: the
    two-axis pixel gather at every kind of ratio (downscale in one axis and upscale in the other, 1-pixel images, frames wider
    than their source) against the oracle"""
    L = mock.lib()
    rng = np.random.default_rng(2024)
    for it in range(40):
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 150))
        W, H = int(rng.integers(1, 90)), int(rng.integers(1, 40))
        cl, rm = int(rng.integers(0, 4)), int(rng.choice([0, 0, 2, 1]))
        aspect, stretch, pad = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        img = orc.frame_hash_noise(w, h, 3000 + it) if it % 4 else orc.frame_bars(w, h, it)
        im = as_image(mock, img)
        c = caps(mock, cl, rm, pad)
        got = mock.take_string(L.ascii_convert_with_capabilities(C.byref(im), W, H, C.byref(c), aspect, stretch, PAL))
        exp = orc.convert_with_caps(img, W, H, cl, rm, pad, aspect, stretch)
        assert got == exp, (it, w, h, W, H, cl, rm, aspect, stretch, pad)


def test_display_passes_and_composite_on_the_mock(mock):
    """the image-space neighbours of the path through the host API on the mock: colour filters and flips as device passes
    (vector and per-pixel kernels), the display ops folded into a frame's descriptor, the materialised grid composite and the
    render straight from its sources"""
    L = mock.lib()
    for (w, h) in ((64, 20), (37, 11)):  # 16-pixel multiples take the vector kernels
        img = orc.frame_hash_noise(w, h, w)
        for flt in (1, 2, 5, 9, 11):
            buf = np.ascontiguousarray(img).copy()
            assert L.asciichat_hip_apply_color_filter(buf.ctypes.data, w, h, 3 * w, flt, None) == 0
            assert np.array_equal(buf, orc.color_filter(img, flt)), (w, h, flt)
        for fx, fy in ((True, False), (False, True), (True, True)):
            src = np.ascontiguousarray(img)
            dst = np.zeros_like(src)
            assert L.asciichat_hip_image_flip(src.ctypes.data, dst.ctypes.data, w, h, fx, fy, None) == 0
            assert np.array_equal(dst, orc.flip(img, fx, fy)), (w, h, fx, fy)
    # display ops folded into the sampler (session_display_convert_to_ascii: flips + colour filter)
    img = orc.frame_hash_noise(160, 120, 9)
    keep = np.ascontiguousarray(img)
    for fx, fy, flt in ((True, False, 0), (False, True, 3), (True, True, 7)):
        f = mock.frame_setup(keep.ctypes.data, 160, 120, 40, 12, 0, False, False, False)
        assert L.achip_frame_set_display_ops(C.byref(f), fx, fy, flt) == 0
        plan = mock.Plan(1, orc.PALETTE_STANDARD, [f])
        out = np.zeros(plan.stride, dtype=np.uint8)
        ln = np.zeros(1, dtype=np.uint32)
        plan.render(out.ctypes.data, plan.stride, ln.ctypes.data)
        pre = orc.flip(img, fx, fy)
        pre = orc.color_filter(pre, flt) if flt else pre
        assert out[:int(ln[0])].tobytes() == orc.convert_with_caps(pre, 40, 12, 3, 0, False, False, False), (fx, fy, flt)
        plan.close()
    # grid composite: four sources -> 2x2
    imgs = [orc.frame_hash_noise(120, 90, 50 + i) if i % 2 else orc.frame_bars(120, 90, i) for i in range(4)]
    keeps = [np.ascontiguousarray(i) for i in imgs]
    n = len(imgs)
    ptrs = (C.c_void_p * n)(*[k.ctypes.data for k in keeps])
    ws, hs = (C.c_int * n)(*[120] * n), (C.c_int * n)(*[90] * n)
    comp = mock.Composite()
    L.achip_composite_setup(C.byref(comp), ptrs, ws, hs, n, 80, 24)
    ref_canvas = orc.composite(imgs, 80, 24)
    canvas = np.zeros(ref_canvas.shape, dtype=np.uint8)
    assert L.asciichat_hip_composite(C.byref(comp), canvas.ctypes.data, None) == 0
    assert np.array_equal(canvas, ref_canvas)
    comp_dev = C.c_void_p()
    assert L.asciichat_hip_composite_upload(C.byref(comp), C.byref(comp_dev)) == 0
    for mode, cl, rm in ((1, 3, 0), (5, 3, 2)):
        hh = 48 if rm == 2 else 24
        f = mock.frame_setup(None, 80, 48, 80, hh, rm, True, True, False)
        f.comp = comp_dev.value
        plan = mock.Plan(mode, orc.PALETTE_STANDARD, [f])
        out = np.zeros(plan.stride, dtype=np.uint8)
        ln = np.zeros(1, dtype=np.uint32)
        plan.render(out.ctypes.data, plan.stride, ln.ctypes.data)
        assert out[:int(ln[0])].tobytes() == orc.convert_with_caps(ref_canvas, 80, hh, cl, rm, True, True, False), mode
        plan.close()
    L.asciichat_hip_free(comp_dev)


def test_plans_and_packed_output_on_the_mock(mock):
    """plan_create / update / render / render_packed through the emulator: mixed geometries in one batch, exact-length
    frames behind 16-byte aligned offsets"""
    imgs = [orc.frame_hash_noise(120 + 8 * i, 90 + 4 * i, 40 + i) for i in range(5)]
    keep = [np.ascontiguousarray(im) for im in imgs]
    frames = [mock.frame_setup(k.ctypes.data, k.shape[1], k.shape[0], 40, 12, 0, False, False, False) for k in keep]
    plan = mock.Plan(1, orc.PALETTE_STANDARD, frames)
    n, stride = len(frames), plan.stride
    slab = np.zeros(n * stride, dtype=np.uint8)
    ln = np.zeros(n, dtype=np.uint32)
    dst = np.zeros(n * stride, dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.uint64)
    ln2 = np.zeros(n, dtype=np.uint32)
    # 40x12 truecolor frames fit the kernel's LDS image: ONE launch writes them at their exact lengths, the slab stays untouched
    assert plan.exact_length
    plan.render_packed(slab.ctypes.data, stride, ln.ctypes.data, dst.ctypes.data, dst.size, off.ctypes.data, ln2.ctypes.data)
    assert not slab.any()
    for i in range(n):
        exp = orc.convert_with_caps(imgs[i], 40, 12, 3, 0, False, False, False)
        assert int(ln[i]) == len(exp)
        assert int(off[i]) % 16 == 0 and dst[int(off[i]):int(off[i]) + int(ln2[i])].tobytes() == exp
    # ... and with that form switched off: render + pack_frames, frames in the slab too and in frame order in dst
    plan.set_exact_length(0)
    assert not plan.exact_length
    dst[:] = 0
    plan.render_packed(slab.ctypes.data, stride, ln.ctypes.data, dst.ctypes.data, dst.size, off.ctypes.data, ln2.ctypes.data)
    for i in range(n):
        exp = orc.convert_with_caps(imgs[i], 40, 12, 3, 0, False, False, False)
        assert slab[i * stride:i * stride + int(ln[i])].tobytes() == exp
        assert int(off[i + 1]) == int(off[i]) + ((len(exp) + 15) & ~15) and dst[int(off[i]):int(off[i]) + int(ln2[i])].tobytes() == exp
    # ... and a caller that passes NO offsets relies on that order: it never gets the one-launch form (completion order),
    # neither by default nor when exact lengths are forced on
    for force in (-1, 1):
        plan.set_exact_length(force)
        slab[:] = 0
        dst[:] = 0xEE
        plan.render_packed(slab.ctypes.data, stride, ln.ctypes.data, dst.ctypes.data, dst.size, None, None)
        at = 0
        for i in range(n):
            exp = orc.convert_with_caps(imgs[i], 40, 12, 3, 0, False, False, False)
            assert dst[at:at + len(exp)].tobytes() == exp, (force, i)
            assert slab[i * stride:i * stride + len(exp)].tobytes() == exp  # the two-pass form wrote the slab
            at += (len(exp) + 15) & ~15
    plan.close()


def test_send_side_rules_of_the_plan(mock):
    """plan.c's own choices for the send side since round 6's wire audit (profiles/r06_wire_audit.txt): the checksum rides the render
    only for small frames -- truecolor foreground up to 8192 cells a frame (from dense sources: while a wave has one block), the other
    per-cell modes up to 12288 (dense: 8192) -- and the length-first form is the plan's own for dense sources and sources up to 1920
    pixels wide; both still follow set_fused_crc / set_exact_length."""
    def plan_for(mode, W, H, dense, src_w=333):
        img = np.ascontiguousarray(orc.frame_hash_noise(W if dense else src_w, H if dense else 50, 5))
        keep.append(img)
        frames = [mock.frame_setup(img.ctypes.data, img.shape[1], img.shape[0], W, H, 0, False, False, False) for _ in range(3)]
        p = mock.Plan(mode, orc.PALETTE_STANDARD, frames)
        p.set_variant(17)  # (whole frames: three frames alone would be cut into row bands)
        return p
    keep = []
    for mode, W, H, dense, fused in ((1, 80, 24, False, True), (1, 160, 45, False, True), (1, 200, 60, False, False),
                                      (1, 80, 24, True, True), (1, 120, 40, True, False),
                                      (2, 200, 60, False, True), (2, 320, 90, False, False), (2, 160, 45, True, True), (2, 200, 60, True, False)):
        p = plan_for(mode, W, H, dense)
        assert p.fused_crc == fused, (mode, W, H, dense)
        p.set_fused_crc(1)
        assert p.fused_crc
        p.set_fused_crc(0)
        assert not p.fused_crc
        p.close()
    for dense, src_w, auto in ((True, 0, True), (False, 1920, True), (False, 1921, False)):
        p = plan_for(1, 200, 60, dense, src_w)
        assert p.length_first  # the plan MAY take the form ...
        n, stride = 3, p.stride
        slab, ln = np.zeros(n * stride, np.uint8), np.zeros(n, np.uint32)
        dst, off, lo = np.zeros(n * stride + 16, np.uint8), np.zeros(n + 1, np.uint64), np.zeros(n, np.uint32)
        dbase = dst.ctypes.data + (-dst.ctypes.data % 16)
        p.render_packed(slab.ctypes.data, stride, ln.ctypes.data, dbase, n * stride, off.ctypes.data, lo.ctypes.data)
        assert (not slab.any()) == auto, (dense, src_w)  # ... and takes it by itself (the slab stays untouched) for these sources only
        p.close()


def test_wire_stage_on_the_mock(mock):
    """plan_render_packets through plan.c's own choice between the fused form (the CRC rides the stream kernel's drain) and a
    second pass over the slab (the rows kernel, row bands): frame CRC-32C, the 24-byte ascii_frame_packet_t headers and the
    packet CRCs against the oracle (lib/network/crc32.c:171-189, lib/network/acip/server.c:186-214)"""
    imgs = [orc.frame_hash_noise(120 + 8 * i, 90 + 4 * i, 70 + i) for i in range(4)]
    keep = [np.ascontiguousarray(im) for im in imgs]
    dims = [(40, 12), (33, 17), (20, 10), (40, 12)]
    for mode, cl, rm, variant, fused in ((1, 3, 0, 17, True), (2, 2, 0, 16, True), (1, 3, 0, -1, None), (5, 3, 2, 25, False)):
        frames = [mock.frame_setup(k.ctypes.data, k.shape[1], k.shape[0], w, h, rm, False, False, False)
                  for k, (w, h) in zip(keep, dims)]
        plan = mock.Plan(mode, orc.PALETTE_STANDARD, frames)
        if variant >= 0:
            plan.set_variant(variant)
        if fused is not None:
            assert plan.fused_crc == fused, (mode, variant)
        n, stride = len(frames), plan.stride
        out = np.zeros(n * stride, dtype=np.uint8)
        ln = np.zeros(n, dtype=np.uint32)
        d32 = np.array(dims, dtype=np.uint32)
        crc = np.zeros(n, dtype=np.uint32)
        hdr = np.zeros(n * 24, dtype=np.uint8)
        pkt = np.zeros(n, dtype=np.uint32)
        plan.render_packets(out.ctypes.data, stride, ln.ctypes.data, d32.ctypes.data, crc.ctypes.data, hdr.ctypes.data,
                            pkt.ctypes.data)
        for i in range(n):
            exp = orc.convert_with_caps(imgs[i], dims[i][0], dims[i][1], cl, rm, False, False, False)
            assert out[i * stride:i * stride + int(ln[i])].tobytes() == exp, (mode, variant, i)
            want_crc = orc.crc32c(exp)
            assert int(crc[i]) == want_crc, (mode, variant, i)
            h = struct.pack(">IIIIII", dims[i][0], dims[i][1], len(exp), 0, want_crc, 0)
            assert hdr[24 * i:24 * i + 24].tobytes() == h, (mode, variant, i)
            assert int(pkt[i]) == orc.crc32c(h + exp), (mode, variant, i)
        # the same plus the compaction: one more launch (pack) behind a fused render, ONE pass that checksums and packs otherwise
        out2, ln2, crc2, hdr2, pkt2 = np.zeros_like(out), np.zeros_like(ln), np.zeros_like(crc), np.zeros_like(hdr), np.zeros_like(pkt)
        dst = np.zeros(n * stride + 16, dtype=np.uint8)
        dbase = dst.ctypes.data + (-dst.ctypes.data % 16)
        off = np.zeros(n + 1, dtype=np.uint64)
        lo = np.zeros(n, dtype=np.uint32)
        plan.render_packets_packed(out2.ctypes.data, stride, ln2.ctypes.data, d32.ctypes.data, crc2.ctypes.data, hdr2.ctypes.data,
                                   pkt2.ctypes.data, dbase, n * stride, off.ctypes.data, lo.ctypes.data)
        assert np.array_equal(ln2, ln) and np.array_equal(crc2, crc) and np.array_equal(hdr2, hdr) and np.array_equal(pkt2, pkt)
        assert np.array_equal(lo, ln)
        assert plan.exact_length == (mode != 5), (mode, variant)  # the per-cell plans: ONE launch, no slab
        assert out2.any() != plan.exact_length
        dv = np.ctypeslib.as_array((C.c_uint8 * (n * stride)).from_address(dbase))
        spans = sorted((int(off[i]), int(off[i]) + ((int(ln[i]) + 15) & ~15)) for i in range(n))
        assert spans[0][0] == 0 and all(spans[i][1] == spans[i + 1][0] for i in range(n - 1)) and spans[-1][1] == int(off[n])
        for i in range(n):
            assert int(off[i]) % 16 == 0
            assert dv[int(off[i]):int(off[i]) + int(ln[i])].tobytes() == out[i * stride:i * stride + int(ln[i])].tobytes(), (mode, variant, i)
        plan.close()


def test_length_first_exact_length_frames_on_the_mock(mock):
    """plan.c's choice of the LENGTH-FIRST form (frames beyond the 48 KB of the LDS-image form; render_stream.hpp LF): by itself
    for dense sources (the sampled images: ratio 1.0) and -- since round 6's wire audit -- for sources up to 1920 pixels wide, on
    request for any single source (here: 2000 pixels wide, a second gather of which costs more than a pack pass), never without
    off_out; checksums, headers and packet CRCs of the frames where they lie (crc32c over offsets) equal those of render + pass."""
    W, H = 200, 60
    for dense in (True, False):
        imgs = [orc.frame_hash_noise(W if dense else 2000, H if dense else 75, 90 + i) for i in range(3)]
        keep = [np.ascontiguousarray(im) for im in imgs]
        frames = [mock.frame_setup(k.ctypes.data, k.shape[1], k.shape[0], W, H, 0, False, False, False) for k in keep]
        want = [orc.convert_with_caps(im, W, H, 3, 0, False, False, False) for im in imgs]
        plan = mock.Plan(1, orc.PALETTE_STANDARD, frames)
        plan.set_variant(17)  # (whole frames: three frames alone would be cut into row bands)
        assert not plan.exact_length and plan.length_first and plan.stride > 48 * 1024
        n, stride = len(frames), plan.stride
        d32 = np.array([(W, H)] * n, dtype=np.uint32)
        ref = None
        for setting in (0, -1, 1):
            plan.set_exact_length(setting)
            assert plan.length_first == (setting != 0)
            for wire in (False, True):
                slab = np.zeros(n * stride, dtype=np.uint8)
                ln, crc, hdr, pkt = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(24 * n, np.uint8), np.zeros(n, np.uint32)
                dst = np.zeros(n * stride + 16, dtype=np.uint8)
                dbase = dst.ctypes.data + (-dst.ctypes.data % 16)
                off, lo = np.zeros(n + 1, np.uint64), np.zeros(n, np.uint32)
                if wire:
                    plan.render_packets_packed(slab.ctypes.data, stride, ln.ctypes.data, d32.ctypes.data, crc.ctypes.data, hdr.ctypes.data,
                                               pkt.ctypes.data, dbase, n * stride, off.ctypes.data, lo.ctypes.data)
                else:
                    plan.render_packed(slab.ctypes.data, stride, ln.ctypes.data, dbase, n * stride, off.ctypes.data, lo.ctypes.data)
                assert (not slab.any()) == (setting == 1 or (setting == -1 and dense)), (dense, setting, wire)
                dv = np.ctypeslib.as_array((C.c_uint8 * (n * stride)).from_address(dbase))
                for i in range(n):
                    assert int(ln[i]) == int(lo[i]) == len(want[i]) and int(off[i]) % 16 == 0
                    assert dv[int(off[i]):int(off[i]) + len(want[i])].tobytes() == want[i], (dense, setting, wire, i)
                if wire:
                    for i in range(n):
                        assert int(crc[i]) == orc.crc32c(want[i])
                        h = struct.pack(">IIIIII", W, H, len(want[i]), 0, int(crc[i]), 0)
                        assert hdr[24 * i:24 * i + 24].tobytes() == h and int(pkt[i]) == orc.crc32c(h + want[i])
        plan.set_exact_length(1)  # no off_out: the ordered two-pass layout, whatever the setting
        slab, ln, dst = np.zeros(n * stride, np.uint8), np.zeros(n, np.uint32), np.zeros(n * stride, np.uint8)
        plan.render_packed(slab.ctypes.data, stride, ln.ctypes.data, dst.ctypes.data, dst.size, None, None)
        at = 0
        for i in range(n):
            assert dst[at:at + len(want[i])].tobytes() == want[i] and slab[i * stride:i * stride + len(want[i])].tobytes() == want[i]
            at += (len(want[i]) + 15) & ~15
        plan.close()
        del ref


def test_frame_table_publish_forms_on_the_mock(mock):
    """frame_table.c on the CPU: whole blobs, sampled rows, a tick's sampled pixels in one batch (the targets are the render
    descriptors of clients of three geometries), single-slot and batch publishes alternating on the same slots,
    latest_frames, more batches than the event ring holds -- renders against the oracle"""
    geos = [(160, 120), (97, 61), (160, 120), (64, 48), (160, 120)]
    n = len(geos)
    table = mock.FrameTable(n)
    slots = (C.c_int * n)(*range(n))
    frames = (mock.Frame * n)(*[mock.frame_setup(None, w, h, 40, 12, 0, False, False, False) for (w, h) in geos])
    for step in range(12):
        imgs = [orc.frame_hash_noise(w, h, 1000 + 31 * step + i) for i, (w, h) in enumerate(geos)]
        bufs = [C.create_string_buffer(struct.pack(">II", im.shape[1], im.shape[0]) + np.ascontiguousarray(im).tobytes(), 8 + im.size)
                for im in imgs]
        if step % 3 == 0:
            for i in range(n):
                table.publish(i, bufs[i].raw[:8 + imgs[i].size])
        elif step % 3 == 1:
            for i in range(n):
                table.publish_rows(i, (C.addressof(bufs[i]), len(bufs[i])), [frames[i]])
        else:
            table.publish_rows_batch(slots, [(C.addressof(b), len(b)) for b in bufs], frames)
        assert table.latest_frames(slots, frames) == n
        plan = mock.Plan(1, orc.PALETTE_STANDARD, list(frames))
        out = np.zeros(n * plan.stride, dtype=np.uint8)
        ln = np.zeros(n, dtype=np.uint32)
        plan.render(out.ctypes.data, plan.stride, ln.ctypes.data)
        for i in range(n):
            exp = orc.convert_with_caps(imgs[i], 40, 12, 3, 0, False, False, False)
            assert out[i * plan.stride:i * plan.stride + int(ln[i])].tobytes() == exp, (step, i)
        plan.close()
    with pytest.raises(RuntimeError):  # nobody's descriptor describes a 50x50 frame
        b = C.create_string_buffer(struct.pack(">II", 50, 50) + bytes(50 * 50 * 3), 8 + 7500)
        table.publish_rows_batch((C.c_int * 1)(0), [(C.addressof(b), len(b))], frames)
    table.close()


def _blob(im):
    return C.create_string_buffer(struct.pack(">II", im.shape[1], im.shape[0]) + np.ascontiguousarray(im).tobytes(), 8 + im.size)


def test_frame_table_sampled_image_ingest_on_the_mock(mock):
    """frame_dense.c on the CPU: a tick's clients staged as the images their targets sample (stage from several threads +
    commit, and the one-call batch on the library's ingest pool), rendered from those images -- against the oracle on the
    ORIGINAL frames.  Clients of three geometries, flips and a tint in the descriptors, aspect + padding, a client that stops
    sending (its frame is carried forward while the ring turns over twice), the same descriptor array reused tick after tick
    (latest_frames rewrites it), whole-blob publishes alternating with staged ones on one slot, and the refusals."""
    geos = [(160, 120), (97, 61), (160, 120), (64, 48), (160, 120), (160, 120)]
    n = len(geos)
    table = mock.FrameTable(n)
    slots = (C.c_int * n)(*range(n))

    def targets():
        t = [mock.frame_setup(None, w, h, 40, 12, 0, i == 4, i == 4, False) for i, (w, h) in enumerate(geos)]
        mock.lib().achip_frame_set_display_ops(C.byref(t[1]), True, False, 0)   # flip_x
        mock.lib().achip_frame_set_display_ops(C.byref(t[2]), True, True, 3)    # both flips + a tint
        return t

    def expect(i, im):
        f = targets()[i]
        src = im
        if f.ops & 1:
            src = src[:, ::-1]
        if f.ops & 2:
            src = src[::-1]
        if i == 2:
            src = orc.color_filter(np.ascontiguousarray(src), 3)
        return orc.convert_with_caps(np.ascontiguousarray(src), 40, 12, 3, 0, i == 4, i == 4, False)

    frames = (mock.Frame * n)(*targets())
    last = [None] * n
    for step in range(11):
        imgs = [orc.frame_hash_noise(w, h, 7000 + 37 * step + i) for i, (w, h) in enumerate(geos)]
        bufs = [_blob(im) for im in imgs]
        live = [i for i in range(n) if not (i == 5 and step >= 2)]  # client 5 stops sending after two ticks
        tg = targets()
        if step % 2 == 0:  # receive threads stage their own blobs, one commit for the tick
            ths = [threading.Thread(target=table.stage, args=(i, (C.addressof(bufs[i]), len(bufs[i])), tg[i])) for i in live]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            table.commit()
        else:
            table.publish_sampled_batch([slots[i] for i in live], [(C.addressof(bufs[i]), len(bufs[i])) for i in live],
                                        [tg[i] for i in live])
        if step == 5:  # a whole blob in between: the slot's latest frame is the full one again
            table.publish(3, bufs[3].raw[:8 + imgs[3].size])
            frames[3] = tg[3]
        for i in live:
            last[i] = imgs[i]
        assert table.latest_frames(slots, frames) == n, step
        if step != 5:
            # on the sampled images: rows always, columns too when the target reads at most half of them (not 64 -> 40)
            assert all(frames[i].src_h == frames[i].out_h and frames[i].y_ratio == 1 << 16 for i in range(n)), step
            assert all(frames[i].src_w == frames[i].out_w and frames[i].x_ratio == 1 << 16 for i in range(n) if i != 3), step
        plan = mock.Plan(1, orc.PALETTE_STANDARD, list(frames))
        out = np.zeros(n * plan.stride, dtype=np.uint8)
        ln = np.zeros(n, dtype=np.uint32)
        plan.render(out.ctypes.data, plan.stride, ln.ctypes.data)
        for i in range(n):
            assert out[i * plan.stride:i * plan.stride + int(ln[i])].tobytes() == expect(i, last[i]), (step, i)
        plan.close()
    # a descriptor that asks for something else than what was staged gets no source; the rest keep theirs
    other = (mock.Frame * n)(*targets())
    other[0] = mock.frame_setup(None, 160, 120, 30, 10, 0, False, False, False)
    assert table.latest_frames(slots, other) == n - 1 and not other[0].src
    with pytest.raises(RuntimeError):  # no full frame behind a staged slot
        table.latest(0)
    b = _blob(orc.frame_hash_noise(50, 50, 1))
    with pytest.raises(RuntimeError):  # the target does not describe a 50x50 frame
        table.stage(0, (C.addressof(b), len(b)), targets()[0])
    with pytest.raises(RuntimeError):  # a target that takes the frame as it is: nothing to compact
        table.stage(0, (C.addressof(b), len(b)), mock.frame_setup(None, 50, 50, 50, 50, 0, False, False, False))
    with pytest.raises(RuntimeError):  # one target, or one per blob
        table.publish_sampled_batch([0, 1, 2], [(C.addressof(b), len(b))] * 3, targets()[:2])
    table.commit()  # nothing staged: a no-op
    assert mock.lib().asciichat_hip_ingest_threads() >= 1
    table.close()


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_combiner_under_sanitizers(san):
    """tests/mockhip/dropin_threads_mock.c: the host C compiled with the sanitizer, an arithmetic stand-in for the kernels (no
    fibers), 6 / 24 / 48 threads x 300 self-checking calls with coalescing forced: callers that poll, callers that sleep"""
    exe = mockgpu.build_thread_harness(san)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=0")
    for budget, nthreads in (("1000", "6"), ("2", "24"), ("2", "48")):
        env["ASCIICHAT_HIP_CPU_BUDGET"] = budget
        p = subprocess.run([exe, nthreads, "300"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, (san, budget, p.stdout.decode()[-400:], p.stderr.decode()[-3000:])
        assert b"ok:" in p.stdout


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_frame_table_locking_under_sanitizers(san):
    """tests/mockhip/frame_table_threads_mock.c: four publishers (whole blobs, sampled rows, whole-tick batches over
    overlapping slot sets named in descending order) against four readers (latest, latest_frames) -- frame_table.c's own
    synchronisation (slot locks in ascending order under the batch lock, the ring of batch events, reader bookkeeping) under the
    sanitizer; afterwards every slot renders to the line of a published image"""
    exe = mockgpu.build_thread_harness(san, "frame_table_threads_mock")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=0")
    p = subprocess.run([exe, "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, (san, p.stdout.decode()[-400:], p.stderr.decode()[-3000:])
    assert b"ok:" in p.stdout
