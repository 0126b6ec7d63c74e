"""The display path's full-frame passes as stand-alone kernels: colour filter and flips."""
import ctypes as C
import os
import struct
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


# ---- compaction of a rendered slab ("compacted per-rank buffers, lengths first") ------------------------------
def packed_reference(slab, stride, lens):
    """off[i] = sum_{j<i} round16(len_ok[j]); frame i's bytes at off[i]"""
    off, out = [], bytearray()
    for i, l in enumerate(lens):
        l = 0 if l >= 0xFFFFFFF0 else int(l)
        off.append(len(out))
        out += slab[i * stride:i * stride + l].tobytes()
        out += bytes((-l) % 16)
    return off + [len(out)], bytes(out)


def test_pack_frames_kernel_emulated():
    rng = np.random.default_rng(5)
    for n, stride, slices in ((1, 64, 1), (5, 256, 1), (37, 1024, 3), (300, 128, 2), (3, 16384, 4)):
        slab = rng.integers(1, 256, n * stride, dtype=np.uint8)
        lens = rng.integers(0, stride + 1, n).astype(np.uint32)
        if n > 2:
            lens[1] = 0xFFFFFFFF  # a frame that overflowed its slot takes no room
            lens[2] = 0
        off_ref, bytes_ref = packed_reference(slab, stride, lens)
        cap = n * stride
        dst = np.zeros(cap, dtype=np.uint8)
        off = np.zeros(n + 1, dtype=np.uint64)
        lo = np.zeros(n, dtype=np.uint32)
        emu.lib().emu_pack(slab.ctypes.data, stride, lens.ctypes.data, n, dst.ctypes.data, cap, off.ctypes.data,
                           lo.ctypes.data, slices)
        assert list(off) == off_ref and np.array_equal(lo, lens), (n, stride)
        for i in range(n):  # the bytes of every frame (padding bytes are unspecified)
            l = 0 if lens[i] >= 0xFFFFFFF0 else int(lens[i])
            assert dst[off_ref[i]:off_ref[i] + l].tobytes() == bytes_ref[off_ref[i]:off_ref[i] + l], (n, stride, i)
        # a destination that is too small: frames that fit are copied, the total still tells the caller
        small = int(off_ref[n // 2 + 1]) if n > 1 else 0
        dst2 = np.zeros(cap, dtype=np.uint8)
        emu.lib().emu_pack(slab.ctypes.data, stride, lens.ctypes.data, n, dst2.ctypes.data, small, off.ctypes.data, None, slices)
        assert int(off[n]) == off_ref[n] and not dst2[small:].any()
        # a capacity that is NOT a multiple of 16 -- the exact sum of the lengths up to some frame.  Frames
        # travel in whole 16-byte groups, so a frame whose last group would cross the capacity is not copied and nothing
        # is stored at or behind dst + capacity
        for k in range(n):
            l = 0 if lens[k] >= 0xFFFFFFF0 else int(lens[k])
            exact = off_ref[k] + l
            if exact % 16 == 0:
                continue
            guard = np.full(cap + 16, 0xEE, dtype=np.uint8)
            emu.lib().emu_pack(slab.ctypes.data, stride, lens.ctypes.data, n, guard.ctypes.data, exact, off.ctypes.data, None, slices)
            assert (guard[exact:] == 0xEE).all(), (n, stride, k)
            for i in range(k):  # every frame whose last group fits arrived
                li = 0 if lens[i] >= 0xFFFFFFF0 else int(lens[i])
                if off_ref[i] + (li + 15) // 16 * 16 <= exact:
                    assert guard[off_ref[i]:off_ref[i] + li].tobytes() == bytes_ref[off_ref[i]:off_ref[i] + li]
            break


@pytest.mark.gpu
def test_pack_frames_gpu_device_and_mapped_host_destinations():
    import torch

    from __graft_entry__ import load_package

    pkg = load_package()
    torch.cuda.set_device(0)
    imgs = [orc.frame_hash_noise(192, 108, 70 + i) if i % 4 else orc.frame_bars(192, 108, i) for i in range(40)]
    dev = [torch.from_numpy(i).cuda() for i in imgs]
    frames = [pkg.frame_setup(d.data_ptr(), 192, 108, 80, 24, 0, False, False, False) for d in dev]
    plan = pkg.Plan(pkg.MODE_TRUE_FG, orc.PALETTE_STANDARD, frames)
    n, stride = len(imgs), plan.stride
    slab = torch.zeros(n * stride, dtype=torch.uint8, device="cuda")
    ln = torch.zeros(n, dtype=torch.int32, device="cuda")
    exp = [orc.convert_with_caps(im, 80, 24, 3, 0, False, False, False) for im in imgs]
    st = torch.cuda.current_stream().cuda_stream
    # (a) device destination
    dst = torch.zeros(n * stride, dtype=torch.uint8, device="cuda")
    off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    plan.render_packed(slab.data_ptr(), stride, ln.data_ptr(), dst.data_ptr(), dst.numel(), off.data_ptr(), None, st)
    torch.cuda.synchronize()
    o, l, d = off.cpu().numpy(), ln.cpu().numpy(), dst.cpu().numpy()
    assert int(o[n]) == sum((len(e) + 15) // 16 * 16 for e in exp) and int(o[n]) < n * stride
    for i in range(n):
        assert int(l[i]) == len(exp[i]) and d[int(o[i]):int(o[i]) + int(l[i])].tobytes() == exp[i], i
    # (b) mapped pinned host destination: the kernel's stores are the transfer; tables in the same block
    tab = 8 * (n + 1) + 4 * n
    tab = (tab + 15) // 16 * 16
    hb = pkg.HostBuffer(tab + n * stride)
    hb.view()[:] = 0
    plan.render_packed(slab.data_ptr(), stride, ln.data_ptr(), hb.dev + tab, n * stride, hb.dev, hb.dev + 8 * (n + 1), st)
    torch.cuda.synchronize()
    v = hb.view()
    o2 = v[:8 * (n + 1)].view(np.uint64)
    l2 = v[8 * (n + 1):8 * (n + 1) + 4 * n].view(np.uint32)
    # (the device destination above took the one-launch form: frames in completion order; into mapped host memory the plan
    # renders and packs: frames in frame order -- the offsets differ, the lengths, the total and every frame's bytes do not)
    assert int(o2[n]) == int(o[n]) and np.array_equal(l2, l.astype(np.uint32))
    assert all(int(o2[i + 1]) == int(o2[i]) + (int(l2[i]) + 15) // 16 * 16 for i in range(n))
    for i in range(n):
        assert v[tab + int(o2[i]):tab + int(o2[i]) + int(l2[i])].tobytes() == exp[i], i
    hb.close()
    plan.close()


def _sample_all(img, f):
    """what the device sampler reads for every cell of the out_w x out_h image (render_kernels.hpp sample_frame_raw)"""
    h, w, _ = img.shape
    xs = np.minimum((np.arange(f.out_w, dtype=np.uint64) * f.x_ratio) >> 16, w - 1).astype(np.int64)
    ys = np.minimum((np.arange(f.out_h, dtype=np.uint64) * f.y_ratio) >> 16, h - 1).astype(np.int64)
    if f.ops & 1:
        xs = w - 1 - xs
    if f.ops & 2:
        ys = h - 1 - ys
    return img[ys][:, xs]


class _SampleSet(C.Structure):
    _fields_ = [("w", C.c_uint32), ("h", C.c_uint32), ("n_rows", C.c_int), ("n_cols", C.c_int),
                ("rows", C.POINTER(C.c_uint32)), ("cols", C.POINTER(C.c_uint32))]


@pytest.mark.parametrize("case", [
    (1920, 1080, [(80, 24, 0, 0)]),                                  # sampled pixels: 80 x 24 of 1920 x 1080
    (1920, 1080, [(80, 24, 0, 0), (60, 20, 2, 1), (100, 37, 0, 2)]),  # three targets, flips: union of rows x union of columns
    (333, 201, [(100, 37, 0, 3), (80, 24, 2, 0)]),                   # odd geometry
    (333, 201, [(100, 37, 0, 0), (200, 60, 0, 0)]),                  # a wide target: whole rows
    (40, 30, [(80, 24, 0, 0)]),                                      # upscale: every row and column
], ids=["pixels", "union", "odd", "rows", "upscale"])
def test_ingest_sample_set_and_scatter_emulated(case):
    """frame_table_publish_rows_batch without a GPU: the host's sample set + packed block (achip_host.c) and the scatter
    kernel under the emulator put exactly what every target's sampler reads into a full-geometry frame."""
    w, h, targets = case
    L = emu.lib()
    rng = np.random.default_rng(w + 7 * len(targets))
    n_clients = 3
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(n_clients)]
    tf = (emu.Frame * len(targets))()
    for i, (tw, th, rm, ops) in enumerate(targets):
        assert L.achip_frame_setup(C.byref(tf[i]), None, w, h, tw, th, rm, False, False, True) == 0
        tf[i].ops = ops
    S = _SampleSet()
    L.achip_sample_set_build.argtypes = [C.POINTER(_SampleSet), C.POINTER(emu.Frame), C.c_int, C.c_uint32, C.c_uint32]
    L.achip_sample_set_block_bytes.restype = C.c_size_t
    L.achip_sample_set_block_bytes.argtypes = [C.POINTER(_SampleSet)]
    L.achip_sample_set_pack.restype = None
    L.achip_sample_set_pack.argtypes = [C.POINTER(_SampleSet), C.c_void_p, C.c_void_p]
    L.achip_sample_set_free.argtypes = [C.POINTER(_SampleSet)]
    L.emu_scatter_rows_batch.restype = None
    L.emu_scatter_rows_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
    assert L.achip_sample_set_build(C.byref(S), tf, len(targets), w, h) == 0
    need_cols = sorted({int(x) for f in tf for x in (_sample_all(np.arange(w)[None, :, None].repeat(h, 0), f)[0, :, 0])})
    need_rows = sorted({int(y) for f in tf for y in (_sample_all(np.arange(h)[:, None, None].repeat(w, 1), f)[:, 0, 0])})
    assert [S.rows[i] for i in range(S.n_rows)] == need_rows
    if 2 * len(need_cols) <= w:
        assert [S.cols[i] for i in range(S.n_cols)] == need_cols
    else:
        assert S.n_cols == 0
    blk_bytes = L.achip_sample_set_block_bytes(C.byref(S))
    head = (n_clients * 32 + 15) // 16 * 16
    staged = np.zeros(head + n_clients * blk_bytes, dtype=np.uint8)
    frames = [np.full((h, w, 3), 0xEE, dtype=np.uint8) for _ in range(n_clients)]
    for c in range(n_clients):
        off = head + c * blk_bytes
        L.achip_sample_set_pack(C.byref(S), imgs[c].ctypes.data, staged.ctypes.data + off)
        rec = np.zeros(8, dtype=np.uint32)
        rec[0], rec[1] = frames[c].ctypes.data & 0xFFFFFFFF, frames[c].ctypes.data >> 32
        rec[2], rec[3], rec[4], rec[5] = off, S.n_rows, 3 * w, S.n_cols
        staged[32 * c:32 * c + 32] = rec.view(np.uint8)
    for slices in (1, 2):
        L.emu_scatter_rows_batch(staged.ctypes.data, n_clients, S.n_rows, slices)
        for c in range(n_clients):
            for f in tf:
                assert np.array_equal(_sample_all(frames[c], f), _sample_all(imgs[c], f))
            arrived = (frames[c] == imgs[c]).all(axis=2)
            expect = len(need_rows) * (len(need_cols) if S.n_cols else w)
            assert expect <= int(arrived.sum()) <= expect + w * h // 200  # (+ chance matches with the 0xEE fill: none in practice)
    L.achip_sample_set_free(C.byref(S))


@pytest.fixture(params=[0, 1], ids=["spans+finish", "one launch"])
def one_launch(request):
    """the span form followed by crc32c_finish_kernel / finishing its frames itself (the last span to arrive combines)"""
    emu.lib().emu_set_crc_one_launch(request.param)
    yield request.param
    emu.lib().emu_set_crc_one_launch(0)


@pytest.mark.parametrize("force", [None, (3, 1), (5, 2)], ids=["frame-kernel", "spans-3x4K", "spans-5x8K"])
def test_checksum_and_pack_in_one_pass_emulated(force, one_launch):
    """crc_kernels.hpp COPY instantiations: the pass that checksums a slab also compacts it.  Checksums, headers and packet
    CRCs as the plain pass computes them (and the oracle), offsets and bytes as pack_frames_kernel lays them out -- for
    frames of every length class (empty, < 16 bytes, ends on / next to a group or span boundary, an overflowed slot), through
    the one-workgroup-per-frame kernel and through the span kernels (forced small spans), and with a destination too small."""
    L = emu.lib()
    L.emu_crc32c_pack.restype = None
    L.emu_crc32c_pack.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(11)
    span = force[1] * 4096 if force else 0
    sizes = [0, 1, 15, 16, 17, 4095, 4096, 4097, 12000, 0xFFFFFFFF]
    if force:
        sizes += [span - 1, span, span + 1, 2 * span + 5, force[0] * span - 3, force[0] * span]
    stride = (max(s for s in sizes if s < 0xFFFFFFF0) + 16 + 15) & ~15
    n = len(sizes)
    slab = rng.integers(1, 256, n * stride + 16, dtype=np.uint8)
    base = slab.ctypes.data + (-slab.ctypes.data % 16)
    view = np.ctypeslib.as_array((C.c_uint8 * (n * stride)).from_address(base))
    lens = np.array(sizes, dtype=np.uint32)
    dims = np.array([(80 + i, 24 + i) for i in range(n)], dtype=np.uint32)
    off_ref, bytes_ref = packed_reference(view, stride, lens)
    fp, fr = force if force else (0, 0)
    for cap in (n * stride, off_ref[n // 2], off_ref[2] + sizes[2], off_ref[5] + sizes[5], off_ref[7] + sizes[7]):
        crc = np.zeros(n, dtype=np.uint32)
        hdr = np.zeros(n * 24, dtype=np.uint8)
        pkt = np.zeros(n, dtype=np.uint32)
        dst = np.zeros(n * stride + 16, dtype=np.uint8)
        dbase = dst.ctypes.data + (-dst.ctypes.data % 16)
        dview = np.ctypeslib.as_array((C.c_uint8 * (n * stride)).from_address(dbase))
        off = np.zeros(n + 1, dtype=np.uint64)
        lo = np.zeros(n, dtype=np.uint32)
        mx = stride if not force else force[0] * span
        L.emu_crc32c_pack(base, stride, lens.ctypes.data, mx, n, fp, fr, dims.ctypes.data, crc.ctypes.data, hdr.ctypes.data,
                          pkt.ctypes.data, dbase, cap, off.ctypes.data, lo.ctypes.data)
        assert list(off) == off_ref and np.array_equal(lo, lens)
        for i, s in enumerate(sizes):
            l = 0 if s >= 0xFFFFFFF0 else s
            frame = view[i * stride:i * stride + l].tobytes()
            want = 0 if s >= 0xFFFFFFF0 else orc.crc32c(frame)
            assert int(crc[i]) == want, (force, i, s)
            w, h = (0, 0) if s >= 0xFFFFFFF0 else (int(dims[i][0]), int(dims[i][1]))
            hd = struct.pack(">IIIIII", w, h, l, 0, want, 0)
            assert hdr[24 * i:24 * i + 24].tobytes() == hd, (force, i, s)
            if s < 0xFFFFFFF0:
                assert int(pkt[i]) == orc.crc32c(hd + frame), (force, i, s)
            if off_ref[i] + (l + 15) // 16 * 16 <= cap:  # frames travel in whole groups: the last one must fit too
                assert dview[off_ref[i]:off_ref[i] + l].tobytes() == frame, (force, i, s, cap)
        if cap < n * stride:  # nothing at or behind the capacity is touched, whatever its alignment
            assert not dview[cap:].any()
