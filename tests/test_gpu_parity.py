"""-m gpu: the HIP path, called through the C-ABI of libasciichat_hip.so, must be byte-identical to the
oracle (integer pipeline: the bar is bit-exact).  Small cases are compared in full; BASELINE.json's
full-size batches are compared on a sample of frames plus size-independent properties."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import orc  # noqa: E402
from achip_ctypes import (ALL_MODES, MODE_16_FG, MODE_256_FG, MODE_CAPS, MODE_HB_256, MODE_HB_TRUE, MODE_NAMES, MODE_16_DITHER_BG,  # noqa: E402
                          MODE_TRUE_BG, MODE_TRUE_FG)


@pytest.fixture(scope="module")
def gpu():
    import torch

    from __graft_entry__ import load_package

    pkg = load_package()
    assert torch.cuda.is_available(), "these tests need a GPU"
    assert pkg.lib().asciichat_hip_device_count() > 0, "libasciichat_hip.so sees no HIP device"
    torch.cuda.set_device(0)
    return pkg, torch


def oracle_convert(img, mode, W, H, palette, wants_padding=False, use_aspect=False):
    if mode == MODE_TRUE_BG:
        return orc.print_truecolor_bg(orc.resize_nn(img, W, H), palette)
    cl, rm = MODE_CAPS[mode]
    return orc.convert_with_caps(img, W, H, cl, rm, wants_padding, use_aspect, False, palette)


def render_batch(gpu, mode, imgs, W, H, palette=orc.PALETTE_STANDARD, wants_padding=False, use_aspect=False,
                 variant=-1, dims=None, split=None, repeat=1, want_parts=None):
    """imgs: list of HxWx3 uint8 numpy arrays -> list of bytes via the batch C-ABI."""
    pkg, torch = gpu
    rm = MODE_CAPS.get(mode, (3, 0))[1]
    dev = [torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in imgs]
    frames = []
    for k, (i, d) in enumerate(zip(imgs, dev)):
        w, h = (W, H) if dims is None else dims[k]
        f = pkg.frame_setup(d.data_ptr(), i.shape[1], i.shape[0], w, h, rm, wants_padding, use_aspect, False)
        assert f is not None
        frames.append(f)
    plan = pkg.Plan(mode, palette, frames)
    if variant >= 0:
        plan.set_variant(variant)
    if split is not None:
        plan.set_split(split)
    if want_parts is not None:
        assert plan.parts == want_parts, (plan.parts, want_parts)
    out = torch.full((len(imgs) * plan.stride,), 0xEE, dtype=torch.uint8, device="cuda")
    ln = torch.zeros(len(imgs), dtype=torch.int32, device="cuda")
    for _ in range(repeat):  # multi-workgroup frames: every launch is a new epoch of the hand-off words
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    host = out.cpu().numpy()
    lens = ln.cpu().numpy().astype(np.uint32)
    res = []
    for k in range(len(imgs)):
        assert lens[k] < 0xFFFFFFF0, f"frame {k}: kernel error code {lens[k]:#x}"
        res.append(host[k * plan.stride:k * plan.stride + int(lens[k])].tobytes())
        assert host[k * plan.stride + int(lens[k])] == 0  # NUL after the frame
    plan.close()
    return res


def render_descs(gpu, mode, frames, palette=orc.PALETTE_STANDARD):
    """Render ready-made descriptors (device sources) through a plan -> list of bytes."""
    pkg, torch = gpu
    plan = pkg.Plan(mode, palette, frames)
    n = len(frames)
    out = torch.zeros(n * plan.stride, dtype=torch.uint8, device="cuda")
    ln = torch.zeros(n, dtype=torch.int32, device="cuda")
    plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    host, lens = out.cpu().numpy(), ln.cpu().numpy().astype(np.uint32)
    plan.close()
    return [host[k * plan.stride:k * plan.stride + int(lens[k])].tobytes() for k in range(n)]


TORTURE = orc.frame_torture()


def geometry_built(pkg, variant):
    """frame geometry 2, stream geometry 19 and the rows kernel's fused CRC exist only in -DACHIP_ALL_GEOMETRIES builds of
    the library (render_variants.h: no plan takes them by itself); scripts/gpu_r5_*.sh run this suite on both builds"""
    L = pkg.lib()
    L.achip_variant_block.restype = C.c_int
    L.achip_variant_block.argtypes = [C.c_int]
    return L.achip_variant_block(variant) > 0


def rows_crc_built(pkg):
    L = pkg.lib()
    L.achip_variant_has_crc.restype = C.c_int
    L.achip_variant_has_crc.argtypes = [C.c_int]
    return bool(L.achip_variant_has_crc(24))


@pytest.mark.parametrize("mode", ALL_MODES, ids=MODE_NAMES)
def test_torture_all_modes_all_variants(gpu, mode):
    # every geometry of the phase kernel that the product library carries (the 64-thread test geometry exists in the
    # emulator build only), and for the per-cell modes every geometry of the stream kernel
    variants = (2, 1, 0, 4) + ((16, 17, 18, 19) if mode in (MODE_TRUE_FG, 2, 3, MODE_TRUE_BG) else ())
    if mode in (5, 6, 7, 8):  # the half-block modes have no instantiations in the 512- / 256-thread geometries
        variants = (0, 4)
    if mode in (0, 5, 6, 7, 8):  # the run-structured modes: both geometries of the rows kernel (rows up to 256 / 448 cells)
        variants += (25, 24, 26)
    for variant in variants:
        if not geometry_built(gpu[0], variant):
            continue
        for (W, H) in [(80, 24), (97, 31), (200, 60)]:
            got = render_batch(gpu, mode, [TORTURE], W, H, variant=variant)[0]
            assert got == oracle_convert(TORTURE, mode, W, H, orc.PALETTE_STANDARD), (MODE_NAMES[mode], variant, W, H)


RUN_MODES = [0, 5, 6, 7, 8]


def _run_image(w, h, kind):
    """inputs with run structure across the whole width (tests/test_kernels_emulated.py run_frames)"""
    img = np.zeros((h, w, 3), np.uint8)
    if kind == "blocks":
        xs, ys = np.meshgrid(np.arange(w), np.arange(h))
        b = (xs // 7 + 3 * (ys // 5)) % 8
        img[...] = np.stack([30 * b, 255 - 30 * b, (b * 77) % 256], axis=-1).astype(np.uint8)
        img[b % 4 == 0] = 0
    elif kind == "flat":
        img[:] = (200, 120, 40)
    elif kind == "nearblack":  # equal keys, different raw rgb: the run head's raw rgb decides transparency (halfblock.c:357,476)
        img[:] = np.random.default_rng(5).integers(0, 3, (h, w, 3))
        img[:, ::97] = 0
    return img


@pytest.mark.parametrize("mode", RUN_MODES, ids=[MODE_NAMES[m] for m in RUN_MODES])
def test_rows_wider_than_one_block_take_the_segment_geometries(gpu, mode):
    """Round 6: rows beyond 448 cells on the rows kernel, cut into segments (render_rows.hpp WIDE;
    geometries 27 / 29).  The torture image and images whose runs cross every segment boundary at 600x60 and 1000x40
    (+ the widest row the reference resizes to, 3840 cells), every geometry that takes them and the automatic choice,
    against the oracle; a 64-frame launch of 640x90 renders every frame to the same bytes (scalar ascii.c:204 admits terminals
    of up to 10 000 columns)."""
    pkg, torch = gpu
    hb = mode in (5, 6, 7, 8)
    for (W, H) in [(600, 60), (1000, 40), (449, 7)]:
        Ht = H // 2 if hb else H  # (keeps the oracle's share of the test short: H text rows of mono, H / 2 of half blocks)
        for img in (TORTURE, _run_image(W, 2 * Ht, "blocks"), _run_image(W, 2 * Ht, "flat"), _run_image(W, 2 * Ht, "nearblack"),
                    np.zeros((2 * Ht, W, 3), np.uint8)):
            exp = oracle_convert(img, mode, W, Ht, orc.PALETTE_STANDARD)
            for variant in (27, 29, -1, 0):
                got = render_batch(gpu, mode, [img], W, Ht, variant=variant)[0]
                assert got == exp, (MODE_NAMES[mode], W, Ht, variant, img.shape)
    for (W, variant) in ((3840, 27), (2560, 29), (3840, -1)):
        img = _run_image(W, 4, "blocks")
        img[2:, 100:2500] = (7, 7, 200)  # one run of 2400 cells over eight segments in the second text row
        Ht = 2 if hb else 4
        assert render_batch(gpu, mode, [img], W, Ht, variant=variant)[0] == oracle_convert(img, mode, W, Ht, orc.PALETTE_STANDARD), (MODE_NAMES[mode], W, variant)
    # aspect fit + padding: pad cells in front of every row, whole segments of them
    img = orc.frame_hash_noise(300, 400, 9)
    exp = oracle_convert(img, mode, 900, 30, orc.PALETTE_STANDARD, True, True)
    for variant in (27, 29, -1):
        assert render_batch(gpu, mode, [img], 900, 30, wants_padding=True, use_aspect=True, variant=variant)[0] == exp, (MODE_NAMES[mode], variant)
    # a whole-frame launch chooses a segment geometry by itself and every frame of it is the oracle's
    imgs = [orc.frame_hash_noise(640, 180, 100 + k) if k % 3 else _run_image(640, 180, ("blocks", "flat", "nearblack")[k % 9 // 3]) for k in range(64)]
    dev = [torch.from_numpy(i).cuda() for i in imgs]
    rm = MODE_CAPS[mode][1]
    frames = [pkg.frame_setup(d.data_ptr(), 640, 180, 640, 90 if hb else 180, rm, False, False, False) for d in dev]
    plan = pkg.Plan(mode, orc.PALETTE_STANDARD, frames)
    plan.set_concurrency(4)  # four launches in flight: a share of 64 CUs, a frame per CU of it -- whole frames
    assert plan.variant in (27, 29), plan.variant
    plan.close()
    got = render_descs(gpu, mode, frames)
    for k in (0, 1, 2, 3, 6, 63):
        assert got[k] == oracle_convert(imgs[k], mode, 640, 90 if hb else 180, orc.PALETTE_STANDARD), (MODE_NAMES[mode], k)
    assert all(got[k] == got[k % 9] for k in range(9, 64) if k % 3 == 0)  # (the structured images repeat with period nine)


def test_word_built_sgrs_at_every_field_length_and_alignment(gpu):
    """round 5: truecolor SGRs leave the registers as aligned dword ORs (render_kernels.hpp word_sgr).  Pixels whose channels
    have one, two and three decimal digits in every combination, odd widths so that tokens start at every byte alignment;
    with and without runs (cells without SGRs, lone half blocks, repeat counts between word-built SGRs), a batch of frames
    per launch, every rows / stream geometry of the build"""
    vals = np.array([0, 5, 9, 10, 55, 99, 100, 200, 255], np.uint8)
    rng = np.random.default_rng(11)

    def image(w, h, rep):
        a = vals[rng.integers(0, len(vals), (h, (w + rep - 1) // rep, 3))]
        return np.ascontiguousarray(np.repeat(a, rep, axis=1)[:, :w])

    for (mode, variants, rows_per_cell) in ((MODE_HB_TRUE, (24, 25, 26, 4), 2), (MODE_TRUE_FG, (16, 17, 18, 4), 1)):
        for (W, H) in ((97, 7), (200, 9), (61, 5)):
            imgs = [image(W, H * rows_per_cell, rep) for rep in (1, 1, 2, 5, 1, 3)]
            exp = [oracle_convert(im, mode, W, H, orc.PALETTE_STANDARD) for im in imgs]
            for variant in variants:
                if not geometry_built(gpu[0], variant):
                    continue
                assert render_batch(gpu, mode, imgs, W, H, variant=variant) == exp, (MODE_NAMES[mode], variant, W, H)
        imgs = [image(120, 8 * rows_per_cell, rep) for rep in (1, 2)]  # aspect + padding: pad cells in front of every row
        exp = [oracle_convert(im, mode, 120, 8, orc.PALETTE_STANDARD, True, True) for im in imgs]
        assert render_batch(gpu, mode, imgs, 120, 8, wants_padding=True, use_aspect=True) == exp


def test_every_geometry_renders_a_batch_to_the_same_bytes(gpu):
    """The geometry policy's audit (scripts/gpu_policy_audit.py) as a test: a plan's bytes must not depend on the
    geometry it takes.  Batches of 16 and 200 frames at two terminal sizes away from the BASELINE shapes, four modes: every
    geometry that can carry the plan -- whole frames on the stream / rows / phase kernels, row bands, the shared-out form --
    against the automatic choice, whose first and last frames are checked against the oracle."""
    pkg, torch = gpu
    n_max = 200
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    src = torch.randint(0, 256, (n_max, 270, 480, 3), dtype=torch.uint8, device="cuda", generator=g)
    host = {k: np.ascontiguousarray(src[k].cpu().numpy()) for k in (0, 15, n_max - 1)}
    st = torch.cuda.current_stream().cuda_stream
    for (mode, cl, rm) in ((0, 0, 0), (MODE_256_FG, 2, 0), (MODE_TRUE_FG, 3, 0), (MODE_HB_TRUE, 3, 2)):
        cell = mode in (MODE_256_FG, MODE_TRUE_FG)
        forced = ([(16, -1), (17, -1), (18, 0), (19, -1)] if cell else [(25, -1), (24, -1)]) + [(4, -1), (4, 0), (0, -1), (4, 3)]
        if mode not in (MODE_HB_TRUE,):
            forced += [(1, -1), (1, 0)]
        forced = [(v, sp) for (v, sp) in forced if geometry_built(pkg, v)]
        for (W, H) in ((120, 40), (200, 60)):
            for n in (16, n_max):
                frames = [pkg.frame_setup(src[k].data_ptr(), 480, 270, W, H, rm, False, False, False) for k in range(n)]
                ref = None
                for (variant, split) in [(-1, None)] + forced:
                    plan = pkg.Plan(mode, orc.PALETTE_STANDARD, frames)
                    try:
                        if split is not None:
                            plan.set_split(split)
                        if variant >= 0:
                            plan.set_variant(variant)
                    except RuntimeError:  # this geometry cannot carry the plan
                        plan.close()
                        continue
                    out = torch.full((n * plan.stride,), 0xEE, dtype=torch.uint8, device="cuda")
                    ln = torch.zeros(n, dtype=torch.int32, device="cuda")
                    plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), st)
                    torch.cuda.synchronize()
                    lens = ln.cpu().numpy().astype(np.uint32)
                    assert (lens < 0xFFFFFFF0).all(), (mode, W, H, n, variant, split)
                    view = out.view(n, plan.stride)
                    if ref is None:
                        for k in (0, n - 1):
                            want = orc.convert_with_caps(host[k], W, H, cl, rm, False, False, False)
                            assert view[k, :int(lens[k])].cpu().numpy().tobytes() == want, (mode, W, H, n, "automatic", k)
                        ref = (view.clone(), lens)
                    else:
                        assert (lens == ref[1]).all(), (mode, W, H, n, variant, split)
                        m = int(lens.max())
                        valid = torch.arange(m, device="cuda")[None, :] < torch.from_numpy(lens.astype(np.int64)).cuda()[:, None]
                        assert bool(((view[:, :m] == ref[0][:, :m]) | ~valid).all()), (mode, W, H, n, variant, split)
                    plan.close()


def test_wire_stage_forms_agree_away_from_the_baseline_shapes(gpu):
    """The send side's audit (scripts/gpu_wire_audit.py) as a test.  (1) Small and mid batches of frames larger
    than 80x24 -- where the plain render is shared out over workgroups or cut into bands and the checksum pass meets frames
    above 128 KB: plan_render_packets fused / stand-alone / the plan's choice and plan_render_packets_packed in its three
    forms leave the oracle's checksums, headers, packet CRCs and frames.  (2) The stand-alone pass on both sides of its cost
    model (achip_crc_parts): the one-workgroup kernel on buffers above 128 KB and the spans + the whole-wave finish kernel,
    lengths that are not multiples of 16, error lengths."""
    pkg, torch = gpu
    L = pkg.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    src = torch.randint(0, 256, (16, 270, 480, 3), dtype=torch.uint8, device="cuda", generator=g)
    host = [np.ascontiguousarray(src[k].cpu().numpy()) for k in range(16)]
    for (mode, cl, rm, W, H, n) in ((MODE_TRUE_FG, 3, 0, 320, 90, 1), (MODE_TRUE_FG, 3, 0, 200, 60, 16), (MODE_256_FG, 2, 0, 160, 45, 3),
                                    (MODE_HB_TRUE, 3, 2, 200, 60, 8), (0, 0, 0, 320, 90, 2), (MODE_TRUE_FG, 3, 0, 120, 40, 16)):
        frames = [pkg.frame_setup(src[k].data_ptr(), 480, 270, W, H, rm, False, False, False) for k in range(n)]
        want = [orc.convert_with_caps(host[k], W, H, cl, rm, False, False, False) for k in range(n)]
        plan = pkg.Plan(mode, orc.PALETTE_STANDARD, frames)
        stride = plan.stride
        dims = torch.tensor([[W, H]] * n, dtype=torch.int32, device="cuda")
        for fused in (-1, 0, 1):
            for exact in (None, -1, 0, 1):
                plan.set_fused_crc(fused)
                slab = torch.full((n * stride,), 0xEE, dtype=torch.uint8, device="cuda")
                ln = torch.zeros(n, dtype=torch.int32, device="cuda")
                crc = torch.zeros(n, dtype=torch.int32, device="cuda")
                hdr = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
                pkt = torch.zeros(n, dtype=torch.int32, device="cuda")
                if exact is None:
                    plan.render_packets(slab.data_ptr(), stride, ln.data_ptr(), dims.data_ptr(), crc.data_ptr(), hdr.data_ptr(), pkt.data_ptr(), st)
                else:
                    plan.set_exact_length(exact)
                    dst = torch.full((n * stride,), 0xEE, dtype=torch.uint8, device="cuda")
                    off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
                    lo = torch.zeros(n, dtype=torch.int32, device="cuda")
                    plan.render_packets_packed(slab.data_ptr(), stride, ln.data_ptr(), dims.data_ptr(), crc.data_ptr(), hdr.data_ptr(),
                                               pkt.data_ptr(), dst.data_ptr(), n * stride, off.data_ptr(), lo.data_ptr(), st)
                torch.cuda.synchronize()
                lens = ln.cpu().numpy().astype(np.uint32)
                crc_h, pkt_h, hdr_h = crc.cpu().numpy().astype(np.uint32), pkt.cpu().numpy().astype(np.uint32), hdr.cpu().numpy()
                if exact is not None:
                    o, l, d = off.cpu().numpy(), lo.cpu().numpy().astype(np.uint32), dst.cpu().numpy()
                for k in range(n):
                    tag = (mode, W, H, n, fused, exact, k)
                    assert int(lens[k]) == len(want[k]), tag
                    eh, ep = orc.ascii_frame_packet(want[k], W, H)
                    assert int(crc_h[k]) == orc.crc32c(want[k]) and hdr_h[24 * k:24 * k + 24].tobytes() == eh and int(pkt_h[k]) == ep, tag
                    if exact is not None:
                        assert int(l[k]) == len(want[k]) and d[int(o[k]):int(o[k]) + len(want[k])].tobytes() == want[k], tag
        plan.close()
    L.achip_crc_parts.restype = C.c_int
    L.achip_crc_parts.argtypes = [C.c_uint32, C.c_int]
    seen = set()
    for (nbytes, nb) in ((277000 - 3, 16), (277000 - 3, 128), (663936 - 7, 1), (540000 + 1, 40), (1 << 20, 3), (131072 + 16, 2), (131072 - 1, 5)):
        stride = (nbytes + 127) // 128 * 128
        buf = torch.randint(0, 256, (nb * stride,), dtype=torch.uint8, device="cuda", generator=g)
        lens_h = np.full(nb, nbytes, dtype=np.uint32)
        lens_h[nb // 2] = nbytes // 3 + 5          # a short buffer among long ones
        if nb > 2:
            lens_h[nb - 1] = 0xFFFFFFF7            # an error length: CRC 0, a header of zeros
        ln = torch.tensor(lens_h.astype(np.int64).tolist(), dtype=torch.int64, device="cuda").to(torch.int32)  # (error codes wrap to negative ints)
        dims = torch.tensor([[97, 31]] * nb, dtype=torch.int32, device="cuda")
        crc = torch.zeros(nb, dtype=torch.int32, device="cuda")
        hdr = torch.full((nb * 24,), 0xEE, dtype=torch.uint8, device="cuda")
        pkt = torch.zeros(nb, dtype=torch.int32, device="cuda")
        rc = L.asciichat_hip_frame_packets(C.c_void_p(buf.data_ptr()), C.c_size_t(stride), C.c_void_p(ln.data_ptr()), C.c_uint32(stride), nb,
                                           C.c_void_p(dims.data_ptr()), C.c_void_p(crc.data_ptr()), C.c_void_p(hdr.data_ptr()), C.c_void_p(pkt.data_ptr()),
                                           C.c_void_p(st))
        assert rc == 0, pkg.last_error()
        torch.cuda.synchronize()
        seen.add(L.achip_crc_parts(stride, nb) == 1)
        hb, crc_h, pkt_h, hdr_h = buf.cpu().numpy(), crc.cpu().numpy().astype(np.uint32), pkt.cpu().numpy().astype(np.uint32), hdr.cpu().numpy()
        for k in range(nb):
            if lens_h[k] >= 0xFFFFFFF0:
                eh, ep = orc.ascii_frame_packet(b"", 0, 0)
                assert int(crc_h[k]) == 0 and hdr_h[24 * k:24 * k + 24].tobytes() == eh, (nbytes, nb, k)
                continue
            fr = hb[k * stride:k * stride + int(lens_h[k])].tobytes()
            eh, ep = orc.ascii_frame_packet(fr, 97, 31)
            assert int(crc_h[k]) == orc.crc32c(fr) and hdr_h[24 * k:24 * k + 24].tobytes() == eh and int(pkt_h[k]) == ep, (nbytes, nb, k)
    assert seen == {True, False} or os.environ.get("ASCIICHAT_HIP_CRC_FRAME_MAX")  # both kernels ran


def test_slots_at_every_line_phase(gpu):
    """The drains map lanes to 16-byte groups from a 128-byte line boundary of the slot's ADDRESS (round 4: whole lines per
    store instruction; the rows kernel carries partial lines from slice to slice, the phase kernel's carry is moved by the
    thread that drained group 0).  Slabs that start at each 16-byte phase of a line, strides that walk the slots through
    the other phases, every kernel family: the oracle's bytes, the NUL behind each frame, and nothing else written."""
    pkg, torch = gpu
    imgs = [orc.frame_hash_noise(120, 90, i) for i in range(3)] + [TORTURE]
    dev = [torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in imgs]
    st = torch.cuda.current_stream().cuda_stream
    for (mode, variant, dims) in [(MODE_TRUE_FG, 17, [(80, 24), (97, 31), (3, 2), (200, 60)]),
                                  (MODE_256_FG, 16, [(80, 24), (61, 7), (1, 1), (130, 9)]),
                                  (MODE_HB_TRUE, 25, [(80, 24), (60, 7), (33, 40), (100, 50)]),
                                  (MODE_HB_TRUE, 24, [(440, 3), (97, 31), (5, 60), (400, 120)]),
                                  (0, 25, [(100, 9), (37, 11), (128, 4), (80, 24)]),
                                  (MODE_HB_TRUE, 4, [(80, 24), (97, 31), (460, 5), (10, 150)]),
                                  (0, 1, [(80, 24), (97, 31), (300, 20), (1, 130)]),
                                  (MODE_TRUE_FG, 2, [(80, 24), (130, 1), (64, 65), (200, 60)])]:
        rm = MODE_CAPS.get(mode, (3, 0))[1]
        if not geometry_built(pkg, variant):  # (frame geometry 2 in a default build: its 512-thread sibling)
            variant = 1
        frames = [pkg.frame_setup(d.data_ptr(), i.shape[1], i.shape[0], w, h, rm, False, False, False)
                  for i, d, (w, h) in zip(imgs, dev, dims)]
        want = [oracle_convert(i, mode, w, h, orc.PALETTE_STANDARD) for i, (w, h) in zip(imgs, dims)]
        plan = pkg.Plan(mode, orc.PALETTE_STANDARD, frames)
        plan.set_variant(variant)
        n = len(frames)
        ln = torch.zeros(n, dtype=torch.int32, device="cuda")
        for phase in range(8):
            stride = plan.stride + 16 * (1 + 2 * phase)  # odd multiples of 16: slot k sits at phase + k * (1 + 2 phase) mod 8
            out = torch.full((n * stride + 512,), 0xEE, dtype=torch.uint8, device="cuda")
            lead = (-out.data_ptr()) % 128 + 128 + 16 * phase
            plan.render(out.data_ptr() + lead, stride, ln.data_ptr(), st)
            torch.cuda.synchronize()
            host, lens = out.cpu().numpy(), ln.cpu().numpy().astype(np.uint32)
            assert (host[:lead] == 0xEE).all(), (mode, variant, phase)
            for k in range(n):
                o = lead + k * stride
                assert int(lens[k]) == len(want[k]) and host[o:o + len(want[k])].tobytes() == want[k], (mode, variant, phase, k)
                assert host[o + len(want[k])] == 0 and (host[o + len(want[k]) + 1:o + stride] == 0xEE).all(), (mode, variant, phase, k)
            assert (host[lead + n * stride:] == 0xEE).all(), (mode, variant, phase)
        plan.close()


@pytest.mark.parametrize("mode", [m for m in ALL_MODES if m != MODE_TRUE_BG],
                         ids=[MODE_NAMES[m] for m in ALL_MODES if m != MODE_TRUE_BG])
def test_aspect_and_padding(gpu, mode):
    for (W, H) in [(80, 24), (97, 31), (60, 40), (300, 20)]:
        got = render_batch(gpu, mode, [TORTURE], W, H, wants_padding=True, use_aspect=True)[0]
        assert got == oracle_convert(TORTURE, mode, W, H, orc.PALETTE_STANDARD, True, True), (MODE_NAMES[mode], W, H)


SPLITTABLE = [m for m in ALL_MODES if m != MODE_16_DITHER_BG]


@pytest.mark.parametrize("mode", SPLITTABLE, ids=[MODE_NAMES[m] for m in SPLITTABLE])
def test_multi_workgroup_frames(gpu, mode):
    """A frame cut into row bands rendered by several workgroups (device-side length hand-off): same bytes."""
    for (W, H, split, pad, n) in [(80, 24, 0, False, 1), (80, 24, 1, True, 3), (97, 31, 4, True, 2), (200, 60, 3, False, 5),
                                  (400, 120, 0, False, 2), (80, 24, 0, False, 40)]:
        aspect = pad and mode != MODE_TRUE_BG
        imgs = [TORTURE] + [orc.frame_hash_noise(160, 120, 7 + k) for k in range(n - 1)]
        got = render_batch(gpu, mode, imgs, W, H, wants_padding=pad, use_aspect=aspect, split=split, repeat=3)
        for k, img in enumerate(imgs):
            assert got[k] == oracle_convert(img, mode, W, H, orc.PALETTE_STANDARD, pad, aspect), (MODE_NAMES[mode], W, H, k)
    # policy.  Run-structured modes: one 160x48 frame -> 48 single-row workgroups of the phase kernel, and never a frame small
    # enough for one block per wave of the rows kernel (80x24: profiles/r04_small_batch_variants.txt) unless asked to.
    # Per-cell modes: a small launch shares the frame's blocks out over four-wave workgroups of the stream kernel, one block
    # per wave (PARTS, profiles/r04_small_batch_parts.txt).  Never when asked not to; row bands when asked for rows.
    cell = mode in (MODE_TRUE_FG, MODE_256_FG, MODE_16_FG, MODE_TRUE_BG)
    per_block = 127 if mode == MODE_TRUE_FG else 128  # truecolor-fg blocks carry a ghost slot

    def shared_out(cells):  # four blocks per workgroup, but up to sixteen workgroups for a small frame
        blocks = -(-cells // per_block)
        return max(-(-blocks // 4), min(blocks, 16))

    # (round 6, visit Q: rows of 129-512 cells of the short-token modes and of truecolor half blocks are cut into segments of at
    # most 128 cells, whole rows per four-wave workgroup of the rows kernel -- geometry 32, 24 workgroups of two rows here; the
    # 256- / 16-colour half blocks keep their 48 one-row bands)
    render_batch(gpu, mode, [TORTURE], 160, 48, split=0, want_parts=shared_out(160 * 48) if cell else 24 if mode in (0, 5, 8) else 48)
    render_batch(gpu, mode, [TORTURE], 160, 48, split=-1, want_parts=1)
    # (round 6: a lone 80x24 frame of a run-structured mode is 24 one-row blocks of the rows kernel shared out over six
    # four-wave workgroups -- render_rows.hpp PARTS, profiles/r06_small_rows_parts.txt; until round 5 whole on geometry 25 in
    # the short-token modes, 24 one-row bands of the phase kernel in the coloured half-block modes)
    render_batch(gpu, mode, [TORTURE], 80, 24, split=0, want_parts=shared_out(80 * 24) if cell else 6)
    render_batch(gpu, mode, [TORTURE], 80, 24, split=2, want_parts=12)


@pytest.mark.parametrize("mode", [MODE_TRUE_FG, MODE_256_FG, MODE_16_FG, MODE_TRUE_BG, 0, 5, 6, 7, 8],
                         ids=["true_fg", "256_fg", "16_fg", "true_bg", "mono", "hb_true", "hb_256", "hb_16", "hb_mono"])
def test_small_launches_share_frames_out_over_workgroups(gpu, mode):
    """PARTS instantiations of the stream kernel (geometry 18) and, round 6, of the rows kernel (geometry 31): what small
    launches of the per-cell / the run-structured modes take by themselves.  Lone frames, small and ragged batches, padding,
    three launches on the same hand-off words (a new epoch each), a range of the plan's frames, and the wire-stage entry
    points of such a plan (which launch whole frames instead)."""
    pkg, torch = gpu
    per_block = 127 if mode == MODE_TRUE_FG else 128
    run_mode = mode in (0, 5, 6, 7, 8)
    rm = MODE_CAPS[mode][1] if run_mode else 0
    for (W, H, n, pad) in [(80, 24, 1, False), (80, 24, 9, False), (160, 48, 9, False), (97, 31, 5, True), (80, 24, 64, False),
                           (200, 60, 2, False), (300, 20, 3, False), (512, 9, 1, False), (257, 13, 2, True), (129, 50, 1, False)]:
        aspect = pad and mode != MODE_TRUE_BG
        imgs = [TORTURE] + [orc.frame_hash_noise(160, 120, 70 + k) for k in range(n - 1)]
        got = render_batch(gpu, mode, imgs, W, H, wants_padding=pad, use_aspect=aspect, repeat=3)
        for k, img in enumerate(imgs):
            assert got[k] == oracle_convert(img, mode, W, H, orc.PALETTE_STANDARD, pad, aspect), (MODE_NAMES[mode], W, H, n, k)
    # the geometry really is the shared-out one
    dev = torch.from_numpy(np.ascontiguousarray(TORTURE)).cuda()
    fr = [pkg.frame_setup(dev.data_ptr(), TORTURE.shape[1], TORTURE.shape[0], 80, 24, rm, False, False, False) for _ in range(9)]
    plan = pkg.Plan(mode, orc.PALETTE_STANDARD, fr)
    # (run-structured modes: 24 one-row blocks, one per wave of six four-wave workgroups)
    assert (plan.variant, plan.parts) == ((31, 6) if run_mode else (18, min(-(-1920 // per_block), 16)))
    if run_mode:  # rows of 129-512 cells: segments of at most 128 cells, whole rows per workgroup (geometry 32 = WIDE + PARTS)
        mid = pkg.Plan(mode, orc.PALETTE_STANDARD, [pkg.frame_setup(dev.data_ptr(), TORTURE.shape[1], TORTURE.shape[0], 160, 48, rm, False, False, False)] * 2)
        assert (mid.variant, mid.parts) == ((32, 24) if mode in (0, 5, 8) else (4, 48))  # (256 / 16 colours: row bands, measured)
        mid.close()
    want = oracle_convert(TORTURE, mode, 80, 24, orc.PALETTE_STANDARD)
    # frames [2, 7) only
    out = torch.full((9 * plan.stride,), 0xEE, dtype=torch.uint8, device="cuda")
    ln = torch.full((9,), -7, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), st, first=2, count=5)
    torch.cuda.synchronize()
    host, lens = out.cpu().numpy(), ln.cpu().numpy()
    for k in range(5):
        assert int(lens[k]) == len(want) and host[k * plan.stride:k * plan.stride + len(want)].tobytes() == want, k
    assert (lens[5:] == -7).all() and (host[5 * plan.stride:] == 0xEE).all()
    # the frame checksum of such a plan: ONE whole-frame launch with the checksum riding it (not render + a second pass)
    if mode == MODE_TRUE_FG:
        assert plan.fused_crc
    crc = torch.zeros(9, dtype=torch.int32, device="cuda")
    plan.render_crc(out.data_ptr(), plan.stride, ln.data_ptr(), crc.data_ptr(), st)
    torch.cuda.synchronize()
    assert (crc.cpu().numpy().astype(np.uint32) == orc.crc32c(want)).all()
    assert (ln.cpu().numpy() == len(want)).all()
    plan.close()


def test_graph_replay_of_small_plans_captures_whole_frames(gpu):
    """asciichat_hip_schedule_*: a captured launch cannot carry the per-launch epoch of frames shared out over workgroups, so
    small plans of the per-cell modes (stream geometry 18) and -- round 6 -- of the run-structured modes (rows geometries 31 / 32)
    are captured in their whole-frame geometry; row-band plans are refused as before."""
    pkg, torch = gpu
    imgs = [orc.frame_hash_noise(160, 120, 300 + k) for k in range(4)]
    dev = [torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in imgs]
    for mode, rm, shared, W, H in ((MODE_TRUE_FG, 0, 18, 80, 24), (0, 0, 31, 80, 24), (MODE_HB_TRUE, 2, 31, 80, 24), (0, 0, 32, 160, 48), (MODE_HB_TRUE, 2, 32, 300, 20)):
        plans = []
        for k in range(2):
            fr = [pkg.frame_setup(dev[(k + j) % 4].data_ptr(), 160, 120, W, H, rm, False, False, False) for j in range(3)]
            plans.append(pkg.Plan(mode, orc.PALETTE_STANDARD, fr))
        assert all(p.parts > 1 and p.variant == shared for p in plans), (mode, [(p.variant, p.parts) for p in plans])
        stride = plans[0].stride
        out = [torch.full((3 * stride,), 0xEE, dtype=torch.uint8, device="cuda") for _ in range(2)]
        ln = [torch.zeros(3, dtype=torch.int32, device="cuda") for _ in range(2)]
        lanes = [torch.cuda.current_stream(), torch.cuda.Stream()]
        sched = pkg.Schedule(plans, [o.data_ptr() for o in out], [x.data_ptr() for x in ln], stride, [s.cuda_stream for s in lanes])
        for _ in range(3):  # replays of ONE captured graph
            for o in out:
                o.fill_(0xEE)
            torch.cuda.synchronize()
            sched.replay(0, 2, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            for k in range(2):
                host, lens = out[k].cpu().numpy(), ln[k].cpu().numpy()
                for j in range(3):
                    want = oracle_convert(imgs[(k + j) % 4], mode, W, H, orc.PALETTE_STANDARD)
                    assert int(lens[j]) == len(want) and host[j * stride:j * stride + len(want)].tobytes() == want, (mode, k, j)
        sched.close()
        for p in plans:
            p.close()
    # a row-band plan (one 256-colour half-block frame of rows beyond the shared-out rows geometry's 128 cells) still cannot be captured
    f = pkg.frame_setup(dev[0].data_ptr(), 160, 120, 160, 48, 2, False, False, False)
    band = pkg.Plan(MODE_HB_256, orc.PALETTE_STANDARD, [f])
    assert band.parts > 1 and band.variant < 16
    o = torch.zeros(band.stride, dtype=torch.uint8, device="cuda")
    x = torch.zeros(1, dtype=torch.int32, device="cuda")
    sched = pkg.Schedule([band], [o.data_ptr()], [x.data_ptr()], band.stride, [torch.cuda.current_stream().cuda_stream])
    with pytest.raises(RuntimeError):
        sched.replay(0, 1, torch.cuda.current_stream().cuda_stream)
    sched.close()
    band.close()


def test_split_exclusions_and_ragged_parts(gpu):
    # multi-byte glyphs in truecolor-fg and the serial dither stay whole-frame
    render_batch(gpu, 1, [TORTURE], 80, 24, palette=orc.PALETTE_BLOCKS, split=0, want_parts=1)
    render_batch(gpu, MODE_16_DITHER_BG, [TORTURE], 80, 24, split=0, want_parts=1)
    # but a multi-byte palette splits fine in the other modes
    for mode in (0, 2, 3):
        got = render_batch(gpu, mode, [TORTURE], 80, 24, palette=orc.PALETTE_BLOCKS, split=5)
        assert got[0] == oracle_convert(TORTURE, mode, 80, 24, orc.PALETTE_BLOCKS)
    # ragged batch: frames with fewer rows than the tallest leave their later workgroups idle
    imgs = [orc.frame_hash_noise(120, 90, i) for i in range(4)]
    dims = [(80, 24), (60, 7), (33, 40), (80, 1)]
    for mode in (1, 2, 5):
        got = render_batch(gpu, mode, imgs, 0, 0, dims=dims, split=4, repeat=2)
        for k, (im, (w, h)) in enumerate(zip(imgs, dims)):
            assert got[k] == oracle_convert(im, mode, w, h, orc.PALETTE_STANDARD), (mode, k)


@pytest.mark.parametrize("palette", [orc.PALETTE_BLOCKS, orc.PALETTE_COOL, orc.PALETTE_DIGITAL, orc.PALETTE_MINIMAL,
                                     "ab", "x", "é漢😀 ."], ids=["blocks", "cool", "digital", "minimal", "ab", "x", "mixed"])
def test_palettes(gpu, palette):
    for mode in (0, 1, 2, 3, 4):
        got = render_batch(gpu, mode, [TORTURE], 61, 17, palette)[0]
        assert got == oracle_convert(TORTURE, mode, 61, 17, palette), MODE_NAMES[mode]


def test_survey_anchor_through_gpu(gpu):
    """The SURVEY 8(c) reference-output anchors, reproduced by the GPU path itself."""
    g = orc.frame_anchor_gradient()
    for mode, (length, fnv) in {2: (22255, 0xBE60A438), 1: (35852, 0x885DA51D), 5: (73802, 0x362719AD)}.items():
        got = render_batch(gpu, mode, [g], 80, 24)[0]
        assert len(got) == length and orc.fnv1a32(got) == fnv
    got = render_batch(gpu, 0, [g], 80, 24, wants_padding=True, use_aspect=True)[0]
    assert len(got) == 1721 and orc.fnv1a32(got) == 0x7D62F78F


def test_synthetic_inputs_mixed_batch(gpu):
    """S-noise / S-smooth / S-bars / S-gray in one ragged batch: different source sizes per frame."""
    imgs = [orc.frame_noise(320, 240, 12345), orc.frame_smooth(640, 480), orc.frame_bars(480, 270, 4),
            orc.frame_gray(200, 100), orc.frame_bars(333, 77, 1), np.zeros((50, 70, 3), np.uint8),
            np.full((9, 9, 3), 255, np.uint8), orc.frame_noise(17, 5, 99)]
    for mode in ALL_MODES:
        got = render_batch(gpu, mode, imgs, 80, 24)
        for k, im in enumerate(imgs):
            assert got[k] == oracle_convert(im, mode, 80, 24, orc.PALETTE_STANDARD), (MODE_NAMES[mode], k)


def test_ragged_output_sizes_and_edges(gpu):
    """Per-frame output dims differ inside one batch; 1x1 outputs; a row as wide as the largest chunk."""
    img = orc.frame_hash_noise(97, 41, 3)
    dims = [(1, 1), (2, 1), (1, 2), (80, 24), (3, 50), (7, 200), (500, 3), (1024, 2), (2048, 2), (3840, 2)]
    for mode in (0, 1, 5, 6, 8, 9):
        got = render_batch(gpu, mode, [img] * len(dims), 0, 0, dims=dims)
        for k, (w, h) in enumerate(dims):
            exp = oracle_convert(img, mode, w, h, orc.PALETTE_STANDARD)
            if exp is None:  # resized image would exceed 3840x2160: the reference returns NULL
                continue
            assert got[k] == exp, (MODE_NAMES[mode], w, h)


def test_overflow_and_bad_descriptor_are_reported(gpu):
    pkg, torch = gpu
    img = torch.from_numpy(orc.frame_noise(64, 64, 5)).cuda()
    f = pkg.frame_setup(img.data_ptr(), 64, 64, 40, 12, 0)
    plan = pkg.Plan(1, orc.PALETTE_STANDARD, [f])
    out = torch.zeros(plan.stride, dtype=torch.uint8, device="cuda")
    ln = torch.zeros(1, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError):  # stride below the plan's bound is refused on the host
        plan.render(out.data_ptr(), 64, ln.data_ptr())
    with pytest.raises(RuntimeError):  # unaligned slab
        plan.render(out.data_ptr() + 4, plan.stride, ln.data_ptr())
    plan.close()
    bad = pkg.Frame()
    with pytest.raises(RuntimeError):
        pkg.Plan(1, orc.PALETTE_STANDARD, [bad])
    with pytest.raises(RuntimeError):
        pkg.Plan(1, "", [f])


# ------------------------------------------------------------------------------------------------
# BASELINE.json full-size configurations
# ------------------------------------------------------------------------------------------------
FULL = [
    ("K2 1080p->80x24 ANSI-256 b256", 1920, 1080, 80, 24, 2, 256, 6),
    ("K2' 1080p->80x24 truecolor b256", 1920, 1080, 80, 24, 1, 256, 6),
    ("K3 4K->200x60 truecolor b256", 3840, 2160, 200, 60, 1, 256, 4),
    ("K5 4K->400x120 half-block truecolor b256", 3840, 2160, 400, 120, 5, 256, 3),  # BASELINE configs[4] at the full batch
]


@pytest.mark.parametrize("name,sw,sh,W,H,mode,batch,nsample", FULL, ids=[f[0].split()[0] for f in FULL])
def test_full_size_batches(gpu, name, sw, sh, W, H, mode, batch, nsample):
    pkg, torch = gpu
    g = torch.Generator(device="cuda")
    g.manual_seed(42)
    frames_t = torch.randint(0, 256, (batch, sh, sw, 3), dtype=torch.uint8, device="cuda", generator=g)
    # make some frames run-heavy / transparent-heavy instead of pure noise
    frames_t[1] = torch.from_numpy(orc.frame_bars(sw, sh, 6)).cuda()
    frames_t[2] = torch.from_numpy(orc.frame_smooth(sw, sh)).cuda()
    frames_t[3] = 0
    frames_t[4] = frames_t[0]  # duplicate input -> identical output (determinism / no cross-frame leakage)
    rm = MODE_CAPS[mode][1]
    descs = [pkg.frame_setup(frames_t.data_ptr() + i * sh * sw * 3, sw, sh, W, H, rm) for i in range(batch)]
    plan = pkg.Plan(mode, orc.PALETTE_STANDARD, descs)
    out = torch.zeros(batch * plan.stride, dtype=torch.uint8, device="cuda")
    ln = torch.zeros(batch, dtype=torch.int32, device="cuda")
    plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    lens = ln.cpu().numpy().astype(np.uint32)
    assert (lens < 0xFFFFFFF0).all()
    rows = H
    check = sorted(set([0, 1, 2, 3, 4, batch - 1] + list(range(5, 5 + nsample))))
    outs = {}
    for k in check:
        got = out[k * plan.stride:k * plan.stride + int(lens[k])].cpu().numpy().tobytes()
        outs[k] = got
        img = frames_t[k].cpu().numpy()
        assert got == oracle_convert(img, mode, W, H, orc.PALETTE_STANDARD), (name, k)
    assert outs[0] == outs[4]
    # size-independent properties on EVERY frame of the batch
    host = out.cpu().numpy().reshape(batch, plan.stride)
    for k in range(batch):
        fr = host[k, :int(lens[k])]
        assert int((fr == 10).sum()) == rows - 1, (name, k)          # one newline between text rows, none after
        assert fr[-4:].tobytes() == b"\033[0m", (name, k)              # colour modes end in a reset
        assert host[k, int(lens[k])] == 0
    # re-render is idempotent
    out2 = torch.zeros_like(out)
    plan.render(out2.data_ptr(), plan.stride, ln.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    plan.close()


OTHER_FULL = [(0, "mono"), (3, "16_fg"), (6, "hb_256"), (7, "hb_16"), (8, "hb_mono"), (MODE_16_DITHER_BG, "dither16_bg")]


@pytest.mark.parametrize("mode,name", OTHER_FULL, ids=[m[1] for m in OTHER_FULL])
def test_full_size_batches_other_renderers(gpu, mode, name):
    """The renderers of SURVEY 8(a) that no BASELINE configuration names (PM, P16, H256, H16, HM, PD) at the metric's shape and full
    batch -- 256 x (1080p -> 80x24): oracle compare of a sample (noise, bars, smooth, black, a duplicate), newline count and
    idempotence on every frame.  (bench.py's legs 1080p_80x24_mono / _dither16_bg / _ansi16 / _halfblock16 are these launches.)"""
    pkg, torch = gpu
    sw, sh, W, H, batch = 1920, 1080, 80, 24, 256
    g = torch.Generator(device="cuda")
    g.manual_seed(44 + mode)
    frames_t = torch.randint(0, 256, (batch, sh, sw, 3), dtype=torch.uint8, device="cuda", generator=g)
    frames_t[1] = torch.from_numpy(orc.frame_bars(sw, sh, 6)).cuda()
    frames_t[2] = torch.from_numpy(orc.frame_smooth(sw, sh)).cuda()
    frames_t[3] = 0
    frames_t[4] = frames_t[0]
    frames_t[5] = torch.from_numpy(orc.frame_gray(sw, sh)).cuda() if hasattr(orc, "frame_gray") else frames_t[2]
    rm = MODE_CAPS[mode][1]
    descs = [pkg.frame_setup(frames_t.data_ptr() + i * sh * sw * 3, sw, sh, W, H, rm) for i in range(batch)]
    plan = pkg.Plan(mode, orc.PALETTE_STANDARD, descs)
    out = torch.zeros(batch * plan.stride, dtype=torch.uint8, device="cuda")
    ln = torch.zeros(batch, dtype=torch.int32, device="cuda")
    plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    lens = ln.cpu().numpy().astype(np.uint32)
    assert (lens < 0xFFFFFFF0).all()
    outs = {}
    for k in (0, 1, 2, 3, 4, 5, 6, 7, batch - 1):
        got = out[k * plan.stride:k * plan.stride + int(lens[k])].cpu().numpy().tobytes()
        outs[k] = got
        assert got == oracle_convert(frames_t[k].cpu().numpy(), mode, W, H, orc.PALETTE_STANDARD), (name, k)
    assert outs[0] == outs[4]
    host = out.cpu().numpy().reshape(batch, plan.stride)
    for k in range(batch):
        fr = host[k, :int(lens[k])]
        assert int((fr == 10).sum()) == H - 1, (name, k)
        assert host[k, int(lens[k])] == 0
    out2 = torch.zeros_like(out)
    plan.render(out2.data_ptr(), plan.stride, ln.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    plan.close()


U8_FULL = [
    ("K2' 1080p->80x24 truecolor b256", 1920, 1080, 80, 24, 256, 5),
    ("K3 4K->200x60 truecolor b256", 3840, 2160, 200, 60, 256, 3),
]


@pytest.mark.parametrize("palette", [orc.PALETTE_BLOCKS, orc.PALETTE_DIGITAL, orc.PALETTE_COOL], ids=["blocks", "digital", "cool"])
@pytest.mark.parametrize("name,sw,sh,W,H,batch,nsample", U8_FULL, ids=[f[0].split()[0] for f in U8_FULL])
def test_full_size_batches_multibyte_palettes(gpu, palette, name, sw, sh, W, H, batch, nsample):
    """truecolor foreground with the reference's multi-byte built-in palettes (palette.h:161-197;
    foreground.c:281-296) at K2' / K3 on the stream kernel's instantiation of its own -- no built-in palette takes the phase
    kernel for a whole-frame launch any more.  Oracle compare of a sample, properties on every frame."""
    pkg, torch = gpu
    g = torch.Generator(device="cuda")
    g.manual_seed(43)
    frames_t = torch.randint(0, 256, (batch, sh, sw, 3), dtype=torch.uint8, device="cuda", generator=g)
    frames_t[1] = torch.from_numpy(orc.frame_bars(sw, sh, 6)).cuda()    # flat areas: ASCII cells whose colour returns
    frames_t[2] = torch.from_numpy(orc.frame_smooth(sw, sh)).cuda()     # behind multi-byte stretches
    frames_t[3] = 0                                                     # all spaces: ONE SGR in the frame
    frames_t[4] = 255                                                   # no ASCII cell at all
    frames_t[5] = frames_t[0]
    descs = [pkg.frame_setup(frames_t.data_ptr() + i * sh * sw * 3, sw, sh, W, H, 0) for i in range(batch)]
    plan = pkg.Plan(MODE_TRUE_FG, palette, descs)
    assert 16 <= plan.variant < 24 and plan.parts == 1, (plan.variant, plan.parts)  # the stream kernel, whole frames
    out = torch.zeros(batch * plan.stride, dtype=torch.uint8, device="cuda")
    ln = torch.zeros(batch, dtype=torch.int32, device="cuda")
    plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    lens = ln.cpu().numpy().astype(np.uint32)
    assert (lens < 0xFFFFFFF0).all()
    outs = {}
    for k in sorted(set([0, 1, 2, 3, 4, 5, batch - 1] + list(range(6, 6 + nsample)))):
        got = out[k * plan.stride:k * plan.stride + int(lens[k])].cpu().numpy().tobytes()
        outs[k] = got
        assert got == oracle_convert(frames_t[k].cpu().numpy(), MODE_TRUE_FG, W, H, palette), (name, k)
    assert outs[0] == outs[5]
    assert outs[3].count(b"\033[38;2;") == 1  # black frame: every glyph a space, one colour, one SGR
    host = out.cpu().numpy().reshape(batch, plan.stride)
    for k in range(batch):
        fr = host[k, :int(lens[k])]
        assert int((fr == 10).sum()) == H - 1, (name, k)
        assert fr[-4:].tobytes() == b"\033[0m", (name, k)
        assert host[k, int(lens[k])] == 0
    # the phase kernel (forced: geometry 4) renders the same bytes
    plan.set_variant(4)
    out2 = torch.zeros_like(out)
    ln2 = torch.zeros_like(ln)
    plan.render(out2.data_ptr(), plan.stride, ln2.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(ln, ln2)
    h2 = out2.cpu().numpy().reshape(batch, plan.stride)
    for k in range(batch):
        assert np.array_equal(h2[k, :int(lens[k])], host[k, :int(lens[k])]), (name, k)
    plan.close()
    # a lone frame and a handful (small launches stay whole frames on the stream kernel: the RLE state never crosses workgroups)
    for nb in (1, 9):
        plan = pkg.Plan(MODE_TRUE_FG, palette, descs[:nb])
        assert 16 <= plan.variant < 24 and plan.parts == 1, (plan.variant, plan.parts)
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        l2 = ln.cpu().numpy().astype(np.uint32)
        for k in range(nb):
            got = out[k * plan.stride:k * plan.stride + int(l2[k])].cpu().numpy().tobytes()
            assert got == (outs[k] if k in outs else oracle_convert(frames_t[k].cpu().numpy(), MODE_TRUE_FG, W, H, palette)), (name, nb, k)
        plan.close()


# ------------------------------------------------------------------------------------------------
# pixel-space composite (K4) and stand-alone resize
# ------------------------------------------------------------------------------------------------
def _composite(pkg, torch, imgs, tw, th):
    dev = [torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in imgs]
    n = len(imgs)
    ptrs = (C.c_void_p * n)(*[d.data_ptr() for d in dev])
    ws = (C.c_int * n)(*[i.shape[1] for i in imgs])
    hs = (C.c_int * n)(*[i.shape[0] for i in imgs])
    comp = pkg.Composite()
    pkg.lib().achip_composite_setup(C.byref(comp), ptrs, ws, hs, n, tw, th)
    return comp, dev


def test_grid9_composite_fused_and_materialised(gpu):
    pkg, torch = gpu
    imgs = [orc.frame_hash_noise(1920, 1080, 10 + i) if i % 2 else orc.frame_bars(1920, 1080, i) for i in range(9)]
    comp, keep = _composite(pkg, torch, imgs, 160, 48)
    assert (comp.cols, comp.rows, comp.cell_w, comp.cell_h) == (3, 3, 53, 32)
    ref_canvas = orc.composite(imgs, 160, 48)
    # materialised canvas
    dst = torch.zeros(96 * 160 * 3, dtype=torch.uint8, device="cuda")
    assert pkg.lib().asciichat_hip_composite(C.byref(comp), dst.data_ptr(), None) == 0
    assert np.array_equal(dst.cpu().numpy().reshape(96, 160, 3), ref_canvas)
    # fused: render straight from the nine sources, canvas never built (convert_composite_to_ascii, stream.c:790-854)
    comp_dev = C.c_void_p()
    assert pkg.lib().asciichat_hip_composite_upload(C.byref(comp), C.byref(comp_dev)) == 0
    for mode in (1, 5, 2, 0):
        cl, rm = MODE_CAPS[mode]
        h = 96 if rm == 2 else 48  # stream.c:831
        f = pkg.frame_setup(None, 160, 96, 160, h, rm, True, True, False)
        f.comp = comp_dev.value
        plan = pkg.Plan(mode, orc.PALETTE_STANDARD, [f])
        out = torch.zeros(plan.stride, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(1, dtype=torch.int32, device="cuda")
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr())
        torch.cuda.synchronize()
        got = out[:int(ln[0].item())].cpu().numpy().tobytes()
        exp = orc.convert_with_caps(ref_canvas, 160, h, cl, rm, True, True, False)
        assert got == exp, MODE_NAMES[mode]
        plan.close()
    pkg.lib().asciichat_hip_free(comp_dev)


def test_resize_kernel(gpu):
    pkg, torch = gpu
    img = orc.frame_hash_noise(1920, 1080, 77)
    src = torch.from_numpy(img).cuda()
    for (dw, dh) in [(80, 24), (53, 30), (1, 1), (1920, 1080), (2500, 1400), (3, 2000)]:
        dst = torch.zeros(dh * dw * 3, dtype=torch.uint8, device="cuda")
        assert pkg.lib().asciichat_hip_resize(src.data_ptr(), 1920, 1080, dst.data_ptr(), dw, dh, None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy().reshape(dh, dw, 3), orc.resize_nn(img, dw, dh)), (dw, dh)


def test_grid9_tile_exchange_path_single_rank(gpu):
    """The multi-GPU grid path (tiles resized by their owners, gathered, composite rendered from the tiles) with
    world=1: same code the ranks run over RCCL, minus the collective."""
    pkg, torch = gpu
    imgs = [orc.frame_hash_noise(1920, 1080, 30 + i) if i % 3 else orc.frame_bars(1920, 1080, i) for i in range(9)]
    local = {k: torch.from_numpy(imgs[k]).cuda() for k in range(9)}
    targets = [(3, 0, True), (3, 2, True), (2, 0, False), (0, 0, True)]
    got = pkg.distributed.render_grid_for_targets(torch, None, pkg, local, [(1920, 1080)] * 9, 160, 48, targets,
                                                  orc.PALETTE_STANDARD, 1, 0)
    canvas = orc.composite(imgs, 160, 48)
    for (cl, rm, pad), g in zip(targets, got):
        h = 96 if rm == 2 else 48
        assert g == orc.convert_with_caps(canvas, 160, h, cl, rm, pad, True, False), (cl, rm, pad)


# ------------------------------------------------------------------------------------------------
# wire stage: CRC-32C + ascii_frame_packet_t headers of a rendered slab
# ------------------------------------------------------------------------------------------------
def test_wire_stage_crc_and_headers(gpu):
    pkg, torch = gpu
    L = pkg.lib()
    stream = torch.cuda.current_stream().cuda_stream
    imgs = [TORTURE] + [orc.frame_hash_noise(320, 240, 3 + k) for k in range(6)]
    dims = [(80, 24), (60, 7), (33, 40), (80, 1), (200, 60), (97, 31), (1, 1)]
    for mode in (1, 2, 0, 5):
        rm = MODE_CAPS[mode][1]
        dev = [torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in imgs]
        frames = [pkg.frame_setup(d.data_ptr(), i.shape[1], i.shape[0], w, h, rm) for i, d, (w, h) in zip(imgs, dev, dims)]
        plan = pkg.Plan(mode, orc.PALETTE_STANDARD, frames)
        n = len(imgs)
        out = torch.full((n * plan.stride,), 0xEE, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(n, dtype=torch.int32, device="cuda")
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), stream)
        d32 = torch.tensor(dims, dtype=torch.int32, device="cuda")
        crc = torch.zeros(n, dtype=torch.int32, device="cuda")
        hdr = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
        pkt = torch.zeros(n, dtype=torch.int32, device="cuda")
        rc = L.asciichat_hip_frame_packets(out.data_ptr(), plan.stride, ln.data_ptr(), plan.stride, n, d32.data_ptr(),
                                           crc.data_ptr(), hdr.data_ptr(), pkt.data_ptr(), stream)
        assert rc == 0, pkg.last_error()
        torch.cuda.synchronize()
        host, lens = out.cpu().numpy(), ln.cpu().numpy().astype(np.uint32)
        crc_h, pkt_h, hdr_h = crc.cpu().numpy().astype(np.uint32), pkt.cpu().numpy().astype(np.uint32), hdr.cpu().numpy()
        for k in range(n):
            fr = host[k * plan.stride:k * plan.stride + int(lens[k])].tobytes()
            assert fr == oracle_convert(imgs[k], mode, dims[k][0], dims[k][1], orc.PALETTE_STANDARD), (mode, k)
            eh, ep = orc.ascii_frame_packet(fr, *dims[k])
            assert int(crc_h[k]) == orc.crc32c(fr), (mode, k)
            assert hdr_h[24 * k:24 * k + 24].tobytes() == eh and int(pkt_h[k]) == ep, (mode, k)
        plan.close()
    # large fixed-length buffers (ingest payloads): multi-span path, incl. a length that is not a multiple of 16
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    for nbytes, nb in ((1920 * 1080 * 3, 3), (131073, 2), (640 * 480 * 3 + 5, 2)):
        stride = (nbytes + 15) & ~15
        buf = torch.randint(0, 256, (nb * stride,), dtype=torch.uint8, device="cuda", generator=g)
        crc = torch.zeros(nb, dtype=torch.int32, device="cuda")
        rc = L.asciichat_hip_crc32c(buf.data_ptr(), stride, None, nbytes, nbytes, nb, crc.data_ptr(), stream)
        assert rc == 0, pkg.last_error()
        torch.cuda.synchronize()
        hb = buf.cpu().numpy()
        for k in range(nb):
            assert int(crc.cpu().numpy().astype(np.uint32)[k]) == orc.crc32c(hb[k * stride:k * stride + nbytes].tobytes())
    # bad arguments are refused on the host
    assert L.asciichat_hip_crc32c(buf.data_ptr() + 4, stride, None, 16, 16, 1, crc.data_ptr(), stream) != 0
    assert L.asciichat_hip_crc32c(None, stride, None, 16, 16, 1, crc.data_ptr(), stream) != 0


@pytest.mark.gpu
def test_render_with_fused_frame_crc(gpu):
    """plan_render_crc: the frame CRC-32C rides the stream kernel's drain for whole-frame launches of the per-cell
    modes; any other plan renders, then checksums the slab.  Bytes, checksums, headers and packet CRCs against the
    oracle either way (asciichat_crc32 lib/network/crc32.c:95-190, acip_send_ascii_frame server.c:186-214)."""
    pkg, torch = gpu
    L = pkg.lib()
    stream = torch.cuda.current_stream().cuda_stream
    imgs = [TORTURE] + [orc.frame_hash_noise(320, 240, 3 + k) for k in range(6)]
    dims = [(80, 24), (60, 7), (33, 40), (80, 1), (200, 60), (97, 31), (1, 1)]
    # variant -1: the plan's own choice (row bands for so few frames: not fused); 16 / 17 carry the fused CRC
    for mode, variant, want_fused, pad in ((1, 17, True, False), (2, 17, True, False), (3, 17, True, True), (4, 17, True, False),
                                           (1, 16, True, True), (2, 16, True, False), (4, 16, True, False),
                                           (1, 18, False, False), (1, -1, None, False), (0, -1, False, False),
                                           (5, -1, False, False),
                                           # the rows kernel (run-structured modes, whole frames) carries the fused CRC too
                                           (0, 25, True, False), (5, 25, True, True), (6, 25, True, False), (7, 24, True, True),
                                           (8, 24, True, False), (5, 24, True, False)):
        rm = MODE_CAPS.get(mode, (3, 0))[1]
        dev = [torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in imgs]
        asp = pad and mode != 4
        frames = [pkg.frame_setup(d.data_ptr(), i.shape[1], i.shape[0], w, h, rm, pad and mode != 4, asp, False)
                  for i, d, (w, h) in zip(imgs, dev, dims)]
        plan = pkg.Plan(mode, orc.PALETTE_STANDARD, frames)
        if variant >= 0:
            plan.set_variant(variant)
        if variant >= 24:  # the rows kernel carries the fused CRC (in builds with every geometry) but plans do not pick
            assert not plan.fused_crc  # it by themselves (slower there); asked for in a default build, the stand-alone pass runs
            plan.set_fused_crc(1)
            want_fused = rows_crc_built(pkg)
        if mode == 1 and variant in (16, 17):  # round 6's wire audit: beside the lean loop of truecolor foreground the fused form pays for
            assert not plan.fused_crc           # small frames only (<= 8192 cells; this batch holds a 200x60 frame): asked for here
            plan.set_fused_crc(1)
        if want_fused is not None:
            assert plan.fused_crc == want_fused, (mode, variant, plan.variant)
        n = len(imgs)
        out = torch.full((n * plan.stride,), 0xEE, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(n, dtype=torch.int32, device="cuda")
        crc = torch.full((n,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        for _ in range(2):  # a second launch finds the LDS words of the first one gone, the slab already written
            plan.render_crc(out.data_ptr(), plan.stride, ln.data_ptr(), crc.data_ptr(), stream)
        d32 = torch.tensor(dims, dtype=torch.int32, device="cuda")
        hdr = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
        pkt = torch.zeros(n, dtype=torch.int32, device="cuda")
        assert L.asciichat_hip_packets_from_crc(ln.data_ptr(), crc.data_ptr(), n, d32.data_ptr(), hdr.data_ptr(),
                                                pkt.data_ptr(), stream) == 0, pkg.last_error()
        # the same in ONE call: frames + checksums + headers + packet CRCs (one launch when the CRC is fused)
        out1 = torch.full((n * plan.stride,), 0xEE, dtype=torch.uint8, device="cuda")
        ln1 = torch.zeros(n, dtype=torch.int32, device="cuda")
        crc1 = torch.full((n,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        hdr1 = torch.full((n * 24,), 0xEE, dtype=torch.uint8, device="cuda")
        pkt1 = torch.full((n,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        plan.render_packets(out1.data_ptr(), plan.stride, ln1.data_ptr(), d32.data_ptr(), crc1.data_ptr(), hdr1.data_ptr(),
                            pkt1.data_ptr(), stream)
        torch.cuda.synchronize()
        assert torch.equal(ln1, ln) and torch.equal(crc1, crc) and torch.equal(hdr1, hdr) and torch.equal(pkt1, pkt), (mode, variant)
        assert torch.equal(out1, out) or all(
            torch.equal(out1[k * plan.stride:k * plan.stride + int(ln[k])], out[k * plan.stride:k * plan.stride + int(ln[k])])
            for k in range(n)), (mode, variant)
        host, lens = out.cpu().numpy(), ln.cpu().numpy().astype(np.uint32)
        crc_h, pkt_h, hdr_h = crc.cpu().numpy().astype(np.uint32), pkt.cpu().numpy().astype(np.uint32), hdr.cpu().numpy()
        for k in range(n):
            fr = host[k * plan.stride:k * plan.stride + int(lens[k])].tobytes()
            assert fr == oracle_convert(imgs[k], mode, dims[k][0], dims[k][1], orc.PALETTE_STANDARD, pad and mode != 4, asp), (mode, k)
            eh, ep = orc.ascii_frame_packet(fr, *dims[k])
            assert int(crc_h[k]) == orc.crc32c(fr), (mode, variant, k)
            assert hdr_h[24 * k:24 * k + 24].tobytes() == eh and int(pkt_h[k]) == ep, (mode, variant, k)
        plan.close()
    # a slab whose slots are smaller than the plan needs is refused on the host (the in-kernel overflow path -> CRC 0 is
    # covered under the emulator, tests/test_kernels_emulated.py)
    d = torch.from_numpy(np.ascontiguousarray(imgs[1])).cuda()
    f = pkg.frame_setup(d.data_ptr(), 320, 240, 80, 24, 0)
    plan = pkg.Plan(1, orc.PALETTE_STANDARD, [f])
    small = 1024
    out = torch.zeros(small, dtype=torch.uint8, device="cuda")
    ln = torch.zeros(1, dtype=torch.int32, device="cuda")
    crc = torch.full((1,), 77, dtype=torch.int32, device="cuda")
    assert L.asciichat_hip_plan_render_crc(plan._h, out.data_ptr(), small, ln.data_ptr(), crc.data_ptr(), stream) != 0  # stride < plan's
    plan.close()


# ------------------------------------------------------------------------------------------------
# ingest: device-resident latest-frame table
# ------------------------------------------------------------------------------------------------
def test_frame_table_ingest(gpu):
    import struct
    pkg, torch = gpu
    stream = torch.cuda.current_stream().cuda_stream

    def blob_of(img, extra=b""):
        return struct.pack(">II", img.shape[1], img.shape[0]) + np.ascontiguousarray(img).tobytes() + extra

    table = pkg.FrameTable(3)
    assert table.latest(0, stream)[0] is None  # no frame yet: has_video = false
    imgs = [TORTURE, orc.frame_hash_noise(640, 480, 5), orc.frame_bars(320, 200, 2)]
    for s_, im in enumerate(imgs):
        table.publish(s_, blob_of(im, b"trailing bytes are ignored" if s_ == 1 else b""), stream)
    # every "render thread" describes the same device frames -- no per-thread copies
    for mode in (1, 2, 5):
        rm = MODE_CAPS[mode][1]
        frames = []
        for s_, im in enumerate(imgs):
            ptr, w, h, gen = table.latest(s_, stream)
            assert (w, h, gen) == (im.shape[1], im.shape[0], 1) and ptr
            frames.append(pkg.frame_setup(ptr, w, h, 80, 24, rm))
        plan = pkg.Plan(mode, orc.PALETTE_STANDARD, frames * 4)  # 4 target clients looking at the same 3 sources
        n = len(frames) * 4
        out = torch.zeros(n * plan.stride, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(n, dtype=torch.int32, device="cuda")
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), stream)
        torch.cuda.synchronize()
        host, lens = out.cpu().numpy(), ln.cpu().numpy().astype(np.uint32)
        for k in range(n):
            got = host[k * plan.stride:k * plan.stride + int(lens[k])].tobytes()
            assert got == oracle_convert(imgs[k % 3], mode, 80, 24, orc.PALETTE_STANDARD), (mode, k)
        plan.close()
    # a new frame of another size replaces slot 0; the previous device frame stays intact for work in flight
    old_ptr = table.latest(0, stream)[0]
    newer = orc.frame_smooth(200, 100)
    table.publish(0, blob_of(newer), stream)
    ptr, w, h, gen = table.latest(0, stream)
    assert (w, h, gen) == (200, 100, 2) and ptr != old_ptr
    stale = pkg.frame_setup(old_ptr, TORTURE.shape[1], TORTURE.shape[0], 80, 24, 0)  # a render queued before the publish
    assert render_descs(gpu, 1, [stale])[0] == oracle_convert(TORTURE, 1, 80, 24, orc.PALETTE_STANDARD)
    f = pkg.frame_setup(ptr, w, h, 60, 20, 0)
    got = render_descs(gpu, 1, [f])[0]
    assert got == oracle_convert(newer, 1, 60, 20, orc.PALETTE_STANDARD)
    # rejected blobs leave the slot untouched (the reference skips such a frame)
    for bad in (b"", struct.pack(">II", 0, 4) + bytes(12), struct.pack(">II", 4000, 4) + bytes(4000 * 4 * 3),
                struct.pack(">II", 8, 8) + bytes(8 * 8 * 3 - 1)):
        with pytest.raises(RuntimeError):
            table.publish(0, bad, stream)
    assert table.latest(0, stream)[3] == 2
    table.close()


# ------------------------------------------------------------------------------------------------
# one server tick end to end: ingest -> render -> wire stage (SURVEY 8f.2 + hot path + 8f.3)
# ------------------------------------------------------------------------------------------------
def test_server_tick_end_to_end(gpu):
    """What a server tick does with this library: every client's camera blob is published once, all client
    frames are rendered in one launch from the shared device frames, and the ASCII frame packets (header +
    frame + CRCs) come back ready for the socket.  Each packet is then checked the way the reference CLIENT
    checks it (src/client/protocol.c:380-408: header fields in network order, CRC-32C of the frame)."""
    import struct
    pkg, torch = gpu
    L = pkg.lib()
    stream = torch.cuda.current_stream().cuda_stream
    n = 9
    cams = [orc.frame_hash_noise(640, 480, 30 + i) if i % 3 else orc.frame_bars(640, 480, i) for i in range(n)]
    table = pkg.FrameTable(n)
    for i, cam in enumerate(cams):
        table.publish(i, struct.pack(">II", 640, 480) + cam.tobytes(), stream)
    terms = [(80, 24), (120, 40), (200, 60), (33, 7), (80, 24), (97, 31), (160, 48), (20, 10), (80, 25)]
    for mode in (1, 2, 5):  # every client watches client (i+1) % n, in its own terminal size
        cl, rm = MODE_CAPS[mode]
        frames = []
        for i in range(n):
            ptr, w, h, _gen = table.latest((i + 1) % n, stream)
            frames.append(pkg.frame_setup(ptr, w, h, terms[i][0], terms[i][1], rm, True, True, False))
        plan = pkg.Plan(mode, orc.PALETTE_STANDARD, frames)
        out = torch.zeros(n * plan.stride, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(n, dtype=torch.int32, device="cuda")
        dims = torch.tensor(terms, dtype=torch.int32, device="cuda")
        crc = torch.zeros(n, dtype=torch.int32, device="cuda")
        hdr = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
        pkt = torch.zeros(n, dtype=torch.int32, device="cuda")
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), stream)
        assert L.asciichat_hip_frame_packets(out.data_ptr(), plan.stride, ln.data_ptr(), plan.stride, n, dims.data_ptr(),
                                             crc.data_ptr(), hdr.data_ptr(), pkt.data_ptr(), stream) == 0
        torch.cuda.synchronize()
        host, lens, hdrs = out.cpu().numpy(), ln.cpu().numpy().astype(np.uint32), hdr.cpu().numpy()
        pkts = pkt.cpu().numpy().astype(np.uint32)
        for i in range(n):
            payload = hdrs[24 * i:24 * i + 24].tobytes() + host[i * plan.stride:i * plan.stride + int(lens[i])].tobytes()
            # --- the receiving client's checks ---
            width, height, original_size, compressed_size, checksum, flags = struct.unpack(">6I", payload[:24])
            frame = payload[24:]
            assert (width, height, compressed_size, flags) == (terms[i][0], terms[i][1], 0, 0)
            assert original_size == len(frame) and orc.crc32c(frame) == checksum
            assert orc.crc32c(payload) == int(pkts[i])  # packet_header_t.crc32 over the whole payload
            assert frame == orc.convert_with_caps(cams[(i + 1) % n], terms[i][0], terms[i][1], cl, rm, True, True, False)
        plan.close()
    table.close()


def test_uniform_batches_pass_the_descriptor_by_value(gpu):
    """Equally sized frames in one slab (and single frames): the plan hands the common descriptor to the kernel with
    its arguments; bytes identical to the descriptor-array path, for whole batches, sub-ranges and row bands."""
    pkg, torch = gpu
    imgs = [orc.frame_hash_noise(320, 200, 50 + k) for k in range(6)]
    slab = torch.from_numpy(np.ascontiguousarray(np.stack(imgs))).cuda()
    for mode in (MODE_TRUE_FG, MODE_HB_TRUE, 2, 0, MODE_16_DITHER_BG):
        rm = MODE_CAPS.get(mode, (3, 0))[1]
        frames = [pkg.frame_setup(slab.data_ptr() + k * 320 * 200 * 3, 320, 200, 100, 30, rm, True, True, False)
                  for k in range(6)]
        exp = [oracle_convert(i, mode, 100, 30, orc.PALETTE_STANDARD, True, True) for i in imgs]
        plan = pkg.Plan(mode, orc.PALETTE_STANDARD, frames)
        assert plan.uniform
        out = torch.zeros(6 * plan.stride, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(6, dtype=torch.int32, device="cuda")

        def take(first=0, count=6):
            torch.cuda.synchronize()
            h, l = out.cpu().numpy(), ln.cpu().numpy()
            return [h[k * plan.stride:k * plan.stride + int(l[k])].tobytes() for k in range(count)]

        for allow in (True, False, True):
            plan.set_uniform(allow)
            assert plan.uniform == allow
            out.zero_()
            plan.render(out.data_ptr(), plan.stride, ln.data_ptr())
            assert take() == exp, (mode, allow)
        out.zero_()
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), 0, 2, 3)   # frames 2..4 into slots 0..2
        assert take(count=3) == exp[2:5], mode
        if mode != MODE_16_DITHER_BG:
            plan.set_split(4)
            assert plan.parts > 1 and plan.uniform
            out.zero_()
            plan.render(out.data_ptr(), plan.stride, ln.data_ptr())
            assert take() == exp, (mode, "bands")
        # an update that breaks the progression falls back to the array
        frames[1], frames[4] = frames[4], frames[1]
        plan.update(frames)
        assert not plan.uniform
        out.zero_()
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr())
        e2 = list(exp)
        e2[1], e2[4] = e2[4], e2[1]
        assert take() == e2, (mode, "fallback")
        plan.close()


def test_launches_in_flight_on_three_streams(gpu):
    """Three independent batches kept in flight on separate streams (what bench.py times): every plan takes the geometry
    for its share of the GPU (asciichat_hip_plan_set_concurrency) and every frame of every launch stays byte-exact."""
    pkg, torch = gpu
    n = 200
    rng = np.random.default_rng(77)
    base = [orc.frame_hash_noise(160, 120, 500 + k) for k in range(8)] + [orc.frame_bars(160, 120, 3), TORTURE]
    streams = [torch.cuda.Stream() for _ in range(3)]
    for mode in (MODE_TRUE_FG, 2, MODE_HB_TRUE):
        rm = MODE_CAPS[mode][1]
        exp_of = [oracle_convert(i, mode, 80, 24, orc.PALETTE_STANDARD) for i in base]
        plans, outs, lns, picks, keep = [], [], [], [], []
        for s in range(3):
            pick = rng.integers(0, len(base), n)
            dev = [torch.from_numpy(np.ascontiguousarray(base[i])).cuda() for i in range(len(base))]
            frames = [pkg.frame_setup(dev[i].data_ptr(), base[i].shape[1], base[i].shape[0], 80, 24, rm, False, False, False)
                      for i in pick]
            plan = pkg.Plan(mode, orc.PALETTE_STANDARD, frames)
            v0 = plan.variant
            plan.set_concurrency(3)
            if mode != MODE_HB_TRUE:  # per-cell modes: the stream kernel, 1024 threads alone / 512 when sharing the CUs
                assert v0 == 16 and plan.variant == 17, (v0, plan.variant)   # 200 frames on a third of 256 CUs
            else:  # half blocks, whole frames: the rows kernel (three 80-cell rows per 256-slot block)
                assert plan.variant == 25
            plans.append(plan)
            picks.append(pick)
            keep.append(dev)
            outs.append(torch.zeros(n * plan.stride, dtype=torch.uint8, device="cuda"))
            lns.append(torch.zeros(n, dtype=torch.int32, device="cuda"))
        torch.cuda.synchronize()
        for rnd in range(20):
            for s in range(3):
                plans[s].render(outs[s].data_ptr(), plans[s].stride, lns[s].data_ptr(), streams[s].cuda_stream)
        torch.cuda.synchronize()
        for s in range(3):
            h, l = outs[s].cpu().numpy(), lns[s].cpu().numpy()
            st = plans[s].stride
            for k in range(n):
                assert h[k * st:k * st + int(l[k])].tobytes() == exp_of[picks[s][k]], (mode, s, k)
            plans[s].close()


def test_rccl_comm_and_grid_exchange_world1(gpu):
    """The C-ABI's RCCL layer (comm.c) on one GPU: a world-size-1 communicator runs the real ncclAllGather calls of
    the sharded-slab gather and of the grid's tile exchange; the grid rendered from the gathered tiles equals the
    oracle's composite + convert (K4: nine 1080p sources, 3x3 at 160x48), with and without a client that has no video.
    World size > 1 is the driver's to run (bench.py --workload grid9 --gpus N); the partition logic is CPU-tested."""
    pkg, torch = gpu
    comm = pkg.Comm(1, 0, pkg.comm_unique_id())
    st = torch.cuda.current_stream().cuda_stream
    # (1) in-place slab gather: a no-op permutation at world 1, but the calls, sizes and group must be accepted
    imgs = [orc.frame_hash_noise(96, 54, 300 + i) for i in range(3)]
    got0 = render_batch(gpu, MODE_TRUE_FG, imgs, 40, 12)
    dev = [torch.from_numpy(i).cuda() for i in imgs]
    frames = [pkg.frame_setup(d.data_ptr(), 96, 54, 40, 12, 0, False, False, False) for d in dev]
    plan = pkg.Plan(MODE_TRUE_FG, orc.PALETTE_STANDARD, frames)
    slab = torch.zeros(3 * plan.stride, dtype=torch.uint8, device="cuda")
    ln = torch.zeros(3, dtype=torch.int32, device="cuda")
    plan.render(slab.data_ptr(), plan.stride, ln.data_ptr(), st)
    comm.all_gather_slab(slab.data_ptr(), plan.stride, ln.data_ptr(), 3, st)
    torch.cuda.synchronize()
    for i in range(3):
        n = int(ln[i].item())
        assert bytes(slab[i * plan.stride:i * plan.stride + n].cpu().numpy()) == got0[i]
    plan.close()
    # (2) K4 through asciichat_hip_grid_*: tiles resized by the owner, all-gathered, rendered by the fused sampler
    srcs = [orc.frame_hash_noise(1920, 1080, 10 + i) if i % 2 else orc.frame_bars(1920, 1080, i) for i in range(9)]
    dsrc = [torch.from_numpy(s).cuda() for s in srcs]
    for has_video in (None, [True, True, False, True, True, True, False, True, True]):
        grid = pkg.Grid(comm, [(1920, 1080)] * 9, 160, 48, has_video)
        assert all(grid.owner(k) == 0 for k in range(9))
        live = [s if (has_video is None or has_video[k]) else None for k, s in enumerate(srcs)]
        if has_video is None:
            assert (grid.geometry.cols, grid.geometry.rows, grid.geometry.n_src) == (3, 3, 9)
        else:
            assert grid.geometry.n_src == 7
        grid.exchange({k: dsrc[k].data_ptr() for k in range(9)}, st)
        ref = orc.composite(live, 160, 48)
        for mode, (cl, rm) in ((MODE_TRUE_FG, (3, 0)), (MODE_HB_TRUE, (3, 2)), (2, (2, 0))):
            h = 96 if rm == 2 else 48
            fs = []
            for _ in range(5):  # five target clients looking at the same grid
                f = pkg.frame_setup(None, 160, 96, 160, h, rm, True, True, False)
                f.comp = grid.composite_dev
                fs.append(f)
            plan = pkg.Plan(mode, orc.PALETTE_STANDARD, fs)
            out = torch.zeros(5 * plan.stride, dtype=torch.uint8, device="cuda")
            l5 = torch.zeros(5, dtype=torch.int32, device="cuda")
            plan.render(out.data_ptr(), plan.stride, l5.data_ptr(), st)
            torch.cuda.synchronize()
            exp = orc.convert_with_caps(ref, 160, h, cl, rm, True, True, False)
            for i in range(5):
                n = int(l5[i].item())
                assert bytes(out[i * plan.stride:i * plan.stride + n].cpu().numpy()) == exp, (mode, i, has_video is None)
            plan.close()
        # the one-GPU direct form: no tiles, no resize, no collective -- plans sample the clients' frames themselves and a
        # tick only names the sources (asciichat_hip_grid_set_direct); new pointers on the second tick
        grid.set_direct(True)
        f = pkg.frame_setup(None, 160, 96, 160, 48, 0, True, True, False)
        f.comp = grid.composite_dev
        plan = pkg.Plan(MODE_TRUE_FG, orc.PALETTE_STANDARD, [f] * 9)
        out = torch.zeros(9 * plan.stride, dtype=torch.uint8, device="cuda")
        l9 = torch.zeros(9, dtype=torch.int32, device="cuda")
        rolled = dsrc[1:] + dsrc[:1]
        for tick, cur in enumerate((dsrc, dsrc, rolled)):
            grid.exchange({k: cur[k].data_ptr() for k in range(9)}, st)
            plan.render(out.data_ptr(), plan.stride, l9.data_ptr(), st)
            torch.cuda.synchronize()
            cur_np = srcs if tick < 2 else srcs[1:] + srcs[:1]
            live2 = [s if (has_video is None or has_video[k]) else None for k, s in enumerate(cur_np)]
            exp = orc.convert_with_caps(orc.composite(live2, 160, 48), 160, 48, 3, 0, True, True, False)
            for i in (0, 8):
                n = int(l9[i].item())
                assert bytes(out[i * plan.stride:i * plan.stride + n].cpu().numpy()) == exp, ("direct", tick, i)
        plan.close()
        grid.close()
    comm.close()


def test_frame_table_publish_rows_uploads_only_sampled_rows(gpu):
    """frame_table_publish_rows: only the rows that the named targets sample cross PCIe; renders of
    those targets are byte-identical to the oracle (and to a full publish); two target heights and a flipped target share
    one publish; odd widths (row pitch not a multiple of 16) take the byte path of the scatter kernel."""
    import struct
    pkg, torch = gpu
    stream = torch.cuda.current_stream().cuda_stream
    for (w, h) in ((1920, 1080), (333, 201)):
        img = orc.frame_hash_noise(w, h, 77)
        blob = struct.pack(">II", w, h) + np.ascontiguousarray(img).tobytes()
        table = pkg.FrameTable(1)
        t_a = pkg.frame_setup(None, w, h, 80, 24, 0, False, False, False)         # 80x24 foreground
        t_b = pkg.frame_setup(None, w, h, 100, 37, 2, True, True, False)          # half-block, aspect + padding
        t_c = pkg.frame_setup(None, w, h, 80, 24, 0, False, False, False)
        assert pkg.lib().achip_frame_set_display_ops(C.byref(t_c), False, True, 0) == 0   # flipped vertically
        for round_ in range(3):  # both buffers of the slot, and a buffer reused
            table.publish_rows(0, blob, [t_a, t_b, t_c], stream)
            ptr, pw, ph, gen = table.latest(0, stream)
            assert (pw, ph) == (w, h)
            # the rows nobody named are not what the blob holds: count rows that arrived
            dev_img = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
            assert pkg.lib().asciichat_hip_resize(ptr, w, h, dev_img.data_ptr(), w, h, stream) == 0  # identity: a copy
            torch.cuda.synchronize()
            same = (dev_img.cpu().numpy() == img).all(axis=(1, 2))
            assert 24 <= int(same.sum()) <= 24 + 2 * 37 + 24, int(same.sum())
            for tmpl, mode, cl, rm, args in ((t_a, 1, 3, 0, (80, 24, False, False)), (t_b, 5, 3, 2, (100, 37, True, True))):
                f = pkg.Frame.from_buffer_copy(tmpl)
                f.src = ptr
                got = render_descs(gpu, mode, [f])[0]
                assert got == orc.convert_with_caps(img, args[0], args[1], cl, rm, args[2], args[3], False), (w, h, mode, round_)
            f = pkg.Frame.from_buffer_copy(t_c)
            f.src = ptr
            assert render_descs(gpu, 1, [f])[0] == orc.convert_with_caps(orc.flip(img, False, True), 80, 24, 3, 0, False, False, False)
        # a target that does not describe this frame is refused
        bad = pkg.frame_setup(None, w, h + 1, 80, 24, 0, False, False, False)
        with pytest.raises(RuntimeError):
            table.publish_rows(0, blob, [bad], stream)
        table.close()


def test_frame_table_publish_rows_batch(gpu):
    """frame_table_publish_rows_batch: a tick's clients in ONE packed block, ONE DMA, ONE scatter launch -- the renders of
    every client equal the oracle's (and a full publish's), across three ticks (both buffers of every slot, staging
    parity reused), clients of two widths in one batch, duplicates / wrong heights refused."""
    import struct
    pkg, torch = gpu
    stream = torch.cuda.current_stream().cuda_stream
    n = 12
    dims = [(1920, 1080) if i % 3 else (640, 1080) for i in range(n)]  # same height (what the targets describe), two widths
    table = pkg.FrameTable(n + 2)
    t_fg = pkg.frame_setup(None, 1920, 1080, 80, 24, 0, False, False, False)
    t_hb = pkg.frame_setup(None, 1920, 1080, 60, 20, 2, False, False, False)
    keep = []
    for tick in range(3):
        imgs = [orc.frame_hash_noise(w, h, 300 + 17 * tick + i) for i, (w, h) in enumerate(dims)]
        blobs = [struct.pack(">II", im.shape[1], im.shape[0]) + np.ascontiguousarray(im).tobytes() for im in imgs]
        bufs = [C.create_string_buffer(b, len(b)) for b in blobs]
        keep.append(bufs)
        slots = [(5 * i + tick) % (n + 2) for i in range(n)]
        assert len(set(slots)) == n
        table.publish_rows_batch(slots, [(C.addressof(b), len(b)) for b in bufs], [t_fg, t_hb], stream)
        for mode, tmpl, (W, H, cl, rm) in ((1, t_fg, (80, 24, 3, 0)), (5, t_hb, (60, 20, 3, 2))):
            frames = []
            for i, sl in enumerate(slots):
                ptr, w, h, gen = table.latest(sl, stream)
                assert (w, h) == dims[i] and ptr
                f = pkg.frame_setup(ptr, w, h, W, H, rm, False, False, False)
                frames.append(f)
            got = render_descs(gpu, mode, frames)
            for i in range(n):
                assert got[i] == orc.convert_with_caps(imgs[i], W, H, cl, rm, False, False, False), (tick, mode, i)
    # sampled PIXELS (targets at most half as wide as the source) next to sampled ROWS (a wider target joins: every column
    # is needed), a horizontally flipped target, odd geometry; descriptors filled by frame_table_latest_frames
    w, h = 333, 201
    t_a = pkg.frame_setup(None, w, h, 100, 37, 0, False, False, False)
    t_b = pkg.frame_setup(None, w, h, 80, 24, 2, False, False, False)
    t_x = pkg.frame_setup(None, w, h, 100, 37, 0, False, False, False)
    assert pkg.lib().achip_frame_set_display_ops(C.byref(t_x), True, False, 0) == 0   # flipped horizontally
    t_wide = pkg.frame_setup(None, w, h, 200, 60, 0, False, False, False)
    for targets in ([t_a, t_b, t_x], [t_a, t_b, t_x, t_wide]):
        imgs = [orc.frame_hash_noise(w, h, 900 + i + len(targets)) for i in range(5)]
        bufs = [C.create_string_buffer(struct.pack(">II", w, h) + np.ascontiguousarray(im).tobytes(), 8 + im.size) for im in imgs]
        keep.append(bufs)
        slots = (C.c_int * 5)(2, 4, 6, 8, 10)
        table.publish_rows_batch(slots, [(C.addressof(b), len(b)) for b in bufs], targets, stream)
        for tmpl, mode, exp in ((t_a, 1, lambda im: orc.convert_with_caps(im, 100, 37, 3, 0, False, False, False)),
                                (t_b, 5, lambda im: orc.convert_with_caps(im, 80, 24, 3, 2, False, False, False)),
                                (t_x, 1, lambda im: orc.convert_with_caps(orc.flip(im, True, False), 100, 37, 3, 0, False, False, False)),
                                (t_wide, 1, lambda im: orc.convert_with_caps(im, 200, 60, 3, 0, False, False, False)))[:len(targets)]:
            frames = (pkg.Frame * 5)(*[pkg.Frame.from_buffer_copy(tmpl) for _ in range(5)])
            assert table.latest_frames(slots, frames, stream) == 5
            got = render_descs(gpu, mode, list(frames))
            for i in range(5):
                assert got[i] == exp(imgs[i]), (len(targets), mode, i)
        if len(targets) == 3:  # what arrived: only sampled pixels (the rest of the buffer is not the blob's)
            ptr = table.latest(2, stream)[0]
            dev_img = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
            assert pkg.lib().asciichat_hip_resize(ptr, w, h, dev_img.data_ptr(), w, h, stream) == 0
            torch.cuda.synchronize()
            same = (dev_img.cpu().numpy() == imgs[0]).all(axis=2)
            assert 100 * 37 <= int(same.sum()) < w * h // 3, int(same.sum())
    # a descriptor set up for another geometry gets no source (the client changed its resolution)
    stale = (pkg.Frame * 1)(pkg.frame_setup(None, 640, 480, 80, 24, 0, False, False, False))
    assert table.latest_frames((C.c_int * 1)(2), stale, stream) == 0 and not stale[0].src
    b0 = keep[0][0]
    with pytest.raises(RuntimeError):  # a slot named twice
        table.publish_rows_batch([1, 1], [(C.addressof(b0), len(b0))] * 2, [t_fg], stream)
    bad = pkg.frame_setup(None, 1920, 1081, 80, 24, 0, False, False, False)
    with pytest.raises(RuntimeError):  # a target that does not describe the blobs
        table.publish_rows_batch([0], [(C.addressof(b0), len(b0))], [bad], stream)
    table.close()


def test_exact_length_frames_full_batch(gpu):
    """BASELINE's metric workload (256 x 1080p -> 80x24 truecolor) through plan_render_packets_packed as ONE launch: the
    render writes every frame at its exact length (claiming its place with an atomic add: frames lie in completion order),
    with checksums and headers; repeated launches on the plan's cursor words; also into mapped host memory, also ANSI-256,
    aspect + padding.  Against the two-launch form for all 256 frames and against the oracle for a sample."""
    pkg, torch = gpu
    stream = torch.cuda.current_stream().cuda_stream
    n, sw, sh = 256, 1920, 1080
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    frames_t = torch.randint(0, 256, (n, sh, sw, 3), dtype=torch.uint8, device="cuda", generator=g)
    frames_t[1] = torch.from_numpy(orc.frame_bars(sw, sh, 6)).cuda()  # short frames next to long ones
    frames_t[2] = 0
    for mode, cl, (W, H), asp in ((1, 3, (80, 24), False), (2, 2, (80, 24), False), (1, 3, (80, 24), True)):
        descs = [pkg.frame_setup(frames_t.data_ptr() + i * sh * sw * 3, sw, sh, W, H, 0, asp, asp, False) for i in range(n)]
        plan = pkg.Plan(mode, orc.PALETTE_STANDARD, descs)
        assert plan.exact_length and plan.stride <= 48 * 1024
        stride = plan.stride
        d32 = torch.tensor([[W, H]] * n, dtype=torch.int32, device="cuda")
        tab = (8 * (n + 1) + 4 * n + 15) // 16 * 16

        def run(one_launch, host_dst, force=True):
            # 1 = wherever the plan qualifies; -1 = the automatic choice: device destinations only
            plan.set_exact_length((1 if force else -1) if one_launch else 0)
            slab = torch.full((n * stride,), 0xEE, dtype=torch.uint8, device="cuda")
            ln = torch.zeros(n, dtype=torch.int32, device="cuda")
            crc = torch.zeros(n, dtype=torch.int32, device="cuda")
            hdr = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
            pkt = torch.zeros(n, dtype=torch.int32, device="cuda")
            if host_dst:
                hb = pkg.HostBuffer(tab + n * stride)
                dst, off_p, len_p = hb.dev + tab, hb.dev, hb.dev + 8 * (n + 1)
            else:
                buf = torch.zeros(tab + n * stride, dtype=torch.uint8, device="cuda")
                dst, off_p, len_p = buf.data_ptr() + tab, buf.data_ptr(), buf.data_ptr() + 8 * (n + 1)
            for _ in range(3):  # the cursor words are re-armed by every launch
                plan.render_packets_packed(slab.data_ptr(), stride, ln.data_ptr(), d32.data_ptr(), crc.data_ptr(), hdr.data_ptr(),
                                           pkt.data_ptr(), dst, n * stride, off_p, len_p, stream)
            torch.cuda.synchronize()
            v = hb.view().copy() if host_dst else buf.cpu().numpy()
            if host_dst:
                hb.close()
            off = v[:8 * (n + 1)].view(np.uint64)
            pl = v[8 * (n + 1):8 * (n + 1) + 4 * n].view(np.uint32)
            fr = [v[tab + int(off[i]):tab + int(off[i]) + int(pl[i])].tobytes() for i in range(n)]
            spans = sorted((int(off[i]), int(off[i]) + (int(pl[i]) + 15) // 16 * 16) for i in range(n))
            assert spans[0][0] == 0 and all(spans[i][1] == spans[i + 1][0] for i in range(n - 1)) and spans[-1][1] == int(off[n])
            assert bool((slab == 0xEE).all()) == (one_launch and (force or not host_dst))
            return fr, ln.cpu().numpy(), crc.cpu().numpy(), hdr.cpu().numpy(), pkt.cpu().numpy(), pl

        ref = run(False, False)
        for host_dst, force in ((False, True), (True, True), (False, False), (True, False)):
            got = run(True, host_dst, force)  # (not forced: one launch into device memory, render + pack into host memory)
            assert got[0] == ref[0], (mode, host_dst, force)
            for a, b in zip(got[1:], ref[1:]):
                assert np.array_equal(a, b), (mode, host_dst, force)
        for i in (0, 1, 2, 3, 100, 255):
            exp = orc.convert_with_caps(frames_t[i].cpu().numpy(), W, H, cl, 0, asp, asp, False)
            assert ref[0][i] == exp and int(ref[2][i]) & 0xFFFFFFFF == orc.crc32c(exp), (mode, i)
        plan.close()


def test_length_first_exact_length_frames_through_the_plan(gpu):
    """Frames beyond the 48 KB of the LDS-image form leave ONE launch at their exact lengths (render_stream.hpp LF, round 6):
    plan_render_packed / plan_render_packets_packed take the form by themselves for sampled-image (dense) sources and sources up to
    1080p, on request (set_exact_length 1) for any single source, never without off_out; bytes, lengths, checksums, headers and packet CRCs equal
    those of render + pass, frames tile the destination (completion order)."""
    pkg, torch = gpu
    stream = torch.cuda.current_stream().cuda_stream
    W, H, n = 200, 60, 200  # (from three quarters of a frame per CU on, plans launch whole frames: the form's precondition)
    dense = [orc.frame_hash_noise(W, H, 70 + i) for i in range(24)]
    dense[3][:] = 0
    dense[4] = orc.frame_smooth(W, H)
    hd = [orc.frame_hash_noise(1920, 1080, 7 + i) for i in range(5)]     # sources up to 1080p: by itself too (round 6's wire audit)
    big = [orc.frame_hash_noise(2048, 1152, 17 + i) for i in range(5)]   # wider ones: a second gather costs more than the pack pass
    for pool, auto_takes_it in ((dense, True), (hd, True), (big, False)):
        dev = [torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in pool]
        imgs = [pool[i % len(pool)] for i in range(n)]
        frames = [pkg.frame_setup(dev[i % len(pool)].data_ptr(), imgs[i].shape[1], imgs[i].shape[0], W, H, 0, False, False, False) for i in range(n)]
        m = len(frames)
        want_pool = [orc.convert_with_caps(i, W, H, 3, 0, False, False, False) for i in pool]
        want = [want_pool[i % len(pool)] for i in range(n)]
        plan = pkg.Plan(MODE_TRUE_FG, orc.PALETTE_STANDARD, frames)
        assert plan.stride > 48 * 1024 and 16 <= plan.variant <= 17
        dims = torch.tensor([[W, H]] * m, dtype=torch.int32, device="cuda")
        ref = None
        for setting in (0, -1, 1):
            plan.set_exact_length(setting)
            for wire in (False, True):
                slab = torch.full((m * plan.stride,), 0xEE, dtype=torch.uint8, device="cuda")
                ln = torch.zeros(m, dtype=torch.int32, device="cuda")
                crc = torch.zeros(m, dtype=torch.int32, device="cuda")
                hdr = torch.zeros(m * 24, dtype=torch.uint8, device="cuda")
                pkt = torch.zeros(m, dtype=torch.int32, device="cuda")
                dst = torch.full((m * plan.stride,), 0xEE, dtype=torch.uint8, device="cuda")
                off = torch.zeros(m + 1, dtype=torch.int64, device="cuda")
                plen = torch.zeros(m, dtype=torch.int32, device="cuda")
                for _ in range(2):  # the second call runs on the cursor words the first one re-armed
                    if wire:
                        plan.render_packets_packed(slab.data_ptr(), plan.stride, ln.data_ptr(), dims.data_ptr(), crc.data_ptr(), hdr.data_ptr(),
                                                   pkt.data_ptr(), dst.data_ptr(), dst.numel(), off.data_ptr(), plen.data_ptr(), stream)
                    else:
                        plan.render_packed(slab.data_ptr(), plan.stride, ln.data_ptr(), dst.data_ptr(), dst.numel(), off.data_ptr(), plen.data_ptr(), stream)
                    torch.cuda.synchronize()
                o, l, d = off.cpu().numpy().astype(np.uint64), plen.cpu().numpy().astype(np.uint32), dst.cpu().numpy()
                one_launch = bool((slab.cpu().numpy() == 0xEE).all())  # the slab is never written by the length-first form
                assert one_launch == (setting == 1 or (setting == -1 and auto_takes_it)), (setting, wire, auto_takes_it)
                spans = sorted((int(o[i]), int(o[i]) + (len(want[i]) + 15) // 16 * 16) for i in range(m))
                assert spans[0][0] == 0 and all(spans[i][1] == spans[i + 1][0] for i in range(m - 1)) and spans[-1][1] == int(o[m])
                for i in range(m):
                    assert int(l[i]) == len(want[i]) and d[int(o[i]):int(o[i]) + int(l[i])].tobytes() == want[i], (setting, wire, i)
                if wire:
                    got = (crc.cpu().numpy().tobytes(), hdr.cpu().numpy().tobytes(), pkt.cpu().numpy().tobytes())
                    if ref is None:
                        ref = got
                        cc = crc.cpu().numpy().astype(np.uint32)
                        for i in range(len(pool)):
                            assert int(cc[i]) == orc.crc32c(want[i]), i
                    assert got == ref, (setting, "checksums / headers / packet CRCs")
        # without off_out the documented (ordered) two-pass layout stays, whatever the setting
        plan.set_exact_length(1)
        slab = torch.full((m * plan.stride,), 0xEE, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(m, dtype=torch.int32, device="cuda")
        dst = torch.zeros(m * plan.stride, dtype=torch.uint8, device="cuda")
        plan.render_packed(slab.data_ptr(), plan.stride, ln.data_ptr(), dst.data_ptr(), dst.numel(), 0, 0, stream)
        torch.cuda.synchronize()
        d, pos = dst.cpu().numpy(), 0
        for i in range(m):
            assert d[pos:pos + len(want[i])].tobytes() == want[i], i
            pos += (len(want[i]) + 15) // 16 * 16
        plan.close()


def test_packed_without_offsets_keeps_the_documented_order(gpu):
    """A caller that passes no off_out relies on pack_frames' layout (frame i behind the 16-byte rounded lengths of the
    frames before it).  The one-launch form lays frames out in completion order, so such a call must not take it -- not by
    default and not under set_exact_length(1)."""
    pkg, torch = gpu
    stream = torch.cuda.current_stream().cuda_stream
    n, sw, sh, W, H = 64, 640, 360, 80, 24
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    frames_t = torch.randint(0, 256, (n, sh, sw, 3), dtype=torch.uint8, device="cuda", generator=g)
    frames_t[3] = 0  # short frames between long ones: completion order differs from index order
    frames_t[7] = torch.from_numpy(orc.frame_bars(sw, sh, 6)).cuda()
    descs = [pkg.frame_setup(frames_t.data_ptr() + i * sh * sw * 3, sw, sh, W, H, 0, False, False, False) for i in range(n)]
    plan = pkg.Plan(1, orc.PALETTE_STANDARD, descs)
    assert plan.exact_length
    stride = plan.stride
    exp = [orc.convert_with_caps(frames_t[i].cpu().numpy(), W, H, 3, 0, False, False, False) for i in range(n)]
    d32 = torch.tensor([[W, H]] * n, dtype=torch.int32, device="cuda")
    for force in (-1, 1):
        plan.set_exact_length(force)
        for wire in (False, True):
            slab = torch.zeros(n * stride, dtype=torch.uint8, device="cuda")
            ln = torch.zeros(n, dtype=torch.int32, device="cuda")
            dst = torch.full((n * stride,), 0xEE, dtype=torch.uint8, device="cuda")
            if wire:
                crc = torch.zeros(n, dtype=torch.int32, device="cuda")
                hdr = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
                plan.render_packets_packed(slab.data_ptr(), stride, ln.data_ptr(), d32.data_ptr(), crc.data_ptr(), hdr.data_ptr(),
                                           None, dst.data_ptr(), n * stride, None, None, stream)
            else:
                plan.render_packed(slab.data_ptr(), stride, ln.data_ptr(), dst.data_ptr(), n * stride, None, None, stream)
            torch.cuda.synchronize()
            v = dst.cpu().numpy()
            lens = ln.cpu().numpy().astype(np.uint32)
            at = 0
            for i in range(n):
                assert int(lens[i]) == len(exp[i])
                assert v[at:at + len(exp[i])].tobytes() == exp[i], (force, wire, i)
                at += (len(exp[i]) + 15) // 16 * 16
            if wire:
                assert [int(c) & 0xFFFFFFFF for c in crc.cpu().numpy()] == [orc.crc32c(e) for e in exp]
    plan.close()


def test_frame_table_sampled_image_ingest(gpu):
    """frame_dense.c on the MI355X: a tick's clients staged as the images their targets sample
    -- one pinned block, ONE DMA, NO kernel -- and rendered from those images: bytes equal the oracle's on the original
    1080p / 4K frames.  256 clients of one geometry in one batch (the descriptors then differ by a constant pitch only:
    a uniform launch), per-blob stage() + commit, half blocks, a flipped + tinted client, aspect + padding, the descriptor
    array reused across ticks, a client that stops sending while the ring of blocks turns over."""
    import struct
    pkg, torch = gpu
    stream = torch.cuda.current_stream().cuda_stream
    w, h = 1920, 1080
    n = 256
    distinct = 8
    table = pkg.FrameTable(n)
    slots = (C.c_int * n)(*range(n))
    tmpl = pkg.frame_setup(None, w, h, 80, 24, 0, False, False, False)
    frames = (pkg.Frame * n)(*[pkg.Frame.from_buffer_copy(tmpl) for _ in range(n)])
    keep = []
    for tick in range(3):
        imgs = [orc.frame_hash_noise(w, h, 5000 + 13 * tick + k) for k in range(distinct)]
        bufs = [C.create_string_buffer(struct.pack(">II", w, h) + np.ascontiguousarray(im).tobytes(), 8 + im.size) for im in imgs]
        keep.append(bufs)
        table.publish_sampled_batch(slots, [(C.addressof(bufs[i % distinct]), len(bufs[i % distinct])) for i in range(n)],
                                    [tmpl], stream)
        assert table.latest_frames(slots, frames, stream) == n
        assert frames[0].src_w == 80 and frames[0].src_h == 24 and frames[0].x_ratio == 1 << 16
        pitch = frames[1].src - frames[0].src
        assert pitch == 80 * 24 * 3 and all(frames[i + 1].src - frames[i].src == pitch for i in range(n - 1))
        got = render_descs(gpu, MODE_TRUE_FG, list(frames))
        exp = [oracle_convert(im, MODE_TRUE_FG, 80, 24, orc.PALETTE_STANDARD) for im in imgs]
        for i in range(n):
            assert got[i] == exp[i % distinct], (tick, i)
    table.close()
    # mixed clients, staged one by one: half blocks of a 4K frame, flips + tint, aspect + padding
    geos = [((3840, 2160), (400, 120, 3, 2), False), ((1920, 1080), (80, 24, 3, 0), False), ((1920, 1080), (80, 24, 3, 0), True),
            ((1280, 720), (100, 37, 2, 0), False), ((1920, 1080), (80, 24, 3, 2), False)]
    m = len(geos)
    table = pkg.FrameTable(m)
    sl = (C.c_int * m)(*range(m))

    def target(i):
        (sw, sh), (W, H, cl, rm), asp = geos[i]
        f = pkg.frame_setup(None, sw, sh, W, H, rm, asp, asp, False)
        if i == 1:
            assert pkg.lib().achip_frame_set_display_ops(C.byref(f), True, True, 5) == 0
        return f

    def expect(i, im):
        (sw, sh), (W, H, cl, rm), asp = geos[i]
        if i == 1:
            im = orc.color_filter(orc.flip(im, True, True), 5)
        return orc.convert_with_caps(im, W, H, cl, rm, asp, asp, False)

    fr = (pkg.Frame * m)(*[target(i) for i in range(m)])
    last = [None] * m
    for tick in range(7):
        live = [i for i in range(m) if not (i == 4 and tick >= 1)]  # client 4 sends once: carried forward twice over
        imgs = {i: orc.frame_hash_noise(geos[i][0][0], geos[i][0][1], 6000 + 7 * tick + i) for i in live}
        bufs = {i: C.create_string_buffer(struct.pack(">II", im.shape[1], im.shape[0]) + np.ascontiguousarray(im).tobytes(), 8 + im.size)
                for i, im in imgs.items()}
        keep.append(bufs)
        for i in live:
            table.stage(i, (C.addressof(bufs[i]), len(bufs[i])), target(i))
            last[i] = imgs[i]
        table.commit(stream)
        assert table.latest_frames(sl, fr, stream) == m
        for i in range(m):
            mode = pkg.lib().achip_mode_from_caps(geos[i][1][2], geos[i][1][3])
            assert render_descs(gpu, mode, [fr[i]])[0] == expect(i, last[i]), (tick, i)
    table.close()


def test_frame_table_sampled_images_wait_for_queued_readers(gpu):
    """The ring of sampled-image blocks: a render of tick A's images is queued behind a busy consumer stream, then as many
    ticks as the ring holds are committed on another stream -- the last of them DMAs into the block tick A's images live in.
    That DMA must go behind the queued render (the table records an event on every stream it handed pointers of the block)."""
    import struct
    pkg, torch = gpu
    n, w, h = 6, 1920, 1080
    ticks = [[orc.frame_hash_noise(w, h, 8000 + 10 * t + i) for i in range(n)] for t in range(5)]
    bufs = [[C.create_string_buffer(struct.pack(">II", w, h) + np.ascontiguousarray(im).tobytes(), 8 + im.size) for im in tk]
            for tk in ticks]
    tmpl = pkg.frame_setup(None, w, h, 80, 24, 0, False, False, False)
    slots = (C.c_int * n)(*range(n))
    busy_src = torch.from_numpy(np.ascontiguousarray(orc.frame_hash_noise(3840, 2160, 5))).cuda()
    pub, cons = torch.cuda.Stream(), torch.cuda.Stream()
    exp = [oracle_convert(im, MODE_TRUE_FG, 80, 24, orc.PALETTE_STANDARD) for im in ticks[0]]
    for round_ in range(4):
        table = pkg.FrameTable(n)
        table.publish_sampled_batch(slots, [(C.addressof(b), len(b)) for b in bufs[0]], [tmpl], pub.cuda_stream)
        frames = (pkg.Frame * n)(*[pkg.Frame.from_buffer_copy(tmpl) for _ in range(n)])
        assert table.latest_frames(slots, frames, cons.cuda_stream) == n
        first_src = frames[0].src
        busy = pkg.Plan(MODE_HB_TRUE, orc.PALETTE_STANDARD,
                        [pkg.frame_setup(busy_src.data_ptr(), 3840, 2160, 400, 120, 2, False, False, False)] * 256)
        bout = torch.empty(256 * busy.stride, dtype=torch.uint8, device="cuda")
        bln = torch.zeros(256, dtype=torch.int32, device="cuda")
        for _ in range(8):
            busy.render(bout.data_ptr(), busy.stride, bln.data_ptr(), cons.cuda_stream)
        plan = pkg.Plan(MODE_TRUE_FG, orc.PALETTE_STANDARD, list(frames))
        out = torch.zeros(n * plan.stride, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(n, dtype=torch.int32, device="cuda")
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), cons.cuda_stream)   # queued, not yet running
        for t in range(1, 5):  # four more ticks: the fourth lands in tick A's block
            table.publish_sampled_batch(slots, [(C.addressof(b), len(b)) for b in bufs[t]], [tmpl], pub.cuda_stream)
        torch.cuda.synchronize()
        host, lens = out.cpu().numpy(), ln.cpu().numpy()
        for k in range(n):
            assert host[k * plan.stride:k * plan.stride + int(lens[k])].tobytes() == exp[k], (round_, k)
        again = (pkg.Frame * n)(*[pkg.Frame.from_buffer_copy(tmpl) for _ in range(n)])
        assert table.latest_frames(slots, again, cons.cuda_stream) == n and again[0].src == first_src  # the ring came round
        got = render_descs(gpu, MODE_TRUE_FG, list(again))
        for k in range(n):
            assert got[k] == oracle_convert(ticks[4][k], MODE_TRUE_FG, 80, 24, orc.PALETTE_STANDARD), (round_, k)
        table.forget_stream(cons.cuda_stream)
        plan.close()
        busy.close()
        table.close()


def test_frame_table_upload_waits_for_queued_readers(gpu):
    """the publish after next on a slot overwrites the buffer that renders handed out by latest() may still
    be reading.  A consumer stream is kept busy, a render of frame A is queued behind that work, then B and C are
    published on another stream (C lands in A's buffer): the upload must go behind the queued render."""
    import struct
    pkg, torch = gpu
    A, B, Cc = (orc.frame_hash_noise(1920, 1080, 900 + i) for i in range(3))

    def blob_of(img):
        return struct.pack(">II", img.shape[1], img.shape[0]) + np.ascontiguousarray(img).tobytes()

    blobs = [blob_of(A), blob_of(B), blob_of(Cc)]
    exp = oracle_convert(A, MODE_TRUE_FG, 80, 24, orc.PALETTE_STANDARD)
    busy_src = torch.from_numpy(np.ascontiguousarray(orc.frame_hash_noise(3840, 2160, 5))).cuda()
    pub, cons = torch.cuda.Stream(), torch.cuda.Stream()
    for round_ in range(6):
        table = pkg.FrameTable(1)
        table.publish(0, blobs[0], pub.cuda_stream)
        ptr, w, h, gen = table.latest(0, cons.cuda_stream)
        # keep the consumer stream busy: 4K -> 400x120 half-block batches (~250 us each)
        busy = pkg.Plan(MODE_HB_TRUE, orc.PALETTE_STANDARD,
                        [pkg.frame_setup(busy_src.data_ptr(), 3840, 2160, 400, 120, 2, False, False, False)] * 256)
        bout = torch.empty(256 * busy.stride, dtype=torch.uint8, device="cuda")
        bln = torch.zeros(256, dtype=torch.int32, device="cuda")
        for _ in range(8):
            busy.render(bout.data_ptr(), busy.stride, bln.data_ptr(), cons.cuda_stream)
        plan = pkg.Plan(MODE_TRUE_FG, orc.PALETTE_STANDARD, [pkg.frame_setup(ptr, w, h, 80, 24, 0, False, False, False)] * 64)
        out = torch.zeros(64 * plan.stride, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(64, dtype=torch.int32, device="cuda")
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), cons.cuda_stream)   # queued, not yet running
        table.publish(0, blobs[1], pub.cuda_stream)
        table.publish(0, blobs[2], pub.cuda_stream)                               # overwrites A's buffer
        torch.cuda.synchronize()
        host, lens = out.cpu().numpy(), ln.cpu().numpy()
        for k in range(64):
            assert host[k * plan.stride:k * plan.stride + int(lens[k])].tobytes() == exp, (round_, k)
        p2, w2, h2, g2 = table.latest(0, cons.cuda_stream)
        assert g2 == 3 and p2 == ptr  # double buffering: C sits where A was
        table.forget_stream(cons.cuda_stream)
        plan.close()
        busy.close()
        table.close()


def test_frame_table_batch_publish_waits_for_queued_readers_and_mixes_with_single(gpu):
    """The batch form of the test above: a tick's buffers are guarded by ONE event per batch (not one per slot) and the
    reader streams of all its slots are waited for once each.  A render of tick A's frames is queued behind a busy consumer
    stream; ticks B and C are batch-published on another stream (C overwrites A's buffers): the render must still see A.
    Then single-slot and batch publishes alternate on one slot (own event <-> batch event hand-over), and more batches than
    the event ring holds."""
    import struct
    pkg, torch = gpu
    n = 6
    w, h = 1920, 1080
    ticks = [[orc.frame_hash_noise(w, h, 4000 + 10 * t + i) for i in range(n)] for t in range(3)]
    bufs = [[C.create_string_buffer(struct.pack(">II", w, h) + np.ascontiguousarray(im).tobytes(), 8 + im.size) for im in tk]
            for tk in ticks]
    tmpl = pkg.frame_setup(None, w, h, 80, 24, 0, False, False, False)
    slots = (C.c_int * n)(*range(n))
    busy_src = torch.from_numpy(np.ascontiguousarray(orc.frame_hash_noise(3840, 2160, 5))).cuda()
    pub, cons = torch.cuda.Stream(), torch.cuda.Stream()

    def publish(table, t):
        table.publish_rows_batch(slots, [(C.addressof(b), len(b)) for b in bufs[t]], [tmpl], pub.cuda_stream)

    exp = [oracle_convert(im, MODE_TRUE_FG, 80, 24, orc.PALETTE_STANDARD) for im in ticks[0]]
    for round_ in range(4):
        table = pkg.FrameTable(n)
        publish(table, 0)
        frames = (pkg.Frame * n)(*[pkg.Frame.from_buffer_copy(tmpl) for _ in range(n)])
        assert table.latest_frames(slots, frames, cons.cuda_stream) == n
        busy = pkg.Plan(MODE_HB_TRUE, orc.PALETTE_STANDARD,
                        [pkg.frame_setup(busy_src.data_ptr(), 3840, 2160, 400, 120, 2, False, False, False)] * 256)
        bout = torch.empty(256 * busy.stride, dtype=torch.uint8, device="cuda")
        bln = torch.zeros(256, dtype=torch.int32, device="cuda")
        for _ in range(8):
            busy.render(bout.data_ptr(), busy.stride, bln.data_ptr(), cons.cuda_stream)
        plan = pkg.Plan(MODE_TRUE_FG, orc.PALETTE_STANDARD, list(frames))
        out = torch.zeros(n * plan.stride, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(n, dtype=torch.int32, device="cuda")
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), cons.cuda_stream)   # queued, not yet running
        publish(table, 1)
        publish(table, 2)                                                           # overwrites tick A's buffers
        torch.cuda.synchronize()
        host, lens = out.cpu().numpy(), ln.cpu().numpy()
        for k in range(n):
            assert host[k * plan.stride:k * plan.stride + int(lens[k])].tobytes() == exp[k], (round_, k)
        table.forget_stream(cons.cuda_stream)
        plan.close()
        busy.close()
        table.close()
    # own event <-> batch event on the same slot, and more batches than the ring of batch events holds
    table = pkg.FrameTable(n)
    st = torch.cuda.current_stream().cuda_stream
    for step in range(24):
        t = step % 3
        if step % 4 == 3:
            table.publish(0, bufs[t][0].raw[:8 + w * h * 3], st)
            for i in range(1, n):
                table.publish_rows(i, (C.addressof(bufs[t][i]), len(bufs[t][i])), [tmpl], st)
        else:
            publish(table, t)
        frames = (pkg.Frame * n)(*[pkg.Frame.from_buffer_copy(tmpl) for _ in range(n)])
        assert table.latest_frames(slots, frames, st) == n
        got = render_descs(gpu, MODE_TRUE_FG, list(frames))
        for k in range(n):
            assert got[k] == oracle_convert(ticks[t][k], MODE_TRUE_FG, 80, 24, orc.PALETTE_STANDARD), (step, k)
    table.close()


def test_render_packets_packed_one_pass(gpu):
    """plan_render_packets_packed / frame_packets_packed: checksums, headers, packet CRCs AND the frames at their exact lengths
    in mapped host memory.  Behind a fused render it is render + pack; behind any other plan ONE pass over the slab
    checksums and packs (crc_kernels.hpp COPY instantiations: the one-workgroup-per-frame kernel for frames up to 128 KB, the
    span kernels above).  Everything must equal plan_render_packets + pack_frames."""
    pkg, torch = gpu
    stream = torch.cuda.current_stream().cuda_stream
    src = torch.from_numpy(np.ascontiguousarray(orc.frame_hash_noise(1920, 1080, 77))).cuda()
    big = torch.from_numpy(np.ascontiguousarray(orc.frame_hash_noise(3840, 2160, 78))).cuda()
    cases = [  # mode, render_mode, source, (w, h) list, forced variant
        (1, 0, src, [(80, 24), (60, 7), (1, 1), (40, 30), (80, 24), (33, 11)], -1),  # frames <= 48 KB: ONE launch, exact lengths
        (2, 0, src, [(80, 24), (100, 20), (5, 5)], 17),                 # ... also from a 512-thread plan (the launch is geometry 16)
        (1, 0, src, [(80, 24), (60, 7), (1, 1), (132, 43)], 17),        # a 136 KB bound: length-first (forced below) + checksums in place
        (5, 2, src, [(80, 24), (100, 37), (33, 17), (80, 24)], -1),     # rows kernel / bands: one pass, frames < 128 KB
        (0, 0, src, [(80, 24), (200, 60), (10, 5)], -1),
        (5, 2, big, [(400, 120), (380, 100), (80, 24)], -1),            # 1.8 MB frames: the span kernels
    ]
    for case_no, (mode, rm, img, dims, variant) in enumerate(cases):
        ih, iw = img.shape[0], img.shape[1]
        frames = [pkg.frame_setup(img.data_ptr(), iw, ih, w, h, rm, False, False, False) for (w, h) in dims]
        plan = pkg.Plan(mode, orc.PALETTE_STANDARD, frames)
        if variant >= 0:
            plan.set_variant(variant)
        n, stride = len(frames), plan.stride
        d32 = torch.tensor(dims, dtype=torch.int32, device="cuda")

        def bufs():
            return (torch.full((n * stride,), 0xEE, dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"),
                    torch.zeros(n, dtype=torch.int32, device="cuda"), torch.zeros(n * 24, dtype=torch.uint8, device="cuda"),
                    torch.zeros(n, dtype=torch.int32, device="cuda"))

        tab = (8 * (n + 1) + 4 * n + 15) // 16 * 16
        out_a, ln_a, crc_a, hdr_a, pkt_a = bufs()
        hb_a = pkg.HostBuffer(tab + n * stride)
        plan.render_packets(out_a.data_ptr(), stride, ln_a.data_ptr(), d32.data_ptr(), crc_a.data_ptr(), hdr_a.data_ptr(),
                            pkt_a.data_ptr(), stream)
        pkg.pack_frames(out_a.data_ptr(), stride, ln_a.data_ptr(), n, hb_a.dev + tab, n * stride, hb_a.dev, hb_a.dev + 8 * (n + 1), stream)
        out_b, ln_b, crc_b, hdr_b, pkt_b = bufs()
        hb_b = pkg.HostBuffer(tab + n * stride)
        plan.set_exact_length(1)  # (the automatic choice keeps mapped host destinations on the two-launch form)
        plan.render_packets_packed(out_b.data_ptr(), stride, ln_b.data_ptr(), d32.data_ptr(), crc_b.data_ptr(), hdr_b.data_ptr(),
                                   pkt_b.data_ptr(), hb_b.dev + tab, n * stride, hb_b.dev, hb_b.dev + 8 * (n + 1), stream)
        torch.cuda.synchronize()
        assert torch.equal(ln_a, ln_b) and torch.equal(crc_a, crc_b) and torch.equal(hdr_a, hdr_b) and torch.equal(pkt_a, pkt_b), (mode, dims)
        va, vb = hb_a.view(), hb_b.view()
        off_a, off_b = va[:8 * (n + 1)].view(np.uint64), vb[:8 * (n + 1)].view(np.uint64)
        la = va[8 * (n + 1):8 * (n + 1) + 4 * n].view(np.uint32)
        assert np.array_equal(la, vb[8 * (n + 1):8 * (n + 1) + 4 * n].view(np.uint32))
        one_launch = plan.exact_length or plan.length_first  # (frames beyond 48 KB: the length-first form, round 6)
        assert plan.exact_length == (case_no < 2) and plan.length_first == (case_no == 2), (case_no, plan.stride)
        if one_launch:
            # the render wrote the frames itself: the slab was never touched, the frames tile the destination in SOME order
            assert bool((out_b == 0xEE).all())
            spans = sorted((int(off_b[i]), int(off_b[i]) + (int(la[i]) + 15) // 16 * 16) for i in range(n))
            assert spans[0][0] == 0 and all(spans[i][1] == spans[i + 1][0] for i in range(n - 1)) and spans[-1][1] == int(off_b[n])
            assert int(off_b[n]) == int(off_a[n])
        else:
            assert np.array_equal(off_a, off_b)
        ih_np = img.cpu().numpy()
        for i, (w, h) in enumerate(dims):
            a = va[tab + int(off_a[i]):tab + int(off_a[i]) + int(la[i])].tobytes()
            assert a == vb[tab + int(off_b[i]):tab + int(off_b[i]) + int(la[i])].tobytes(), (mode, i)
            if w * h <= 200 * 60:  # (the oracle takes its time on the largest frames: they are compared with the two-pass form)
                cl = {1: 3, 2: 2, 5: 3, 0: 0}[mode]
                assert a == orc.convert_with_caps(ih_np, w, h, cl, rm, False, False, False), (mode, i)
        hb_a.close()
        hb_b.close()
        plan.close()
