#!/usr/bin/env python3
"""Randomised calls of the batch C-ABI's stand-alone entry points on the GPU box, each checked against the oracle:
asciichat_hip_crc32c (random lengths / strides incl. 0 and multi-span sizes), the frame table (valid and invalid blobs),
asciichat_hip_resize, asciichat_hip_composite, asciichat_hip_apply_color_filter, asciichat_hip_image_flip."""
import ctypes as C
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from test_random_differential import random_image  # noqa: E402


def main():
    import torch

    from __graft_entry__ import load_package

    pkg = load_package()
    L = pkg.lib()
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    st = torch.cuda.current_stream().cuda_stream
    n_crc = n_tab = n_img = 0
    for rnd in range(rounds):
        # ---- CRC-32C over random buffers ----
        n = int(rng.integers(1, 12))
        mx = int(rng.choice([1, 15, 16, 17, 4095, 4096, 4097, 40000, 131072, 131073, 300001]))
        lens = rng.integers(0, mx + 1, n).astype(np.uint32)
        lens[int(rng.integers(0, n))] = mx
        stride = (mx + 15 + 16 * int(rng.integers(0, 3))) & ~15
        buf = rng.integers(0, 256, n * stride, dtype=np.uint8)
        d = torch.from_numpy(buf).cuda()
        dl = torch.from_numpy(lens.view(np.int32)).cuda()
        crc = torch.zeros(n, dtype=torch.int32, device="cuda")
        use_len = bool(rng.integers(0, 2))
        rc = L.asciichat_hip_crc32c(d.data_ptr(), stride, dl.data_ptr() if use_len else None, 0 if use_len else mx, mx, n,
                                    crc.data_ptr(), st)
        assert rc == 0, pkg.last_error()
        torch.cuda.synchronize()
        got = crc.cpu().numpy().astype(np.uint32)
        for k in range(n):
            ln = int(lens[k]) if use_len else mx
            assert int(got[k]) == orc.crc32c(buf[k * stride:k * stride + ln].tobytes()), ("crc", rnd, k, ln, mx, use_len)
            n_crc += 1
        # ---- frame table ----
        table = pkg.FrameTable(3)
        for _ in range(6):
            slot = int(rng.integers(0, 3))
            w, h = int(rng.choice([0, 1, 2, 64, 333, 3840, 3841, 4097])), int(rng.choice([0, 1, 2, 48, 201, 2160, 2161]))
            real = w * h * 3 if 0 < w * h <= 4000 * 300 else 12
            extra = int(rng.choice([0, 0, 5, -1]))
            payload = rng.integers(0, 256, max(0, real + extra), dtype=np.uint8).tobytes()
            blob = struct.pack(">II", w, h) + payload
            want = orc.frame_blob_accept(blob, False)
            before = table.latest(slot, st)
            try:
                table.publish(slot, blob, st)
                ok = True
            except RuntimeError:
                ok = False
            assert ok == (want is not None), ("blob", w, h, len(blob))
            after = table.latest(slot, st)
            if ok:
                assert (after[1], after[2], after[3]) == (w, h, before[3] + 1)
                img = np.frombuffer(payload[:w * h * 3], dtype=np.uint8).reshape(h, w, 3)
                f = pkg.frame_setup(after[0], w, h, 40, 12, 0)
                plan = pkg.Plan(1, orc.PALETTE_STANDARD, [f])
                out = torch.zeros(plan.stride, dtype=torch.uint8, device="cuda")
                l1 = torch.zeros(1, dtype=torch.int32, device="cuda")
                plan.render(out.data_ptr(), plan.stride, l1.data_ptr(), st)
                torch.cuda.synchronize()
                assert out[:int(l1[0].item())].cpu().numpy().tobytes() == orc.convert_with_caps(img, 40, 12, 3, 0)
                plan.close()
            else:
                assert after == before
            n_tab += 1
        table.close()
        # ---- image-space entry points ----
        sw, sh = int(rng.integers(1, 700)), int(rng.integers(1, 400))
        img = random_image(rng, sw, sh)
        src = torch.from_numpy(np.ascontiguousarray(img)).cuda()
        dw, dh = int(rng.integers(1, 500)), int(rng.integers(1, 300))
        dst = torch.zeros(dh * dw * 3, dtype=torch.uint8, device="cuda")
        assert L.asciichat_hip_resize(src.data_ptr(), sw, sh, dst.data_ptr(), dw, dh, st) == 0
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy().reshape(dh, dw, 3), orc.resize_nn(img, dw, dh)), ("resize", sw, sh, dw, dh)
        flt = int(rng.integers(0, 12))
        work = src.clone()
        assert L.asciichat_hip_apply_color_filter(work.data_ptr(), sw, sh, 3 * sw, flt, st) == 0
        torch.cuda.synchronize()
        assert np.array_equal(work.cpu().numpy(), orc.color_filter(img, flt)), ("filter", sw, sh, flt)
        fx, fy = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        flipped = torch.zeros_like(src)
        assert L.asciichat_hip_image_flip(src.data_ptr(), flipped.data_ptr(), sw, sh, fx, fy, st) == 0
        torch.cuda.synchronize()
        assert np.array_equal(flipped.cpu().numpy(), orc.flip(img, fx, fy)), ("flip", sw, sh, fx, fy)
        k = int(rng.integers(1, 10))
        imgs = [random_image(rng, int(rng.integers(8, 500)), int(rng.integers(8, 300))) for _ in range(k)]
        devs = [torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in imgs]
        tw, th = int(rng.integers(20, 200)), int(rng.integers(8, 70))
        comp = pkg.Composite()
        L.achip_composite_setup(C.byref(comp), (C.c_void_p * k)(*[d_.data_ptr() for d_ in devs]),
                                (C.c_int * k)(*[i.shape[1] for i in imgs]), (C.c_int * k)(*[i.shape[0] for i in imgs]), k, tw, th)
        canvas = torch.zeros(2 * th * tw * 3, dtype=torch.uint8, device="cuda")
        assert L.asciichat_hip_composite(C.byref(comp), canvas.data_ptr(), None) == 0
        assert np.array_equal(canvas.cpu().numpy().reshape(2 * th, tw, 3), orc.composite(imgs, tw, th)), ("composite", k, tw, th)
        n_img += 4
    print(f"api fuzz OK: {n_crc} CRCs, {n_tab} frame-table publishes, {n_img} image-space calls match the oracle")


if __name__ == "__main__":
    main()
