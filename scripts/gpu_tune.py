#!/usr/bin/env python3
"""GPU tuning aid: per-workload, per-geometry kernel time and per-phase cycle breakdown.
Usage (on the GPU box): python scripts/gpu_tune.py [--batch 256] [--workloads a,b] > gpurun_out/tune.txt"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--workloads", default="1080p_80x24_truecolor,1080p_80x24_ansi256,4k_200x60_truecolor,4k_400x120_halfblock")
    ap.add_argument("--variants", default="0,1,2")
    ap.add_argument("--splits", default="-1", help="rows per workgroup: -1 whole frames, 0 automatic, n = n text rows")
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    import torch

    from __graft_entry__ import load_package

    pkg = load_package()
    torch.cuda.set_device(0)
    names = ["setup", "gather", "heads", "lengths", "scan", "emit", "drain", "total"]
    for wl in args.workloads.split(","):
        sw, sh, W, H, cl, rm = bench.WORKLOADS[wl]
        frames_t = bench.make_frames(torch, args.batch, sw, sh, 1234)
        plan, mode = bench.build_plan(pkg, frames_t, W, H, cl, rm)
        out = torch.empty(args.batch * plan.stride, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(args.batch, dtype=torch.int32, device="cuda")
        prof = torch.zeros(args.batch * 8, dtype=torch.int64, device="cuda")
        for v in [int(x) for x in args.variants.split(",")]:
          for sp in [int(x) for x in args.splits.split(",")]:
            try:
                plan.set_variant(v)
                plan.set_split(sp)
            except RuntimeError as e:
                print(f"{wl} variant {v} split {sp}: skipped ({e})")
                continue
            for _ in range(5):
                plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            ms = bench.kernel_time_events(torch, [plan], out, ln, args.reps)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), torch.cuda.current_stream().cuda_stream)
            e1.record()
            torch.cuda.synchronize()
            b2b = e0.elapsed_time(e1) / args.reps
            lens = ln.cpu().numpy().astype("uint32")
            assert (lens < 0xFFFFFFF0).all(), "kernel reported an error code"
            rows = 2 * H if rm == 2 else H
            alg = int(lens.sum()) + args.batch * 3 * W * rows
            head = (f"{wl:26s} v{plan.variant} parts {plan.parts:3d} (req v{v} split {sp:2d}) kernel {ms*1e3:8.1f} us "
                    f"(b2b {b2b*1e3:8.1f} us) alg {alg/1e6:8.2f} MB -> {alg/(ms*1e-3)/1e9:7.1f} GB/s")
            if plan.parts > 1:
                print(head)
                continue
            plan.render_profiled(out.data_ptr(), plan.stride, ln.data_ptr(), prof.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            p = prof.cpu().numpy().reshape(args.batch, 8).astype("float64")
            mean = p.mean(axis=0)
            mx = p.max(axis=0)
            print(head + "  | cycles/frame mean: " +
                  " ".join(f"{n}={int(m)}" for n, m in zip(names, mean[:8])) + f" | max total={int(mx[7])}")
        plan.close()
        del frames_t, out
        torch.cuda.empty_cache()


def stream_passes():
    """HBM-streaming neighbours: colour filter (in place) and flips (out of place) on a batch of 1080p frames."""
    import torch

    from __graft_entry__ import load_package

    pkg = load_package()
    L = pkg.lib()
    b, h, w = 64, 1080, 1920
    src = torch.randint(0, 256, (b, h, w, 3), dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    nbytes = b * h * w * 3
    stream = torch.cuda.current_stream().cuda_stream

    def timeit(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    # the whole batch is one tightly packed buffer of b*h rows
    ms = timeit(lambda: L.asciichat_hip_apply_color_filter(src.data_ptr(), w, b * h, 3 * w, 3, stream))
    print(f"color_filter  {b}x1080p in place : {ms*1e3:8.1f} us  {2*nbytes/(ms*1e-3)/1e9:7.1f} GB/s (read+write) "
          f"= {2*nbytes/(ms*1e-3)/8e12*100:4.1f} % of 8 TB/s")
    for fx, fy in ((1, 0), (0, 1), (1, 1)):
        ms = timeit(lambda: L.asciichat_hip_image_flip(src.data_ptr(), dst.data_ptr(), w, b * h, fx, fy, stream))
        print(f"flip x={fx} y={fy} {b}x1080p          : {ms*1e3:8.1f} us  {2*nbytes/(ms*1e-3)/1e9:7.1f} GB/s (read+write) "
              f"= {2*nbytes/(ms*1e-3)/8e12*100:4.1f} % of 8 TB/s")
    ms = timeit(lambda: dst.copy_(src))
    print(f"torch copy_ (reference point)      : {ms*1e3:8.1f} us  {2*nbytes/(ms*1e-3)/1e9:7.1f} GB/s")


def wire_stage():
    """CRC-32C + packet headers: on a rendered K2 slab, and on 1080p payloads (HBM-bound)."""
    import torch

    from __graft_entry__ import load_package

    pkg = load_package()
    L = pkg.lib()
    stream = torch.cuda.current_stream().cuda_stream

    def timeit(fn, reps=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for wl in ("1080p_80x24_truecolor", "4k_200x60_truecolor"):
        sw, sh, W, H, cl, rm = bench.WORKLOADS[wl]
        b = 256
        frames_t = bench.make_frames(torch, b, sw, sh, 1234)
        plan, mode = bench.build_plan(pkg, frames_t, W, H, cl, rm)
        out = torch.empty(b * plan.stride, dtype=torch.uint8, device="cuda")
        ln = torch.zeros(b, dtype=torch.int32, device="cuda")
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), stream)
        torch.cuda.synchronize()
        total = int(ln.cpu().numpy().astype("uint32").sum())
        dims = torch.tensor([[W, H]] * b, dtype=torch.int32, device="cuda")
        crc = torch.zeros(b, dtype=torch.int32, device="cuda")
        hdr = torch.zeros(b * 24, dtype=torch.uint8, device="cuda")
        pkt = torch.zeros(b, dtype=torch.int32, device="cuda")
        ms = timeit(lambda: L.asciichat_hip_frame_packets(out.data_ptr(), plan.stride, ln.data_ptr(), plan.stride, b,
                                                          dims.data_ptr(), crc.data_ptr(), hdr.data_ptr(), pkt.data_ptr(), stream))
        ms_r = timeit(lambda: plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), stream))
        print(f"frame_packets {wl} b{b}: {ms*1e3:8.1f} us for {total/1e6:.2f} MB of frames -> {total/(ms*1e-3)/1e9:7.1f} GB/s "
              f"(render alone {ms_r*1e3:.1f} us)")
        plan.close()
        del frames_t, out
    for nb in (8, 64):
        nbytes = 1920 * 1080 * 3
        buf = torch.randint(0, 256, (nb * nbytes,), dtype=torch.uint8, device="cuda")
        crc = torch.zeros(nb, dtype=torch.int32, device="cuda")
        ms = timeit(lambda: L.asciichat_hip_crc32c(buf.data_ptr(), nbytes, None, nbytes, nbytes, nb, crc.data_ptr(), stream), 20)
        print(f"crc32c {nb} x 1080p payload: {ms*1e3:8.1f} us -> {nb*nbytes/(ms*1e-3)/1e9:7.1f} GB/s = "
              f"{nb*nbytes/(ms*1e-3)/8e12*100:4.1f} % of 8 TB/s")


def dropin_latency():
    """End-to-end latency of the reference's own call (ascii_convert_with_capabilities, host image in, malloc'd
    string out) through libasciichat_hip.so, next to the CPU oracle on the same host."""
    import ctypes as C
    import time

    import numpy as np
    import torch  # noqa: F401

    from __graft_entry__ import load_package
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc

    pkg = load_package()
    L = pkg.lib()
    pal = orc.PALETTE_STANDARD.encode()
    for (sw, sh, W, H, cl, rm, name) in ((1920, 1080, 80, 24, 3, 0, "1080p->80x24 truecolor"), (3840, 2160, 200, 60, 3, 0, "4K->200x60 truecolor"),
                                       (3840, 2160, 400, 120, 3, 2, "4K->400x120 half-block"), (640, 480, 80, 24, 0, 0, "640x480->80x24 mono")):
        img = orc.frame_hash_noise(sw, sh, 5)
        caps = pkg.TermCaps()
        caps.color_level, caps.render_mode, caps.utf8_support = cl, rm, True
        arr = np.ascontiguousarray(img)
        pageable = pkg.Image(sw, sh, arr.ctypes.data, 0)
        L.image_new_from_pool.restype = C.POINTER(pkg.Image)
        pooled = L.image_new_from_pool(sw, sh)
        C.memmove(pooled.contents.pixels, arr.ctypes.data, arr.size)

        def call(im):
            p = L.ascii_convert_with_capabilities(im, W, H, C.byref(caps), False, False, pal)
            s = pkg.take_string(p)
            return s

        exp = orc.convert_with_caps(img, W, H, cl, rm)
        res = {}
        for label, im in (("pageable image", C.byref(pageable)), ("pool (pinned) image", pooled)):
            assert call(im) == exp
            for _ in range(20):
                call(im)
            t0 = time.perf_counter()
            reps = 200
            for _ in range(reps):
                call(im)
            res[label] = (time.perf_counter() - t0) / reps * 1e6
        t0 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            orc.convert_with_caps(img, W, H, cl, rm)
        cpu = (time.perf_counter() - t0) / reps * 1e6
        print(f"drop-in {name:26s}: " + "  ".join(f"{k} {v:7.1f} us" for k, v in res.items()) + f"   CPU oracle (1 thread) {cpu:8.1f} us")
        L.image_destroy_to_pool(pooled)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--stream-passes":
        stream_passes()
    elif len(sys.argv) > 1 and sys.argv[1] == "--wire-stage":
        wire_stage()
    elif len(sys.argv) > 1 and sys.argv[1] == "--dropin":
        dropin_latency()
    else:
        main()
