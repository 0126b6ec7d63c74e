#!/usr/bin/env python3
"""Small batches: BASELINE configs[0] (ONE 640x480 -> 80x24 mono frame) and the nine-target grid
of configs[3], one launch at a time, through every geometry that can carry them -- the automatic choice (row bands of the
phase kernel), whole frames on the phase kernel, and whole frames on the wave-autonomous kernels (rows kernel for mono,
stream kernel for truecolor).  GPU time per launch by HIP events over back-to-back launches on one stream.  GPU box only."""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import orc  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
L = pkg.lib()
torch.cuda.set_device(0)
cur = torch.cuda.current_stream()


def time_plan(plan, n, reps=300):
    out = torch.empty(n * plan.stride, dtype=torch.uint8, device="cuda")
    ln = torch.zeros(n, dtype=torch.int32, device="cuda")
    for _ in range(20):
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), cur.cuda_stream)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), cur.cuda_stream)
        e0.record(cur)
        for _ in range(reps):
            plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), cur.cuda_stream)
        e1.record(cur)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    lens = ln.cpu().numpy().astype("uint32")
    assert (lens < 0xFFFFFFF0).all()
    return statistics.median(ts), out, lens


def sweep(title, mode, descs, palette, expect, combos):
    print(f"# {title}")
    for label, variant, split in combos:
        plan = pkg.Plan(mode, palette, descs)
        try:
            if split is not None:
                plan.set_split(split)
            if variant >= 0:
                plan.set_variant(variant)
        except RuntimeError as e:
            print(f"  {label:34s}: not available ({str(e)[:60]})")
            plan.close()
            continue
        t, out, lens = time_plan(plan, len(descs))
        got = bytes(out[:int(lens[0])].cpu().numpy())
        print(f"  {label:34s}: {t:7.2f} us per launch   variant {plan.variant} parts {plan.parts}   bytes ok: {got == expect}")
        plan.close()


# K1: one 640x480 frame -> 80x24 mono, stretch (ascii_convert's call, host.c:696)
img = bench.make_frames(torch, 1, 640, 480, 77)[0]
host = np.ascontiguousarray(img.cpu().numpy())
f = pkg.frame_setup(img.data_ptr(), 640, 480, 80, 24, 0, False, False, False)
exp = orc.convert_with_caps(host, 80, 24, 0, 0, False, False, False)
sweep("configs[0]: 640x480 -> 80x24 mono, ONE frame per launch", 0, [f], bench.PALETTE_STANDARD, exp,
      [("automatic", -1, None), ("phase kernel 4, whole frame", 4, -1), ("phase kernel 0, whole frame", 0, -1),
       ("rows kernel 25 (4 slots)", 25, -1), ("rows kernel 24 (7 slots)", 24, -1),
       ("phase kernel 4, 12-row bands", 4, 12), ("phase kernel 4, 6-row bands", 4, 6), ("phase kernel 4, 2-row bands", 4, 2)])
# the same as truecolor (stream kernel)
exp = orc.convert_with_caps(host, 80, 24, 3, 0, False, False, False)
sweep("640x480 -> 80x24 truecolor, ONE frame per launch", 1, [f], bench.PALETTE_STANDARD, exp,
      [("automatic", -1, None), ("stream 16", 16, -1), ("stream 17", 17, -1), ("stream 18", 18, -1), ("stream 19", 19, -1),
       ("phase kernel 4, whole frame", 4, -1)])
# K4: nine 1080p sources -> 3x3 grid at 160x48, nine target clients, sampled directly from the sources
n, sw, sh, tw, th = 9, 1920, 1080, 160, 48
grid = pkg.Grid(None, [(sw, sh)] * n, tw, th)
src = {k: bench.make_frames(torch, 1, sw, sh, 4321 + k)[0] for k in range(n)}
grid.set_direct(True)
grid.exchange({k: t.data_ptr() for k, t in src.items()}, cur.cuda_stream)
torch.cuda.synchronize()
descs = []
for _ in range(9):
    d = pkg.frame_setup(None, tw, 2 * th, tw, th, 0, True, True, False)
    d.comp = grid.composite_dev
    descs.append(d)
allsrc = [np.ascontiguousarray(src[k].cpu().numpy()) for k in range(n)]
exp = orc.convert_with_caps(orc.composite(allsrc, tw, th), tw, th, 3, 0, True, True, False)
sweep("configs[3]: nine targets of the 3x3 grid at 160x48 truecolor, one launch (render only, sources sampled directly)", 1,
      descs, bench.PALETTE_STANDARD, exp,
      [("automatic", -1, None), ("stream 16", 16, -1), ("stream 17", 17, -1), ("stream 18", 18, -1), ("stream 19", 19, -1),
       ("phase kernel 4, whole frame", 4, -1), ("phase kernel 4, 8-row bands", 4, 8), ("phase kernel 4, 4-row bands", 4, 4)])

# the policy change checked at batch sizes between one frame and a frame per CU: whole frames (automatic now) against the
# row bands the automatic choice used to make
for nb in (8, 64, 128):
    imgs = bench.make_frames(torch, nb, 640, 480, 99)
    fr = [pkg.frame_setup(imgs.data_ptr() + i * 640 * 480 * 3, 640, 480, 80, 24, 0, False, False, False) for i in range(nb)]
    host0 = np.ascontiguousarray(imgs[0].cpu().numpy())
    for mode, cl, label in ((1, 3, "truecolor"), (0, 0, "mono")):
        exp = orc.convert_with_caps(host0, 80, 24, cl, 0, False, False, False)
        bands = max(1, 24 // max(1, (256 + nb - 1) // nb))
        sweep(f"{nb} x (640x480 -> 80x24 {label}) per launch", mode, fr, bench.PALETTE_STANDARD, exp,
              [("automatic", -1, None), (f"phase kernel 4, {bands}-row bands", 4, bands), ("phase kernel 4, whole frame", 4, -1)])
