#!/usr/bin/env python3
"""Truecolor foreground between a quarter of a frame and a frame per CU, one launch at a time: the automatic choice (row bands of
the phase kernel below the whole-frame threshold) against whole frames on the stream kernel (geometry 16), 1080p sources.
Where is the crossover since the stream kernel's lean loop (round 6)?  GPU box only.  usage: gpu_truecolor_half_cu.py [share]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, orc
from __graft_entry__ import load_package
pkg = load_package(); torch.cuda.set_device(0); cur = torch.cuda.current_stream()
share = int(sys.argv[1]) if len(sys.argv) > 1 else 1
def time_plan(plan, n, reps=200):
    out = torch.empty(n * plan.stride, dtype=torch.uint8, device="cuda"); ln = torch.zeros(n, dtype=torch.int32, device="cuda")
    for _ in range(20): plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), cur.cuda_stream)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for _ in range(reps): plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), cur.cuda_stream)
        e1.record(cur); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts), out, ln.cpu().numpy().astype("uint32")
for mode, nm, cl in ((1, "truecolor", 3), (2, "ansi256", 2)):
    for (W, H) in ((120, 40), (160, 45), (200, 60), (238, 70), (320, 90)):
        for nb in (64, 80, 96, 112, 128, 160):
            imgs = bench.make_frames(torch, nb, 1920, 1080, 5)
            fr = [pkg.frame_setup(imgs.data_ptr() + i * 1920 * 1080 * 3, 1920, 1080, W, H, 0, False, False, False) for i in range(nb)]
            want = orc.convert_with_caps(np.ascontiguousarray(imgs[0].cpu().numpy()), W, H, cl, 0, False, False, False)
            row = []
            for label, variant in (("auto", -1), ("stream 16 whole", 16)):
                plan = pkg.Plan(mode, bench.PALETTE_STANDARD, fr)
                if share > 1: plan.set_concurrency(share)
                if variant >= 0: plan.set_variant(variant)
                t, out, lens = time_plan(plan, nb)
                ok = bytes(out[:int(lens[0])].cpu().numpy()) == want
                row.append(f"{label} {t:6.2f}{'' if ok else ' WRONG'} (v{plan.variant} p{plan.parts})")
                plan.close()
            print(f"{nm:9s} {W}x{H} {nb:4d} frames: " + " | ".join(row), flush=True)
            del imgs
