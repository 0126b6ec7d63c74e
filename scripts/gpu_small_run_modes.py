#!/usr/bin/env python3
"""Small launches of the run-structured modes: whole frames on the rows kernel against row bands of the phase kernel, one and
eight 80x24 frames per launch (the policy's rule for frames of one block per wave was measured on mono only).  GPU box only."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, orc
from __graft_entry__ import load_package
pkg = load_package(); torch.cuda.set_device(0); cur = torch.cuda.current_stream()
def time_plan(plan, n, reps=300):
    out = torch.empty(n * plan.stride, dtype=torch.uint8, device="cuda"); ln = torch.zeros(n, dtype=torch.int32, device="cuda")
    for _ in range(20): plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), cur.cuda_stream)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for _ in range(reps): plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), cur.cuda_stream)
        e1.record(cur); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts), out, ln.cpu().numpy().astype("uint32")
MODES = ((0, "mono", 0, 0), (5, "half-block truecolor", 3, 2), (6, "half-block 256", 2, 2), (7, "half-block 16", 1, 2), (8, "half-block mono", 0, 2))
SIZES = [(640, 480, 80, 24), (1920, 1080, 80, 24), (1920, 1080, 120, 40)]
if os.environ.get("SMALL_SIZES"):  # e.g. SMALL_SIZES=160x48,200x60 (1080p sources): rows beyond the shared-out geometry 31
    SIZES = [(1920, 1080, int(t.split("x")[0]), int(t.split("x")[1])) for t in os.environ["SMALL_SIZES"].split(",")]
for (sw, sh, W, H) in SIZES:
    for nb in ((1, 8, 64) if len(sys.argv) < 2 else tuple(int(a) for a in sys.argv[1:])):
        imgs = bench.make_frames(torch, nb, sw, sh, 5)
        host0 = np.ascontiguousarray(imgs[0].cpu().numpy())
        for mode, nm, cl, rm in MODES:
            fr = [pkg.frame_setup(imgs.data_ptr() + i * sw * sh * 3, sw, sh, W, H, rm, False, False, False) for i in range(nb)]
            want = orc.convert_with_caps(host0, W, H, cl, rm, False, False, False)
            row = []
            labels = (("auto", -1, None), ("rows 25 whole", 25, -1), ("phase 4 whole", 4, -1), ("bands 1", 4, 1), ("bands 2", 4, 2), ("bands 3", 4, 3), ("bands 4", 4, 4), ("bands 6", 4, 6))
            if os.environ.get("ONLY_AUTO"):  # round 6: the shared-out rows form (geometry 31) under ASCIICHAT_HIP_ROWS_PARTS=1 / N
                labels = labels[:1]
            for label, variant, split in labels:
                plan = pkg.Plan(mode, bench.PALETTE_STANDARD, fr)
                try:
                    if split is not None: plan.set_split(split)
                    if variant >= 0: plan.set_variant(variant)
                except RuntimeError:
                    row.append(f"{label} n/a"); plan.close(); continue
                t, out, lens = time_plan(plan, nb)
                ok = bytes(out[:int(lens[0])].cpu().numpy()) == want
                row.append(f"{label} {t:6.2f}{'' if ok else ' WRONG'} (v{plan.variant} p{plan.parts})")
                plan.close()
            print(f"{nb:3d} x ({sw}x{sh} -> {W}x{H} {nm:22s}): " + " | ".join(row))
