#!/usr/bin/env python3
"""Policy audit: does achip_choose_geometry pick the fastest geometry away from the five BASELINE shapes?  For a grid of
terminal sizes x batch sizes x modes (1080p sources; typical terminals, not only 80x24 / 200x60 / 400x120) every geometry
that can carry the plan is forced in turn -- whole frames on the stream / rows / phase kernels, row bands of the phase kernel
-- and timed like the automatic choice: HIP events over back-to-back launches on ONE stream (what a lone server tick is) and,
with --inflight, four plans in flight through bench.py's C-issued schedule (what a saturated server is).  Every forced geometry's frames are compared with the
automatic choice's on the GPU (all kernels must agree byte for byte; the automatic choice's first frame is checked against
the oracle).  Prints one line per case: the automatic choice, the best, the regret.  GPU box only.
--wide (round 6): rows beyond one block of the rows kernel -- the segment geometries 27 / 29 next to the phase kernel's.
usage: gpu_policy_audit.py [--inflight] [--quick] [--other-modes] [--4k] [--grid] [--dense] [--wide] [--modes=name,...]"""
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
INFLIGHT = "--inflight" in sys.argv
QUICK = "--quick" in sys.argv
streams = [None] * 4 if INFLIGHT else [torch.cuda.current_stream()]  # (--inflight: bench.Runner owns the lanes)


def time_plan(plans, n, stride, reps):
    if INFLIGHT:  # bench.py's own schedule: the steps issued from C (asciichat_hip_render_many), four lanes of ONE stream pool
        r = bench.Runner(torch, pkg, plans, n, len(plans))
        r.issue(2 * len(plans))
        torch.cuda.synchronize()
        ts = [r.gpu_ms_per_step(max(reps, 4 * len(plans))) * 1e3 for _ in range(3)]
        torch.cuda.synchronize()
        return statistics.median(ts), r.outs[0], r.lns[0]
    outs = [torch.empty(n * stride, dtype=torch.uint8, device="cuda") for _ in plans]
    lns = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in plans]

    def burst(k):
        for j in range(k):
            i = j % len(plans)
            plans[i].render(outs[i].data_ptr(), stride, lns[i].data_ptr(), streams[i].cuda_stream)

    burst(2 * len(plans))
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if len(plans) == 1:
            burst(1)
            e0.record(streams[0])
            burst(reps)
            e1.record(streams[0])
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
        else:  # wall clock over the burst: the streams' events do not order across streams
            import time
            t0 = time.perf_counter()
            burst(reps)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / reps * 1e6)
    return statistics.median(ts), outs[0], lns[0]


MODES = [(0, "mono", 0, 0), (2, "ansi256", 2, 0), (1, "truecolor", 3, 0), (5, "hb_true", 3, 2)]
if "--other-modes" in sys.argv:  # the rest of the dispatcher's table (ascii.c:955-1002)
    MODES = [(3, "ansi16", 1, 0), (6, "hb_256", 2, 2), (7, "hb_16", 1, 2), (8, "hb_mono", 0, 2)]
for a in sys.argv:  # --modes=hb_true,truecolor: only these rows of the table
    if a.startswith("--modes="):
        MODES = [m for m in MODES if m[1] in a[8:].split(",")]
SIZES = [(80, 24), (120, 40), (160, 45), (200, 60), (238, 70), (320, 90)]
BATCHES = [1, 4, 16, 64, 128, 192, 256]
if QUICK:
    SIZES, BATCHES = [(120, 40), (200, 60)], [1, 16, 256]
SRC_W, SRC_H = (3840, 2160) if "--4k" in sys.argv else (1920, 1080)
if "--4k" in sys.argv:
    SIZES = [(200, 60), (320, 90), (400, 120)]
WIDE = "--wide" in sys.argv
if WIDE:
    SIZES = [(512, 60), (640, 90), (1000, 40), (1920, 54)]
    BATCHES = [1, 8, 64, 128, 192, 256, 512]
if os.environ.get("AUDIT_SIZES"):  # targeted grids: AUDIT_SIZES=80x24,640x90 AUDIT_BATCHES=384,512,1024
    SIZES = [tuple(int(v) for v in t.split("x")) for t in os.environ["AUDIT_SIZES"].split(",")]
if os.environ.get("AUDIT_BATCHES"):
    BATCHES = [int(v) for v in os.environ["AUDIT_BATCHES"].split(",")]
DENSE = "--dense" in sys.argv
GRID = "--grid" in sys.argv  # the targets are composite frames: the 3x3 grid of nine 1080p sources (stream.c:523-854), sampled directly
frames_t = bench.make_frames(torch, 9 if GRID else 256, SRC_W, SRC_H, 4242)  # (larger batches repeat the sources)
host0 = np.ascontiguousarray(frames_t[0].cpu().numpy())
if GRID:
    BATCHES = [1, 4, 9, 32, 64, 256]
    SIZES = [(80, 24), (120, 40), (160, 48), (200, 60), (238, 70)]
    MODES = [(0, "mono", 0, 0), (1, "truecolor", 3, 0), (5, "hb_true", 3, 2)]
regrets = []
print(f"# {SRC_W}x{SRC_H} sources, {'four launches in flight through the C-issued schedule of bench.py (HIP events on every lane)' if INFLIGHT else 'one stream, back to back (HIP events)'}; us per launch")
for (mode, mname, cl, rm) in MODES:
    cell = mode in (1, 2, 3, 4)
    forced = ([("stream 16", 16, -1), ("stream 17", 17, -1), ("stream 18", 18, -1), ("stream 18 shared", 18, 0), ("stream 19", 19, -1)] if cell
              else [("rows 27", 27, -1), ("rows 29", 29, -1)] if WIDE else [("rows 25", 25, -1), ("rows 24", 24, -1), ("rows 26", 26, -1)])
    forced += [("phase 4 whole", 4, -1), ("phase 4 bands", 4, 0), ("phase 1 whole", 1, -1), ("phase 0 whole", 0, -1), ("phase 0 bands", 0, 0)]
    for (W, H) in SIZES:
        for n in BATCHES:
            if GRID:
                grid = pkg.Grid(None, [(SRC_W, SRC_H)] * 9, W, H)
                grid.set_direct(True)
                grid.exchange({k: frames_t[k].data_ptr() for k in range(9)}, torch.cuda.current_stream().cuda_stream)
                descs = []
                for _ in range(n):  # every client looks at the same grid (stream.c:790-854: aspect + padding on)
                    f = pkg.frame_setup(None, W, 2 * H, W, H, rm, True, True, False)
                    f.comp = grid.composite_dev
                    descs.append(f)
            elif DENSE:  # what a server tick renders after the sampled-image ingest: the source IS the image the target samples
                hs = 2 * H if rm == 2 else H
                dense_t = bench.make_frames(torch, n, W, hs, 99)
                host0 = np.ascontiguousarray(dense_t[0].cpu().numpy())
                descs = [pkg.frame_setup(dense_t[k].data_ptr(), W, hs, W, H, rm, False, False, False) for k in range(n)]
            else:
                descs = [pkg.frame_setup(frames_t[k % 256].data_ptr(), SRC_W, SRC_H, W, H, rm, False, False, False) for k in range(n)]
            bytes_per_frame = W * H * (41 if mode == 5 else 20)
            reps = max(8, min(300, int(4e8 / max(1, bytes_per_frame * n) / 50)))
            res = []
            ref = None
            for (label, variant, split) in [("automatic", -1, None)] + forced:
                plans = []
                try:
                    for _ in streams:
                        p = pkg.Plan(mode, bench.PALETTE_STANDARD, descs)
                        plans.append(p)
                        if INFLIGHT:
                            p.set_concurrency(len(streams))  # the caller's hint: this plan shares the GPU with three more
                        if split is not None:
                            p.set_split(split)
                        if variant >= 0:
                            p.set_variant(variant)
                except RuntimeError:
                    for p in plans:
                        p.close()
                    continue
                stride = plans[0].stride
                t, out, ln = time_plan(plans, n, stride, reps)
                lens = ln.cpu().numpy().astype("uint32")
                if not (lens < 0xFFFFFFF0).all():
                    for p in plans:
                        p.close()
                    continue
                view = out.view(n, stride)
                if ref is None:
                    ref = (view.clone(), lens.copy())
                    if not GRID:  # (the grid's bytes against the oracle's composite: tests/test_gpu_parity.py)
                        exp = orc.convert_with_caps(host0, W, H, cl, rm, False, False, False)
                        assert view[0, :int(lens[0])].cpu().numpy().tobytes() == exp, (mname, W, H, n, "automatic choice differs from the oracle")
                    label = f"automatic = v{plans[0].variant} parts {plans[0].parts}"
                else:
                    assert (lens == ref[1]).all(), (mname, W, H, n, label, "lengths differ")
                    m = int(lens.max())
                    idx = torch.arange(m, device="cuda")[None, :] < torch.from_numpy(lens.astype("int64")).cuda()[:, None]
                    assert bool(((view[:, :m] == ref[0][:, :m]) | ~idx).all()), (mname, W, H, n, label, "bytes differ from the automatic choice's")
                res.append((t, label, plans[0].variant, plans[0].parts))
                for p in plans:
                    p.close()
            if GRID:
                torch.cuda.synchronize()
                grid.close()
            auto = res[0]
            best = min(res, key=lambda r: r[0])
            regret = auto[0] / best[0] - 1.0
            regrets.append((regret, mname, W, H, n, auto, best))
            flag = "  <-- REGRET" if regret > 0.08 and auto[0] - best[0] > 0.4 else ""
            print(f"{mname:10s} {W:3d}x{H:<3d} batch {n:3d}: {auto[1]:28s} {auto[0]:8.2f} | best {best[1]:16s} {best[0]:8.2f} (+{100 * regret:5.1f} %){flag}   all: "
                  + " ".join(f"{r[1].split(' = ')[0]}={r[0]:.1f}" for r in res[1:]), flush=True)
worst = sorted(regrets, key=lambda r: -r[0])[:10]
print("# ten largest regrets")
for (regret, mname, W, H, n, auto, best) in worst:
    print(f"#   {mname} {W}x{H} batch {n}: automatic {auto[0]:.2f} us, {best[1]} {best[0]:.2f} us (+{100 * regret:.1f} %)")
