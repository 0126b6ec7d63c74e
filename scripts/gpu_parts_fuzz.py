#!/usr/bin/env python3
"""Randomised check of the shared-out forms (PARTS) of the stream kernel and, round 6, of the rows kernel on the GPU: random
modes (--rows: the run-structured ones), target sizes, paddings, palettes and batch sizes through plans whose part count is the
policy's or forced (ASCIICHAT_HIP_STREAM_PARTS / ASCIICHAT_HIP_ROWS_PARTS, read once per process: run once per value), every
frame compared byte-for-byte with the oracle, three launches per plan (epochs).
usage: gpu_parts_fuzz.py <seed> <plans> [--rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import orc
from __graft_entry__ import load_package
pkg = load_package(); torch.cuda.set_device(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(seed)
ROWS = "--rows" in sys.argv
CAPS = {1: (3, 0), 2: (2, 0), 3: (1, 0), 4: None, 0: (0, 0), 5: (3, 2), 6: (2, 2), 7: (1, 2), 8: (0, 2)}
PALS = [orc.PALETTE_STANDARD, orc.PALETTE_BLOCKS, "ab", "é漢😀 ."]
srcs = [orc.frame_torture(), orc.frame_hash_noise(160, 120, 5), orc.frame_bars(64, 48, 2), orc.frame_smooth(90, 70), orc.frame_hash_noise(7, 5, 9)]
dev = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in srcs]
st = torch.cuda.current_stream().cuda_stream
shared, frames_checked, hist = 0, 0, {}
for it in range(rounds):
    mode = int(rng.choice([0, 5, 6, 7, 8] if ROWS else [1, 2, 3, 4]))
    pal = orc.PALETTE_STANDARD if mode == 1 else PALS[int(rng.integers(0, len(PALS)))]
    n = int(rng.choice([1, 1, 2, 3, 5, 9]))
    big = rng.random() < 0.25
    W, H = (int(rng.integers(60, 201 if not ROWS else 521)), int(rng.integers(20, 71))) if big else (int(rng.integers(1, 100 if not ROWS else 129)), int(rng.integers(1, 40)))
    pad = mode != 4 and rng.random() < 0.4
    ragged = rng.random() < 0.3
    fr, want = [], []
    for k in range(n):
        i = int(rng.integers(0, len(srcs)))
        w, h = (W, H) if not ragged else (max(1, W - int(rng.integers(0, 7))), max(1, H - int(rng.integers(0, 5))))
        fr.append(pkg.frame_setup(dev[i].data_ptr(), srcs[i].shape[1], srcs[i].shape[0], w, h, CAPS[mode][1] if mode != 4 else 0, pad, pad, False))
        if mode == 4:
            want.append(orc.print_truecolor_bg(orc.resize_nn(srcs[i], w, h), pal))
        else:
            want.append(orc.convert_with_caps(srcs[i], w, h, CAPS[mode][0], CAPS[mode][1], pad, pad, False, pal))
    plan = pkg.Plan(mode, pal, fr)
    hist[(plan.variant, plan.parts > 1)] = hist.get((plan.variant, plan.parts > 1), 0) + 1
    shared += plan.parts > 1 and plan.variant in (18, 31, 32)
    out = torch.full((n * plan.stride,), 0xEE, dtype=torch.uint8, device="cuda")
    ln = torch.zeros(n, dtype=torch.int32, device="cuda")
    for rep in range(3):
        plan.render(out.data_ptr(), plan.stride, ln.data_ptr(), st)
    torch.cuda.synchronize()
    host, lens = out.cpu().numpy(), ln.cpu().numpy().astype(np.uint32)
    for k in range(n):
        got = host[k * plan.stride:k * plan.stride + int(lens[k])].tobytes() if lens[k] < 0xFFFFFFF0 else int(lens[k])
        assert got == want[k], (it, mode, W, H, n, k, pad, plan.variant, plan.parts, pal[:4])
        frames_checked += 1
    plan.close()
print(f"parts fuzz OK ({'run-structured' if ROWS else 'per-cell'} modes, ASCIICHAT_HIP_STREAM_PARTS={os.environ.get('ASCIICHAT_HIP_STREAM_PARTS', '')!r} ASCIICHAT_HIP_ROWS_PARTS={os.environ.get('ASCIICHAT_HIP_ROWS_PARTS', '')!r}): {rounds} plans, {frames_checked} frames "
      f"byte-identical to the oracle, {shared} plans shared out over workgroups; (geometry, multi-workgroup) counts {sorted(hist.items())}")
