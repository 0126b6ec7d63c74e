#!/usr/bin/env python3
"""The end-to-end server tick (bench.py tick_e2e: blobs in the pinned pool -> publish_sampled_batch -> latest_frames -> plan_update ->
plan_render_packets_packed into mapped host memory -> the tick's synchronisation) at the client counts and terminal sizes a real
server runs, not only 256 x 80x24: ms per tick sequential and with two ticks in flight, frames/s, PCIe bytes down.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
torch.cuda.set_device(0)
print("# clients x terminal: sequential tick (us) [publish part], two ticks in flight (us per tick), PCIe bytes down per tick")
for grid in ((80, 24), (120, 40), (160, 45), (200, 60)):
    for n in (1, 9, 32, 64, 256):
        if n * grid[0] * grid[1] * 20 > 60e6:
            continue
        r = bench.tick_e2e(torch, pkg, n=n, distinct=min(32, max(2, n)), ticks=(2, 6), forms=("sampled_images",), grid=grid)
        s, p = r["sampled_images"], r["sampled_images_pipelined"]
        print(f"{n:4d} x {grid[0]:3d}x{grid[1]:<3d}: {s['ms_per_tick'] * 1e3:8.1f} us [{s['publish_ms_per_tick'] * 1e3:6.1f}]   pipelined {p['ms_per_tick'] * 1e3:8.1f} us   "
              f"{s['pcie_bytes_down_per_tick'] / 1e3:9.1f} KB down = {s['pcie_bytes_down_per_tick'] / max(1e-9, p['ms_per_tick']) / 1e6:6.1f} GB/s pipelined", flush=True)
