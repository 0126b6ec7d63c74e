#!/usr/bin/env python3
"""Busy time per launch with NO profiler attached: K steps issued from C on S streams, every
launch of the stream kernel writing its waves' entry / last-store timestamps (100 MHz wall clock, asciichat_hip_render_many_
profiled).  Per burst: the union of the launches' [first wave entry, last wave's final stamp] intervals / K = the time the
GPU spent per launch with at least one of them running -- what rocprofv3's kernel trace gives when it does not disturb the
overlap -- next to the HIP-event time per step of the SAME schedule with the production (unstamped) kernels.
usage: gpu_busy_stamps.py [workload] [streams] [K]   (stream-kernel workloads only: per-cell modes)"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
torch.cuda.set_device(0)
name = sys.argv[1] if len(sys.argv) > 1 else "1080p_80x24_truecolor"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
K = int(sys.argv[3]) if len(sys.argv) > 3 else 400
sw, sh, W, H, cl, rm = bench.WORKLOADS[name]
nsets = 12 if sw < 3000 else 4
sets = [bench.make_frames(torch, 256, sw, sh, 50 + s) for s in range(nsets)]
plans = [bench.build_plan(pkg, t, W, H, cl, rm)[0] for t in sets]
for p in plans:
    p.set_concurrency(S)
waves = pkg.lib().achip_variant_block(plans[0].variant) // 64
run = bench.Runner(torch, pkg, plans, 256, S)
run.issue(40)
torch.cuda.synchronize()
event_ms = statistics.median(run.gpu_ms_per_step(K) for _ in range(5))
stride = 256 * waves * 8
prof = torch.zeros(K * stride, dtype=torch.int64, device="cuda")
busy, inflight, dur = [], [], []
for it in range(8):
    prof.zero_()
    torch.cuda.synchronize()
    run.sched.issue_profiled(0, K, prof.data_ptr(), stride)
    run.sched.wait()
    torch.cuda.synchronize()
    a = prof.cpu().numpy().reshape(K, 256, waves, 8)
    act = a[:, :, :, 7] != 0
    iv = sorted((int(a[k, :, :, 0][act[k]].min()), int(a[k, :, :, 7][act[k]].max())) for k in range(K))
    total, lo, hi = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > hi:
            total += hi - lo
            lo, hi = s, e
        else:
            hi = max(hi, e)
    total += hi - lo
    if it >= 2:
        busy.append(total / 100.0 / K)  # 100 MHz ticks -> us
        inflight.append(sum(e - s for s, e in iv) / total)
        dur.append(sum(e - s for s, e in iv) / 100.0 / K)
alg = None
print(json.dumps({"workload": name, "kernel_variant": plans[0].variant, "launches_in_flight_requested": S, "steps_per_burst": K,
                  "busy_us_per_launch_device_stamps": statistics.median(busy), "avg_in_flight": statistics.median(inflight),
                  "avg_first_entry_to_last_store_us": statistics.median(dur),
                  "hip_event_us_per_step_unstamped_kernels": event_ms * 1e3,
                  "note": "stamped kernels carry eight global timestamp stores per wave: slightly slower than production"}))
