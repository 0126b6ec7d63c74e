#!/usr/bin/env python3
"""Exact-length frames from the render kernel (PACK instantiations) against render (+ fused wire stage) + pack_frames:
GPU time per 256-frame step by HIP events, S = 1 and 4 launches in flight, with and without the wire stage, device
destination and mapped host destination.  Decides the plan's automatic choice.  GPU box only."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
torch.cuda.set_device(0)
name = sys.argv[1] if len(sys.argv) > 1 else "1080p_80x24_truecolor"
sw, sh, W, H, cl, rm = bench.WORKLOADS[name]
batch, nsets = 256, 12
sets = [bench.make_frames(torch, batch, sw, sh, 1234 + 7919 * s) for s in range(nsets)]
plans = [bench.build_plan(pkg, t, W, H, cl, rm)[0] for t in sets]
stride = plans[0].stride
dims = torch.tensor([[W, H]] * batch, dtype=torch.int32, device="cuda")
tab = (8 * (batch + 1) + 4 * batch + 15) // 16 * 16
print(f"# {name}: 256 frames per step, us per step (HIP events over 200 steps, median of 3)")
for S in (1, 4):
    for p in plans:
        p.set_concurrency(S)
    lanes = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(S - 1)]
    slabs = [torch.empty(batch * stride, dtype=torch.uint8, device="cuda") for _ in range(S)]
    lns = [torch.zeros(batch, dtype=torch.int32, device="cuda") for _ in range(S)]
    crcs = [torch.zeros(batch, dtype=torch.int32, device="cuda") for _ in range(S)]
    hdrs = [torch.zeros(batch * 24, dtype=torch.uint8, device="cuda") for _ in range(S)]
    pkts = [torch.zeros(batch, dtype=torch.int32, device="cuda") for _ in range(S)]
    devd = [torch.empty(tab + batch * stride, dtype=torch.uint8, device="cuda") for _ in range(S)]
    hbs = [pkg.HostBuffer(tab + batch * stride) for _ in range(S)]
    for host in (False, True):
        for wire in (True, False):
            res = {}
            for exact in (1, 0):
                for p in plans:
                    p.set_exact_length(exact)

                def step(k):
                    s, p = k % S, plans[k % nsets]
                    base = hbs[s].dev if host else devd[s].data_ptr()
                    st = lanes[s].cuda_stream
                    if wire:
                        p.render_packets_packed(slabs[s].data_ptr(), stride, lns[s].data_ptr(), dims.data_ptr(), crcs[s].data_ptr(),
                                                hdrs[s].data_ptr(), pkts[s].data_ptr(), base + tab, batch * stride, base,
                                                base + 8 * (batch + 1), st)
                    else:
                        p.render_packed(slabs[s].data_ptr(), stride, lns[s].data_ptr(), base + tab, batch * stride, base,
                                        base + 8 * (batch + 1), st)

                def timed(steps):
                    for k in range(2 * S):
                        step(k)
                    torch.cuda.synchronize()
                    b = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
                    e = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
                    for s in range(S):
                        step(s)
                        b[s].record(lanes[s])
                    for k in range(steps):
                        step(k)
                    for s in range(S):
                        e[s].record(lanes[s])
                    torch.cuda.synchronize()
                    return max(b[s].elapsed_time(e[t]) for s in range(S) for t in range(S)) / steps * 1e3

                res[exact] = statistics.median(timed(60 if host else 200) for _ in range(3))
            print(f"  in flight {S}  {'host  ' if host else 'device'} destination  {'frames + checksums + headers' if wire else 'frames only                 '}:"
                  f"  ONE launch {res[1]:8.2f}   render + pack {res[0]:8.2f}")
            for p in plans:
                p.set_exact_length(-1)
    for hb in hbs:
        hb.close()
