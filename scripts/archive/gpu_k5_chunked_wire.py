#!/usr/bin/env python3
"""Experiment (the cheap half): the wire stage behind the rows kernel as a second pass that reads
the slab out of the Infinity Cache instead of HBM.  4K -> 400x120 half blocks write 472 MB per 256 frames -- more than the
256 MB cache holds -- so the stand-alone checksum pass re-reads all of it from HBM (~94 us).  Here the step is cut into
chunks of C frames: render chunk k (plan_render_range), checksum chunk k (frame_packets on that part of the slab) -- the
chunk's 118 MB (C = 64) are still on die when its checksum pass reads them.  Same stream, S streams of whole steps in flight.
GPU box only.  usage: gpu_k5_chunked_wire.py [workload] [streams]"""
import ctypes as C
import statistics
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
import orc  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
L = pkg.lib()
torch.cuda.set_device(0)
name = sys.argv[1] if len(sys.argv) > 1 else "4k_400x120_halfblock"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sw, sh, W, H, cl, rm = bench.WORKLOADS[name]
batch, nsets = 256, 4
sets = [bench.make_frames(torch, batch, sw, sh, 1234 + 7919 * s) for s in range(nsets)]
plans = []
for t in sets:
    p, mode = bench.build_plan(pkg, t, W, H, cl, rm)
    p.set_concurrency(S)
    plans.append(p)
stride = plans[0].stride
lanes = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(S - 1)]
outs = [torch.empty(batch * stride, dtype=torch.uint8, device="cuda") for _ in range(S)]
lns = [torch.zeros(batch, dtype=torch.int32, device="cuda") for _ in range(S)]
crcs = [torch.zeros(batch, dtype=torch.int32, device="cuda") for _ in range(S)]
hdrs = [torch.zeros(batch * 24, dtype=torch.uint8, device="cuda") for _ in range(S)]
pkts = [torch.zeros(batch, dtype=torch.int32, device="cuda") for _ in range(S)]
pks = [torch.empty(batch * stride, dtype=torch.uint8, device="cuda") for _ in range(S)]
offs = [torch.zeros(batch + 1, dtype=torch.int64, device="cuda") for _ in range(S)]
plens = [torch.zeros(batch, dtype=torch.int32, device="cuda") for _ in range(S)]
dims = torch.tensor([[W, H]] * batch, dtype=torch.int32, device="cuda")


def step(k, chunk, packed):
    s, p = k % S, plans[k % nsets]
    st = lanes[s].cuda_stream
    if chunk == 0:  # the library's own forms
        if packed:
            p.render_packets_packed(outs[s].data_ptr(), stride, lns[s].data_ptr(), dims.data_ptr(), crcs[s].data_ptr(),
                                    hdrs[s].data_ptr(), pkts[s].data_ptr(), pks[s].data_ptr(), batch * stride,
                                    offs[s].data_ptr(), plens[s].data_ptr(), st)
        else:
            p.render_packets(outs[s].data_ptr(), stride, lns[s].data_ptr(), dims.data_ptr(), crcs[s].data_ptr(),
                             hdrs[s].data_ptr(), pkts[s].data_ptr(), st)
        return
    for f0 in range(0, batch, chunk):
        n = min(chunk, batch - f0)
        rc = L.asciichat_hip_plan_render_range(p._h, f0, n, outs[s].data_ptr() + f0 * stride, stride, lns[s].data_ptr() + 4 * f0, st)
        assert rc == 0, pkg.last_error()
        rc = L.asciichat_hip_frame_packets(outs[s].data_ptr() + f0 * stride, stride, lns[s].data_ptr() + 4 * f0, stride, n,
                                           dims.data_ptr() + 8 * f0, crcs[s].data_ptr() + 4 * f0, hdrs[s].data_ptr() + 24 * f0,
                                           pkts[s].data_ptr() + 4 * f0, st)
        assert rc == 0, pkg.last_error()


def timed(chunk, packed, steps):
    for k in range(2 * S):
        step(k, chunk, packed)
    torch.cuda.synchronize()
    b = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    for s in range(S):
        step(s, chunk, packed)
        b[s].record(lanes[s])
    for k in range(steps):
        step(k, chunk, packed)
    for s in range(S):
        e[s].record(lanes[s])
    torch.cuda.synchronize()
    return max(b[s].elapsed_time(e[t]) for s in range(S) for t in range(S)) / steps


steps = 16 if sw > 3000 else 80
step(0, 0, False)
torch.cuda.synchronize()
ref = crcs[0].cpu().numpy().copy()
print(f"# {name}, {S} launches in flight, {steps} steps per timing; ms per 256-frame step (frames + checksums + headers)")
for chunk in (0, 256, 128, 64, 32, 16):
    t = statistics.median(timed(chunk, False, steps) for _ in range(3))
    step(0, chunk, False)
    torch.cuda.synchronize()
    same = (crcs[0].cpu().numpy() == ref).all()
    print(f"chunk {chunk or 'library':>8}: {t * 1e3:8.1f} us   checksums equal: {bool(same)}")
t = statistics.median(timed(0, True, steps) for _ in range(3))
print(f"library packed : {t * 1e3:8.1f} us")
