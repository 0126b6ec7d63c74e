#!/usr/bin/env python3
"""Length-first exact-length frames (ONE launch: the stream kernel's lean loop run twice, render_stream.hpp LENGTH-FIRST)
against what plans do for frames beyond the 48 KB of the one-launch form: render into the slab + ONE pass that packs
(asciichat_hip_plan_render_packed) / checksums and packs (plan_render_packets_packed).  HIP-event time per 256-frame step, 1 and 4 launches in
flight; every frame of the length-first output is compared with the two-launch output.  VERDICT r5 next 6.  GPU box only.
(The forms are selected with plan_set_exact_length: 0 = render + pass, 1 = length-first wherever it applies.)"""
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
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
L = pkg.lib()
L.asciichat_hip_plan_render_length_first.restype = C.c_int
L.asciichat_hip_plan_render_length_first.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
torch.cuda.set_device(0)
names = sys.argv[1:] or ["sampled_200x60_truecolor", "4k_200x60_truecolor"]
batch = 256
for name in names:
    sw, sh, W, H, cl, rm = bench.WORKLOADS[name]
    nsets = 4 if sw > 3000 else 12
    sets = [bench.make_frames(torch, batch, sw, sh, 1234 + 7919 * s) for s in range(nsets)]
    plans = [bench.build_plan(pkg, t, W, H, cl, rm)[0] for t in sets]
    stride = plans[0].stride
    dims = torch.tensor([[W, H]] * batch, dtype=torch.int32, device="cuda")
    tab = (8 * (batch + 1) + 4 * batch + 15) // 16 * 16
    print(f"# {name}: 256 frames per step (frames of ~{stride} bytes at most), us per step, HIP events over 120 steps, median of 3; variant {plans[0].variant}")
    for S in (1, 4):
        for p in plans:
            p.set_concurrency(S)
        lanes = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(S - 1)]
        slabs = [torch.empty(batch * stride, dtype=torch.uint8, device="cuda") for _ in range(S)]
        lns = [torch.zeros(batch, dtype=torch.int32, device="cuda") for _ in range(S)]
        crcs = [torch.zeros(batch, dtype=torch.int32, device="cuda") for _ in range(S)]
        hdrs = [torch.zeros(batch * 24, dtype=torch.uint8, device="cuda") for _ in range(S)]
        pkts = [torch.zeros(batch, dtype=torch.int32, device="cuda") for _ in range(S)]
        devd = [torch.zeros(tab + batch * stride, dtype=torch.uint8, device="cuda") for _ in range(S)]

        def step(kind, k):
            s, p = k % S, plans[k % nsets]
            p.set_exact_length(0 if kind in ("render+pack", "render+checksum_pack") else 1)  # the two-pass forms / length-first
            base = devd[s].data_ptr()
            st = lanes[s].cuda_stream
            if kind == "render":
                p.render(slabs[s].data_ptr(), stride, lns[s].data_ptr(), st)
            elif kind == "render+pack":
                p.render_packed(slabs[s].data_ptr(), stride, lns[s].data_ptr(), base + tab, batch * stride, base, base + 8 * (batch + 1), st)
            elif kind in ("render+checksum_pack", "length_first+checksum"):
                p.render_packets_packed(slabs[s].data_ptr(), stride, lns[s].data_ptr(), dims.data_ptr(), crcs[s].data_ptr(), hdrs[s].data_ptr(),
                                        pkts[s].data_ptr(), base + tab, batch * stride, base, base + 8 * (batch + 1), st)
            else:
                rc = L.asciichat_hip_plan_render_length_first(p._h, lns[s].data_ptr(), base + tab, batch * stride, base, base + 8 * (batch + 1), st)
                if rc != 0:
                    raise RuntimeError(f"plan_render_length_first failed ({rc}): {pkg.last_error()}")

        def frames_of(s):
            torch.cuda.synchronize()
            raw = devd[s].cpu().numpy()
            off = raw[:8 * (batch + 1)].view(np.uint64)
            ln = raw[8 * (batch + 1):8 * (batch + 1) + 4 * batch].view(np.uint32)
            return [raw[tab + int(off[i]):tab + int(off[i]) + int(ln[i])].tobytes() for i in range(batch)], int(off[batch])

        # byte identity of the two forms on every frame of one batch
        step("render+pack", 0)
        ref, tot_ref = frames_of(0)
        devd[0].zero_()
        step("length_first", 0)
        got, tot = frames_of(0)
        assert got == ref and tot == tot_ref, "length-first output differs from render + pack"

        def timed(kind, steps=120):
            for k in range(2 * S):
                step(kind, k)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for ln_ in lanes[1:]:
                ln_.wait_stream(lanes[0])
            ev[0].record(lanes[0])
            for k in range(steps):
                step(kind, k)
            for ln_ in lanes[1:]:
                lanes[0].wait_stream(ln_)
            ev[1].record(lanes[0])
            torch.cuda.synchronize()
            return ev[0].elapsed_time(ev[1]) / steps * 1e3

        res = {kind: statistics.median(timed(kind) for _ in range(3)) for kind in ("render", "render+pack", "length_first", "render+checksum_pack", "length_first+checksum")}
        print(f"  {S} in flight: render alone {res['render']:8.2f}   render + pack {res['render+pack']:8.2f}   LENGTH-FIRST (one launch) {res['length_first']:8.2f}"
              f"   render + checksum-and-pack {res['render+checksum_pack']:8.2f}   LENGTH-FIRST + checksum in place {res['length_first+checksum']:8.2f}   [identical bytes, {tot} bytes packed]")
    del sets, plans
    torch.cuda.empty_cache()
