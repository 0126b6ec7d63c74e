#!/usr/bin/env python3
"""What the stand-alone wire pass costs behind a small launch: render alone, asciichat_hip_frame_packets alone (HIP events,
back to back on one stream).  usage: gpu_wire_one.py W H n [W H n ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
L = pkg.lib()
torch.cuda.set_device(0)
cur = torch.cuda.current_stream()
st = cur.cuda_stream
fr = bench.make_frames(torch, 256, 1920, 1080, 1)


def timed(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for _ in range(reps):
        fn()
    e1.record(cur)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


a = sys.argv[1:]
for i in range(0, len(a), 3):
    W, H, n = int(a[i]), int(a[i + 1]), int(a[i + 2])
    descs = [pkg.frame_setup(fr[k].data_ptr(), 1920, 1080, W, H, 0, False, False, False) for k in range(n)]
    plan = pkg.Plan(1, bench.PALETTE_STANDARD, descs)
    stride = plan.stride
    slab = torch.zeros(n * stride, dtype=torch.uint8, device="cuda")
    ln = torch.zeros(n, dtype=torch.int32, device="cuda")
    dims = torch.tensor([[W, H]] * n, dtype=torch.int32, device="cuda")
    crc = torch.zeros(n, dtype=torch.int32, device="cuda")
    hdr = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
    pkt = torch.zeros(n, dtype=torch.int32, device="cuda")
    t_r = timed(lambda: plan.render(slab.data_ptr(), stride, ln.data_ptr(), st))
    max_len = int(ln.max().item())
    vp, sz, u32 = C.c_void_p, C.c_size_t, C.c_uint32

    def wire(ml):
        rc = L.asciichat_hip_frame_packets(vp(slab.data_ptr()), sz(stride), vp(ln.data_ptr()), u32(ml), C.c_int(n), vp(dims.data_ptr()),
                                           vp(crc.data_ptr()), vp(hdr.data_ptr()), vp(pkt.data_ptr()), vp(st))
        assert rc == 0, rc

    t_w = timed(lambda: wire(stride))
    t_w2 = timed(lambda: wire((max_len + 15) // 16 * 16))
    plan.set_fused_crc(0)
    t_p = timed(lambda: plan.render_packets(slab.data_ptr(), stride, ln.data_ptr(), dims.data_ptr(), crc.data_ptr(), hdr.data_ptr(), pkt.data_ptr(), st))
    print(f"{W}x{H} x {n}: variant {plan.variant} parts {plan.parts} stride {stride} longest frame {max_len}: render {t_r:.1f} us | frame_packets "
          f"(max_len = stride: {L.achip_crc_parts(u32(stride), C.c_int(n))} spans) {t_w:.1f} us, (max_len = longest frame: {L.achip_crc_parts(u32(max_len), C.c_int(n))} spans) {t_w2:.1f} us | "
          f"render_packets, separate {t_p:.1f} us", flush=True)
    plan.close()
