#!/usr/bin/env python3
"""The stand-alone checksum pass (asciichat_hip_frame_packets) over buffer length x buffer count, one-workgroup kernel against
64 KB spans + finish kernel: run once per ASCIICHAT_HIP_CRC_FRAME_MAX value (read once per process).  HIP events, back to back."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import orc  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
L = pkg.lib()
torch.cuda.set_device(0)
cur = torch.cuda.current_stream()
st = cur.cuda_stream
vp, sz, u32 = C.c_void_p, C.c_size_t, C.c_uint32
print(f"# ASCIICHAT_HIP_CRC_FRAME_MAX={os.environ.get('ASCIICHAT_HIP_CRC_FRAME_MAX', '(default)')}: us per call of frame_packets")
g = torch.Generator(device="cuda")
g.manual_seed(5)
for length in (65536 - 40, 131072 - 40, 262144 - 40, 540000, 1048576 - 40, 1845408, 4 * 1048576 - 40):
    stride = (length + 127) // 128 * 128
    row = f"{length:8d} B:"
    for n in (1, 4, 16, 64, 256):
        if n * stride > (2 << 30):
            continue
        buf = torch.randint(0, 256, (n * stride,), dtype=torch.uint8, device="cuda", generator=g)
        ln = torch.full((n,), length, dtype=torch.int32, device="cuda")
        dims = torch.tensor([[80, 24]] * n, dtype=torch.int32, device="cuda")
        crc = torch.zeros(n, dtype=torch.int32, device="cuda")
        hdr = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
        pkt = torch.zeros(n, dtype=torch.int32, device="cuda")

        def call():
            rc = L.asciichat_hip_frame_packets(vp(buf.data_ptr()), sz(stride), vp(ln.data_ptr()), u32(stride), C.c_int(n), vp(dims.data_ptr()),
                                               vp(crc.data_ptr()), vp(hdr.data_ptr()), vp(pkt.data_ptr()), vp(st))
            assert rc == 0, rc

        for _ in range(3):
            call()
        torch.cuda.synchronize()
        reps = max(5, min(100, int(2e9 / (n * stride) / 20)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for _ in range(reps):
            call()
        e1.record(cur)
        torch.cuda.synchronize()
        got = int(crc[0].item()) & 0xFFFFFFFF
        assert got == orc.crc32c(buf[:length].cpu().numpy().tobytes()), (length, n)
        row += f"  n={n:3d}: {e0.elapsed_time(e1) / reps * 1e3:8.1f}"
    print(row, flush=True)
