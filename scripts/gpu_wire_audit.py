#!/usr/bin/env python3
"""Wire-stage policy audit (the sibling of gpu_policy_audit.py): which form should the send side of a plan take?
  packets: plan_render_packets -- the frame CRC fused into the render (set_fused_crc 1) or render + the stand-alone kernel (0),
           against the plan's own choice (-1);
  packed : plan_render_packets_packed into DEVICE memory -- exact-length frames from the render kernel itself
           (set_exact_length 1), render + pack / checksum-and-pack (0), against the plan's own choice (-1).
Terminal sizes x batch sizes x modes, 1080p sources, launches back to back on one stream (HIP events).  Every form's
checksums, headers and packed frames are compared with the automatic form's (the automatic form's first frame and CRC with the
oracle).  GPU box only.  usage: gpu_wire_audit.py [--quick] [--dense] [--modes=truecolor,ansi256]"""
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
torch.cuda.set_device(0)
cur = torch.cuda.current_stream()
QUICK = "--quick" in sys.argv
MODES = [(0, "mono", 0, 0), (2, "ansi256", 2, 0), (1, "truecolor", 3, 0), (5, "hb_true", 3, 2)]
SIZES = [(80, 24), (120, 40), (160, 45), (200, 60), (320, 90)]
BATCHES = [1, 16, 64, 128, 256]
if QUICK:
    SIZES, BATCHES = [(80, 24), (200, 60)], [16, 256]
SRC_W, SRC_H = 1920, 1080
DENSE = "--dense" in sys.argv
if "--modes" in " ".join(sys.argv):
    MODES = [m for m in MODES if m[1] in [a[8:] for a in sys.argv if a.startswith("--modes=")][0].split(",")]
frames_t = bench.make_frames(torch, 256, SRC_W, SRC_H, 4242)
host0 = np.ascontiguousarray(frames_t[0].cpu().numpy())


def timed(fn, reps):
    fn()
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record(cur)
        for _ in range(reps):
            fn()
        e1.record(cur)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts)


worst = []
print("# 1080p sources, one stream, back to back (HIP events); us per launch.  packets = render + frame CRCs + headers + packet CRCs; "
      "packed = the same + frames at their exact lengths in device memory")
for (mode, mname, cl, rm) in MODES:
    for (W, H) in SIZES:
        for n in BATCHES:
            if DENSE:  # what a server tick renders after the sampled-image ingest: the source IS the image the target samples
                hs = 2 * H if rm == 2 else H
                dense_t = bench.make_frames(torch, n, W, hs, 99)
                host0 = np.ascontiguousarray(dense_t[0].cpu().numpy())
                descs = [pkg.frame_setup(dense_t[k].data_ptr(), W, hs, W, H, rm, False, False, False) for k in range(n)]
            else:
                descs = [pkg.frame_setup(frames_t[k].data_ptr(), SRC_W, SRC_H, W, H, rm, False, False, False) for k in range(n)]
            plan = pkg.Plan(mode, bench.PALETTE_STANDARD, descs)
            stride = plan.stride
            reps = max(8, min(200, int(4e8 / max(1, W * H * (41 if mode == 5 else 20) * n) / 60)))
            slab = torch.zeros(n * stride, dtype=torch.uint8, device="cuda")
            ln = torch.zeros(n, dtype=torch.int32, device="cuda")
            dims = torch.tensor([[W, H]] * n, dtype=torch.int32, device="cuda")
            crc = torch.zeros(n, dtype=torch.int32, device="cuda")
            hdr = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
            pkt = torch.zeros(n, dtype=torch.int32, device="cuda")
            dst = torch.zeros(n * stride, dtype=torch.uint8, device="cuda")
            off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
            lo = torch.zeros(n, dtype=torch.int32, device="cuda")
            st = cur.cuda_stream

            def packets():
                plan.render_packets(slab.data_ptr(), stride, ln.data_ptr(), dims.data_ptr(), crc.data_ptr(), hdr.data_ptr(), pkt.data_ptr(), st)

            def packed():
                plan.render_packets_packed(slab.data_ptr(), stride, ln.data_ptr(), dims.data_ptr(), crc.data_ptr(), hdr.data_ptr(),
                                           pkt.data_ptr(), dst.data_ptr(), n * stride, off.data_ptr(), lo.data_ptr(), st)

            def snapshot(with_frames):
                torch.cuda.synchronize()
                s = [crc.cpu().numpy().copy(), hdr.cpu().numpy().copy(), pkt.cpu().numpy().copy()]
                if with_frames:
                    o, l, d = off.cpu().numpy(), lo.cpu().numpy().astype("uint32"), dst.cpu().numpy()
                    s.append([d[int(o[k]):int(o[k]) + int(l[k])].tobytes() for k in (0, n // 2, n - 1)])
                return s

            res_a, res_b = [], []
            ref_a = ref_b = None
            for label, fused in (("automatic", -1), ("separate", 0), ("fused", 1)):
                plan.set_fused_crc(fused)
                is_fused = plan.fused_crc
                t = timed(packets, reps)
                s = snapshot(False)
                if ref_a is None:
                    ref_a = s
                    exp = orc.convert_with_caps(host0, W, H, cl, rm, False, False, False)
                    assert int(np.uint32(s[0][0])) == orc.crc32c(exp), (mname, W, H, n, "frame CRC differs from the oracle's")
                    label = f"automatic={'fused' if is_fused else 'separate'}"
                else:
                    assert all((a == b).all() for a, b in zip(s, ref_a)), (mname, W, H, n, label, "wire stage differs")
                res_a.append((t, label))
            plan.set_fused_crc(-1)
            for label, ex in (("automatic", -1), ("two passes", 0), ("one launch", 1)):
                plan.set_exact_length(ex)
                t = timed(packed, reps)
                s = snapshot(True)
                if ref_b is None:
                    ref_b = s
                    label = "automatic"
                else:
                    assert all((a == b).all() for a, b in zip(s[:3], ref_b[:3])) and s[3] == ref_b[3], (mname, W, H, n, label, "packed output differs")
                res_b.append((t, label))
            plan.set_exact_length(-1)
            plan.close()
            line = f"{mname:10s} {W:3d}x{H:<3d} batch {n:3d}:"
            for name, res in (("packets", res_a), ("packed", res_b)):
                auto, best = res[0], min(res, key=lambda r: r[0])
                regret = auto[0] / best[0] - 1.0
                worst.append((regret, mname, W, H, n, name, auto, best))
                flag = " <-- REGRET" if regret > 0.08 and auto[0] - best[0] > 0.4 else ""
                line += f"  {name}: " + " ".join(f"{lab}={t:.1f}" for t, lab in res) + f" (+{100 * regret:.1f} %){flag} |"
            print(line, flush=True)
print("# ten largest regrets")
for (regret, mname, W, H, n, name, auto, best) in sorted(worst, key=lambda r: -r[0])[:10]:
    print(f"#   {mname} {W}x{H} batch {n} {name}: {auto[1]} {auto[0]:.2f} us, {best[1]} {best[0]:.2f} us (+{100 * regret:.1f} %)")
