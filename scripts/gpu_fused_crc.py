"""What does the wire stage cost behind the render?  In one run, on the headline batch (256 frames per
step, 12 input sets), per 256-frame step on one stream and round-robin on 4 streams:
  (a) render only                                   plan.render
  (b) render + stand-alone CRC/packet kernel       plan.render ; asciichat_hip_frame_packets   (a second pass over the slab)
  (c) render with the CRC riding the drain + hdrs  plan.render_crc ; asciichat_hip_packets_from_crc
  (d) all of it in the render launch                plan.render_packets
and checks (c)'s and (d)'s checksums / headers / packet CRCs against (b)'s.  Calls are issued from Python here (ctypes, ~3 us each): the one-stream rows
of (b) and (c) are two calls per step and may be issue-bound; the 4-stream rows show the device cost."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


def main():
    pkg = load_package()
    L = pkg.lib()
    torch.cuda.set_device(0)
    for name in sys.argv[1:] or ["1080p_80x24_truecolor"]:
        sw, sh, W, H, cl, rm = bench.WORKLOADS[name]
        nsets = 12
        sets = [bench.make_frames(torch, 256, sw, sh, 300 + s) for s in range(nsets)]
        plans = [bench.build_plan(pkg, t, W, H, cl, rm)[0] for t in sets]
        stride = plans[0].stride
        n = 256
        print(f"{name}: variant {plans[0].variant}, fused crc available: {plans[0].fused_crc}", flush=True)
        dims = torch.tensor([[W, H]] * n, dtype=torch.int32, device="cuda")
        for nstreams, variant in ((1, -1), (4, -1), (1, 17), (4, 17)):
            if variant >= 0:
                for p in plans:
                    p.set_variant(variant)
            print(f"  -- geometry {plans[0].variant}", flush=True)
            streams = [torch.cuda.Stream() for _ in range(nstreams)]
            outs = [torch.empty(n * stride, dtype=torch.uint8, device="cuda") for _ in range(nstreams)]
            lns = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(nstreams)]
            crcs = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(nstreams)]
            hdrs = [torch.zeros(n * 24, dtype=torch.uint8, device="cuda") for _ in range(nstreams)]
            pkts = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(nstreams)]

            def step(kind, k):
                s = k % nstreams
                st = streams[s].cuda_stream
                p = plans[k % nsets]
                if kind == "a":
                    p.render(outs[s].data_ptr(), stride, lns[s].data_ptr(), st)
                elif kind == "b":
                    p.render(outs[s].data_ptr(), stride, lns[s].data_ptr(), st)
                    L.asciichat_hip_frame_packets(outs[s].data_ptr(), stride, lns[s].data_ptr(), stride, n, dims.data_ptr(),
                                                  crcs[s].data_ptr(), hdrs[s].data_ptr(), pkts[s].data_ptr(), st)
                elif kind == "c":
                    p.render_crc(outs[s].data_ptr(), stride, lns[s].data_ptr(), crcs[s].data_ptr(), st)
                    L.asciichat_hip_packets_from_crc(lns[s].data_ptr(), crcs[s].data_ptr(), n, dims.data_ptr(),
                                                     hdrs[s].data_ptr(), pkts[s].data_ptr(), st)
                else:
                    p.render_packets(outs[s].data_ptr(), stride, lns[s].data_ptr(), dims.data_ptr(), crcs[s].data_ptr(),
                                     hdrs[s].data_ptr(), pkts[s].data_ptr(), st)

            ref = {}
            for rnd in range(2):
                for kind in ("a", "b", "c", "c-only", "d"):
                    steps = 240
                    kk = "c" if kind == "c-only" else kind
                    if kind == "c-only":
                        def stepf(k):
                            s = k % nstreams
                            plans[k % nsets].render_crc(outs[s].data_ptr(), stride, lns[s].data_ptr(), crcs[s].data_ptr(),
                                                        streams[s].cuda_stream)
                    else:
                        def stepf(k, kk=kk):
                            step(kk, k)
                    for k in range(24):
                        stepf(k)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for st in streams:
                        st.wait_event(e0)
                    for k in range(steps):
                        stepf(k)
                    for st in streams:
                        e = torch.cuda.Event()
                        e.record(st)
                        torch.cuda.current_stream().wait_event(e)
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1000 / steps
                    print(f"  streams {nstreams} round {rnd} ({kind:6s}): {us:7.2f} us per 256-frame step", flush=True)
                    if kind in ("b", "c", "d"):
                        # last step on stream 0 rendered plan (steps - nstreams ...) -- compare b vs c on one fixed plan
                        step(kk, 0)
                        torch.cuda.synchronize()
                        got = (crcs[0].cpu().numpy().copy(), pkts[0].cpu().numpy().copy(), hdrs[0].cpu().numpy().copy())
                        if kind == "b":
                            ref = got
                        else:
                            same = all((a == b).all() for a, b in zip(ref, got))
                            print(f"    fused checksums / packet CRCs / headers == stand-alone kernel's: {same}", flush=True)
                            assert same
        for p in plans:
            p.close()


if __name__ == "__main__":
    main()
