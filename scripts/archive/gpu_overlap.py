"""Can consecutive 256-frame steps overlap each other's phases?  One step is gather (HBM-bound burst) followed by
tokenise/emit (latency-bound, HBM idle).  Measures, per kernel geometry, 256-frame steps issued (a) back to back on one
stream, (b) round-robin on 2 / 3 streams (independent batches, separate output slabs), (c) as one launch of 512 / 1024
frames.  Fresh input batch every step (4 sets per stream)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


def main():
    pkg = load_package()
    torch.cuda.set_device(0)
    name = sys.argv[1] if len(sys.argv) > 1 else "1080p_80x24_truecolor"
    sw, sh, W, H, cl, rm = bench.WORKLOADS[name]
    nsets = int(os.environ.get("OVERLAP_NSETS", "12"))
    sets = [bench.make_frames(torch, 256, sw, sh, 300 + s) for s in range(nsets)]
    steps = 480 if sw < 3000 else 120
    big_too = len(sys.argv) > 2
    variants = [int(v) for v in os.environ.get("OVERLAP_VARIANTS", "4,1,2").split(",")]
    stream_counts = [int(v) for v in os.environ.get("OVERLAP_STREAMS", "1,2,3,4,6").split(",")]
    for variant in variants:
        for nstreams in stream_counts:
            plans = [bench.build_plan(pkg, t, W, H, cl, rm)[0] for t in sets]
            try:
                for p in plans:
                    p.set_variant(variant)
            except RuntimeError:
                continue
            # ONE pool of streams for the whole process (bench.py's): HIP maps streams onto hardware queues round robin,
            # and a fresh set of stream objects per configuration made the first configuration of a run measure 20-30 %
            # slow (profiles/r02_geometry_ab.txt shows the artefact next to the repeat)
            while len(bench._LANE_POOL) < nstreams - 1:
                bench._LANE_POOL.append(torch.cuda.Stream())
            streams = [torch.cuda.current_stream()] + bench._LANE_POOL[:nstreams - 1]
            outs = [torch.empty(256 * plans[0].stride, dtype=torch.uint8, device="cuda") for _ in range(nstreams)]
            lns = [torch.zeros(256, dtype=torch.int32, device="cuda") for _ in range(nstreams)]

            sched = pkg.Schedule(plans, [o.data_ptr() for o in outs], [l.data_ptr() for l in lns], plans[0].stride,
                                 [st.cuda_stream for st in streams])
            sched.issue(0, 48)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for st in streams:
                st.wait_event(e0)
            import time
            t0 = time.perf_counter()
            sched.issue(0, steps)  # the K launches are issued from C (asciichat_hip_render_many)
            t_issue = time.perf_counter() - t0
            for st in streams:
                e = torch.cuda.Event()
                e.record(st)
                torch.cuda.current_stream().wait_event(e)
            e1.record()
            torch.cuda.synchronize()
            t_wall = time.perf_counter() - t0
            us = e0.elapsed_time(e1) * 1000 / steps
            print(f"{name} variant {variant} streams {nstreams}: {us:7.2f} us per 256-frame step ({256 / us:6.2f} M frames/s)"
                  f"  [host issue {t_issue * 1e6 / steps:5.2f} us/launch, wall {t_wall * 1e6 / steps:6.2f} us/step]", flush=True)
            for p in plans:
                p.close()
    # one launch of 512 / 1024 frames (what a server with more clients would submit)
    for nb in ((512, 1024) if big_too else ()):
        big = [torch.cat(sets[i * (nb // 256):(i + 1) * (nb // 256)]) for i in range(nsets * 256 // nb)]
        for variant in (-1, 4, 1, 2):
            plans = [bench.build_plan(pkg, t, W, H, cl, rm)[0] for t in big]
            for p in plans:
                if variant >= 0:
                    p.set_variant(variant)
            out = torch.empty(nb * plans[0].stride, dtype=torch.uint8, device="cuda")
            ln = torch.zeros(nb, dtype=torch.int32, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            for k in range(16):
                plans[k % len(plans)].render(out.data_ptr(), plans[0].stride, ln.data_ptr(), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 200
            e0.record()
            for k in range(n):
                plans[k % len(plans)].render(out.data_ptr(), plans[0].stride, ln.data_ptr(), st)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / n
            print(f"{name} one launch of {nb} frames, variant {plans[0].variant}: {us:7.2f} us = {us * 256 / nb:6.2f} us per 256 frames "
                  f"({nb / us:6.2f} M frames/s)", flush=True)
            for p in plans:
                p.close()
        del big


if __name__ == "__main__":
    main()
