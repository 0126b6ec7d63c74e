"""Does the headline number lean on re-reading the same 256 source frames every step (the sampled sectors of one
batch, ~31 MB, fit the 256 MB Infinity Cache)?  Time the K2 workloads while rotating through S independent input
sets (S x 1.6 GB), so that with S >= 2 no step finds its sectors where the previous step left them."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


def main():
    pkg = load_package()
    torch.cuda.set_device(0)
    stream = torch.cuda.current_stream().cuda_stream
    for name in sys.argv[1:] or ["1080p_80x24_truecolor", "1080p_80x24_ansi256", "4k_200x60_truecolor"]:
        sw, sh, W, H, cl, rm = bench.WORKLOADS[name]
        for S, uniform in ((1, True), (4, True), (4, False), (4, True), (4, False), (8, True)):
            if S * 256 * sw * sh * 3 > 60e9:
                continue
            sets = [bench.make_frames(torch, 256, sw, sh, 100 + s) for s in range(S)]
            plans = [bench.build_plan(pkg, t, W, H, cl, rm)[0] for t in sets]
            for p in plans:
                p.set_uniform(uniform)
            out = torch.empty(256 * plans[0].stride, dtype=torch.uint8, device="cuda")
            ln = torch.zeros(256, dtype=torch.int32, device="cuda")
            for k in range(40):
                plans[k % S].render(out.data_ptr(), plans[0].stride, ln.data_ptr(), stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            steps = 400
            e0.record()
            for k in range(steps):
                plans[k % S].render(out.data_ptr(), plans[0].stride, ln.data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / steps
            print(f"{name:24s} input sets {S} descriptor {'by value' if plans[0].uniform else 'fetched '}: {us:7.2f} us per 256-frame step  ({256 / us:6.2f} M frames/s)", flush=True)
            for p in plans:
                p.close()
            del sets, plans


if __name__ == "__main__":
    main()
