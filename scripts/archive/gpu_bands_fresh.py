"""Row bands at batch 256 with a fresh input batch every step: does splitting every frame over 2-4 workgroups (several
smaller workgroups per CU, phases overlapping) beat one 1024-thread workgroup per frame now that the gather is HBM-bound?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


def main():
    pkg = load_package()
    torch.cuda.set_device(0)
    for name in sys.argv[1:] or ["1080p_80x24_truecolor", "1080p_80x24_ansi256"]:
        sw, sh, W, H, cl, rm = bench.WORKLOADS[name]
        nsets = 4
        sets = [bench.make_frames(torch, 256, sw, sh, 300 + s) for s in range(nsets)]
        st = torch.cuda.current_stream().cuda_stream
        for variant, split in ((4, 0), (1, 0), (2, 0), (4, 12), (1, 12), (2, 12), (1, 8), (2, 8), (1, 6), (2, 6), (2, 4), (2, 3)):
            plans = [bench.build_plan(pkg, t, W, H, cl, rm)[0] for t in sets]
            try:
                for p in plans:
                    p.set_variant(variant)
                    p.set_split(split)
            except RuntimeError as e:
                print(f"{name} variant {variant} rows/band {split}: {e}")
                continue
            out = torch.empty(256 * plans[0].stride, dtype=torch.uint8, device="cuda")
            ln = torch.zeros(256, dtype=torch.int32, device="cuda")
            for k in range(40):
                plans[k % nsets].render(out.data_ptr(), plans[0].stride, ln.data_ptr(), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 400
            e0.record()
            for k in range(n):
                plans[k % nsets].render(out.data_ptr(), plans[0].stride, ln.data_ptr(), st)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / n
            print(f"{name} variant {plans[0].variant} rows/band {split:2d} -> {plans[0].parts} workgroups per frame: {us:7.2f} us per "
                  f"256-frame step ({256 / us:6.2f} M frames/s)", flush=True)
            for p in plans:
                p.close()


if __name__ == "__main__":
    main()
