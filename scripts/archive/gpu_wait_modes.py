"""Wall time of a K-step burst (barrier + synchronize on both sides, as bench.py's regions) for one completion-wait
mode (env ASCIICHAT_HIP_WAIT_MODE (experimental builds only), read once per process) and S streams."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from __graft_entry__ import load_package

pkg = load_package()
torch.cuda.set_device(0)
sw, sh, W, H, cl, rm = bench.WORKLOADS["1080p_80x24_truecolor"]
for S in (4, 2, 1):
    sets = [bench.make_frames(torch, 256, sw, sh, 50 + s) for s in range(12)]
    plans = [bench.build_plan(pkg, t, W, H, cl, rm)[0] for t in sets]
    for p in plans:
        p.set_concurrency(S)
    run = bench.Runner(torch, pkg, plans, 256, S)
    run.issue(40)
    torch.cuda.synchronize()
    for K in (1, 4, 20, 100):
        parts = []
        for _ in range(40):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); run.issue(K); t1 = time.perf_counter(); run.sched.wait(); t2 = time.perf_counter()
            torch.cuda.synchronize(); t3 = time.perf_counter()
            parts.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
        m = [statistics.median(p[i] for p in parts) * 1e6 for i in range(4)]
        print(f"mode {os.environ.get('ASCIICHAT_HIP_WAIT_MODE (experimental builds only)', 'default')} streams {S} K={K:3d}: issue {m[0]:6.1f} wait {m[1]:6.1f} "
              f"sync {m[2]:5.1f} total {m[3]:6.1f} us = {m[3] / K:6.2f} us/step", flush=True)
    run.sched.close()
    for p in plans:
        p.close()
