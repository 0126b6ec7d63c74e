"""Device-side timeline of a short burst, without a profiler attached: K steps issued from C on S streams after a
synchronize (as bench.py's timed regions), every launch writing its waves' entry / last-store timestamps (100 MHz wall
clock).  Per launch index: start = earliest wave entry, end = latest final stamp, relative to the burst's first start;
medians over the bursts.  Also the host's wall time per burst and when (host clock) the issue loop returned."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from __graft_entry__ import load_package

pkg = load_package()
torch.cuda.set_device(0)
sw, sh, W, H, cl, rm = bench.WORKLOADS["1080p_80x24_truecolor"]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sets = [bench.make_frames(torch, 256, sw, sh, 50 + s) for s in range(12)]
plans = [bench.build_plan(pkg, t, W, H, cl, rm)[0] for t in sets]
for p in plans:
    p.set_concurrency(S)
waves = pkg.lib().achip_variant_block(plans[0].variant) // 64
run = bench.Runner(torch, pkg, plans, 256, S)
run.issue(40)
torch.cuda.synchronize()
stride = 256 * waves * 8
prof = torch.zeros(K * stride, dtype=torch.int64, device="cuda")
starts, ends, walls, issues = [], [], [], []
for it in range(24):
    prof.zero_()
    torch.cuda.synchronize()
    time.sleep(0.001)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.sched.issue_profiled(0, K, prof.data_ptr(), stride)
    t1 = time.perf_counter()
    run.sched.wait()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    a = prof.cpu().numpy().reshape(K, 256, waves, 8)
    act = a[:, :, :, 7] != 0
    st = np.array([a[k, :, :, 0][act[k]].min() for k in range(K)])
    en = np.array([a[k, :, :, 7][act[k]].max() for k in range(K)])
    if it >= 4:
        starts.append((st - st.min()) / 100.0)
        ends.append((en - st.min()) / 100.0)
        walls.append((t2 - t0) * 1e6)
        issues.append((t1 - t0) * 1e6)
starts, ends = np.array(starts), np.array(ends)
print(f"# variant {plans[0].variant}, {S} streams, bursts of {K} steps (kernels with timestamps on: slightly slower than production)")
print(f"# host: wall {statistics.median(walls):.1f} us per burst, issue loop returns after {statistics.median(issues):.1f} us;"
      f" device: first start -> last end {np.median(ends.max(axis=1)):.1f} us")
print("# step  stream  start_us   end_us  duration_us")
for k in range(K):
    print(f"  {k:3d}   {k % S:3d}   {np.median(starts[:, k]):8.1f} {np.median(ends[:, k]):8.1f}  {np.median(ends[:, k] - starts[:, k]):8.1f}")
