import os, sys, time, statistics, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
mode = os.environ.get("BENCH_SPIN", "")
if mode:
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(int(mode)))
import torch
import bench
from __graft_entry__ import load_package
pkg = load_package()
torch.cuda.set_device(0)
sw, sh, W, H, cl, rm = bench.WORKLOADS["1080p_80x24_truecolor"]
S = 4
sets = [bench.make_frames(torch, 256, sw, sh, 50 + s) for s in range(12)]
plans = [bench.build_plan(pkg, t, W, H, cl, rm)[0] for t in sets]
for p in plans:
    p.set_concurrency(S)
run = bench.Runner(torch, pkg, plans, 256, S)
run.issue(40); torch.cuda.synchronize()
for K in (20, 100):
    w = [run.region(K) for _ in range(60)]
    # variant: no spin wait, only torch synchronize
    def region2(K):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run.issue(K); torch.cuda.synchronize(); return time.perf_counter() - t0
    w2 = [region2(K) for _ in range(60)]
    print(f"flags={mode or 'default'} K={K}: spin-wait + synchronize {statistics.median(w)*1e6:.1f} us; synchronize only {statistics.median(w2)*1e6:.1f} us")
