"""Where does the fixed cost of a timed region go?  (issue from C / spin wait / torch.cuda.synchronize)"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from __graft_entry__ import load_package

pkg = load_package()
torch.cuda.set_device(0)
sw, sh, W, H, cl, rm = bench.WORKLOADS["1080p_80x24_truecolor"]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sets = [bench.make_frames(torch, 256, sw, sh, 50 + s) for s in range(3 * S)]
plans = [bench.build_plan(pkg, t, W, H, cl, rm)[0] for t in sets]
for p in plans:
    p.set_concurrency(S)
run = bench.Runner(torch, pkg, plans, 256, S)
run.issue(40)
torch.cuda.synchronize()
def med(f, n=40):
    return statistics.median(f() for _ in range(n)) * 1e6
def t_sync():
    t0 = time.perf_counter(); torch.cuda.synchronize(); return time.perf_counter() - t0
def t_wait_idle():
    t0 = time.perf_counter(); run.sched.wait(); return time.perf_counter() - t0
print(f"streams {S}: idle torch.cuda.synchronize {med(t_sync):.1f} us, idle spin wait {med(t_wait_idle):.1f} us")
for K in (1, 4, 20, 100):
    parts = []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); run.issue(K); t1 = time.perf_counter(); run.sched.wait(); t2 = time.perf_counter()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        parts.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
    m = [statistics.median(p[i] for p in parts) * 1e6 for i in range(4)]
    print(f"K={K:3d}: issue {m[0]:7.1f} us, spin wait {m[1]:7.1f} us, final synchronize {m[2]:6.1f} us, total {m[3]:7.1f} us = {m[3]/K:6.2f} us/step")

# the same K steps as ONE captured HIP graph (asciichat_hip_schedule_*), replayed on lane 0 and spin-waited there
import ctypes as C
lane0 = (C.c_void_p * 1)(run.lanes[0].cuda_stream)
for K in (1, 4, 20, 100):
    run.sched.graph(0, K)
    parts = []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); run.sched.replay(0, K, run.lanes[0].cuda_stream); t1 = time.perf_counter()
        pkg.lib().asciichat_hip_streams_wait(lane0, 1); t2 = time.perf_counter()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        parts.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
    m = [statistics.median(p[i] for p in parts) * 1e6 for i in range(4)]
    print(f"graph K={K:3d}: launch {m[0]:7.1f} us, spin wait {m[1]:7.1f} us, final synchronize {m[2]:6.1f} us, total {m[3]:7.1f} us = {m[3]/K:6.2f} us/step")
