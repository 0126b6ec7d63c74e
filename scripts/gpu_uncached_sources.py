#!/usr/bin/env python3
"""Where do full-frame sources belong?  The metric's launch samples 3 bytes every 72 (1080p -> 80 columns): from ordinary device
memory (hipMalloc: cached in the L2, MTYPE RW) every sample costs a 128-byte line fill -- 36.4 MB per launch against 1.5 MB
consumed, and those fills are what the launch's time is made of (46 MB through a fabric that fills lines at ~6.5 TB/s).  This
script renders the same batches from sources allocated with hipExtMallocWithFlags -- default (0), fine-grained (1), UNCACHED (3:
the L2 does not keep such lines, so a read can be a 32- / 64-byte request) -- and times the launch four in flight and one at a
time; output bytes are compared with the default allocation's.  GPU box only.
usage: gpu_uncached_sources.py [workload] [flags ...]     (ONLY_FLAG=<n> REPS=<n>: one allocation kind, for a counter pass)"""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from __graft_entry__ import load_package
pkg = load_package(); torch.cuda.set_device(0)
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipFree.argtypes = [C.c_void_p]
NAMES = {0: "default", 1: "fine-grained", 3: "uncached"}

class Slab:  # what bench.build_plan needs of a tensor
    def __init__(self, ptr, shape): self.ptr, self.shape = ptr, shape
    def data_ptr(self): return self.ptr

def alloc_sets(flag, nsets, batch, sw, sh, seed=1234):
    sets = []
    for s in range(nsets):
        t = bench.make_frames(torch, batch, sw, sh, seed + 7919 * s, "noise")
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), t.numel(), flag)
        assert rc == 0 and p.value, ("hipExtMallocWithFlags", flag, rc)
        assert hip.hipMemcpy(p, C.c_void_p(t.data_ptr()), t.numel(), 3) == 0
        sets.append(Slab(p.value, tuple(t.shape)))
        del t
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    return sets

def measure(name, flag, nsets=12, batch=256, streams=4, reps=3):
    sw, sh, W, H, cl, rm = bench.WORKLOADS[name]
    sets = alloc_sets(flag, nsets, batch, sw, sh)
    plans = []
    for t in sets:
        plan, mode = bench.build_plan(pkg, t, W, H, cl, rm, False, bench.palette_of(name))
        plan.set_concurrency(streams); plans.append(plan)
    run = bench.Runner(torch, pkg, plans, batch, streams)
    run.issue(24); torch.cuda.synchronize()
    many = [run.gpu_ms_per_step(400) * 1e3 for _ in range(reps)]
    out0 = run.outs[0].clone(); run.step = 0; run.issue(1); torch.cuda.synchronize()
    first = bytes(run.outs[0][:plans[0].stride * 4].cpu().numpy()); lens = run.lns[0].cpu().numpy().copy()
    for p in plans: p.set_concurrency(1)
    one = bench.Runner(torch, pkg, plans, batch, 1); one.issue(8)
    ones = [one.gpu_ms_per_step(200) * 1e3 for _ in range(reps)]
    same = bench.Runner(torch, pkg, plans[:1], batch, 1); same.issue(8)
    sames = [same.gpu_ms_per_step(200) * 1e3 for _ in range(reps)]
    v = plans[0].variant
    for r in (run, one, same): r.sched.close()
    for p in plans: p.close()
    for t in sets: hip.hipFree(C.c_void_p(t.ptr))
    return many, ones, sames, first, lens, v

if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "1080p_80x24_truecolor"
    flags = [int(a) for a in sys.argv[2:]] or [0, 3, 1, 0, 3]
    if os.environ.get("ONLY_FLAG"):
        flags = [int(os.environ["ONLY_FLAG"])]
    nsets = int(os.environ.get("NSETS", "12"))
    ref = None
    for flag in flags:
        many, ones, sames, first, lens, v = measure(name, flag, nsets=nsets, reps=int(os.environ.get("REPS", "3")))
        if ref is None: ref = (first, lens)
        same_bytes = first == ref[0] and (lens == ref[1]).all()
        print(f"{name} sources {NAMES[flag]:12s} (geometry {v}): four in flight {min(many):6.2f}-{max(many):5.2f} us per launch | one at a time "
              f"{min(ones):6.2f}-{max(ones):5.2f} | the same batch every step {min(sames):6.2f}-{max(sames):5.2f} | output {'identical' if same_bytes else 'DIFFERS'}", flush=True)
