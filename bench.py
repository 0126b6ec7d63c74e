#!/usr/bin/env python3
"""bench.py -- frames/s of the MI355X render path on BASELINE.json's metric.

A "step" is one pass of the hot path (one asciichat_hip_plan_render launch) over one batch of
device-resident synthetic frames.  Default workload = the configuration the metric is quoted on:
batch=256 x 1080p -> 80x24 truecolor foreground (BASELINE.json `metric`; `configs[1]` is the same
shape in ANSI-256, reported under "other_workloads").  One process per GPU; frames are independent,
so N>1 shards the batch across ranks with no data-path collective (weak scaling: 256 frames/rank).

Prints ONE JSON line on rank 0 (see the task contract): value = whole-job frames/s with inputs already
resident in HBM; roofline = algorithmic bytes / kernel time vs the 8 TB/s HBM peak; cpu_baseline = the
CPU oracle (a port of the reference's scalar path) timed on this host on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PALETTE_STANDARD = "   ...',;:clodxkO0KXNWM"  # PALETTE_CHARS_STANDARD (palette.h:161)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec

WORKLOADS = {
    # name: (src_w, src_h, W, H, color_level, render_mode, mode_name)
    "1080p_80x24_truecolor": (1920, 1080, 80, 24, 3, 0),
    "1080p_80x24_ansi256": (1920, 1080, 80, 24, 2, 0),
    "4k_200x60_truecolor": (3840, 2160, 200, 60, 3, 0),
    "4k_400x120_halfblock": (3840, 2160, 400, 120, 3, 2),
    "640x480_80x24_mono": (640, 480, 80, 24, 0, 0),
}


def make_frames(torch, batch, w, h, seed):
    """Device-resident synthetic S-noise-like frames: uniform random RGB24 (every cell changes colour,
    no runs -- the worst case for output size, as in BASELINE.md's 'noise' rows)."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.randint(0, 256, (batch, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)


def build_plan(pkg, frames_t, W, H, cl, rm):
    b, h, w, _ = frames_t.shape
    mode = pkg.lib().achip_mode_from_caps(cl, rm)
    descs = []
    base = frames_t.data_ptr()
    for i in range(b):
        f = pkg.frame_setup(base + i * h * w * 3, w, h, W, H, rm, False, False, False)
        descs.append(f)
    return pkg.Plan(mode, PALETTE_STANDARD, descs), mode


def time_steps(torch, plans, outs, lns, streams, steps, warmup, dist=None):
    """K launches: launch k renders input batch k % len(plans) on stream k % len(streams) into that stream's own output
    slab (len(plans) is a multiple of len(streams), so a plan always runs on the same stream).  One stream = launches
    back to back; several = that many independent batches in flight.  Returns wall seconds, the GPU time of the whole
    region (events on the current stream, which every launch stream is fenced against on both sides) and the average
    duration of ONE launch (events on each launch stream around its share of the launches)."""
    P, S = len(plans), len(streams)
    assert P % S == 0
    cur = torch.cuda.current_stream()

    def launch(k):
        s = k % S
        plans[k % P].render(outs[s].data_ptr(), plans[0].stride, lns[s].data_ptr(), streams[s].cuda_stream)

    for k in range(warmup):
        launch(k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    b = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    t0 = time.perf_counter()
    e0.record(cur)
    for s in range(S):
        if streams[s] != cur:
            streams[s].wait_event(e0)
        b[s].record(streams[s])
    for k in range(steps):
        launch(k)
    for s in range(S):
        e[s].record(streams[s])
        if streams[s] != cur:
            cur.wait_event(e[s])
    e1.record(cur)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1)
    per_launch = [b[s].elapsed_time(e[s]) / len(range(s, steps, S)) for s in range(S) if s < steps]
    return wall, gpu_ms, sum(per_launch) / len(per_launch)


def launch_durations(torch, plans, outs, lns, streams, n):
    """Average duration of ONE launch while len(streams) launches are in flight: a HIP event pair around every launch,
    on the stream it is launched on (the stream is busy, so the first event completes when the previous kernel of that
    stream ends).  This is the per-kernel figure rocprofv3 --kernel-trace --stats reports."""
    P, S = len(plans), len(streams)
    pairs = []
    for k in range(n + 2 * S):
        s = k % S
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record(streams[s])
        plans[k % P].render(outs[s].data_ptr(), plans[0].stride, lns[s].data_ptr(), streams[s].cuda_stream)
        b.record(streams[s])
        pairs.append((a, b))
    torch.cuda.synchronize()
    d = [a.elapsed_time(b) for a, b in pairs[2 * S:]]  # the first launches start on idle streams
    return sum(d) / len(d)


def kernel_time_events(torch, plans, out, ln, reps):
    """Per-launch duration: HIP events recorded on the launch stream around EACH launch (idle stream
    in between), averaged -- contains the launch latency the back-to-back figure hides."""
    stream = torch.cuda.current_stream().cuda_stream
    tot = 0.0
    for k in range(reps):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        plans[k % len(plans)].render(out.data_ptr(), plans[0].stride, ln.data_ptr(), stream)
        b.record()
        b.synchronize()
        tot += a.elapsed_time(b)
    return tot / reps


def run_workload(torch, pkg, name, batch, steps, warmup, dist=None, seed=1234, variant=-1, input_sets=4, streams=1,
                 serial_leg=True):
    """`streams` independent batches are kept in flight on separate HIP streams (the reference's model is one render
    thread per client, src/server/render.c:1233): a launch is a gather burst (HBM-bound) followed by token work
    (latency-bound, HBM idle), and launches in flight overlap those phases (profiles/r01_overlap.txt).
    input_sets x streams independent batches of source frames are rendered round-robin: a video tick never renders the
    frames of the tick before, and the sampled lines of ONE batch (36 MB at 1080p->80x24) would otherwise be served by
    the 256 MB Infinity Cache from the second step on (profiles/r01_input_sets.txt)."""
    sw, sh, W, H, cl, rm = WORKLOADS[name]
    nsets = input_sets * streams
    sets = [make_frames(torch, batch, sw, sh, seed + 7919 * s) for s in range(nsets)]
    plans = []
    for t in sets:
        plan, mode = build_plan(pkg, t, W, H, cl, rm)
        plan.set_concurrency(streams)
        if variant >= 0:
            plan.set_variant(variant)
        plans.append(plan)
    cur = torch.cuda.current_stream()
    lanes = [cur] + [torch.cuda.Stream() for _ in range(streams - 1)]
    outs = [torch.empty(batch * plans[0].stride, dtype=torch.uint8, device="cuda") for _ in range(streams)]
    lns = [torch.zeros(batch, dtype=torch.int32, device="cuda") for _ in range(streams)]
    rows = 2 * H if rm == 2 else H
    out_bytes = []
    for g in range(0, nsets, streams):  # exact output bytes of every set (sampled RGB consumed + exact output)
        for s in range(streams):
            plans[g + s].render(outs[s].data_ptr(), plans[0].stride, lns[s].data_ptr(), lanes[s].cuda_stream)
        torch.cuda.synchronize()
        for s in range(streams):
            lens = lns[s].cpu().numpy().astype("uint32")
            assert (lens < 0xFFFFFFF0).all(), "kernel reported overflow/bad descriptor"
            out_bytes.append(int(lens.sum()))
    wall, gpu_ms, stream_ms = time_steps(torch, plans, outs, lns, lanes, steps, warmup, dist)
    launch_ms = launch_durations(torch, plans, outs, lns, lanes, max(60, steps // 2))
    alg_bytes = int(sum(out_bytes) / len(out_bytes)) + batch * 3 * W * rows
    res = dict(name=name, mode=pkg.MODE_NAMES[mode], batch=batch, wall_s=wall, gpu_ms=gpu_ms, steps=steps,
               out_bytes_per_frame=sum(out_bytes) / len(out_bytes) / batch, alg_bytes_per_launch=alg_bytes,
               launch_ms=launch_ms, stream_ms_per_launch=stream_ms, ms_per_step_gpu=gpu_ms / steps,
               variant=plans[0].variant, input_sets=nsets,
               streams=streams, frames=sets, plan=plans[0], plans=plans, out=outs[0], ln=lns[0], serial=None)
    if serial_leg:
        # the same steps issued back to back on ONE stream (each plan re-chooses its geometry for the whole GPU), next
        # to it the per-launch event-pair time (with launch latency) and the same batch rendered every step
        for plan in plans:
            plan.set_concurrency(1)
            if variant >= 0:
                plan.set_variant(variant)
        n1 = max(40, steps // 2)
        _, g1, l1 = time_steps(torch, plans, outs[:1], lns[:1], lanes[:1], n1, 8, None)
        pair = kernel_time_events(torch, plans, outs[0], lns[0], 20)
        _, gh, _ = time_steps(torch, plans[:1], outs[:1], lns[:1], lanes[:1], n1, 8, None)
        res["serial"] = {"kernel_ms": g1 / n1, "frames_per_s": batch * n1 / (g1 * 1e-3), "kernel_variant": plans[0].variant,
                         "kernel_ms_event_pair": pair, "kernel_ms_same_batch_every_step": gh / n1,
                         "roofline_frac": alg_bytes / (g1 / n1 * 1e-3) / 1e9 / HBM_PEAK_GBS}
    return res


def time_with_d2h(torch, plan, n, steps):
    """the same steps with the output copied to pinned host memory after every launch -- slab and
    lengths share one allocation so that they cross PCIe in ONE transfer (a second, tiny copy behind the big one
    costs more than the big one).  The PCIe-inclusive rate, never `value`."""
    stream = torch.cuda.current_stream().cuda_stream
    slab = n * plan.stride
    buf = torch.empty(slab + 4 * n, dtype=torch.uint8, device="cuda")
    host = torch.empty(buf.shape, dtype=buf.dtype, pin_memory=True)
    out_ptr, len_ptr = buf.data_ptr(), buf.data_ptr() + slab

    def step():
        plan.render(out_ptr, plan.stride, len_ptr, stream)
        host.copy_(buf, non_blocking=True)

    for _ in range(30):  # the first transfers into a fresh pinned block are slow
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, int(buf.numel())


def run_grid9(torch, pkg, steps, warmup):
    """BASELINE configs[3]: nine 1080p sources -> the 3x3 grid at 160x48 for each of the nine clients.  One launch
    renders the nine client frames straight from the sources (the W x 2H canvas of create_multi_source_composite
    is virtual); with 9 frames the launch is cut into row bands automatically."""
    import ctypes as C
    n, sw, sh, tw, th = 9, 1920, 1080, 160, 48
    src = make_frames(torch, n, sw, sh, 4321)
    ptrs = (C.c_void_p * n)(*[src.data_ptr() + i * sh * sw * 3 for i in range(n)])
    ws, hs = (C.c_int * n)(*([sw] * n)), (C.c_int * n)(*([sh] * n))
    comp = pkg.Composite()
    pkg.lib().achip_composite_setup(C.byref(comp), ptrs, ws, hs, n, tw, th)
    comp_dev = C.c_void_p()
    assert pkg.lib().asciichat_hip_composite_upload(C.byref(comp), C.byref(comp_dev)) == 0
    descs = []
    for _ in range(n):  # every client looks at the same grid (stream.c:790-854: aspect + padding on)
        f = pkg.frame_setup(None, tw, 2 * th, tw, th, 0, True, True, False)
        f.comp = comp_dev.value
        descs.append(f)
    plan = pkg.Plan(pkg.lib().achip_mode_from_caps(3, 0), PALETTE_STANDARD, descs)
    out = torch.empty(n * plan.stride, dtype=torch.uint8, device="cuda")
    ln = torch.zeros(n, dtype=torch.int32, device="cuda")
    wall, gpu_ms, _ = time_steps(torch, [plan], [out], [ln], [torch.cuda.current_stream()], steps, warmup, None)
    lens = ln.cpu().numpy().astype("uint32")
    assert (lens < 0xFFFFFFF0).all()
    cells = int(sum(d.out_w * d.out_h for d in descs))
    res = {"frames_per_s": n * steps / wall, "kernel_ms": gpu_ms / steps, "out_bytes_per_frame": float(lens.mean()),
           "alg_bytes_per_launch": int(lens.sum()) + 3 * cells, "kernel_variant": plan.variant, "bands_per_frame": plan.parts}
    res["roofline_GBps"] = res["alg_bytes_per_launch"] / (res["kernel_ms"] * 1e-3) / 1e9
    res["roofline_frac"] = res["roofline_GBps"] / HBM_PEAK_GBS
    plan.close()
    pkg.lib().asciichat_hip_free(comp_dev)
    return res


def cpu_baseline(name, budget_s=12.0):
    """Times the CPU oracle (port of the reference's scalar path) on this host: 1 thread and all cores,
    on a bounded sample of the same workload (same frame shape, S-noise input, same caps)."""
    import numpy as np

    import orc

    sw, sh, W, H, cl, rm = WORKLOADS[name]
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    L = orc.lib()
    pal = orc.PALETTE_STANDARD.encode()
    nb = C.c_uint64()
    # calibrate
    t = L.orc_bench_convert(img.ctypes.data, sw, sh, W, H, cl, rm, pal, 20, 1, C.byref(nb))
    per = max(t / 20, 1e-7)
    it1 = max(50, int(budget_s * 0.4 / per))
    t1 = L.orc_bench_convert(img.ctypes.data, sw, sh, W, H, cl, rm, pal, it1, 1, C.byref(nb))
    cores = os.cpu_count() or 1
    th = min(cores, 256)
    itn = max(20, int(budget_s * 0.6 / per))
    tn = L.orc_bench_convert(img.ctypes.data, sw, sh, W, H, cl, rm, pal, itn, th, C.byref(nb))
    return {
        "value": it1 / t1, "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": f"{it1} x ({sw}x{sh}->{W}x{H}, color_level={cl}, render_mode={rm}, uniform-noise frame) on 1 thread; "
                  f"{itn} per thread on {th} threads",
        "all_cores": {"value": itn * th / tn, "cores": th},
    }


def load_pmc_traffic(workload, variant, batch):
    """profiles/pmc_traffic.json: {workload: {"variant": v, "batch": b, "fetch_size_kb": F, "write_size_kb": W, ...}}"""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        ent = json.load(open(path)).get(workload)
    except Exception:
        return None
    if not ent or ent.get("variant") != variant or ent.get("batch") != batch:
        return None
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  On gfx950 FETCH_SIZE tallies 128-byte line fetches at 64 B
    # (MI355X_MICROARCH.md, "HBM"); calibrated for this kernel's access pattern by scripts/ubench/sparse_fetch.hip
    # (profiles/r01_ubench_sparse_fetch.txt): a lone dword load moves one whole 128-byte line, so the x2 correction
    # applies to the sparse gathers as well (and 2 x FETCH_SIZE equals the bytes of the sampled source rows).
    fetch, write = ent["fetch_size_kb"] * 1024.0, ent["write_size_kb"] * 1024.0
    return {"hbm_bytes": 2 * fetch + write, "fetch_bytes_raw": fetch, "fetch_bytes_x2_corrected": 2 * fetch,
            "write_bytes": write, "source": ent.get("source", "profiles/pmc_traffic.json")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--workload", default="1080p_80x24_truecolor", choices=sorted(WORKLOADS))
    ap.add_argument("--others", default="1080p_80x24_ansi256,4k_200x60_truecolor,4k_400x120_halfblock",
                    help="comma list of extra workloads reported under other_workloads (N=1 only); '' = none")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-d2h", action="store_true", help="skip the PCIe-inclusive with_d2h leg (profiling runs)")
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--input-sets", type=int, default=4,
                    help="independent batches of source frames per stream, rendered round-robin (1 = the same batch every step)")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent batches kept in flight on separate HIP streams (1 = one launch at a time)")
    ap.add_argument("--no-hot", action="store_true",
                    help="skip the one-launch-at-a-time / same-batch comparison legs (profiling runs: keeps the kernel trace to the timed launches)")
    args = ap.parse_args()

    import torch

    from __graft_entry__ import load_package

    pkg = load_package()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available() or pkg.lib().asciichat_hip_device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (this path has no CPU fallback)")
    # one process per GPU; ASCIICHAT_BENCH_BACKEND=gloo lets several ranks share a GPU for testing the N > 1 control flow
    backend = os.environ.get("ASCIICHAT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as d

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            d.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            d.init_process_group(backend)
        dist = d

    res = run_workload(torch, pkg, args.workload, args.batch, args.steps, args.warmup, dist, seed=1234 + rank,
                       variant=args.variant, input_sets=args.input_sets, streams=args.streams,
                       serial_leg=not args.no_hot)
    wall = res["wall_s"]
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    total_frames = args.batch * args.steps * world
    value = total_frames / wall

    # Roofline of the frame kernel.  With `launches_in_flight` launches overlapping, what a launch COSTS is the GPU time of
    # the timed region (HIP events fenced against every launch stream) divided by K: `kernel_ms`, and
    # achieved = B_alg / kernel_ms.  `kernel_ms_on_stream` is the time a launch occupies its own stream (a HIP event
    # pair around every launch, in a leg right after the timed region): its begin->end duration plus the wait for CUs
    # behind the other streams' workgroups.  rocprofv3 --kernel-trace reports the begin->end part; the committed trace
    # (profiles/r01_bench_kernel_stats.csv, summarised by scripts/trace_overlap.py in profiles/bench_trace_overlap.json)
    # gives average duration / average launches in flight = busy time per launch, which is what agrees with
    # `kernel_ms`.  One stream: all three coincide (`one_launch_at_a_time`; 14.009 vs 14.011 us in round 1).
    eff_ms = res["ms_per_step_gpu"]
    achieved = res["alg_bytes_per_launch"] / (eff_ms * 1e-3) / 1e9
    line = {
        "metric": "frames/sec, 1080p->80x24 truecolor (batch of independent client frames)",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8",
        "data": f"synthetic (uniform-random RGB24 frames generated on device, resident in HBM; {res['input_sets']} independent "
                "batches rendered round-robin so no step re-reads the frames of the step before)",
        "config": {"workload": args.workload, "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                   "src": f"{WORKLOADS[args.workload][0]}x{WORKLOADS[args.workload][1]}",
                   "grid": f"{WORKLOADS[args.workload][2]}x{WORKLOADS[args.workload][3]}", "mode": res["mode"],
                   "parallelism": f"frames sharded over {world} rank(s), no data-path collective",
                   "kernel_variant": res["variant"], "input_sets": res["input_sets"],
                   "launches_in_flight": args.streams},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "alg_bytes_per_launch": res["alg_bytes_per_launch"], "kernel_ms": eff_ms,
                     "launches_in_flight": args.streams, "kernel_ms_on_stream": res["launch_ms"],
                     "kernel_ms_note": "kernel_ms = GPU time of the timed region / K while launches_in_flight launches "
                                       "overlap (= what one launch costs); rocprofv3 reports each dispatch begin-to-end "
                                       "(rocprof_kernel_trace.avg_duration_us), and that divided by the average number "
                                       "in flight is the same busy time per launch (busy_us_per_launch)",
                     "out_bytes_per_frame": res["out_bytes_per_frame"]},
    }
    if res["serial"] is not None:
        line["one_launch_at_a_time"] = res["serial"]
    # HBM traffic per launch comes from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a
    # pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"); scripts/pmc_run.sh collects them and the summary is
    # committed under profiles/.  When a summary for this workload and kernel geometry exists it is reported here.
    trace = os.path.join(ROOT, "profiles", "bench_trace_overlap.json")
    if os.path.exists(trace):
        try:
            line["roofline"]["rocprof_kernel_trace"] = json.load(open(trace))
        except Exception:
            pass
    traffic = load_pmc_traffic(args.workload, res["variant"], args.batch)
    if traffic is not None:
        line["roofline"]["traffic"] = traffic["hbm_bytes"]
        line["roofline"]["traffic_detail"] = traffic
    if rank == 0 and world == 1:
        if not args.no_d2h:
            d2h_s, d2h_bytes = time_with_d2h(torch, res["plan"], args.batch, max(50, args.steps // 4))
            line["with_d2h"] = {"ms_per_step": d2h_s * 1e3, "frames_per_s": args.batch / d2h_s,
                                "copied_bytes_per_step": d2h_bytes, "GBps": d2h_bytes / d2h_s / 1e9,
                                "note": "whole fixed-stride slab + lengths to pinned host memory after every launch; PCIe-bound"}
        if not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(args.workload)
        others = {}
        for name in [s for s in args.others.split(",") if s]:
            if name == args.workload:
                continue
            del res
            torch.cuda.empty_cache()
            b = args.batch
            res = run_workload(torch, pkg, name, b, max(40, args.steps // 4), 8, None, input_sets=args.input_sets,
                               streams=args.streams, serial_leg=not args.no_hot)
            k = res["ms_per_step_gpu"]
            a = res["alg_bytes_per_launch"] / (k * 1e-3) / 1e9
            others[name] = {"frames_per_s": b * res["steps"] / res["wall_s"], "kernel_ms": k,
                            "kernel_ms_on_stream": res["launch_ms"], "launches_in_flight": args.streams,
                            "out_bytes_per_frame": res["out_bytes_per_frame"], "roofline_GBps": a,
                            "roofline_frac": a / HBM_PEAK_GBS, "kernel_variant": res["variant"],
                            "input_sets": res["input_sets"], "one_launch_at_a_time": res["serial"]}
        del res
        torch.cuda.empty_cache()
        if args.others:
            others["grid9_1080p_160x48_truecolor"] = run_grid9(torch, pkg, max(10, args.steps // 10), 3)
        line["other_workloads"] = others
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
