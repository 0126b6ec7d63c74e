#!/usr/bin/env python3
"""README.md's first table from the committed files of the round's last profile visit: profiles/r06_bench_extra_driver_flags_{1,2}.json
(the driver's command, twice), the committed traces (profiles/r06_<name>_kernel_stats.csv + _under_rocprof.json) and
profiles/committed_profile.json.  Prints the table rows; paste them into README.md."""
import csv
import json
import os

P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def rng(vals, fmt):
    vals = [v for v in vals if v is not None]
    if not vals:
        return "—"
    lo, hi = fmt(min(vals)), fmt(max(vals))
    return lo if lo == hi else f"{lo}–{hi}"


def trace_frac(name):
    try:
        line = json.load(open(os.path.join(P, f"r06_{name}_under_rocprof.json")))
        alg = line["roofline"]["alg_bytes_per_launch"]
        best = None
        for r in csv.DictReader(open(os.path.join(P, f"r06_{name}_kernel_stats.csv"))):
            if "achip::render" in r["Name"] and (best is None or int(r["Calls"]) > int(best["Calls"])):
                best = r
        busy = float(best["RunBusyNsPerCall"]) / 1e3
        return alg / (busy * 1e-6) / 8e12, float(best["RunAvgInFlight"]), busy
    except Exception:
        return None


rows = {}
extra = []
for i in (1, 2):
    d = json.load(open(os.path.join(P, f"r06_bench_extra_driver_flags_{i}.json")))
    extra.append(d)

    def add(name, e):
        one = e.get("one_launch_at_a_time") or {}
        rows.setdefault(name, []).append((e.get("kernel_ms"), e.get("frames_per_s"), e.get("roofline_frac"), one.get("kernel_ms"), one.get("roofline_frac")))

    add("1080p_80x24_truecolor", dict(kernel_ms=d["roofline"]["kernel_ms"], frames_per_s=d["value"], roofline_frac=d["roofline"]["frac"],
                                      one_launch_at_a_time=d["one_launch_at_a_time"]))
    for k, e in d["other_workloads"].items():
        if "frames_per_s" in e and "+" not in k:
            add(k, e)
TRACE = {"1080p_80x24_truecolor": "headline_s4_median", "4k_200x60_truecolor": "k3_4k_200x60", "4k_400x120_halfblock": "k5_4k_400x120_hb",
         "sampled_200x60_truecolor": "k3_sampled_200x60", "sampled_400x240_halfblock": "k5_sampled_400x240_hb",
         "1080p_80x24_truecolor_blocks": "u8_1080p_80x24_blocks", "4k_640x180_halfblock": "k6_4k_640x180_hb", "sampled_640x360_halfblock": "k6_sampled_640x360_hb"}
for name, v in rows.items():
    us = rng([x[0] * 1e3 if x[0] else None for x in v], lambda a: f"{a:.2f}" if a < 100 else f"{a:.0f}")
    fps = rng([x[1] for x in v], lambda a: f"{a / 1e6:.2f} M" if a >= 1e6 else f"{a / 1e3:.0f} k")
    fr = rng([x[2] for x in v], lambda a: f"{a:.3f}")
    one = rng([x[3] * 1e3 if x[3] else None for x in v], lambda a: f"{a:.1f}" if a < 100 else f"{a:.0f}")
    onef = rng([x[4] for x in v], lambda a: f"{a:.2f}")
    t = trace_frac(TRACE[name]) if name in TRACE else None
    tr = f"{t[0]:.3f}" + (f" ({t[1]:.1f} in flight under the tracer)" if t[1] < 2.0 else "") if t else "—"
    print(f"| {name} | {us} | {fps} | {fr} | {tr} | {one} ({onef}) |")
cp = json.load(open(os.path.join(P, "committed_profile.json")))
hp = cp["1080p_80x24_truecolor"]["round6"]["headline_traces_busy_us_and_in_flight"]
alg = extra[0]["roofline"]["alg_bytes_per_launch"]
print("headline traces (busy us, in flight):", hp, "-> fractions", [round(alg / (b * 1e-6) / 8e12, 3) for b, _ in hp])
print("frac_timed_region:", [round(d["roofline"].get("frac_timed_region", 0), 3) for d in extra], "cpu:", [round(d["cpu_baseline"]["value"]) for d in extra],
      [round(d["cpu_baseline"]["all_cores"]["value"]) for d in extra], "tick:", [(round(d["tick_e2e"]["sampled_images"]["frames_per_s"]), round(d["tick_e2e"]["sampled_images_pipelined"]["frames_per_s"])) for d in extra],
      "grid nine direct us:", [round(d["other_workloads"]["grid9_1080p_160x48_truecolor"]["nine_targets_direct"]["ms_per_step"] * 1e3, 2) for d in extra])
