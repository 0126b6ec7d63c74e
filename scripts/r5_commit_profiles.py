#!/usr/bin/env python3
"""Copies a profile visit's reduced files (scripts/gpu_r5_profiles.sh -> gpurun_out/<tag>/) into profiles/r05_* and writes
the round-5 entries of profiles/committed_profile.json: per workload the trace's busy time per launch (what bench.py's
`roofline.frac_profile` divides the algorithmic bytes by), the launches in flight under the tracer, the bench line the
profiled run printed, and the counter traffic where a --pmc pass measured it.

usage: r5_commit_profiles.py gpurun_out/<tag> [more tags whose headline_s4 traces also count ...]
A headline trace is taken from the visit whose tracer kept the MOST launches in flight (the tracer serialises short
launches to a box-dependent degree: 1.6-2.9 of the four requested); every candidate is listed in the entry."""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
NAMES = {  # trace name -> (workload key of bench.py, kernel substring)
    "headline_s4": ("1080p_80x24_truecolor", "render_stream_kernel"),
    "headline_s1": ("1080p_80x24_truecolor", "render_stream_kernel"),
    "k3_4k_200x60": ("4k_200x60_truecolor", "render_stream_kernel"),
    "k5_4k_400x120_hb": ("4k_400x120_halfblock", "render_rows_kernel"),
    "k3_sampled_200x60": ("sampled_200x60_truecolor", "render_stream_kernel"),
    "k5_sampled_400x240_hb": ("sampled_400x240_halfblock", "render_rows_kernel"),
    "headline_sampled_80x24": ("sampled_80x24_truecolor", "render_stream_kernel"),
}


def trace_row(d, name, kern):
    f = os.path.join(d, name + "_kernel_stats.csv")
    if not os.path.exists(f):
        return None
    best = None
    for r in csv.DictReader(open(f)):
        if kern in r["Name"] and (best is None or int(r["Calls"]) > int(best["Calls"])):
            best = r
    if best is None:
        return None
    line = {}
    try:
        line = json.load(open(os.path.join(d, name + "_under_rocprof.json")))
    except Exception:
        pass
    return {"kernel": re.sub(r"\(.*", "", best["Name"]).replace("void ", ""), "calls": int(best["Calls"]),
            "rocprof_avg_dispatch_us": float(best["AverageNs"]) / 1e3, "avg_in_flight": float(best["RunAvgInFlight"]),
            "busy_us_per_launch": float(best["RunBusyNsPerCall"]) / 1e3,
            "bench_kernel_ms_in_the_profiled_run": (line.get("roofline") or {}).get("kernel_ms"),
            "alg_bytes_per_launch": (line.get("roofline") or {}).get("alg_bytes_per_launch")}


def main():
    tags = [a.rstrip("/") for a in sys.argv[1:]]
    main_dir = tags[0]
    cp = json.load(open(os.path.join(PROF, "committed_profile.json")))
    for name, (wl, kern) in NAMES.items():
        cands = []  # (directory, trace name, row): the headline is traced several times per visit (headline_s4, _s4b, _s4c)
        for d in tags if name == "headline_s4" else tags[:1]:
            for nm in ([name, name + "b", name + "c"] if name == "headline_s4" else [name]):
                row = trace_row(d, nm, kern)
                if row:
                    cands.append((d, nm, row))
        if not cands:
            continue
        d, nm, row = max(cands, key=lambda c: c[2]["avg_in_flight"]) if name == "headline_s4" else cands[0]
        cands = [(c[0], c[2]) for c in cands]
        files = []
        for suf in ("_kernel_stats.csv", "_trace_overlap.json", "_under_rocprof.json"):
            src = os.path.join(d, nm + suf)
            if os.path.exists(src):
                dst = os.path.join(PROF, "r05_" + nm + suf)
                shutil.copy(src, dst)
                files.append("profiles/" + os.path.basename(dst))
        row["files"] = files
        if len(cands) > 1:
            row["candidates_busy_us_and_in_flight"] = [[round(c[1]["busy_us_per_launch"], 3), round(c[1]["avg_in_flight"], 2)] for c in cands]
        ent = cp.setdefault(wl, {"not_measured_by_this_run": True})
        r5 = ent.setdefault("round5", {"source": "profiles/r05_* (round 5, one MI355X through gpurun, scripts/gpu_r5_profiles.sh: rocprofv3 "
                                                 "--kernel-trace of bench.py, reduced by scripts/trace_stats.py; scripts/r5_commit_profiles.py)",
                                       "not_measured_by_this_run": True})
        r5["one_launch_at_a_time" if name.endswith("_s1") else "four_launches_requested_under_the_tracer"] = row
        if not name.endswith("_s1"):
            ent["frac_profile"] = {"busy_us_per_launch": row["busy_us_per_launch"], "avg_in_flight": row["avg_in_flight"],
                                   "file": files[0] if files else "profiles/",
                                   "note": "busy time per launch = union of the dispatch intervals / launches of the committed trace; "
                                           "the tracer keeps fewer short launches in flight than the unprofiled run does"}
    # counter traffic
    pmc = os.path.join(main_dir, "pmc_summary.txt")
    if os.path.exists(pmc):
        shutil.copy(pmc, os.path.join(PROF, "r05_pmc_summary.txt"))
        vals = {}
        for l in open(pmc):
            m = re.match(r"(\S+)\s+.*?(FETCH_SIZE|WRITE_SIZE)\s+per-dispatch mean\s+([\d.]+)", l)
            if m:
                vals[(m.group(1), m.group(2))] = float(m.group(3))
        for key, wl in (("k5_4k", "4k_400x120_halfblock"), ("k5_sampled", "sampled_400x240_halfblock"), ("headline", "1080p_80x24_truecolor"),
                        ("k3_sampled", "sampled_200x60_truecolor")):
            fe, wr = vals.get((key + "_fetch", "FETCH_SIZE")), vals.get((key + "_write", "WRITE_SIZE"))
            if fe and wr:
                ent = cp.setdefault(wl, {"not_measured_by_this_run": True})
                ent.setdefault("round5", {})["traffic"] = {
                    "fetch_size_kib_per_dispatch": fe, "write_size_kib_per_dispatch": wr,
                    "hbm_bytes_per_launch": 2 * fe * 1024 + wr * 1024, "file": "profiles/r05_pmc_summary.txt",
                    "note": "FETCH_SIZE on gfx950 tallies 128-byte line fills at 64 B (MI355X_MICROARCH.md): x 2"}
                if wl != "1080p_80x24_truecolor" or True:
                    ent["traffic"] = dict(ent["round5"]["traffic"], alg_bytes_per_launch=(ent.get("traffic") or {}).get("alg_bytes_per_launch"))
    json.dump(cp, open(os.path.join(PROF, "committed_profile.json"), "w"), indent=1)
    for wl, ent in cp.items():
        if "frac_profile" in ent:
            print(wl, ent["frac_profile"]["busy_us_per_launch"], ent["frac_profile"]["avg_in_flight"])


if __name__ == "__main__":
    main()
