#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per-kernel statistics WITHOUT rocprofv3's in-process --stats pass: calls, average /
min / max begin-to-end duration per dispatch, share of the summed durations -- and, because several launches are in
flight at once, the time the GPU spent with at least one dispatch of that kernel running (union of the intervals),
the average number in flight and the BUSY TIME PER LAUNCH (union / calls of the longest gap-free run): the figure that
bench.py's `roofline.kernel_ms` has to agree with.
usage: trace_stats.py <kernel_trace.csv> <out.csv> [out.json]"""
import csv
import json
import sys


def union_of(iv):
    iv = sorted(iv)
    total, lo, hi = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > hi:
            total += hi - lo
            lo, hi = a, b
        else:
            hi = max(hi, b)
    return total + hi - lo


def longest_run(iv, gap_ns=50_000):
    iv = sorted(iv)
    runs, cur, reach = [], [iv[0]], iv[0][1]
    for a, b in iv[1:]:
        if a > reach + gap_ns:
            runs.append(cur)
            cur = []
        cur.append((a, b))
        reach = max(reach, b)
    runs.append(cur)
    return max(runs, key=len)


def main():
    by = {}
    for r in csv.DictReader(open(sys.argv[1])):
        by.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    grand = sum(b - a for iv in by.values() for a, b in iv)
    rows = []
    for name, iv in by.items():
        d = [b - a for a, b in iv]
        run = longest_run(iv)
        u = union_of(run)
        rows.append({"Name": name, "Calls": len(iv), "TotalDurationNs": sum(d), "AverageNs": sum(d) / len(d),
                     "Percentage": 100.0 * sum(d) / grand, "MinNs": min(d), "MaxNs": max(d),
                     "RunCalls": len(run), "RunAvgInFlight": sum(b - a for a, b in run) / u, "RunBusyNsPerCall": u / len(run)})
    rows.sort(key=lambda r: -r["TotalDurationNs"])
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            w.writerow({k: (f"{v:.3f}" if isinstance(v, float) else v) for k, v in r.items()})
    frame = [r for r in rows if any(k in r["Name"] for k in ("render_frames_kernel", "render_stream_kernel", "render_rows_kernel"))]
    top = max(frame, key=lambda r: r["Calls"]) if frame else rows[0]
    out = {"kernel": top["Name"].split("(")[0].replace("void ", ""), "launches": top["RunCalls"], "all_launches_in_trace": top["Calls"],
           "avg_duration_us": top["AverageNs"] / 1e3, "avg_in_flight": top["RunAvgInFlight"], "busy_us_per_launch": top["RunBusyNsPerCall"] / 1e3}
    print(json.dumps(out))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
