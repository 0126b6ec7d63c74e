"""Cross-check of the pipelined bench against rocprofv3's kernel trace: with several launches in flight the begin->end
duration rocprofv3 reports per dispatch is not the time a launch costs.  From the trace's start/end timestamps of the
frame kernel: average duration, the time the GPU spent with at least one of them running (union of the intervals), the
average number in flight (sum of durations / union) and the busy time per launch (union / launches) -- the figure
bench.py's `ms_per_launch_effective` (GPU time of the timed region / K) has to agree with.
usage: trace_overlap.py <kernel_trace.csv> [out.json]"""
import csv
import json
import sys


def main():
    rows = [r for r in csv.DictReader(open(sys.argv[1]))
            if any(k in r["Kernel_Name"] for k in ("render_frames_kernel", "render_stream_kernel", "render_rows_kernel"))]
    by = {}
    for r in rows:
        by.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    name, iv = max(by.items(), key=lambda kv: len(kv[1]))  # the kernel the timed region launches
    iv.sort()
    # longest run of launches without an idle gap of more than 50 us between them = warm-up + timed region + event-pair leg
    runs, cur = [], [iv[0]]
    reach = iv[0][1]
    for a, b in iv[1:]:
        if a > reach + 50_000:
            runs.append(cur)
            cur = []
        cur.append((a, b))
        reach = max(reach, b)
    runs.append(cur)
    run = max(runs, key=len)
    union, lo, hi = 0, run[0][0], run[0][1]
    for a, b in run[1:]:
        if a > hi:
            union += hi - lo
            lo, hi = a, b
        else:
            hi = max(hi, b)
    union += hi - lo
    total = sum(b - a for a, b in run)
    out = {"kernel": name.replace("void ", ""), "launches": len(run), "avg_duration_us": total / len(run) / 1e3,
           "avg_in_flight": total / union, "busy_us_per_launch": union / len(run) / 1e3,
           "all_launches_in_trace": len(iv), "avg_duration_all_us": sum(b - a for a, b in iv) / len(iv) / 1e3}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
