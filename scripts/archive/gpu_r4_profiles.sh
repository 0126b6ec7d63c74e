#!/bin/bash
# round 4 profile visit: for EVERY BASELINE config a rocprofv3 kernel trace of the bench command
# with long bursts and no in-process --stats pass (statistics + busy time per launch come from the trace CSV,
# scripts/trace_stats.py), the bench line each profiled run printed, the device-stamp busy time with no profiler attached,
# the tick (sampled-image ingest: no scatter kernel) under the trace, and FETCH / WRITE counter passes of the render from
# sampled images.  Outputs: gpurun_out/<tag>/ -> copied to profiles/r04_* by hand.
TAG=${1:-r4prof}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
COMMON="--warmup 5 --no-cpu --no-d2h --no-hot --no-wire --others none --extra ''"
trace() { # name, bench args...
  local name=$1; shift
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$name -o t -- python $GRAFT_REPO_ROOT/bench.py $COMMON "$@" > $OUT/${name}_under_rocprof.json 2> $OUT/${name}_rocprof.log )
  local csvf=$(find $OUT/t_$name -name "*kernel_trace.csv" | head -1)
  if [ -n "$csvf" ]; then python scripts/trace_stats.py $csvf $OUT/${name}_kernel_stats.csv $OUT/${name}_trace_overlap.json; else echo "$name: no trace"; fi
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${name}_under_rocprof.json")); r=d["roofline"]
    print("$name: line kernel_ms", r["kernel_ms"], "frac", r["frac"], "alg", r["alg_bytes_per_launch"], "value", d["value"])
except Exception as e: print("$name: no line", e)
PY
  rm -rf $OUT/t_$name
}
trace headline_s4 --steps 400 --regions 3 --streams 4
trace headline_s1 --steps 400 --regions 3 --streams 1
trace k2_ansi256 --workload 1080p_80x24_ansi256 --steps 400 --regions 3 --streams 4
trace k3_4k_200x60 --workload 4k_200x60_truecolor --steps 100 --regions 3 --input-sets 4 --streams 4
trace k5_4k_400x120_hb --workload 4k_400x120_halfblock --steps 40 --regions 3 --input-sets 4 --streams 4
trace hb_1080p_80x24 --workload 1080p_80x24_halfblock --steps 400 --regions 3 --streams 4
trace k1_mono_single --workload 640x480_80x24_mono --batch 1 --steps 400 --regions 3 --streams 1
trace sampled_80x24 --workload sampled_80x24_truecolor --steps 400 --regions 3 --streams 4
trace sampled_80x24_s1 --workload sampled_80x24_truecolor --steps 400 --regions 3 --streams 1
trace k4_grid9_256 --workload grid9 --steps 100 --regions 3
trace k4_grid9_nine --workload grid9 --batch 9 --steps 100 --regions 3
# the tick (blobs -> sampled images -> render + wire -> packed host copy) under the trace
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_tick -o t -- python $GRAFT_REPO_ROOT/scripts/gpu_tick_sweep.py sampled_images > $OUT/tick_under_rocprof.json 2> $OUT/tick_rocprof.log )
python scripts/trace_stats.py $(find $OUT/t_tick -name "*kernel_trace.csv" | head -1) $OUT/tick_kernel_stats.csv > /dev/null; rm -rf $OUT/t_tick
# busy time per launch from the kernels' own timestamps, no profiler
for a in "1080p_80x24_truecolor 4" "1080p_80x24_truecolor 1" "1080p_80x24_ansi256 4" "4k_200x60_truecolor 4" "sampled_80x24_truecolor 4"; do
  python scripts/gpu_busy_stamps.py $a 2>/dev/null | tail -1 >> $OUT/busy_stamps.jsonl
done
cat $OUT/busy_stamps.jsonl
# HBM traffic of the render from sampled images (separate counter passes)
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/p_$c -o p -- python $GRAFT_REPO_ROOT/bench.py $COMMON --workload sampled_80x24_truecolor --steps 100 --regions 3 --streams 4 > $OUT/pmc_$c.log 2>&1 )
done
python - <<PY | tee $OUT/pmc_sampled_summary.txt
import csv, glob, collections
print("# rocprofv3 --pmc passes (separate runs, counters only) over: bench.py --workload sampled_80x24_truecolor --steps 100 --streams 4")
for name in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob("$OUT/p_%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for row in csv.DictReader(open(f)):
            kn = row["Kernel_Name"]
            if "render_stream_kernel" in kn:
                k = (kn.split("(")[0].replace("void achip::",""), row["Counter_Name"]); acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for (kn,k),(v,n) in sorted(acc.items()):
            if n >= 20: print(f"{kn[:58]:58s} {k:12s} per-dispatch mean {v/n:14.1f} KiB (n={n})")
PY
rm -rf $OUT/p_FETCH_SIZE $OUT/p_WRITE_SIZE
ls $OUT
