#!/bin/bash
# round 6 profile visit (the round's last build): rocprofv3 kernel traces of bench.py reduced by scripts/trace_stats.py -- the
# headline NINE times (its trace-derived fraction is a statistic now: scripts/r6_commit_profiles.py records every trace and the
# bench line carries the MEDIAN, the best and the count -- VERDICT r5 next 3), the headline one launch at a time, configs[2] /
# configs[4] from 4K sources and from sampled images, the multi-byte-palette legs -- and FETCH_SIZE / WRITE_SIZE passes.
# Outputs: gpurun_out/<tag>/ -> profiles/r06_* through scripts/r6_commit_profiles.py gpurun_out/<tag>
TAG=${1:-r6prof}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 scripts/source_id.py > $OUT/source_id.txt   # what the traces were taken on (bench.py drops a stale frac_profile)
COMMON="--warmup 5 --no-cpu --no-d2h --no-hot --no-wire --others none --batch-sweep= --extra ''"
reduce() { # name
  local csvf=$(find $OUT/t_$1 -name "*kernel_trace.csv" | head -1)
  if [ -n "$csvf" ]; then python scripts/trace_stats.py $csvf $OUT/${1}_kernel_stats.csv $OUT/${1}_trace_overlap.json > /dev/null; else echo "$1: no trace"; fi
  rm -rf $OUT/t_$1
}
trace() { # name, bench args...
  local name=$1; shift
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$name -o t -- python $GRAFT_REPO_ROOT/bench.py $COMMON "$@" > $OUT/${name}_under_rocprof.json 2> $OUT/${name}_rocprof.log )
  reduce $name
  python - <<PY
import json,csv
try:
    d=json.load(open("$OUT/${name}_under_rocprof.json")); r=d["roofline"]
    busy=[(row["Name"][:70], float(row["RunBusyNsPerCall"])/1e3, row["RunAvgInFlight"]) for row in csv.DictReader(open("$OUT/${name}_kernel_stats.csv")) if "achip::render" in row["Name"]]
    print("$name: line kernel_ms", r["kernel_ms"], "frac", r["frac"], "variant", d["config"].get("kernel_variant"), "| trace busy us/launch", busy[:2])
except Exception as e: print("$name: no line", e)
PY
}
for i in 1 2 3 4 5 6 7 8 9; do trace headline_s4_$i --steps 2000 --regions 2 --streams 4; done
trace headline_s1 --steps 400 --regions 3 --streams 1
trace k3_4k_200x60 --workload 4k_200x60_truecolor --steps 100 --regions 3 --input-sets 4 --streams 4
trace k5_4k_400x120_hb --workload 4k_400x120_halfblock --steps 40 --regions 3 --input-sets 4 --streams 4
trace k3_sampled_200x60 --workload sampled_200x60_truecolor --steps 100 --regions 3 --streams 4
trace k3_sampled_200x60_s1 --workload sampled_200x60_truecolor --steps 100 --regions 3 --streams 1
trace k5_sampled_400x240_hb --workload sampled_400x240_halfblock --steps 40 --regions 3 --streams 4
trace headline_sampled_80x24 --workload sampled_80x24_truecolor --steps 400 --regions 3 --streams 4
trace headline_sampled_80x24_s1 --workload sampled_80x24_truecolor --steps 400 --regions 3 --streams 1
trace u8_1080p_80x24_blocks --workload 1080p_80x24_truecolor_blocks --steps 400 --regions 3 --streams 4
trace u8_4k_200x60_cool --workload 4k_200x60_truecolor_cool --steps 100 --regions 3 --input-sets 4 --streams 4
trace k6_sampled_640x360_hb --workload sampled_640x360_halfblock --steps 20 --regions 3 --streams 4
trace k6_sampled_640x360_hb_s1 --workload sampled_640x360_halfblock --steps 20 --regions 3 --streams 1
trace k6_4k_640x180_hb --workload 4k_640x180_halfblock --steps 20 --regions 3 --input-sets 4 --streams 4
trace k1_mono_lone_frame --workload 640x480_80x24_mono --batch 1 --steps 400 --regions 3 --streams 1
pmc() { # name, counter, bench args...
  local name=$1 ctr=$2; shift 2
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/p_$name -o p -- python $GRAFT_REPO_ROOT/bench.py $COMMON --steps 20 --regions 3 "$@" > $OUT/pmc_$name.log 2>&1 )
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p_$name/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "achip::render" in row["Kernel_Name"]:
            k = (row["Kernel_Name"].split("(")[0][-64:], row["Counter_Name"]); acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
for (kn, c), (v, n) in sorted(acc.items()):
    print(f"$name {kn:64s} {c:12s} per-dispatch mean {v/n:14.1f}  (n={n})")
PY
  rm -rf $OUT/p_$name
}
{ pmc headline_fetch FETCH_SIZE --streams 4
  pmc headline_write WRITE_SIZE --streams 4
  pmc k3_sampled_fetch FETCH_SIZE --workload sampled_200x60_truecolor --streams 4
  pmc k3_sampled_write WRITE_SIZE --workload sampled_200x60_truecolor --streams 4
  pmc k5_sampled_fetch FETCH_SIZE --workload sampled_400x240_halfblock --streams 4
  pmc k5_sampled_write WRITE_SIZE --workload sampled_400x240_halfblock --streams 4
  pmc k5_4k_fetch FETCH_SIZE --workload 4k_400x120_halfblock --input-sets 4 --streams 4
  pmc k5_4k_write WRITE_SIZE --workload 4k_400x120_halfblock --input-sets 4 --streams 4
  pmc k6_sampled_fetch FETCH_SIZE --workload sampled_640x360_halfblock --streams 4
  pmc k6_sampled_write WRITE_SIZE --workload sampled_640x360_halfblock --streams 4
  pmc k6_4k_fetch FETCH_SIZE --workload 4k_640x180_halfblock --input-sets 4 --streams 4
  pmc k6_4k_write WRITE_SIZE --workload 4k_640x180_halfblock --input-sets 4 --streams 4; } | tee $OUT/pmc_summary.txt
# the driver's own command, twice (the round's bench line as the driver will see it)
cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --extra $OUT/bench_extra_driver_flags_$i.json > $OUT/bench_driver_stdout_$i.txt 2> $OUT/bench_driver_stderr_$i.txt; tail -c 600 $OUT/bench_driver_stdout_$i.txt; done
