#!/bin/bash
# round 6: nine more kernel traces of the metric's launch into an existing profile visit's directory (headline_s4_10 .. _18), so that the
# trace-derived fraction of the bench line is the median of EIGHTEEN traces taken on two boxes (boxes differ by +-5 % and in how many of the four
# launches the tracer keeps in flight); scripts/r6_commit_profiles.py takes every headline_s4_* file it finds.  usage: gpu_r6_headline_more.sh <tag> [first index: 10]   (EVERY trace taken counts: the statistic is over all headline_s4_* files of the directory)
TAG=${1:-r6prof}; START=${2:-10}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 scripts/source_id.py > $OUT/source_id_more.txt
COMMON="--warmup 5 --no-cpu --no-d2h --no-hot --no-wire --others none --batch-sweep= --extra ''"
for i in $(seq $START $((START + 8))); do
  name=headline_s4_$i
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$name -o t -- python $GRAFT_REPO_ROOT/bench.py $COMMON --steps 2000 --regions 2 --streams 4 > $OUT/${name}_under_rocprof.json 2> $OUT/${name}_rocprof.log )
  csvf=$(find $OUT/t_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$csvf" ] && python scripts/trace_stats.py $csvf $OUT/${name}_kernel_stats.csv $OUT/${name}_trace_overlap.json > /dev/null
  rm -rf $OUT/t_$name
  python3 - <<PY
import csv
for row in csv.DictReader(open("$OUT/${name}_kernel_stats.csv")):
    if "achip::render" in row["Name"]: print("$name busy us", float(row["RunBusyNsPerCall"])/1e3, "in flight", row["RunAvgInFlight"]); break
PY
done
