#!/bin/bash
# round 4, session 2 profile visit: rocprofv3 kernel traces (no in-process --stats: scripts/trace_stats.py reduces the CSV) of the last build --
# the headline, configs[2] and configs[4] after the whole-line drains, 256 frames of 4K -> 200x60 one launch at a time (the audited geometry rule),
# and the send side of small launches (scripts/gpu_wire_one.py: which kernels a lone large frame's render + wire stage are).
# Outputs: gpurun_out/<tag>/ -> copied to profiles/r04s2_* by hand.
TAG=${1:-r4s2prof}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
COMMON="--warmup 5 --no-cpu --no-d2h --no-hot --no-wire --others none --extra ''"
reduce() { # name
  local csvf=$(find $OUT/t_$1 -name "*kernel_trace.csv" | head -1)
  if [ -n "$csvf" ]; then python scripts/trace_stats.py $csvf $OUT/${1}_kernel_stats.csv $OUT/${1}_trace_overlap.json > /dev/null; else echo "$1: no trace"; fi
  rm -rf $OUT/t_$1
}
trace() { # name, bench args...
  local name=$1; shift
  ( cd /tmp && timeout 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$name -o t -- python $GRAFT_REPO_ROOT/bench.py $COMMON "$@" > $OUT/${name}_under_rocprof.json 2> $OUT/${name}_rocprof.log )
  reduce $name
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${name}_under_rocprof.json")); r=d["roofline"]
    print("$name: line kernel_ms", r["kernel_ms"], "frac", r["frac"], "variant", d["config"].get("kernel_variant"), "value", d["value"])
except Exception as e: print("$name: no line", e)
PY
}
trace headline_s4 --steps 400 --regions 3 --streams 4
trace k3_4k_200x60 --workload 4k_200x60_truecolor --steps 100 --regions 3 --input-sets 4 --streams 4
trace k3_4k_200x60_s1 --workload 4k_200x60_truecolor --steps 100 --regions 3 --input-sets 4 --streams 1
trace k5_4k_400x120_hb --workload 4k_400x120_halfblock --steps 40 --regions 3 --input-sets 4 --streams 4
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_wire_small -o t -- python $GRAFT_REPO_ROOT/scripts/gpu_wire_one.py 320 90 1 200 60 16 160 45 64 > $OUT/wire_small_stdout.txt 2> $OUT/wire_small_rocprof.log )
reduce wire_small
grep -v amdgpu $OUT/wire_small_stdout.txt
head -12 $OUT/wire_small_kernel_stats.csv | cut -c1-200
