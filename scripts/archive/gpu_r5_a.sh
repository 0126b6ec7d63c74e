#!/bin/bash
# round 5, visit A: GPU suite on the build with the ABI-only export map, the driver's bench command (new: sampled-image legs
# of configs[2] / configs[4], the batch-size axis), and the round's BASELINE traces of the rows kernel before its VALU diet:
# configs[4] from 4K sources and from sampled images, configs[2] from sampled images.
# Outputs: gpurun_out/<tag>/
TAG=${1:-r5a}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; grep -E "passed|failed|FAILED|rc=" $OUT/pytest.log | tail -8
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_stdout.txt 2> $OUT/bench_driver_stderr.txt; echo "bench rc=$?"
cp bench_extra.json $OUT/bench_extra_driver_flags.json 2>/dev/null
wc -c $OUT/bench_driver_stdout.txt; tail -c 1500 $OUT/bench_driver_stdout.txt; grep "\[bench\]" $OUT/bench_driver_stderr.txt | tail -40
COMMON="--warmup 5 --no-cpu --no-d2h --no-hot --no-wire --others none --batch-sweep= --extra ''"
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
  head -3 $OUT/${name}_kernel_stats.csv | cut -c1-260
}
trace k5_4k_400x120_hb --workload 4k_400x120_halfblock --steps 40 --regions 3 --input-sets 4 --streams 4
trace k5_sampled_400x240_hb --workload sampled_400x240_halfblock --steps 40 --regions 3 --streams 4
trace k3_sampled_200x60 --workload sampled_200x60_truecolor --steps 100 --regions 3 --streams 4
