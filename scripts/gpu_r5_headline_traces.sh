#!/bin/bash
# the headline's kernel trace, again: rocprofv3's tracer serialises short launches to a box- and run-dependent degree (1.6-2.9 of
# the four requested stay in flight); scripts/r5_commit_profiles.py takes the run that kept the most.  usage: <tag> [sets]
TAG=${1:-r5ht}; SETS=${2:-2}; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
COMMON="--warmup 5 --no-cpu --no-d2h --no-hot --no-wire --others none --batch-sweep= --extra ''"
for s in $(seq $SETS); do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_$s; mkdir -p $OUT
  for name in headline_s4 headline_s4b headline_s4c; do
    ( cd /tmp && timeout 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$name -o t -- python $GRAFT_REPO_ROOT/bench.py $COMMON --steps 2000 --regions 2 --streams 4 > $OUT/${name}_under_rocprof.json 2> $OUT/${name}_rocprof.log )
    csvf=$(find $OUT/t_$name -name "*kernel_trace.csv" | head -1)
    [ -n "$csvf" ] && python scripts/trace_stats.py $csvf $OUT/${name}_kernel_stats.csv $OUT/${name}_trace_overlap.json > /dev/null
    rm -rf $OUT/t_$name
    grep render_stream $OUT/${name}_kernel_stats.csv | cut -d, -f1,2,6- | cut -c1-200
  done
done
