#!/bin/bash
# round 2, call zl: rocprofv3 kernel stats of the whole default bench line (every workload) on the final sources
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/zl; mkdir -p $OUT
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o all -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --regions 5 --no-cpu --no-d2h > $OUT/bench_all_under_rocprof.json 2> $OUT/rocprof.log
echo "rc=$? lines $(wc -l < $OUT/bench_all_under_rocprof.json)"
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/bench_all_workloads_kernel_stats.csv
rm -rf $OUT/trace
head -25 $OUT/bench_all_workloads_kernel_stats.csv | cut -c1-170
