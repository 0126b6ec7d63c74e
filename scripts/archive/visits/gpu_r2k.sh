#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for k in 20 100; do
( time timeout 900 python bench.py --steps $k --warmup 5 --others '' --no-cpu --no-d2h ) > $OUT/bench_$k.json 2> $OUT/bench_$k.err
tail -3 $OUT/bench_$k.err
done
