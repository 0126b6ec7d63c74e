#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $OUT/bench_20.json 2> $OUT/bench_20.err
tail -5 $OUT/bench_20.err
( time timeout 900 python bench.py --others '' --no-cpu --no-d2h ) > $OUT/bench_default_main.json 2> $OUT/bench_default.err
tail -4 $OUT/bench_default.err
