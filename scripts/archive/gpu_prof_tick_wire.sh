#!/bin/bash
# rocprofv3 --kernel-trace --stats of (a) the ingest-inclusive tick (bench.tick_e2e: scatter, render + wire stage, pack) and
# (b) the send side behind the half-block workload (bench.py wire_stage leg: rows kernel, stand-alone wire stage, pack, and
# the one-pass checksum + pack)
TAG=${1:-proftick}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
cat > /tmp/tick_only.py <<PY
import json, sys, torch
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
from __graft_entry__ import load_package
pkg = load_package(); torch.cuda.set_device(0)
r = bench.tick_e2e(torch, pkg, ticks=(1, 6))
print(json.dumps({k: v for k, v in r.items() if k != "note"}))
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tick -o tick -- python /tmp/tick_only.py > $OUT/tick_under_rocprof.json 2> $OUT/tick.log
cp $(find $OUT/tick -name "*kernel_stats.csv" | head -1) $OUT/tick_kernel_stats.csv; rm -rf $OUT/tick
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wire -o wire -- python $GRAFT_REPO_ROOT/bench.py --workload 4k_400x120_halfblock --steps 10 --warmup 3 --regions 3 --others none --no-cpu --no-d2h --no-hot --input-sets 4 --streams 4 > $OUT/wire_under_rocprof.json 2> $OUT/wire.log
cp $(find $OUT/wire -name "*kernel_stats.csv" | head -1) $OUT/wire_k5_kernel_stats.csv; rm -rf $OUT/wire
head -12 $OUT/tick_kernel_stats.csv | cut -c1-150; echo; head -12 $OUT/wire_k5_kernel_stats.csv | cut -c1-150
