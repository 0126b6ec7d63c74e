#!/bin/bash
# One gpurun visit: GPU parity tests, smoke, bench, rocprofv3 kernel trace. Logs land in gpurun_out/<tag>/.
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -8 > $OUT/rocminfo.txt
nproc > $OUT/nproc.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/nproc.txt
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -40 | tee $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee $OUT/smoke.log
echo "== bench"; timeout 900 python bench.py --steps 200 --warmup 20 2>&1 | tail -3 | tee $OUT/bench.log
echo "== rocprof"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu --others '' > $GRAFT_REPO_ROOT/$OUT/rocprof_run.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*stats*" | head; for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -12 $f; done
