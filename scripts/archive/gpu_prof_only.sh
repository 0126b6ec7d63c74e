#!/bin/bash
# rocprofv3 kernel trace + PMC passes only (the tail of gpu_check.sh)
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-d2h --no-hot --others '' > $OUT/rocprof_run.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -3 $f | cut -c1-200; done
tail -1 $OUT/rocprof_run.log | cut -c1-400
bash scripts/pmc_run.sh $TAG/pmc 1080p_80x24_truecolor 2>&1 | tail -20 | tee $OUT/pmc_summary.txt
