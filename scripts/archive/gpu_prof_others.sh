#!/bin/bash
# kernel trace over the bench incl. other_workloads, and FETCH/WRITE PMC passes for the two 4K workloads
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_all -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-d2h --no-hot > $OUT/rocprof_all.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/prof_all -name "*kernel_stats.csv"); do cut -c1-160 $f | head -8; done
for wl in 4k_200x60_truecolor 4k_400x120_halfblock; do
  bash scripts/pmc_run.sh $TAG/pmc_$wl $wl 2>&1 | grep -E "FETCH|WRITE|SQ_WAVES|SQ_INSTS_VALU|SQ_INSTS_SALU|BANK_CONFLICT|IDX_ACTIVE|WAIT_ANY" | sed "s/^/$wl /" | tee -a $OUT/pmc_others.txt
done
