#!/bin/bash
# round 4, visit j: kernel traces of the round's late kernels -- the exact-length launches (PACK), the shared-out small launches
# (PARTS) and the row bands small launches of the coloured half-block modes went back to -- reduced by scripts/trace_stats.py
TAG=${1:-r4j}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
trace() { # name, command...
  local name=$1; shift
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$name -o t -- "$@" > $OUT/${name}_stdout.txt 2> $OUT/${name}_rocprof.log )
  local csvf=$(find $OUT/t_$name -name "*kernel_trace.csv" | head -1)
  if [ -n "$csvf" ]; then python scripts/trace_stats.py $csvf $OUT/${name}_kernel_stats.csv $OUT/${name}_trace_overlap.json; else echo "$name: no trace"; fi
  rm -rf $OUT/t_$name
}
trace exact_length python $GRAFT_REPO_ROOT/scripts/gpu_exact_length_timing.py
trace small_launches python $GRAFT_REPO_ROOT/scripts/gpu_small_batch_variants.py
trace small_run_modes python $GRAFT_REPO_ROOT/scripts/gpu_small_run_modes.py 1 8
head -30 $OUT/exact_length_kernel_stats.csv | cut -c1-260
