#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
: > $OUT/exp.txt
for lib in "" gpurun_tmp/lib_exp8.so gpurun_tmp/lib_exp11.so "" gpurun_tmp/lib_exp11.so; do
  echo "## lib=${lib:-product}" | tee -a $OUT/exp.txt
  export OVERLAP_VARIANTS=16,17 OVERLAP_STREAMS=1,4
  ASCIICHAT_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 300 python scripts/gpu_overlap.py 1080p_80x24_truecolor 2>&1 | grep -v amdgpu.ids | tee -a $OUT/exp.txt
  export OVERLAP_VARIANTS=17 OVERLAP_STREAMS=1,4
  ASCIICHAT_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 300 python scripts/gpu_overlap.py 4k_200x60_truecolor 2>&1 | grep -v amdgpu.ids | tee -a $OUT/exp.txt
done
