#!/bin/bash
# round 2, visit C: what bounds the stream kernel at 3-4 launches in flight?  ablations (no HBM writes / no gather / no
# token stores) and Infinity-Cache-resident inputs (4 input sets instead of 12)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
: > $OUT/ablate.txt
export OVERLAP_VARIANTS=17,16 OVERLAP_STREAMS=1,2,4
for lib in "" gpurun_tmp/lib_abl1.so gpurun_tmp/lib_abl2.so gpurun_tmp/lib_abl3.so; do
  for nsets in 12 4; do
    echo "## lib=${lib:-product} input_sets=$nsets" | tee -a $OUT/ablate.txt
    ASCIICHAT_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} OVERLAP_NSETS=$nsets timeout 300 python scripts/gpu_overlap.py 1080p_80x24_truecolor 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ablate.txt
  done
done
