#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export OVERLAP_VARIANTS=16,17 OVERLAP_STREAMS=1,3,4
for dk in 0 1; do
echo "## HIP_FORCE_DEV_KERNARG=$dk" | tee -a $OUT/kernarg.txt
HIP_FORCE_DEV_KERNARG=$dk timeout 300 python scripts/gpu_overlap.py 1080p_80x24_truecolor 2>&1 | grep -v amdgpu.ids | tee -a $OUT/kernarg.txt
HIP_FORCE_DEV_KERNARG=$dk timeout 300 python scripts/gpu_stream_timeline.py 1080p_80x24_truecolor 16 2>&1 | grep -v amdgpu.ids | tee -a $OUT/kernarg.txt
done
