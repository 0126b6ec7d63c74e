#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for v in 16 17; do
timeout 300 python scripts/gpu_stream_timeline.py 1080p_80x24_truecolor $v 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline.txt
done
timeout 300 python scripts/gpu_stream_timeline.py 4k_200x60_truecolor 17 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline.txt
