#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/gpu_stream_timeline.py 1080p_80x24_truecolor 16 2>&1 | grep -v amdgpu.ids | tee $OUT/timeline.txt
timeout 300 python scripts/gpu_stream_timeline.py 1080p_80x24_truecolor 17 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline.txt
