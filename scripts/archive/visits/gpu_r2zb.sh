#!/bin/bash
# round 2, call zb: the stream sweep again with one pool of streams for the whole process (see profiles/r02_geometry_ab.txt)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
OVERLAP_VARIANTS=17 OVERLAP_STREAMS=4 timeout 200 python scripts/gpu_overlap.py 1080p_80x24_truecolor > /dev/null 2>&1   # first touch of the box
OVERLAP_VARIANTS=16,17,1,4 OVERLAP_STREAMS=1,2,3,4,6 timeout 300 python scripts/gpu_overlap.py 1080p_80x24_truecolor 2>&1 | grep -v amdgpu.ids | tee gpurun_out/zb_stream_sweep.txt
OVERLAP_VARIANTS=17,16,1 OVERLAP_STREAMS=1,4 timeout 300 python scripts/gpu_overlap.py 4k_200x60_truecolor 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/zb_stream_sweep.txt
OVERLAP_VARIANTS=17,16,1 OVERLAP_STREAMS=1,4 timeout 300 python scripts/gpu_overlap.py 1080p_80x24_ansi256 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/zb_stream_sweep.txt
