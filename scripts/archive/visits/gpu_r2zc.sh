#!/bin/bash
# round 2, call zc: 256-thread stream geometry (18) and CPL = 1 (19) against 17 with 3-8 launches in flight
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
OVERLAP_VARIANTS=17 OVERLAP_STREAMS=4 timeout 200 python scripts/gpu_overlap.py 1080p_80x24_truecolor > /dev/null 2>&1
OVERLAP_VARIANTS=17,18,19,17,18 OVERLAP_STREAMS=3,4,6 OVERLAP_NSETS=12 timeout 300 python scripts/gpu_overlap.py 1080p_80x24_truecolor 2>&1 | grep -v amdgpu.ids | tee gpurun_out/zc_small_geometries.txt
