#!/bin/bash
# round 2, visit B: load-policy microbenchmark + stream sweep with the launches issued from C
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
: skip ubench
export OVERLAP_VARIANTS=16,17,1 OVERLAP_STREAMS=1,2,3,4,6
timeout 600 python scripts/gpu_overlap.py 1080p_80x24_truecolor 2>&1 | grep -v amdgpu.ids | tee $OUT/overlap_1080p.txt
export OVERLAP_VARIANTS=17,16,1 OVERLAP_STREAMS=1,3,4
timeout 600 python scripts/gpu_overlap.py 4k_200x60_truecolor 2>&1 | grep -v amdgpu.ids | tee $OUT/overlap_4k.txt
