#!/bin/bash
# round 2, visit A: GPU parity suite with the stream kernel in the policy + geometry/stream sweep of the stream variants
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest.log
export OVERLAP_VARIANTS=16,17,18,19,1 OVERLAP_STREAMS=1,2,3,4
timeout 600 python scripts/gpu_overlap.py 1080p_80x24_truecolor 2>&1 | grep -v amdgpu.ids | tee $OUT/overlap_1080p.txt
export OVERLAP_VARIANTS=16,17,19,4,1 OVERLAP_STREAMS=1,3
timeout 600 python scripts/gpu_overlap.py 4k_200x60_truecolor 2>&1 | grep -v amdgpu.ids | tee $OUT/overlap_4k.txt
timeout 600 python scripts/gpu_overlap.py 1080p_80x24_ansi256 2>&1 | grep -v amdgpu.ids | tee $OUT/overlap_256.txt
