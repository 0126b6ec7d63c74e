#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/pytest.log
export OVERLAP_VARIANTS=16,17 OVERLAP_STREAMS=1,2,4
timeout 300 python scripts/gpu_overlap.py 1080p_80x24_truecolor 2>&1 | grep -v amdgpu.ids | tee $OUT/overlap.txt
export OVERLAP_VARIANTS=17 OVERLAP_STREAMS=1,4
timeout 300 python scripts/gpu_overlap.py 4k_200x60_truecolor 2>&1 | grep -v amdgpu.ids | tee -a $OUT/overlap.txt
timeout 300 python scripts/gpu_overlap.py 1080p_80x24_ansi256 2>&1 | grep -v amdgpu.ids | tee -a $OUT/overlap.txt
timeout 300 python scripts/gpu_stream_timeline.py 1080p_80x24_truecolor 16 2>&1 | grep -v amdgpu.ids | tee $OUT/timeline.txt
