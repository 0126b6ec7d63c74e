#!/bin/bash
# round 2, call zg: EXPERIMENT -- stream geometry 21 (8 waves x 4 cells per lane: per-block work amortised over 256 cells)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "torture or full_size" 2>&1 | grep -E "passed|failed" | tee gpurun_out/zg_cpl4.txt
OVERLAP_VARIANTS=17 OVERLAP_STREAMS=4 timeout 200 python scripts/gpu_overlap.py 1080p_80x24_truecolor > /dev/null 2>&1
for W in 1080p_80x24_truecolor 1080p_80x24_ansi256 4k_200x60_truecolor; do
  OVERLAP_VARIANTS=17,21,16,21,17 OVERLAP_STREAMS=1,4 OVERLAP_NSETS=12 timeout 300 python scripts/gpu_overlap.py $W 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/zg_cpl4.txt
done
