#!/bin/bash
# round 2, call zd: EXPERIMENT -- stagger the workgroups of a launch (groups of fidx & 3) so that their gather bursts do not coincide
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
: > gpurun_out/zd_stagger.txt
for S in 0 1 2 4 8 0; do
  echo "## ASCIICHAT_HIP_STAGGER=$S  (delay per group = $S x 512 cycles)" >> gpurun_out/zd_stagger.txt
  ASCIICHAT_HIP_STAGGER=$S OVERLAP_VARIANTS=16,17 OVERLAP_STREAMS=1,4 timeout 200 python scripts/gpu_overlap.py 1080p_80x24_truecolor 2>&1 | grep -v amdgpu.ids >> gpurun_out/zd_stagger.txt
done
cat gpurun_out/zd_stagger.txt
