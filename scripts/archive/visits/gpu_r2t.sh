#!/bin/bash
# round 2, call t: device-side timeline of a 20-step burst (no profiler attached)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
for S in 4 2 1; do
  timeout 300 python scripts/gpu_burst_timeline.py $S 20 > gpurun_out/t_burst_timeline_s$S.txt 2>&1
  cat gpurun_out/t_burst_timeline_s$S.txt
done
