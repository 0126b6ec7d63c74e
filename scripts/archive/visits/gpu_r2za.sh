#!/bin/bash
# round 2, call za: geometry 16 vs 17 with four launches in flight, alternated twice per workload on one box (policy check)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
: > gpurun_out/za_geometry_ab.txt
for W in 1080p_80x24_truecolor 1080p_80x24_ansi256 4k_200x60_truecolor; do
  OVERLAP_VARIANTS=17,16,17,16 OVERLAP_STREAMS=4 OVERLAP_NSETS=12 timeout 300 python scripts/gpu_overlap.py $W 2>&1 | grep -v amdgpu.ids >> gpurun_out/za_geometry_ab.txt
done
cat gpurun_out/za_geometry_ab.txt
