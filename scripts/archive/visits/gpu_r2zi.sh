#!/bin/bash
# round 2, call zi: the half-block workload (K5 shape, phase kernel) over geometries and launches in flight
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
OVERLAP_VARIANTS=4 OVERLAP_STREAMS=2 OVERLAP_NSETS=4 timeout 200 python scripts/gpu_overlap.py 4k_400x120_halfblock > /dev/null 2>&1
OVERLAP_VARIANTS=4,0,1,4 OVERLAP_STREAMS=1,2,3,4 OVERLAP_NSETS=12 timeout 600 python scripts/gpu_overlap.py 4k_400x120_halfblock 2>&1 | grep -v amdgpu.ids | tee gpurun_out/zi_k5_sweep.txt
