#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for s in 4 1; do timeout 300 python scripts/gpu_region_overhead.py $s 2>&1 | grep -v amdgpu.ids | tee -a $OUT/region.txt; done
