#!/bin/bash
# round 4, visit L: the traffic floor of the HBM-bound workloads (scripts/ubench/rows_floor.hip) -> profiles/r04_rows_floor.txt
set -u
O=gpurun_out/r4l; mkdir -p $O
for n in 4; do ./scripts/ubench/rows_floor $n 2>&1 | grep -v amdgpu.ids; done | tee $O/rows_floor.txt
