#!/bin/bash
# round 2, call zm: does hipSetDeviceFlags(spin / yield / blocking) change what a timed region's closing synchronize costs?
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
: > gpurun_out/zm_sync_flags.txt
for m in "" 1 2 4 ""; do BENCH_SPIN=$m timeout 200 python scripts/gpu_sync_flags.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/zm_sync_flags.txt; done
cat gpurun_out/zm_sync_flags.txt
