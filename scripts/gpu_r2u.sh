#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
: > gpurun_out/u_wait_modes.txt
for m in 0 1 2 0 2; do
  ASCIICHAT_HIP_WAIT_MODE=$m timeout 200 python scripts/gpu_wait_modes.py >> gpurun_out/u_wait_modes.txt 2>&1
done
cat gpurun_out/u_wait_modes.txt
