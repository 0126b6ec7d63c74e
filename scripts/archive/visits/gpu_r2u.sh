#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
: > gpurun_out/u_anyorder.txt
for a in 0 1; do
  echo "## ASCIICHAT_HIP_ANYORDER=$a" >> gpurun_out/u_anyorder.txt
  ASCIICHAT_HIP_ANYORDER=$a timeout 200 python scripts/gpu_wait_modes.py >> gpurun_out/u_anyorder.txt 2>&1
  ASCIICHAT_HIP_ANYORDER=$a timeout 200 python scripts/gpu_burst_timeline.py 1 20 >> gpurun_out/u_anyorder.txt 2>&1
done
ASCIICHAT_HIP_ANYORDER=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size or launches_in_flight" 2>&1 | tail -3 >> gpurun_out/u_anyorder.txt
cat gpurun_out/u_anyorder.txt
