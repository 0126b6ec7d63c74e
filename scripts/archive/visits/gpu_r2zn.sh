#!/bin/bash
# round 2, call zn: A/B -- 24-bit multiplies in the fast sampler's address (before / after, alternated twice), GPU parity for the new library
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tee gpurun_out/zn_u24.txt
OVERLAP_VARIANTS=17 OVERLAP_STREAMS=4 timeout 200 python scripts/gpu_overlap.py 1080p_80x24_truecolor > /dev/null 2>&1
for rep in 1 2; do
for lib in gpurun_tmp/libachip_before_u24.so ascii-chat_amd/libasciichat_hip.so; do
  echo "## $lib" >> gpurun_out/zn_u24.txt
  ASCIICHAT_HIP_LIB=$GRAFT_REPO_ROOT/$lib OVERLAP_VARIANTS=16,17 OVERLAP_STREAMS=1,4 timeout 200 python scripts/gpu_overlap.py 1080p_80x24_truecolor 2>&1 | grep -v amdgpu.ids >> gpurun_out/zn_u24.txt
  ASCIICHAT_HIP_LIB=$GRAFT_REPO_ROOT/$lib OVERLAP_VARIANTS=17 OVERLAP_STREAMS=4 OVERLAP_NSETS=4 timeout 200 python scripts/gpu_overlap.py 4k_200x60_truecolor 2>&1 | grep -v amdgpu.ids >> gpurun_out/zn_u24.txt
done
done
cat gpurun_out/zn_u24.txt
