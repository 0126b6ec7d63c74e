#!/bin/bash
# round 2, call zf: EXPERIMENT -- kernarg preload (-mllvm -amdgpu-kernarg-preload-count=N) for the stream kernel's prologue
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
: > gpurun_out/zf_kernarg_preload.txt
for rep in 1 2; do
for lib in ascii-chat_amd/libasciichat_hip.so gpurun_tmp/libachip_kernarg_preload6.so gpurun_tmp/libachip_kernarg_preload12.so; do
  echo "## $lib" >> gpurun_out/zf_kernarg_preload.txt
  ASCIICHAT_HIP_LIB=$GRAFT_REPO_ROOT/$lib OVERLAP_VARIANTS=16,17 OVERLAP_STREAMS=1,4 timeout 200 python scripts/gpu_overlap.py 1080p_80x24_truecolor 2>&1 | grep -v amdgpu.ids >> gpurun_out/zf_kernarg_preload.txt
done
done
ASCIICHAT_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_tmp/libachip_kernarg_preload12.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed" >> gpurun_out/zf_kernarg_preload.txt
cat gpurun_out/zf_kernarg_preload.txt
