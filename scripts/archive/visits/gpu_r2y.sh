#!/bin/bash
# round 2, call y: the round-1 fuzzers on the final build -- drop-in entry points (direct path and every call through the
# combiner), stand-alone batch entry points; plus the C harnesses' threaded pool test and the RCCL world-1 grid test
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 600 python scripts/gpu_dropin_fuzz.py 91 20000 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/y_dropin_fuzz.txt
ASCIICHAT_HIP_COALESCE=1 timeout 600 python scripts/gpu_dropin_fuzz.py 92 6000 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/y_dropin_fuzz_coalesced.txt
timeout 600 python scripts/gpu_api_fuzz.py 93 400 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/y_api_fuzz.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "rccl or buffer_pool or harness or frame_table" 2>&1 | grep -E "passed|failed" | tee gpurun_out/y_pytest.txt
