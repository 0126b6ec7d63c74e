#!/bin/bash
# round 5, visit O: rows geometry 26 (sixteen waves per frame) -- GPU suite on both builds, the one-launch-at-a-time legs of the
# half-block workloads (automatic choice: 26 from dense sources), the soak on the all-geometries build
TAG=${1:-r5o}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_default.log 2>&1; grep -E "passed|failed" $O/pytest_default.log | tail -2
ASCIICHAT_HIP_LIB=$PWD/ascii-chat_amd/lib_all.so timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; grep -E "passed|failed" $O/pytest_all.log | tail -2
HOT=1 bash scripts/gpu_abn.sh ${TAG}_ab "HEAD" "sampled_400x240_halfblock 4k_400x120_halfblock" 2 2>&1 | grep -A6 "^# median"
ASCIICHAT_HIP_LIB=$PWD/ascii-chat_amd/lib_all.so timeout 600 python scripts/gpu_soak.py --seed 9 2>&1 | grep -v amdgpu.ids | tail -2
