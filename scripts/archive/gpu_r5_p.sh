#!/bin/bash
# round 5, visit P: truecolor half-block tokens built as words (ACHIP_ROWS_WORD_EMIT) against the byte stores (lib_w0.so):
# the rows kernel's parity tests on the GPU first, then the interleaved A/B on the half-block workloads
TAG=${1:-r5p}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.log 2>&1; grep -E "passed|failed" $O/pytest_parity.log | tail -2
HOT=1 bash scripts/gpu_abn.sh ${TAG}_ab "HEAD lib_w0.so" "sampled_400x240_halfblock 4k_400x120_halfblock 1080p_80x24_halfblock" 2 2>&1 | grep -A14 "^# median"
