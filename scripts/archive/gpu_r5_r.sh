#!/bin/bash
# round 5, visit R: truecolor / 256-colour SGRs of the stream kernel built as words (HEAD) against byte stores (lib_sw0.so)
TAG=${1:-r5r}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.log 2>&1; grep -E "passed|failed" $O/pytest_parity.log | tail -2
HOT=1 bash scripts/gpu_abn.sh ${TAG}_ab "HEAD lib_sw0.so" "1080p_80x24_truecolor 1080p_80x24_ansi256 4k_200x60_truecolor sampled_80x24_truecolor sampled_200x60_truecolor" 2 2>&1 | grep -A22 "^# median"
