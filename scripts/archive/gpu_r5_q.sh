#!/bin/bash
# round 5, visit Q: word-built truecolor half-block tokens, every everyday token kind (HEAD) against byte stores (lib_w0.so)
# on noise (every cell both SGRs), smooth (a third of the cells a lone half block) and bars (long repeated runs)
TAG=${1:-r5q}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.log 2>&1; grep -E "passed|failed" $O/pytest_parity.log | tail -2
for inp in noise smooth bars; do
  echo "## input $inp"
  INPUT=$inp bash scripts/gpu_abn.sh ${TAG}_$inp "HEAD lib_w0.so" "sampled_400x240_halfblock 1080p_80x24_halfblock" 2 2>&1 | grep -A5 "^# median"
done
