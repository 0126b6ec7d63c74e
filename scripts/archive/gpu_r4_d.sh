#!/bin/bash
set -u
O=gpurun_out/r4d; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -12
python scripts/gpu_exact_length_timing.py 1080p_80x24_truecolor 2>&1 | grep -v amdgpu.ids | tee $O/exact_length_timing.txt
python scripts/gpu_exact_length_timing.py sampled_80x24_truecolor 2>&1 | grep -v amdgpu.ids | tee -a $O/exact_length_timing.txt
