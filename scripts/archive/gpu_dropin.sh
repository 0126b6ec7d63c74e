#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/dropin
timeout 900 python -m pytest tests/test_gpu_dropin.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/dropin/pytest.log
timeout 600 python scripts/gpu_tune.py --dropin 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dropin/latency.txt
