#!/bin/bash
# run selected GPU tests: bash scripts/gpu_one.sh "<pytest -k expression>"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/one
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider -k "$1" 2>&1 | tail -25 | tee gpurun_out/one/pytest.log
