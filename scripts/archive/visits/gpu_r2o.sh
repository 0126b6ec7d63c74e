#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2o
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_dropin.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -12 | grep -v "^$" | tail -8; done
for i in 1 2; do
  timeout 120 scripts/dropin_threads 64 2>&1 | grep -v "render threads" | head -5
done
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -4
