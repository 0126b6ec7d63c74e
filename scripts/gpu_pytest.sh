#!/bin/bash
# the GPU suite alone
TAG=${1:-pytest}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=8 2>&1 | tail -40 | tee $OUT/pytest_gpu.txt
