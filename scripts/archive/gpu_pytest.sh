#!/bin/bash
# the GPU suite alone (optionally -k "$K")
TAG=${1:-pytest}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider ${K:+-k "$K"} 2>&1 | grep -v "^  File\|^E    \+where\|amdgpu.ids" | tail -60 | tee $OUT/pytest_gpu.txt
