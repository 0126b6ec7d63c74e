#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest.log 2>&1
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $OUT/pytest.log | tail -40
