#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/pytest.log
