#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/wire
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "wire_stage" 2>&1 | tail -5 | tee $OUT/pytest.log
timeout 300 python scripts/gpu_tune.py --wire-stage 2>&1 | grep -v amdgpu.ids | tee $OUT/wire.txt
