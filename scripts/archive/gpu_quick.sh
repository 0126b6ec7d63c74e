#!/bin/bash
# quick GPU visit: full GPU parity suite + geometry tune at BASELINE batch + the auto policy at small batches + wire stage
OUT=$GRAFT_REPO_ROOT/gpurun_out/quick
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest.log
timeout 600 python scripts/gpu_tune.py --batch 256 --variants=-1 --splits=0 --reps 200 2>&1 | grep -v amdgpu.ids | tee $OUT/tune_b256.txt
timeout 300 python scripts/gpu_tune.py --wire-stage 2>&1 | grep -v amdgpu.ids | tee $OUT/wire.txt
: > $OUT/tune_small.txt
for b in 1 16 64 128; do
timeout 600 python scripts/gpu_tune.py --batch $b --variants=-1 --splits=-1,0 2>&1 | grep -v amdgpu.ids | sed "s/^/b$b /" | cut -c1-160 | tee -a $OUT/tune_small.txt
done
