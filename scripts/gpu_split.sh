#!/bin/bash
# gpurun visit for the multi-workgroup-frame work: split parity tests, then a geometry x band-height sweep
OUT=$GRAFT_REPO_ROOT/gpurun_out/split
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "multi_workgroup or split_exclusions or full_size" 2>&1 | tail -15 | tee $OUT/pytest.log
: > $OUT/sweep.txt
for b in 1 8 32 64 128 192; do
timeout 300 python scripts/gpu_tune.py --reps 30 --batch $b --variants=2,1,4 --splits=-1,1,2,3,4,6,8,12 --workloads 1080p_80x24_truecolor 2>&1 | grep -v amdgpu.ids | sed "s/^/b$b /" | cut -c1-150 >> $OUT/sweep.txt
timeout 300 python scripts/gpu_tune.py --reps 30 --batch $b --variants=2,1,4 --splits=-1,1,2,3,5,10 --workloads 4k_200x60_truecolor 2>&1 | grep -v amdgpu.ids | sed "s/^/b$b /" | cut -c1-150 >> $OUT/sweep.txt
timeout 300 python scripts/gpu_tune.py --reps 30 --batch $b --variants=2,1,4 --splits=-1,1,2,3,5 --workloads 4k_400x120_halfblock 2>&1 | grep -v amdgpu.ids | sed "s/^/b$b /" | cut -c1-150 >> $OUT/sweep.txt
done
