#!/bin/bash
# A/B of ablation builds (gpurun_tmp/*.so) on one box + wire-stage test and throughput
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
#timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "wire_stage" 2>&1 | tail -15 | tee $OUT/pytest.log
#timeout 300 python scripts/gpu_tune.py --wire-stage 2>&1 | grep -v amdgpu.ids | tee $OUT/wire.txt
for rep in 1 2; do
for lib in ascii-chat_amd/libasciichat_hip.so gpurun_tmp/lib_NOPARTS.so gpurun_tmp/lib_NOOPS.so; do
  ASCIICHAT_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python scripts/gpu_tune.py --batch 256 --variants=-1 --splits=-1 --reps 200 --workloads 1080p_80x24_truecolor,1080p_80x24_ansi256 2>&1 | grep -v amdgpu.ids | sed "s|^|$lib |" | cut -c1-330 | tee -a $OUT/ab.txt
done
done
