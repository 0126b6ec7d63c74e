#!/bin/bash
# A/B of experimental builds (gpurun_tmp/*.so) against the in-tree library on one box
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab
mkdir -p $OUT; : > $OUT/ab.txt
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in ascii-chat_amd/libasciichat_hip.so $(ls gpurun_tmp/*.so); do
  ASCIICHAT_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python scripts/gpu_tune.py --batch 256 --variants=-1 --splits=-1 --reps 100 2>&1 | grep -v amdgpu.ids | sed "s|^|$(basename $lib) |" | cut -c1-330 | tee -a $OUT/ab.txt
done
done
