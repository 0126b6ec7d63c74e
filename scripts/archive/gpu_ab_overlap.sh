#!/bin/bash
# A/B of experimental builds (gpurun_tmp/*.so) against the in-tree library under the pipelined bench pattern
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for w in "$@"; do
for lib in ascii-chat_amd/libasciichat_hip.so $(ls gpurun_tmp/*.so); do
  ASCIICHAT_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 200 python scripts/gpu_overlap.py $w 2>&1 | grep -v amdgpu.ids | grep "variant 1 streams [13]" | sed "s|^|$(basename $lib) |"
done
done | tee gpurun_out/ab/overlap_ab.txt
