#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-dt}; mkdir -p $OUT
gcc -O2 -I include scripts/dropin_threads.c -o scripts/dropin_threads -L ascii-chat_amd -lasciichat_hip -Wl,-rpath,$PWD/ascii-chat_amd -lpthread || exit 1
ulimit -c 0
unset ASCIICHAT_HIP_COALESCE
for pooled in 0 1; do DT_MIN_T=1 DT_POOLED=$pooled timeout 150 ./scripts/dropin_threads 128 2>&1 | grep -v amdgpu.ids | tee -a $OUT/threads.txt | sed 's/1920x1080 -> 80x24 colour 3 mode 0, //'; echo "exit=${PIPESTATUS[0]}"; done
