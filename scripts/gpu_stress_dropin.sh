#!/bin/bash
# extended stress of the drop-in layer on the last build: the self-verifying scaling harness at 200 and 256 threads (roaming and
# confined), mixed sizes / modes per thread (half blocks, 4K sources), the Python thread fuzz at 48 threads in both waiting modes
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-stress}; mkdir -p $OUT; export TMPDIR=/tmp
gcc -O2 -I include scripts/dropin_threads.c -o scripts/dropin_threads -L ascii-chat_amd -lasciichat_hip -Wl,-rpath,$PWD/ascii-chat_amd -lpthread || exit 1
ulimit -c 0
{ for conf in 0 1; do for args in "200" "256" "96 3840 2160 200 60 3 0" "96 1920 1080 100 37 3 2" "64 640 480 80 24 0 0"; do
    set -- $args; T=$1
    echo "## ASCIICHAT_HIP_CONFINE=$conf dropin_threads $args"
    ASCIICHAT_HIP_CONFINE=$conf DT_MIN_T=$T DT_POOLED=$((T % 2)) timeout 240 ./scripts/dropin_threads $args 2>&1 | grep -v amdgpu.ids; echo "exit=${PIPESTATUS[0]}"
  done; done
  for b in 1 1000; do echo "## thread fuzz 48 threads, combiner always, ASCIICHAT_HIP_CPU_BUDGET=$b"; ASCIICHAT_HIP_COALESCE=1 ASCIICHAT_HIP_CPU_BUDGET=$b timeout 400 python scripts/gpu_thread_fuzz.py 48 300 2>&1 | grep -v amdgpu.ids | tail -1; done
} | tee $OUT/stress.txt
