#!/bin/bash
# after a change to combine.c: the full GPU suite, the thread fuzz in both waiting modes (every caller polls / every caller
# sleeps), the self-verifying scaling harness roaming, confined by taskset and confined by ASCIICHAT_HIP_CONFINE=1
cd $GRAFT_REPO_ROOT; TAG=${1:-dtverify}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
bash scripts/gpu_pytest.sh $TAG | tail -4
for b in 1 1000; do echo "## thread fuzz, ASCIICHAT_HIP_CPU_BUDGET=$b"; ASCIICHAT_HIP_COALESCE=1 ASCIICHAT_HIP_CPU_BUDGET=$b timeout 300 python scripts/gpu_thread_fuzz.py 24 400 2>&1 | grep -v amdgpu.ids | tail -3; done | tee $OUT/thread_fuzz.txt
gcc -O2 -I include scripts/dropin_threads.c -o scripts/dropin_threads -L ascii-chat_amd -lasciichat_hip -Wl,-rpath,$PWD/ascii-chat_amd -lpthread || exit 1
ulimit -c 0
for T in 8 32 128; do for pooled in 0 1; do
  echo "## ASCIICHAT_HIP_CONFINE=1 T=$T"; ASCIICHAT_HIP_CONFINE=1 DT_MIN_T=$T DT_POOLED=$pooled timeout 120 ./scripts/dropin_threads $T 2>&1 | grep -v amdgpu.ids
done; done | tee $OUT/confine.txt
