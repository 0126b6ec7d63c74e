#!/bin/bash
# round 4, visit g: where the CPU time of a drop-in call goes (getrusage per call + the combiner's own stamps), and a sweep
# of the combiner's knobs on today's kernels (every line: same box, same minute)
mkdir -p gpurun_out/r4g; cd $GRAFT_REPO_ROOT
gcc -O2 -I include scripts/dropin_threads.c -o scripts/dropin_threads -L ascii-chat_amd -lasciichat_hip -Wl,-rpath,$PWD/ascii-chat_amd -lpthread || exit 1
one() { # T, env...
  local T=$1; shift
  echo "## T=$T $*"
  env "$@" DT_RUSAGE=1 ASCIICHAT_HIP_COMBINE_STATS=1 DT_MIN_T=$T DT_POOLED=0 timeout 120 ./scripts/dropin_threads $T 2>&1 | grep -v amdgpu.ids
}
{
for T in 64 128; do
  one $T X=0
  for v in ASCIICHAT_HIP_CB_SHARE=2 ASCIICHAT_HIP_CB_SHARE=4 ASCIICHAT_HIP_CB_SHARE=6 ASCIICHAT_HIP_CB_LINGER_US=10 ASCIICHAT_HIP_CB_LINGER_US=60 \
           ASCIICHAT_HIP_CB_BLOCK=0 ASCIICHAT_HIP_CB_INFLIGHT=3 ASCIICHAT_HIP_CB_INFLIGHT=10 ASCIICHAT_HIP_CB_SPIN_US=0 ASCIICHAT_HIP_CB_SPIN_US=20 \
           ASCIICHAT_HIP_CB_FANOUT=4 ASCIICHAT_HIP_COMBINE_INPLACE=0; do
    one $T $v
  done
  one $T X=0
done
} > gpurun_out/r4g/dropin_sweep.txt 2>&1
grep -E "^##|calls/s" gpurun_out/r4g/dropin_sweep.txt | paste - - | awk '{print $2, $3, $(NF-12), $(NF-11)}'
