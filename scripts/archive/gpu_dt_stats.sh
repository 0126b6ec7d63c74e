#!/bin/bash
# where a combined drop-in call spends its time (ASCIICHAT_HIP_COMBINE_STATS) per number of calling threads; VARIANTS is a
# list of env settings to compare (default: the build's defaults only); after the drop-in GPU tests
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-dtstats}; mkdir -p $OUT; export TMPDIR=/tmp
K="dropin or drop_in or coalesc or combin or pool" bash scripts/gpu_pytest.sh ${1:-dtstats} | tail -3
gcc -O2 -I include scripts/dropin_threads.c -o scripts/dropin_threads -L ascii-chat_amd -lasciichat_hip -Wl,-rpath,$PWD/ascii-chat_amd -lpthread || exit 1
ulimit -c 0
IFS=';' read -ra VS <<< "${VARIANTS:-X=0}"
for T in ${THREADS:-1 4 16 32 64 128}; do for pooled in ${POOLED:-0 1}; do for v in "${VS[@]}"; do
  echo "## T=$T pooled=$pooled $v" | tee -a $OUT/stats.txt
  thr0=$(awk '/throttled_usec/{print $2}' /sys/fs/cgroup/cpu.stat 2>/dev/null)
  env $v ASCIICHAT_HIP_COMBINE_STATS=1 DT_MIN_T=$T DT_POOLED=$pooled timeout 120 ${TASKSET} ./scripts/dropin_threads $T 2>&1 | grep -v amdgpu.ids | sed 's/1920x1080 -> 80x24 colour 3 mode 0, //' | tee -a $OUT/stats.txt
  thr1=$(awk '/throttled_usec/{print $2}' /sys/fs/cgroup/cpu.stat 2>/dev/null)
  echo "   cgroup throttled for $(( (${thr1:-0} - ${thr0:-0}) / 1000 )) ms during this run" | tee -a $OUT/stats.txt
done; done; done
