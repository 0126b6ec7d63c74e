#!/bin/bash
# drop-in scaling: T render threads calling ascii_convert_with_capabilities (scripts/dropin_threads.c), 1 .. 128 threads,
# pageable and pooled images, with the combiner's own timing of a call (ASCIICHAT_HIP_COMBINE_STATS); first what the box
# gives a process (scripts/cpu_scaling.c, cgroup cpu.max), then the sweep with the process free to roam over every CPU
# the box shows and confined to as many CPUs as its quota is worth (taskset).
TAG=${1:-dropin}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
gcc -O2 -I include scripts/dropin_threads.c -o scripts/dropin_threads -L ascii-chat_amd -lasciichat_hip -Wl,-rpath,$PWD/ascii-chat_amd -lpthread || exit 1
gcc -O2 scripts/cpu_scaling.c -o /tmp/cpu_scaling -lpthread || exit 1
ulimit -c 0
{ echo "# nproc $(nproc); cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; /tmp/cpu_scaling 100000000; } | tee $OUT/threads.txt
QUOTA=$(awk '{ if ($1 == "max") print 0; else print int(($1 + $2 - 1) / $2) }' /sys/fs/cgroup/cpu.max 2>/dev/null); QUOTA=${QUOTA:-0}
run() { # label, command prefix, env...
  local label=$1 prefix=$2; shift 2
  echo "## $label" | tee -a $OUT/threads.txt
  for pooled in 0 1; do for T in ${THREADS:-1 2 4 8 16 32 64 128}; do
    thr0=$(awk '/throttled_usec/{print $2}' /sys/fs/cgroup/cpu.stat 2>/dev/null)
    env "$@" ASCIICHAT_HIP_COMBINE_STATS=1 DT_MIN_T=$T DT_POOLED=$pooled timeout 120 $prefix ./scripts/dropin_threads $T 2>&1 | grep -v amdgpu.ids | tee -a $OUT/threads.txt
    thr1=$(awk '/throttled_usec/{print $2}' /sys/fs/cgroup/cpu.stat 2>/dev/null)
    echo "   (cgroup throttled for $(( (${thr1:-0} - ${thr0:-0}) / 1000 )) ms, summed over CPUs, during this run)" | tee -a $OUT/threads.txt
  done; done
}
run "every CPU the box shows" "" X=0
if [ "$QUOTA" -gt 0 ] && [ "$QUOTA" -lt "$(nproc)" ]; then
  run "confined to the quota: taskset -c 0-$((QUOTA - 1))" "taskset -c 0-$((QUOTA - 1))" X=0
  for v in ${EXTRA:-}; do THREADS="${EXTRA_THREADS:-16 128}" run "confined, $v" "taskset -c 0-$((QUOTA - 1))" $v; done
fi
