#!/bin/bash
# drop-in scaling: T render threads calling ascii_convert_with_capabilities (scripts/dropin_threads.c), 1 .. 128 threads
TAG=${1:-dropin}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
gcc -O2 -I include scripts/dropin_threads.c -o scripts/dropin_threads -L ascii-chat_amd -lasciichat_hip -Wl,-rpath,$PWD/ascii-chat_amd -lpthread || exit 1
nproc | tee $OUT/threads.txt
for mode in default ${EXTRA_MODES}; do
  echo "## ASCIICHAT_HIP_COALESCE=$mode" | tee -a $OUT/threads.txt
  if [ $mode = default ]; then timeout ${TMO:-300} ./scripts/dropin_threads ${MAXT:-128} 2>&1 | grep -v amdgpu.ids | tee -a $OUT/threads.txt
  else ASCIICHAT_HIP_COALESCE=$mode timeout ${TMO:-300} ./scripts/dropin_threads ${MAXT:-128} 2>&1 | grep -v amdgpu.ids | tee -a $OUT/threads.txt; fi
done
