#!/bin/bash
# drop-in layer under T concurrent render threads: adaptive coalescing (default), never, always
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2p; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
echo "# scripts/dropin_threads.c on one MI355X ($(nproc) host threads): T threads each calling ascii_convert_with_capabilities on a 1080p frame (-> 80x24 truecolor) and freeing the string" | tee $OUT/dropin_threads.txt
echo "## default: calls share launches (combine.c) from 24 concurrent callers on" | tee -a $OUT/dropin_threads.txt
timeout 200 scripts/dropin_threads 64 2>&1 | tee -a $OUT/dropin_threads.txt
echo "## ASCIICHAT_HIP_COALESCE=0: every call its own upload + launch + wait (the round-1 path)" | tee -a $OUT/dropin_threads.txt
ASCIICHAT_HIP_COALESCE=0 timeout 200 scripts/dropin_threads 64 2>&1 | tee -a $OUT/dropin_threads.txt
echo "## ASCIICHAT_HIP_COALESCE=1: every call through the combiner" | tee -a $OUT/dropin_threads.txt
ASCIICHAT_HIP_COALESCE=1 timeout 200 scripts/dropin_threads 64 2>&1 | tee -a $OUT/dropin_threads.txt
