#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2n
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_cabi_harness.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/pytest.log
echo "## default (coalescing from 24 concurrent callers on); host threads: $(nproc)" | tee $OUT/dropin_threads.txt
timeout 300 scripts/dropin_threads 64 2>&1 | tee -a $OUT/dropin_threads.txt
echo "## ASCIICHAT_HIP_COALESCE=0 (every call its own launch)" | tee -a $OUT/dropin_threads.txt
ASCIICHAT_HIP_COALESCE=0 timeout 300 scripts/dropin_threads 64 2>&1 | tee -a $OUT/dropin_threads.txt
echo "## ASCIICHAT_HIP_COALESCE=1 (every call through the combiner)" | tee -a $OUT/dropin_threads.txt
ASCIICHAT_HIP_COALESCE=1 timeout 300 scripts/dropin_threads 64 2>&1 | tee -a $OUT/dropin_threads.txt
ASCIICHAT_HIP_COALESCE=1 timeout 600 python -m pytest tests/test_gpu_dropin.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -2 | tee -a $OUT/pytest.log
ASCIICHAT_HIP_COALESCE=1 timeout 600 python scripts/gpu_thread_fuzz.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $OUT/thread_fuzz.txt
ASCIICHAT_HIP_COALESCE=1 timeout 900 python scripts/gpu_dropin_fuzz.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/dropin_fuzz.txt
timeout 600 python scripts/gpu_thread_fuzz.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/thread_fuzz.txt

