#!/bin/bash
# round 2, call zj: regression sweep on the final sources -- soak, fuzzers, thread fuzz, pool harness with pinned frames
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/zj && export TMPDIR=/tmp
O=gpurun_out/zj
timeout 900 python scripts/gpu_soak.py --rounds 100 --seed 57 > $O/soak.txt 2>&1; echo "soak rc=$?"; tail -1 $O/soak.txt
timeout 600 python scripts/gpu_dropin_fuzz.py 58 12000 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/dropin_fuzz.txt
timeout 600 python scripts/gpu_api_fuzz.py 59 300 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/api_fuzz.txt
timeout 600 python scripts/gpu_thread_fuzz.py 32 500 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/thread_fuzz.txt
for i in 1 2 3; do tests/cabi/buffer_pool_test_port gpu | tail -1; done | tee $O/pool_harness.txt
timeout 300 ./scripts/dropin_threads 2>/dev/null | tail -16 > $O/dropin_threads.txt || true
tail -8 $O/dropin_threads.txt
