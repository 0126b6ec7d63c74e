#!/bin/bash
# regression visit after the drop-in / ingest changes: GPU suite, smoke(), the drop-in fuzzer (direct path, then every call
# through the combiner), the API fuzzer (frame table, resize, CRC ...), the thread fuzz in both waiting modes, the
# randomised soak
cd $GRAFT_REPO_ROOT; TAG=${1:-regress}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tee $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | grep -v amdgpu.ids | tail -1 | tee $OUT/smoke.txt
{ echo "## drop-in fuzz, direct path"; timeout 400 python scripts/gpu_dropin_fuzz.py 78 ${N:-3000} 2>&1 | grep -v amdgpu.ids | tail -2
  echo "## drop-in fuzz, every call through the combiner"; ASCIICHAT_HIP_COALESCE=1 timeout 400 python scripts/gpu_dropin_fuzz.py 79 ${N:-3000} 2>&1 | grep -v amdgpu.ids | tail -2
  echo "## API fuzz"; timeout 400 python scripts/gpu_api_fuzz.py 80 200 2>&1 | grep -v amdgpu.ids | tail -2
  for b in 1 1000; do echo "## thread fuzz, combiner always, ASCIICHAT_HIP_CPU_BUDGET=$b"; ASCIICHAT_HIP_COALESCE=1 ASCIICHAT_HIP_CPU_BUDGET=$b timeout 300 python scripts/gpu_thread_fuzz.py 24 400 2>&1 | grep -v amdgpu.ids | tail -1; done
  echo "## thread fuzz, direct path (never combined), 4 and 12 threads: small launches shared out over workgroups from several threads at once"
  for t in 4 12; do ASCIICHAT_HIP_COALESCE=0 timeout 300 python scripts/gpu_thread_fuzz.py $t 400 2>&1 | grep -v amdgpu.ids | tail -1; done
  echo "## shared-out small launches: random plans at the policy's and at forced part counts"
  for pc in 0 2 5 16 64; do ASCIICHAT_HIP_STREAM_PARTS=$pc timeout 300 python scripts/gpu_parts_fuzz.py $((100 + pc)) 200 2>&1 | grep -v amdgpu.ids | tail -1; done
  echo "## soak"; timeout 600 python scripts/gpu_soak.py 2>&1 | grep -v amdgpu.ids | tail -3; } | tee $OUT/regression.txt
