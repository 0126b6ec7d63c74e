#!/bin/bash
# round 5 regression visit (the rows kernel was rebuilt): smoke(), the drop-in fuzzer (direct path / every call through the
# combiner), the API fuzzer, the thread fuzz, the shared-out fuzz on the DEFAULT build; the randomised soak -- which forces
# every geometry, fused CRC on the rows kernel included -- on the -DACHIP_ALL_GEOMETRIES build (lib_all.so)
cd $GRAFT_REPO_ROOT; TAG=${1:-r5regress}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | grep -v amdgpu.ids | tail -1 | tee $OUT/smoke.txt
{ echo "## drop-in fuzz, direct path"; timeout 400 python scripts/gpu_dropin_fuzz.py 78 ${N:-2000} 2>&1 | grep -v amdgpu.ids | tail -2
  echo "## drop-in fuzz, every call through the combiner"; ASCIICHAT_HIP_COALESCE=1 timeout 400 python scripts/gpu_dropin_fuzz.py 79 ${N:-2000} 2>&1 | grep -v amdgpu.ids | tail -2
  echo "## API fuzz"; timeout 400 python scripts/gpu_api_fuzz.py 80 200 2>&1 | grep -v amdgpu.ids | tail -2
  for b in 1 1000; do echo "## thread fuzz, combiner always, ASCIICHAT_HIP_CPU_BUDGET=$b"; ASCIICHAT_HIP_QUIET=1 ASCIICHAT_HIP_COALESCE=1 ASCIICHAT_HIP_CPU_BUDGET=$b timeout 300 python scripts/gpu_thread_fuzz.py 24 400 2>&1 | grep -v amdgpu.ids | tail -1; done
  echo "## shared-out small launches: random plans at the policy's and at forced part counts"
  for pc in 0 5 16; do ASCIICHAT_HIP_STREAM_PARTS=$pc timeout 300 python scripts/gpu_parts_fuzz.py $((100 + pc)) 200 2>&1 | grep -v amdgpu.ids | tail -1; done
  echo "## soak (all-geometries build)"; ASCIICHAT_HIP_LIB=$PWD/ascii-chat_amd/lib_all.so timeout 900 python scripts/gpu_soak.py --seed 5 2>&1 | grep -v amdgpu.ids | tail -3; } | tee $OUT/regression.txt
