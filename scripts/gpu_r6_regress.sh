#!/bin/bash
# round 6 regression visit (the rows kernel gained two forms: rows cut into segments, frames shared out over workgroups): the whole
# GPU suite on the default build and on the -DACHIP_ALL_GEOMETRIES build (lib_all.so), smoke(), the drop-in fuzzer (direct path /
# every call through the combiner), the API fuzzer, the thread fuzz, the shared-out fuzz of both kernels at the policy's and at
# forced part counts, the randomised soak (which now draws rows of 449-1500 cells and forces geometries 27 / 29 too).
cd $GRAFT_REPO_ROOT; TAG=${1:-r6regress}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $OUT/gputests_default.log 2>&1; echo "pytest rc=$?" >> $OUT/gputests_default.log; tail -3 $OUT/gputests_default.log
ASCIICHAT_HIP_LIB=$PWD/ascii-chat_amd/lib_all.so timeout 2400 python -m pytest tests -m gpu -q > $OUT/gputests_all_geometries.log 2>&1; echo "pytest rc=$?" >> $OUT/gputests_all_geometries.log; tail -3 $OUT/gputests_all_geometries.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | grep -v amdgpu.ids | tail -1 | tee $OUT/smoke.txt
{ echo "## drop-in fuzz, direct path"; timeout 400 python scripts/gpu_dropin_fuzz.py 78 ${N:-2000} 2>&1 | grep -v amdgpu.ids | tail -2
  echo "## drop-in fuzz, every call through the combiner"; ASCIICHAT_HIP_COALESCE=1 timeout 400 python scripts/gpu_dropin_fuzz.py 79 ${N:-2000} 2>&1 | grep -v amdgpu.ids | tail -2
  echo "## API fuzz"; timeout 400 python scripts/gpu_api_fuzz.py 80 200 2>&1 | grep -v amdgpu.ids | tail -2
  for b in 1 1000; do echo "## thread fuzz, combiner always, ASCIICHAT_HIP_CPU_BUDGET=$b"; ASCIICHAT_HIP_QUIET=1 ASCIICHAT_HIP_COALESCE=1 ASCIICHAT_HIP_CPU_BUDGET=$b timeout 300 python scripts/gpu_thread_fuzz.py 24 400 2>&1 | grep -v amdgpu.ids | tail -1; done
  echo "## shared-out small launches: random plans at the policy's and at forced part counts"
  for pc in 0 5 16; do ASCIICHAT_HIP_STREAM_PARTS=$pc timeout 300 python scripts/gpu_parts_fuzz.py $((100 + pc)) 200 2>&1 | grep -v amdgpu.ids | tail -1; done
  for pc in 0 2 7 24; do ASCIICHAT_HIP_ROWS_PARTS=$pc timeout 300 python scripts/gpu_parts_fuzz.py $((200 + pc)) 200 --rows 2>&1 | grep -v amdgpu.ids | tail -1; done
  for seed in ${SEEDS:-5 6 7}; do echo "## soak (all-geometries build), seed $seed"; ASCIICHAT_HIP_LIB=$PWD/ascii-chat_amd/lib_all.so timeout 900 python scripts/gpu_soak.py --seed $seed 2>&1 | grep -v amdgpu.ids | tail -3; done
  for seed in ${SEEDS:-5 6 7}; do echo "## soak (default build), seed 1$seed"; timeout 900 python scripts/gpu_soak.py --seed 1$seed 2>&1 | grep -v amdgpu.ids | tail -3; done; } | tee $OUT/regression.txt
