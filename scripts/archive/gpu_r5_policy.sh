#!/bin/bash
# round 5: the geometry policy re-audited after the rows kernel's diet (scripts/gpu_policy_audit.py, quick grids) on the build
# with every geometry: does the automatic choice still pick the fastest now that the rows kernel is 5-7 % faster?
TAG=${1:-r5policy}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
export ASCIICHAT_HIP_LIB=$PWD/ascii-chat_amd/lib_all.so
timeout 420 python scripts/gpu_policy_audit.py --quick 2>&1 | grep -v amdgpu.ids > $O/pass1_quick.txt; tail -3 $O/pass1_quick.txt
timeout 420 python scripts/gpu_policy_audit.py --quick --other-modes 2>&1 | grep -v amdgpu.ids > $O/pass2_quick_other_modes.txt; tail -3 $O/pass2_quick_other_modes.txt
timeout 420 python scripts/gpu_policy_audit.py --quick --4k 2>&1 | grep -v amdgpu.ids > $O/pass3_quick_4k.txt; tail -3 $O/pass3_quick_4k.txt
