#!/bin/bash
# round 6: a long randomised soak on the round's last kernels -- many seeds of scripts/gpu_soak.py on both builds, the parts fuzz of
# both kernels over many seeds, the drop-in fuzz; every frame against the oracle.  usage: gpu_r6_soak_many.sh <tag> <first seed> <count>
cd $GRAFT_REPO_ROOT; TAG=${1:-r6soak}; S0=${2:-400}; N=${3:-20}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for ((s = S0; s < S0 + N; s++)); do
  lib=""; [ $((s % 3)) = 0 ] && lib=$PWD/ascii-chat_amd/lib_all.so
  echo "## soak seed $s ${lib:+(all-geometries build)}"
  ASCIICHAT_HIP_LIB=$lib timeout 900 python scripts/gpu_soak.py --seed $s --rounds 150 2>&1 | grep -v amdgpu.ids | tail -1
done | tee $O/soak.txt
for ((s = S0; s < S0 + N / 2; s++)); do
  timeout 300 python scripts/gpu_parts_fuzz.py $s 200 --rows 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 300 python scripts/gpu_parts_fuzz.py $s 200 2>&1 | grep -v amdgpu.ids | tail -1
done | tee $O/parts_fuzz.txt
timeout 600 python scripts/gpu_dropin_fuzz.py $S0 6000 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/dropin_fuzz.txt
grep -c "soak OK" $O/soak.txt; grep -c "fuzz OK" $O/parts_fuzz.txt
