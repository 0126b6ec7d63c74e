#!/bin/bash
# round 2, call w: randomised soak incl. the stream geometries + fused CRC; thread fuzz with calls coalesced always / by default
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python scripts/gpu_soak.py --rounds 120 --seed 22 > gpurun_out/w_soak.txt 2>&1; echo "soak rc=$?" | tee -a gpurun_out/w_soak.txt
grep -c "crc fused" gpurun_out/w_soak.txt; tail -4 gpurun_out/w_soak.txt
for c in 1 default; do
  if [ $c = default ]; then unset ASCIICHAT_HIP_COALESCE; else export ASCIICHAT_HIP_COALESCE=$c; fi
  timeout 600 python scripts/gpu_thread_fuzz.py 32 600 > gpurun_out/w_thread_fuzz_$c.txt 2>&1; echo "thread fuzz (coalesce $c) rc=$?" | tee -a gpurun_out/w_thread_fuzz_$c.txt
  tail -3 gpurun_out/w_thread_fuzz_$c.txt
done
