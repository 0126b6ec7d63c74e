#!/bin/bash
# round 4, visit h: a frame's blocks shared out over workgroups (stream kernel PARTS, geometry 18): small launches
mkdir -p gpurun_out/r4h; cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > gpurun_out/r4h/pytest_gpu.txt 2>&1; tail -8 gpurun_out/r4h/pytest_gpu.txt
for p in 0 1; do
  echo "## ASCIICHAT_HIP_STREAM_PARTS=$p"
  ASCIICHAT_HIP_STREAM_PARTS=$p python scripts/gpu_small_batch_variants.py 2>&1 | grep -E "^#|automatic"
done > gpurun_out/r4h/small_batch_parts_policy.txt 2>&1
cat gpurun_out/r4h/small_batch_parts_policy.txt
for p in 0 1; do ASCIICHAT_HIP_STREAM_PARTS=$p python bench.py --workload grid9 --steps 200 --warmup 20 --extra gpurun_out/r4h/grid9_parts$p.json 2>/dev/null | tail -1 | cut -c1-600; done
