#!/bin/bash
# round 5: the short-token modes after geometry 26 entered their policy; then the GPU suite and the soak on the final policy
TAG=${1:-r5policy6}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for pass in "--modes=mono" "--modes=mono --inflight" "--modes=mono --dense" "--modes=mono --dense --inflight" \
            "--other-modes --modes=hb_mono" "--other-modes --modes=hb_mono --inflight" "--other-modes --modes=hb_mono --dense" "--other-modes --modes=hb_mono --dense --inflight"; do
  name=$(echo "$pass" | sed 's/--modes=//; s/--//g; s/[ ,]/_/g')
  timeout 400 python scripts/gpu_policy_audit.py $pass 2>&1 | grep -v amdgpu.ids > $O/$name.txt; echo "## $name"; tail -4 $O/$name.txt
done
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_default.log 2>&1; grep -E "passed|failed" $O/pytest_default.log | tail -2
timeout 600 python scripts/gpu_soak.py --seed 11 2>&1 | grep -v amdgpu.ids | tail -2
