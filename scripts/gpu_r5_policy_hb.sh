#!/bin/bash
# round 5: the coloured half-block modes' geometry policy after the word-built SGRs -- full grids (six terminal sizes x seven
# batch sizes; 4K: three sizes), full-frame and dense sources, one launch at a time and four in flight
TAG=${1:-r5policy4}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for pass in "--modes=hb_true" "--modes=hb_true --inflight" "--modes=hb_true --dense" "--modes=hb_true --dense --inflight" "--4k --modes=hb_true" "--4k --modes=hb_true --inflight" \
            "--other-modes --modes=hb_256,hb_16" "--other-modes --modes=hb_256,hb_16 --inflight" "--other-modes --modes=hb_256,hb_16 --dense" "--other-modes --modes=hb_256,hb_16 --dense --inflight"; do
  name=$(echo "$pass" | sed 's/--modes=//; s/--//g; s/[ ,]/_/g')
  timeout 400 python scripts/gpu_policy_audit.py $pass 2>&1 | grep -v amdgpu.ids > $O/$name.txt; echo "## $name"; tail -5 $O/$name.txt
done
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_default.log 2>&1; grep -E "passed|failed" $O/pytest_default.log | tail -2
