#!/bin/bash
# round 5, after the word-built SGRs, second visit: truecolor half blocks from 4K sources, and the coloured half-block modes
# whose tokens are still byte-built (where does the sixteen-wave rows geometry stand for them?)
TAG=${1:-r5policy3}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for pass in "--4k --modes=hb_true" "--4k --modes=hb_true --inflight" "--other-modes --modes=hb_256,hb_16" "--other-modes --modes=hb_256,hb_16 --inflight" "--other-modes --modes=hb_256,hb_16 --dense" "--other-modes --modes=hb_256,hb_16 --dense --inflight"; do
  name=$(echo "$pass" | sed 's/--modes=//; s/--//g; s/[ ,]/_/g')
  timeout 400 python scripts/gpu_policy_audit.py $pass 2>&1 | grep -v amdgpu.ids > $O/$name.txt; echo "## $name"; tail -4 $O/$name.txt
done
