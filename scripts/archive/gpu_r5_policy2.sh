#!/bin/bash
# round 5, after the word-built SGRs: the geometry policy of the modes whose kernels changed (truecolor half blocks on the rows
# kernel, truecolor / ANSI-256 on the stream kernel), full grids: full-frame and dense sources, one launch at a time and four
# in flight
TAG=${1:-r5policy2}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for pass in "" "--inflight" "--dense" "--dense --inflight"; do
  name=hb_true$(echo "$pass" | tr -d ' ' | tr '-' '_')
  timeout 300 python scripts/gpu_policy_audit.py --modes=hb_true $pass 2>&1 | grep -v amdgpu.ids > $O/$name.txt; echo "## $name"; tail -6 $O/$name.txt
done
