#!/bin/bash
# round 5: where does the sixteen-wave rows geometry stand for the short-token modes (mono, mono half blocks)?
TAG=${1:-r5policy5}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for pass in "--modes=mono" "--modes=mono --inflight" "--modes=mono --dense" "--modes=mono --dense --inflight" \
            "--other-modes --modes=hb_mono" "--other-modes --modes=hb_mono --inflight" "--other-modes --modes=hb_mono --dense" "--other-modes --modes=hb_mono --dense --inflight"; do
  name=$(echo "$pass" | sed 's/--modes=//; s/--//g; s/[ ,]/_/g')
  timeout 400 python scripts/gpu_policy_audit.py $pass 2>&1 | grep -v amdgpu.ids > $O/$name.txt; echo "## $name"; tail -6 $O/$name.txt
done
