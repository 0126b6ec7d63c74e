#!/bin/bash
# round 6, visit O: the round-5 policy audit of the run-structured modes again with this round's geometries among the candidates'
# automatic choices (the shared-out rows form for small launches): 1080p and dense sources, one launch at a time and four in flight
TAG=${1:-r6o}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for args in "--modes=mono,hb_true" "--modes=mono,hb_true --inflight" "--modes=mono,hb_true --dense" "--other-modes --modes=hb_256,hb_mono"; do
  name=$(echo $args | tr -d ' =,-'); timeout 2400 python3 scripts/gpu_policy_audit.py $args > $O/audit_$name.txt 2>> $O/stderr.txt; tail -n 12 $O/audit_$name.txt
done
