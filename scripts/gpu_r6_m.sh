#!/bin/bash
# round 6, visit M: policy audit of the segment geometries (rows beyond 448 cells): 27 / 29 / the phase kernel whole and in bands,
# mono and truecolor half blocks, dense and 1080p sources, one launch at a time and four plans in flight
TAG=${1:-r6m}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for args in "--wide --modes=mono,hb_true" "--wide --modes=mono,hb_true --dense" "--wide --modes=mono,hb_true --inflight" "--wide --modes=mono,hb_true --dense --inflight"; do
  name=$(echo $args | tr -d ' =,-'); timeout 1500 python3 scripts/gpu_policy_audit.py $args > $O/audit_$name.txt 2>> $O/stderr.txt; tail -12 $O/audit_$name.txt
done
