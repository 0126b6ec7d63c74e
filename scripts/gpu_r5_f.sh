#!/bin/bash
# round 5, visit F: GPU suite on the default build (all of it), then the profile visit
TAG=${1:-r5f}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_default.log 2>&1; echo "pytest default build rc=$?" >> $O/pytest_default.log; grep -E "passed|failed|FAILED|rc=" $O/pytest_default.log | tail -6
bash scripts/gpu_r5_profiles.sh ${TAG}_prof
