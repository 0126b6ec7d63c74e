#!/bin/bash
# round 5, the evidence visit on the round's last build: GPU suite on both builds, the driver's bench command, the profile visit
TAG=${1:-r5final}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
bash scripts/gpu_r5_e.sh $TAG
bash scripts/gpu_r5_profiles.sh ${TAG}_prof
