#!/bin/bash
# round 6, visit P: is the sampled 200x60 leg slower than in visits A-D (13.5 us there, 16.8-17.0 in the profile visit)?  This
# library (HEAD), the library of the previous session's last commit (lib_r6s1.so) and round 5's (lib_r5.so), interleaved on one box
TAG=${1:-r6p}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
HOT=1 bash scripts/gpu_abn.sh $TAG/ab "HEAD lib_r6s1.so lib_r5.so" "sampled_200x60_truecolor sampled_200x60_truecolor_blocks sampled_80x24_truecolor 1080p_80x24_truecolor" 3 > $O/ab_summary.txt 2>&1; tail -n 26 $O/ab_summary.txt
