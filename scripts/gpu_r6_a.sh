#!/bin/bash
# round 6, visit A: the stream kernel's lean loop (truecolor foreground) against round 5's library (lib_r5.so), interleaved on one
# box: the GPU tests that touch the stream kernel first, then kernel time per launch, four in flight and one at a time.
TAG=${1:-r6a}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "stream or torture or word_built or small_launches or full_size or palettes or aspect" > $O/pytest_stream.log 2>&1; echo "pytest rc=$?" >> $O/pytest_stream.log; tail -4 $O/pytest_stream.log
HOT=1 bash scripts/gpu_abn.sh $TAG "lib_r5.so HEAD" "sampled_200x60_truecolor sampled_80x24_truecolor 1080p_80x24_truecolor 4k_200x60_truecolor" 3
