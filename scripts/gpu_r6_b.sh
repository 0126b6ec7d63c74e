#!/bin/bash
# round 6, visit B: the lean loop with the slot-major order kept for line-apart samples: stream GPU tests, A/B against round 5's
# library, SQ counters of the sampled 200x60 launch (before / after).
TAG=${1:-r6b}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "stream or torture or word_built or small_launches or full_size or palettes or aspect" > $O/pytest_stream.log 2>&1; echo "pytest rc=$?" >> $O/pytest_stream.log; tail -4 $O/pytest_stream.log
HOT=1 bash scripts/gpu_abn.sh $TAG "lib_r5.so HEAD" "sampled_200x60_truecolor sampled_80x24_truecolor 1080p_80x24_truecolor 4k_200x60_truecolor" 2
KERNEL=render_stream_kernel bash scripts/gpu_pmc_rows.sh ${TAG}_pmc_head sampled_200x60_truecolor HEAD
KERNEL=render_stream_kernel bash scripts/gpu_pmc_rows.sh ${TAG}_pmc_r5 sampled_200x60_truecolor lib_r5.so
