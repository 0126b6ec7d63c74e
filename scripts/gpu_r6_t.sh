#!/bin/bash
# round 6, visit T: quant16 without the table walk (render_kernels.hpp: the 16 ANSI colours' structure instead of 16 distances) --
# the 16-colour renderers (P16, H16) and the Floyd-Steinberg renderer (PD) against the library before it (lib_q0.so), interleaved,
# four launches in flight and one at a time; the tests of those modes first; small launches of 16-colour half blocks (where they
# lost to their row bands in visit Q) again with the shared-out segment geometry forced on for them
TAG=${1:-r6t}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -n "passed\|failed" $O/pytest.log | tail -2
HOT=1 bash scripts/gpu_abn.sh $TAG "lib_q0.so HEAD" "1080p_80x24_dither16_bg 1080p_80x24_ansi16 sampled_200x60_ansi16 1080p_80x24_halfblock16 sampled_400x240_halfblock16 sampled_400x240_halfblock" 3 2>&1 | tail -40
for lib in lib_q0.so "" lib_q0.so ""; do echo "## library ${lib:-HEAD}"; ASCIICHAT_HIP_LIB=${lib:+$PWD/ascii-chat_amd/$lib} SMALL_SIZES=160x48,256x30 timeout 600 python3 scripts/gpu_small_run_modes.py 1 4 2>> $O/stderr.txt | grep "half-block 16\|half-block 256" | cut -c1-330; done | tee $O/small_hb16.txt
for seed in 91 92; do timeout 900 python3 scripts/gpu_soak.py --seed $seed --rounds 150 2>> $O/stderr.txt | tail -1; done | tee $O/soak.txt
