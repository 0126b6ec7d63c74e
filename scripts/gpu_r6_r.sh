#!/bin/bash
# round 6, visit R: after visit Q's policy (geometry 32 for mono, mono half blocks, truecolor half blocks; the 256- / 16-colour
# half blocks keep their bands): the whole GPU suite, the parts fuzz, the soak with its wide rows
TAG=${1:-r6r}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for seed in 71 72; do timeout 600 python3 scripts/gpu_parts_fuzz.py $seed 300 --rows 2>> $O/stderr.txt | tail -1; done | tee $O/parts_fuzz.txt
for np in 2 9 64; do ASCIICHAT_HIP_ROWS_PARTS=$np timeout 600 python3 scripts/gpu_parts_fuzz.py 7$np 150 --rows 2>> $O/stderr.txt | tail -1; done | tee -a $O/parts_fuzz.txt
timeout 900 python3 scripts/gpu_soak.py --seed 81 --rounds 150 2>> $O/stderr.txt | tail -2 | tee $O/soak.txt
tail -5 $O/stderr.txt
