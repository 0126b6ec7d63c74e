#!/bin/bash
# round 6, visit Q: the shared-out rows form for rows of 129-512 cells, second attempt (geometry 32 = WIDE + PARTS: a row cut into
# segments of at most 128 cells, whole rows per four-wave workgroup, a segment per wave): the tests that cover it, one / four /
# ten frames of 160x48, 200x60, 256x30, 300x40, 400x30 with every older form beside it, then the automatic choice with the
# geometry switched off (ASCIICHAT_HIP_ROWS_PARTS_WIDE=0: row bands as until now) and on, interleaved; the parts fuzz
TAG=${1:-r6q}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "small_launches or multi_workgroup or graph_replay or rows" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
SMALL_SIZES=160x48,200x60,256x30,300x40,400x30 timeout 1200 python3 scripts/gpu_small_run_modes.py 1 4 10 > $O/small_all_forms.txt 2>> $O/stderr.txt; cut -c1-330 $O/small_all_forms.txt
for sw in 0 1 0 1; do echo "## ASCIICHAT_HIP_ROWS_PARTS_WIDE=$sw"; SMALL_SIZES=160x48,200x60,256x30,300x40,400x30 ONLY_AUTO=1 ASCIICHAT_HIP_ROWS_PARTS_WIDE=$sw timeout 600 python3 scripts/gpu_small_run_modes.py 1 4 10 2>> $O/stderr.txt; done > $O/small_switch.txt; cat $O/small_switch.txt
for np in 24 48; do echo "## ASCIICHAT_HIP_ROWS_PARTS=$np (160x48: two rows / one row per workgroup)"; SMALL_SIZES=160x48 ONLY_AUTO=1 ASCIICHAT_HIP_ROWS_PARTS=$np timeout 600 python3 scripts/gpu_small_run_modes.py 1 4 2>> $O/stderr.txt; done > $O/small_parts_sweep.txt; cat $O/small_parts_sweep.txt
for seed in 61 62; do timeout 600 python3 scripts/gpu_parts_fuzz.py $seed 250 --rows 2>> $O/stderr.txt | tail -1; done | tee $O/parts_fuzz.txt
ASCIICHAT_HIP_ROWS_PARTS_WIDE=0 timeout 600 python3 scripts/gpu_parts_fuzz.py 63 120 --rows 2>> $O/stderr.txt | tail -1 | tee -a $O/parts_fuzz.txt
tail -5 $O/stderr.txt
