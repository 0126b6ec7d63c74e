#!/bin/bash
# round 6, visit N: the shared-out rows form for rows of 129-256 cells (geometry 32: four cell slots, one row per block): one /
# four / sixteen frames of 160x48 and 200x60, every older form beside it, then the automatic choice with the geometry switched
# off (ASCIICHAT_HIP_ROWS_PARTS_WIDE=0: row bands as until now) and on
TAG=${1:-r6n}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "small_launches or multi_workgroup or graph_replay or rows" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
SMALL_SIZES=160x48,200x60,256x30 timeout 900 python3 scripts/gpu_small_run_modes.py 1 4 16 > $O/small_all_forms.txt 2>> $O/stderr.txt; cut -c1-330 $O/small_all_forms.txt
for sw in 0 1 0 1; do echo "## ASCIICHAT_HIP_ROWS_PARTS_WIDE=$sw"; SMALL_SIZES=160x48,200x60,256x30 ONLY_AUTO=1 ASCIICHAT_HIP_ROWS_PARTS_WIDE=$sw timeout 600 python3 scripts/gpu_small_run_modes.py 1 4 16 2>> $O/stderr.txt; done > $O/small_switch.txt; cat $O/small_switch.txt
