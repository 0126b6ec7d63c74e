#!/bin/bash
# round 6, visit J: cell slots per segment of the WIDE geometries (ACHIP_ROWS_WIDE_CPL = 5 / 6 / 7 as A/B builds lib_w5.so /
# HEAD / lib_w7.so) on the 640-cell workloads (two segments of 320 cells: five slots hold them exactly)
TAG=${1:-r6j}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "rows_wider" > $O/pytest_rows.log 2>&1; echo "pytest rc=$?" >> $O/pytest_rows.log; tail -3 $O/pytest_rows.log
HOT=1 bash scripts/gpu_abn.sh $TAG/ab "HEAD lib_w5.so lib_w7.so" "sampled_640x360_halfblock 4k_640x180_halfblock" 2 > $O/ab_summary.txt 2>&1; tail -14 $O/ab_summary.txt
