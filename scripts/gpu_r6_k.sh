#!/bin/bash
# round 6, visit K: small launches of the run-structured modes shared out over four-wave workgroups of the rows kernel
# (render_rows.hpp PARTS, geometry 31): GPU tests, then one / eight 80x24 and 120x40 frames per launch --
# the automatic choice next to every older form, then the shared-out form at forced part counts (ASCIICHAT_HIP_ROWS_PARTS: 1 =
# never) on the product library (four cell slots: three 80-cell rows per block) and on lib_p2.so (two slots: a row per block).
TAG=${1:-r6k}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -k "small_launches or rows or torture or dropin or graph_replay or split or multi_workgroup" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 900 python3 scripts/gpu_small_run_modes.py 1 8 > $O/small_all_forms.txt 2>> $O/stderr.txt; cat $O/small_all_forms.txt
for lib in HEAD lib_p2.so; do
  for P in 1 2 3 4 6 8 12 24; do
    path=""; [ "$lib" != HEAD ] && path=$PWD/ascii-chat_amd/$lib
    echo "## library $lib ASCIICHAT_HIP_ROWS_PARTS=$P"
    ASCIICHAT_HIP_LIB=$path ONLY_AUTO=1 ASCIICHAT_HIP_ROWS_PARTS=$P timeout 600 python3 scripts/gpu_small_run_modes.py 1 8 2>> $O/stderr.txt
  done
done > $O/small_parts_sweep.txt; cat $O/small_parts_sweep.txt
for rep in 1 2 3; do
timeout 300 python3 bench.py --workload 640x480_80x24_mono --batch 1 --others '' --no-cpu --no-wire --no-d2h --batch-sweep '' --steps 100 --warmup 20 --extra $O/extra_k1_$rep.json > $O/line_k1_$rep.txt 2>> $O/stderr.txt
python3 -c "
import json; d=json.load(open('$O/extra_k1_$rep.json')); r=d['roofline']; print('configs[0] lone mono frame: value', d['value'], d['unit'], 'ms_per_step', d['ms_per_step']*1e3, 'us kernel', r['kernel_ms']*1e3, 'us variant', d['config'].get('kernel_variant'), 'verify', (d.get('verify') or {}).get('byte_identical_to_oracle'))"
done | tee $O/k1.txt
