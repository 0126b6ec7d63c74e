#!/bin/bash
# A/B of two builds of the library on one box, interleaved (A B A B): scripts/gpu_ab.sh <tag> <other .so> [workloads...]
# prints kernel_ms (HIP events over the timed launches) and ms_per_step per run
set -u
TAG=$1; OTHER=$2; shift 2
WL=${@:-"4k_400x120_halfblock 4k_200x60_truecolor 1080p_80x24_truecolor 1080p_80x24_halfblock 1080p_80x24_ansi256"}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for w in $WL; do
  for rep in 1 2; do
    for lib in "" "$OTHER"; do
      name=${lib:-HEAD}
      ASCIICHAT_HIP_LIB=${lib:+$PWD/$lib} python3 bench.py --workload $w --others '' --no-cpu --no-wire --no-d2h --no-hot --batch-sweep '' --steps 100 --warmup 20 \
         --extra $O/extra_${w}_$(basename $name)_$rep.json > $O/line_${w}_$(basename $name)_$rep.txt 2>> $O/stderr.txt
      python3 - "$O/extra_${w}_$(basename $name)_$rep.json" "$w" "$name" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(f"{sys.argv[2]:28s} {sys.argv[3]:28s} kernel_ms {r['kernel_ms']*1e3:9.2f} us  ms_per_step {d['ms_per_step']*1e3:9.2f} us  frac {r['frac']:.4f} variant {d['config'].get('kernel_variant')} in flight {r.get('launches_in_flight')}")
PY
    done
  done
done | tee $O/ab.txt
