#!/bin/bash
# round 6, visit I: rows beyond 448 cells on the rows kernel (render_rows.hpp WIDE, geometries 27 / 29): GPU tests, then the
# wide-row bench legs on every geometry that takes them (27 / 29 / the phase kernel's 0 and 4), then the one-block workloads on
# this library next to round 5's (the refactor must not have moved them).
TAG=${1:-r6i}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -k "rows or torture or aspect or full_size" > $O/pytest_rows.log 2>&1; echo "pytest rc=$?" >> $O/pytest_rows.log; tail -4 $O/pytest_rows.log
leg() { # workload, variant
  timeout 600 python3 bench.py --workload $1 --others '' --no-cpu --no-wire --no-d2h --batch-sweep '' --steps 40 --warmup 10 ${2:+--variant $2} --extra $O/extra_$1_${2:-auto}.json > $O/line_$1_${2:-auto}.txt 2>> $O/stderr.txt
  python3 - $O/extra_$1_${2:-auto}.json $1 ${2:-auto} <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']; one=(d.get('one_launch_at_a_time') or {})
    print(f"{sys.argv[2]:34s} variant {str(d['config'].get('kernel_variant')):>3s} ({sys.argv[3]:4s}) kernel {r['kernel_ms']*1e3:8.2f} us  frac {r['frac']:.3f}  one at a time {one.get('kernel_ms',0)*1e3:8.2f} us (variant {one.get('kernel_variant')})  verify {(d.get('verify') or {}).get('byte_identical_to_oracle')}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
}
for rep in 1 2; do
for w in sampled_640x360_halfblock 4k_640x180_halfblock; do for v in "" 27 29 4 0; do leg $w $v; done; done
done | tee $O/legs.txt
HOT=1 bash scripts/gpu_abn.sh $TAG/ab "HEAD lib_r5.so" "sampled_400x240_halfblock 4k_400x120_halfblock 1080p_80x24_halfblock 640x480_80x24_mono" 2 > $O/ab_summary.txt 2>&1; tail -12 $O/ab_summary.txt
