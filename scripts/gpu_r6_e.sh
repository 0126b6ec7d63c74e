#!/bin/bash
# round 6, visit E: what rows wider than a rows-kernel block cost today (the phase kernel's wide geometries), next to 400x120
TAG=${1:-r6e}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
leg() { # workload, extra args
  local w=$1; shift
  timeout 600 python3 bench.py --workload $w --others '' --no-cpu --no-wire --no-d2h --batch-sweep '' --steps 20 --warmup 5 --regions 5 --input-sets 4 --streams 4 "$@" --extra $O/extra_$w.json > $O/line_$w.txt 2>> $O/stderr.txt
  python3 - $O/extra_$w.json $w <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']; one=(d.get('one_launch_at_a_time') or {})
    cells=d.get('cells_per_s',0)
    print(f"{sys.argv[2]:30s} variant {str(d['config'].get('kernel_variant')):>3s} kernel {r['kernel_ms']*1e3:9.2f} us  frac {r['frac']:.3f}  one at a time {one.get('kernel_ms',0)*1e3:9.2f} us (variant {one.get('kernel_variant')}) alg MB/launch {r.get('alg_bytes_per_launch',0)/1e6:.1f} verify {(d.get('verify') or {}).get('byte_identical_to_oracle')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for w in sampled_400x240_halfblock sampled_640x360_halfblock 4k_400x120_halfblock 4k_640x180_halfblock; do leg $w; done | tee $O/legs.txt
tail -5 $O/stderr.txt
