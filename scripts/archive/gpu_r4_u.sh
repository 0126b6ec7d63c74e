#!/bin/bash
# round 4, visit U: the in-flight thresholds of the geometry policy measured with bench.py's own C-issued schedule (four launches in flight from ONE
# stream pool) instead of the audit's Python-issued torch streams: geometry 16 vs 17 and rows vs phase at 64 / 128 / 192 / 256 frames per launch
set -u
O=gpurun_out/r4u; mkdir -p $O; export TMPDIR=/tmp
run() { # workload batch variant
  python3 bench.py --workload $1 --batch $2 --variant $3 --others '' --no-cpu --no-wire --no-d2h --no-hot --steps 100 --warmup 20 --streams 4 --extra $O/x.json > /dev/null 2>> $O/stderr.txt
  python3 - $O/x.json "$1" "$2" "$3" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(f"{sys.argv[2]:24s} batch {sys.argv[3]:>3s} forced {sys.argv[4]:>2s} -> variant {d['config'].get('kernel_variant')}: kernel {r['kernel_ms']*1e3:8.2f} us  step {d['ms_per_step']*1e3:8.2f} us")
PY
}
for b in 64 128 192 256; do for v in -1 16 17; do run 4k_200x60_truecolor $b $v; done; done
for b in 64 128 192 256; do for v in -1 16 17; do run 1080p_80x24_truecolor $b $v; done; done
for b in 64 128 192 256; do for v in -1 24 4; do run 4k_400x120_halfblock $b $v; done; done
for b in 64 128 192 256; do for v in -1 25 4; do run 1080p_80x24_halfblock $b $v; done; done
