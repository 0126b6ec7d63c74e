#!/bin/bash
# round 6, visit D: truecolor foreground with multi-byte palettes on the stream kernel: GPU tests, then the new bench legs next
# to the STANDARD-palette legs and to the phase kernel (geometry 4 forced: what these palettes took until round 5).
TAG=${1:-r6d}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -k "multibyte or palettes or stream or torture" > $O/pytest_u8.log 2>&1; echo "pytest rc=$?" >> $O/pytest_u8.log; tail -4 $O/pytest_u8.log
leg() { # workload, variant
  timeout 300 python3 bench.py --workload $1 --others '' --no-cpu --no-wire --no-d2h --batch-sweep '' --steps 100 --warmup 20 ${2:+--variant $2} --extra $O/extra_$1_${2:-auto}.json > $O/line_$1_${2:-auto}.txt 2>> $O/stderr.txt
  python3 - $O/extra_$1_${2:-auto}.json $1 ${2:-auto} <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']; one=(d.get('one_launch_at_a_time') or {})
    print(f"{sys.argv[2]:34s} variant {str(d['config'].get('kernel_variant')):>3s} ({sys.argv[3]:4s}) kernel {r['kernel_ms']*1e3:8.2f} us  frac {r['frac']:.3f}  one at a time {one.get('kernel_ms',0)*1e3:8.2f} us (variant {one.get('kernel_variant')})  out {d['config'].get('out_bytes_per_frame',0):.0f} B/frame verify {(d.get('verify') or {}).get('byte_identical_to_oracle')}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
}
for rep in 1 2; do
for w in 1080p_80x24_truecolor 1080p_80x24_truecolor_blocks 4k_200x60_truecolor 4k_200x60_truecolor_cool sampled_200x60_truecolor sampled_200x60_truecolor_blocks; do leg $w; done
for w in 1080p_80x24_truecolor_blocks 4k_200x60_truecolor_cool sampled_200x60_truecolor_blocks; do leg $w 4; done
done | tee $O/legs.txt
