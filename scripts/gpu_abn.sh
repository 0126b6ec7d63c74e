#!/bin/bash
# N-way interleaved A/B of builds of the library on one box: scripts/gpu_abn.sh <tag> "<lib .so ...>" "<workloads ...>" [reps]
# ("HEAD" = the product library).  Prints kernel_ms (HIP events over the timed launches) and ms_per_step per run, and the
# per-library median at the end (HOT=1: also one launch at a time; INPUT=smooth|bars|gray: that synthetic input instead of noise).  Variant builds: scripts/build_variant.sh.
set -u
TAG=$1; LIBS=$2; WL=$3; REPS=${4:-2}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for w in $WL; do
  for rep in $(seq $REPS); do
    for lib in $LIBS; do
      path=""; [ "$lib" != HEAD ] && path=$PWD/ascii-chat_amd/$lib
      ASCIICHAT_HIP_LIB=$path timeout 300 python3 bench.py --workload $w --others '' --no-cpu --no-wire --no-d2h $( [ -n "${HOT:-}" ] || echo --no-hot ) --batch-sweep '' --steps 100 --warmup 20 ${INPUT:+--input $INPUT} \
         --extra $O/extra_${w}_${lib}_$rep.json > $O/line_${w}_${lib}_$rep.txt 2>> $O/stderr.txt
      python3 - "$O/extra_${w}_${lib}_$rep.json" "$w" "$lib" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    one = (d.get('one_launch_at_a_time') or {}).get('kernel_ms')
    print(f"{sys.argv[2]:28s} {sys.argv[3]:22s} kernel_ms {r['kernel_ms']*1e3:9.2f} us  ms_per_step {d['ms_per_step']*1e3:9.2f} us  frac {r['frac']:.4f} variant {d['config'].get('kernel_variant')} in flight {r.get('launches_in_flight')}" + (f" one_at_a_time {one*1e3:.2f} us" if one else "") + f" verify {(d.get('verify') or {}).get('byte_identical_to_oracle')}")
except Exception as e:
    print(f"{sys.argv[2]:28s} {sys.argv[3]:22s} FAILED {e}")
PY
    done
  done
done | tee $O/ab.txt
python3 - $O/ab.txt <<'PY'
import sys,collections,statistics
by=collections.defaultdict(list)
for l in open(sys.argv[1]):
    p=l.split()
    if 'kernel_ms' in p: by[(p[0],p[1])].append(float(p[p.index('kernel_ms')+1]))
    if 'one_at_a_time' in p: by[(p[0]+' [one at a time]',p[1])].append(float(p[p.index('one_at_a_time')+1]))
print("# median kernel time per launch, us")
for (w,lib),v in sorted(by.items()): print(f"{w:44s} {lib:22s} {statistics.median(v):9.2f}  ({len(v)} runs: {' '.join('%.2f'%x for x in v)})")
PY
