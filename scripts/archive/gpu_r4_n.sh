#!/bin/bash
# round 4, visit N: the traffic floor with line-aligned shares (ubench), the GPU suite with the line-phase test, the phase kernel A/B
set -u
O=gpurun_out/r4n; mkdir -p $O; export TMPDIR=/tmp
./scripts/ubench/rows_floor 4 2>&1 | grep -v amdgpu.ids > $O/rows_floor.txt; head -18 $O/rows_floor.txt
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -3
for w in 4k_400x120_halfblock 4k_200x60_truecolor; do
  for rep in 1 2; do for lib in "" ascii-chat_amd/lib_a16.so; do
    v=4; [ $w = 4k_200x60_truecolor ] && v=1
    ASCIICHAT_HIP_LIB=${lib:+$PWD/$lib} python3 bench.py --workload $w --others '' --no-cpu --no-wire --no-d2h --no-hot --steps 200 --warmup 20 --streams 4 --variant $v \
       --extra $O/x.json > /dev/null 2>> $O/stderr.txt
    python3 - $O/x.json "$w" "${lib:-HEAD}" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(f"{sys.argv[2]:28s} {sys.argv[3]:28s} kernel_ms {r['kernel_ms']*1e3:9.2f} us  ms_per_step {d['ms_per_step']*1e3:9.2f} us  frac {r['frac']:.4f} variant {d['config'].get('kernel_variant')} in flight {r.get('launches_in_flight')}")
PY
  done; done
done | tee $O/ab_phase.txt
