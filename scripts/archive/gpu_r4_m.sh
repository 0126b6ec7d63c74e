#!/bin/bash
# round 4, visit M: whole-line drains (128-byte windows in the stream and rows kernels, partial lines carried across the slices
# of a block in the rows kernel): suite + soak, then A/B against the 16-byte-aligned build (lib_a16.so: EXTRA=-DACHIP_DRAIN_ALIGN=16u)
set -u
O=gpurun_out/r4m; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -3
timeout 600 python scripts/gpu_soak.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/soak.txt
for w in 4k_400x120_halfblock 1080p_80x24_halfblock 640x480_80x24_mono 4k_200x60_truecolor; do
  for rep in 1 2 3; do for lib in "" ascii-chat_amd/lib_a16.so; do
    ASCIICHAT_HIP_LIB=${lib:+$PWD/$lib} python3 bench.py --workload $w --others '' --no-cpu --no-wire --no-d2h --no-hot --steps 200 --warmup 20 --streams 4 \
       --extra $O/x.json > /dev/null 2>> $O/stderr.txt
    python3 - $O/x.json "$w" "${lib:-HEAD}" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(f"{sys.argv[2]:28s} {sys.argv[3]:28s} kernel_ms {r['kernel_ms']*1e3:9.2f} us  ms_per_step {d['ms_per_step']*1e3:9.2f} us  frac {r['frac']:.4f} variant {d['config'].get('kernel_variant')} in flight {r.get('launches_in_flight')}")
PY
  done; done
done | tee $O/ab.txt
