#!/bin/bash
# round 4, visit P: pack_frames with line-aligned stores + plan strides of whole lines: suite, then the PCIe-inclusive and wire-stage legs A/B
set -u
O=gpurun_out/r4p; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -3
for w in 1080p_80x24_truecolor 4k_400x120_halfblock; do
  for rep in 1 2; do for lib in "" ascii-chat_amd/lib_a16.so; do
    ASCIICHAT_HIP_LIB=${lib:+$PWD/$lib} python3 bench.py --workload $w --others '' --no-cpu --no-hot --steps 100 --warmup 20 --streams 4 \
       --extra $O/x.json > /dev/null 2>> $O/stderr.txt
    python3 - $O/x.json "$w" "${lib:-HEAD}" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']; w=d.get('wire_stage',{})
p=(w.get('packed') or {})
print(f"{sys.argv[2]:24s} {sys.argv[3]:26s} kernel {r['kernel_ms']*1e3:8.2f} us | d2h_packed {d.get('with_d2h_packed',{}).get('frames_per_s',0)/1e6:.3f} M fps | wire_then_pack {p.get('wire_stage_then_pack_frames_ms_per_step',0)*1e3:.2f} packed {p.get('render_packets_packed_ms_per_step',0)*1e3:.2f}")
PY
  done; done
done | tee $O/ab.txt
