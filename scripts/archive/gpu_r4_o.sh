#!/bin/bash
# round 4, visit O: the wire stage behind line-aligned slots (plan stride a multiple of 128) against the previous build
set -u
O=gpurun_out/r4o; mkdir -p $O; export TMPDIR=/tmp
for w in 4k_400x120_halfblock 1080p_80x24_truecolor; do
  for rep in 1 2; do for lib in "" ascii-chat_amd/lib_a16.so; do
    ASCIICHAT_HIP_LIB=${lib:+$PWD/$lib} python3 bench.py --workload $w --others '' --no-cpu --no-d2h --no-hot --steps 100 --warmup 20 --streams 4 \
       --extra $O/x.json > /dev/null 2>> $O/stderr.txt
    python3 - $O/x.json "$w" "${lib:-HEAD}" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']; w=d.get('wire_stage',{})
print(f"{sys.argv[2]:24s} {sys.argv[3]:26s} kernel {r['kernel_ms']*1e3:8.2f} us | wire:", {k:(round(v*1e3,2) if isinstance(v,float) else v) for k,v in w.items() if k.endswith('ms_per_step') or k.endswith('_ms')}, {k:(round(v*1e3,2) if isinstance(v,float) else v) for k,v in (w.get('packed') or {}).items() if k.endswith('ms_per_step')})
PY
  done; done
done | tee $O/wire_ab.txt
