#!/bin/bash
# the one-pass checksum + pack: its GPU tests, then the wire_stage leg (with the packed forms) behind the metric workload and the
# two half-block workloads
cd $GRAFT_REPO_ROOT; TAG=${1:-wirepack}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
K="packed or packets or crc or pack" bash scripts/gpu_pytest.sh $TAG | grep -E "passed|failed|Error|assert" | tail -5
for wl in 1080p_80x24_truecolor 1080p_80x24_halfblock 4k_400x120_halfblock; do
  sets=12; [ $wl = 4k_400x120_halfblock ] && sets=4
  timeout 600 python bench.py --workload $wl --steps 20 --warmup 5 --regions 3 --others none --no-cpu --no-d2h --no-hot --input-sets $sets --streams 4 > $OUT/$wl.json 2> $OUT/$wl.err
  python -c "
import json; d=json.load(open('$OUT/$wl.json')); w=d['wire_stage']
print('$wl', 'variant', w['kernel_variant'], 'fused', w['fused_crc_in_render_kernel'], 'render', round(w['render_ms_per_step']*1e3,1), 'separate', round(w['render_plus_packet_kernel_ms_per_step']*1e3,1), 'fused', round(w['render_with_fused_crc_and_headers_ms_per_step']*1e3,1), 'then_pack', round(w['packed']['wire_stage_then_pack_frames_ms_per_step']*1e3,1), 'packed', round(w['packed']['render_packets_packed_ms_per_step']*1e3,1))" | tee -a $OUT/summary.txt
done
