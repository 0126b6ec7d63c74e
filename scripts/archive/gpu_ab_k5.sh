#!/bin/bash
# A/B on ONE box: the half-block workload (BASELINE configs[4]) and the metric workload with the in-tree library and with
# every gpurun_tmp/*.so (older builds), alternating, three rounds
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-abk5}; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2 3; do
for lib in ascii-chat_amd/libasciichat_hip.so $(ls gpurun_tmp/*.so); do
  for wl in ${WLS:-4k_400x120_halfblock 1080p_80x24_halfblock 1080p_80x24_truecolor}; do
    sets=12; case $wl in 4k_*) sets=4;; esac
    env $([ $lib = ascii-chat_amd/libasciichat_hip.so ] || echo ASCIICHAT_HIP_LIB=$GRAFT_REPO_ROOT/$lib) timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --regions 3 --others none --no-cpu --no-d2h --no-wire --input-sets $sets --streams 4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$(basename $lib)', '$wl', 'variant', d['config']['kernel_variant'], 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'one_at_a_time_us', round(d['one_launch_at_a_time']['kernel_ms']*1e3,2))" | tee -a $OUT/ab.txt
  done
done
done
