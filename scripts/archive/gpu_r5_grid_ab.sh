#!/bin/bash
# A/B of the grid workload (configs[3]) between two builds: scripts/gpu_r5_grid_ab.sh <tag> "<libs>" -- every leg of bench.py's grid9
# (nine targets / 256 targets, tiles + all-gather / direct), two runs each, interleaved
TAG=$1; LIBS=$2; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do for lib in $LIBS; do
  path=""; [ "$lib" != HEAD ] && path=$PWD/ascii-chat_amd/$lib
  ASCIICHAT_HIP_LIB=$path timeout 300 python3 bench.py --workload grid9 --steps 100 --warmup 5 --extra $O/grid_${lib}_$rep.json > $O/grid_${lib}_$rep.txt 2>> $O/stderr.txt
  python3 - $O/grid_${lib}_$rep.json $lib <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
g=d.get('grid9',{})
print(f"{sys.argv[2]:16s}", ' | '.join(f"{k}: {v.get('kernel_ms',0)*1e3:6.2f} us kernel, {v.get('ms_per_step',0)*1e3:6.2f} us step" for k,v in g.items() if isinstance(v,dict) and 'ms_per_step' in v))
PY
done; done | tee $O/grid_ab.txt
