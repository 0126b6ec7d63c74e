#!/bin/bash
# round 5, visit E: the GPU suite on the default build (geometries no plan takes by itself left out) and on the
# -DACHIP_ALL_GEOMETRIES build (lib_all.so: the geometry-equivalence tests force every one), with the world-8 loopback tests;
# then the driver's bench command.
TAG=${1:-r5e}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_default.log 2>&1; echo "pytest default build rc=$?" >> $O/pytest_default.log; grep -E "passed|failed|FAILED|rc=" $O/pytest_default.log | tail -6
ASCIICHAT_HIP_LIB=$PWD/ascii-chat_amd/lib_all.so timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_all_geometries.log 2>&1; echo "pytest all-geometries build rc=$?" >> $O/pytest_all_geometries.log; grep -E "passed|failed|FAILED|rc=" $O/pytest_all_geometries.log | tail -6
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_stdout.txt 2> $O/bench_driver_stderr.txt; echo "bench rc=$?"
cp bench_extra.json $O/bench_extra_driver_flags.json 2>/dev/null
wc -c $O/bench_driver_stdout.txt; head -c 900 $O/bench_driver_stdout.txt; echo; TAG=$TAG python3 - <<'PY'
import json, os
d=json.load(open('gpurun_out/' + os.environ['TAG'] + '/bench_extra_driver_flags.json'))
print({k:(round(v['frames_per_s']),round(v['kernel_ms']*1e3,2),round(v['roofline_frac'],3)) for k,v in d['other_workloads'].items() if isinstance(v,dict) and 'frames_per_s' in v and '+' not in k})
print(d.get('wire_stage',{}).get('render_ms_per_step'), {k:v for k,v in d.get('wire_stage',{}).items() if k.endswith('ms_per_step')})
PY
