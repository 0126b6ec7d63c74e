#!/bin/bash
# round 4, visit A: the GPU suite, the driver's exact bench command (stdout captured as the driver sees it), the tick sweep
set -u
O=gpurun_out/r4a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_stdout.txt 2> $O/bench_driver_stderr.txt; echo "bench rc=$?"
cp bench_extra.json $O/bench_extra_driver_flags.json 2>/dev/null
wc -c $O/bench_driver_stdout.txt; cat $O/bench_driver_stdout.txt
for th in 1 2 4 8; do
  ASCIICHAT_HIP_INGEST_THREADS=$th python scripts/gpu_tick_sweep.py > $O/tick_threads_$th.json 2>> $O/tick.err
done
ASCIICHAT_HIP_INGEST_ZERO_COPY=1 python scripts/gpu_tick_sweep.py sampled_images > $O/tick_zero_copy.json 2>> $O/tick.err
ASCIICHAT_HIP_INGEST_SPIN_US=400 python scripts/gpu_tick_sweep.py sampled_images > $O/tick_spin400.json 2>> $O/tick.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4a/tick_*.json')):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], {k:(round(v['frames_per_s']), round(v.get('publish_ms_per_tick',0),3), round(v['ms_per_tick'],3)) for k,v in d.items() if isinstance(v,dict) and 'frames_per_s' in v})
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -5 $O/tick.err
