#!/bin/bash
# round 3, last evidence visit (the render kernels are those of gpu_r3_final.sh's traces and counters; what changed since is
# host code: drop-in layer, ingest, bench legs): GPU suite, smoke(), bench.py --gpus 2 spawning its own ranks (gloo on the
# shared GPU; RCCL must refuse), the bench line under the driver's flags and the default line
TAG=${1:-r3final2}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tee $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $OUT/summary.txt
ASCIICHAT_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu --no-d2h --no-hot --no-wire --others none > $OUT/bench_n2_gloo_shared_gpu.json 2> $OUT/bench_n2.err
echo "N=2 flow (bench.py spawned its own ranks): rc=$? $(python -c "import json;d=json.load(open('$OUT/bench_n2_gloo_shared_gpu.json'));print('n_gpus', d['n_gpus'], 'multi_gpu keys', sorted(d['multi_gpu']))")" | tee -a $OUT/summary.txt
python bench.py --gpus 2 --steps 5 > /dev/null 2> $OUT/bench_n2_refused.err; echo "N=2 over RCCL on one device: rc=$? ($(tail -1 $OUT/bench_n2_refused.err))" | tee -a $OUT/summary.txt
t0=$(date +%s); timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
echo "driver-flags bench: rc=$? stdout lines $(wc -l < $OUT/bench_driver_flags.json), $(( $(date +%s) - t0 )) s" | tee -a $OUT/summary.txt
python -c "import json; d=json.load(open('$OUT/bench_driver_flags.json')); print('driver flags: value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['all_cores'], 'tick', {k: round(v['frames_per_s']) for k, v in d['tick_e2e'].items() if isinstance(v, dict)})" | tee -a $OUT/summary.txt
t0=$(date +%s); timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default bench: rc=$? stdout lines $(wc -l < $OUT/bench_default.json), $(( $(date +%s) - t0 )) s" | tee -a $OUT/summary.txt
python -c "import json; d=json.load(open('$OUT/bench_default.json')); print('default: value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])" | tee -a $OUT/summary.txt
