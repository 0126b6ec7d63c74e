#!/bin/bash
# round 4, visit C: GPU suite (exact-length frames, hand KATs, N = 2 bench), the driver's bench command, the small-batch
# geometry sweep after the policy change, drop-in calls/s with the threads confined by default
set -u
O=gpurun_out/r4c; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_stdout.txt 2> $O/bench_driver_stderr.txt; echo "bench rc=$?"
cp bench_extra.json $O/bench_extra_driver_flags.json 2>/dev/null
wc -c $O/bench_driver_stdout.txt; python - <<'PY'
import json
d=json.load(open('gpurun_out/r4c/bench_extra_driver_flags.json'))
print('value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])
print('wire_stage', json.dumps(d.get('wire_stage'))[:1500])
print('with_d2h_packed', d.get('with_d2h_packed'))
print('tick', {k:v.get('frames_per_s') for k,v in d.get('tick_e2e',{}).items() if isinstance(v,dict)})
for k,v in d['other_workloads'].items():
    if 'frames_per_s' in v: print(k, round(v['frames_per_s']), round(v['kernel_ms']*1e3,2), 'us', v.get('kernel_variant'), (v.get('one_launch_at_a_time') or {}).get('kernel_ms'))
PY
python scripts/gpu_small_batch_variants.py 2>&1 | grep -v amdgpu.ids > $O/small_batch_variants.txt; cat $O/small_batch_variants.txt
gcc -O2 -I include scripts/dropin_threads.c -o scripts/dropin_threads -L ascii-chat_amd -lasciichat_hip -Wl,-rpath,$PWD/ascii-chat_amd -lpthread
{ echo "# nproc $(nproc); cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
  for T in 1 4 16 32 64 128; do for pooled in 0 1; do
    echo "## default environment (threads confined by the library when a quota is below the mask), T=$T pooled=$pooled"
    DT_MIN_T=$T DT_POOLED=$pooled timeout 120 ./scripts/dropin_threads $T 2>&1 | grep -v amdgpu.ids
  done; done
  for T in 64 128; do
    echo "## ASCIICHAT_HIP_CONFINE=0 (roaming), T=$T pooled=0"
    ASCIICHAT_HIP_CONFINE=0 DT_MIN_T=$T DT_POOLED=0 timeout 120 ./scripts/dropin_threads $T 2>&1 | grep -v amdgpu.ids
  done; } > $O/dropin_threads.txt 2>&1
grep -E "^##|calls/s" $O/dropin_threads.txt
