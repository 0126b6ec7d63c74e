#!/bin/bash
# round 4, visit E: GPU suite, the driver's bench command on the build with the faster stand-alone frame checksum, drop-in
# calls/s with the frame left in the caller's buffer (DT_INTO=1) next to the malloc'd form
set -u
O=gpurun_out/r4e; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -8
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_stdout.txt 2> $O/bench_driver_stderr.txt; echo "bench rc=$?"
cp bench_extra.json $O/bench_extra_driver_flags.json 2>/dev/null
wc -c $O/bench_driver_stdout.txt; python - <<'PY'
import json
d=json.load(open('gpurun_out/r4e/bench_extra_driver_flags.json'))
print('value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])
w=d.get('wire_stage',{}); print('wire', {k:v for k,v in w.items() if k.endswith('ms_per_step')}, w.get('packed'))
for k,v in d['other_workloads'].items():
    if isinstance(v,dict) and 'wire_stage' in v:
        ww=v['wire_stage']; print(k, {a:b for a,b in ww.items() if a.endswith('ms_per_step')}, {a:b for a,b in ww.get('packed',{}).items() if a.endswith('ms_per_step')})
print('tick', {k:round(v.get('frames_per_s')) for k,v in d.get('tick_e2e',{}).items() if isinstance(v,dict)})
PY
gcc -O2 -I include scripts/dropin_threads.c -o scripts/dropin_threads -L ascii-chat_amd -lasciichat_hip -Wl,-rpath,$PWD/ascii-chat_amd -lpthread
{ echo "# nproc $(nproc); cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
  for T in 1 32 64 128; do for into in 0 1; do
    echo "## T=$T pageable images, DT_INTO=$into"
    DT_INTO=$into DT_MIN_T=$T DT_POOLED=0 timeout 120 ./scripts/dropin_threads $T 2>&1 | grep -v amdgpu.ids
  done; done; } > $O/dropin_into.txt 2>&1
grep -E "^##|calls/s" $O/dropin_into.txt
