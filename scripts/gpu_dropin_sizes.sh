#!/bin/bash
# round 4, session 2: the drop-in entry point away from 80x24 -- ascii_convert_with_capabilities from 1 and 16 render threads at four terminal sizes,
# truecolor and half blocks, next to the CPU port on the same box (oracle_bench: one thread)
set -u
O=gpurun_out/dropin_sizes; mkdir -p $O
gcc -O2 -I include scripts/dropin_threads.c -o scripts/dropin_threads -L ascii-chat_amd -lasciichat_hip -Wl,-rpath,$PWD/ascii-chat_amd -lpthread
for sz in "80 24" "120 40" "160 45" "200 60"; do for rm in 0 2; do for T in 1 16; do
  DT_MIN_T=$T DT_POOLED=0 timeout 60 ./scripts/dropin_threads $T 1920 1080 $sz 3 $rm 2>&1 | grep "calls/s"
done; done; done | tee $O/dropin_sizes.txt
timeout 200 python3 - <<'PY' | tee -a $O/dropin_sizes.txt
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench
print("# CPU port, the same calls through bench.py's cpu_baseline leg: us per call on one thread, calls/s on the CPUs the box grants")
for (W, H) in ((80, 24), (120, 40), (160, 45), (200, 60)):
    for rm in (0, 2):
        bench.WORKLOADS["_tmp"] = (1920, 1080, W, H, 3, rm)
        r = bench.cpu_baseline("_tmp", budget_s=1.5)
        print(f"1920x1080 -> {W}x{H} colour 3 mode {rm}: {1e6 / r['value']:8.1f} us per call on one thread; {r['all_cores']['value']:9.0f} calls/s on {r['all_cores']['cores']} threads")
PY
