#!/bin/bash
# round 2, call v: after the gfx950_ops split -- GPU suite, then the bench line under the driver's flags
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/v_pytest.txt
tail -3 gpurun_out/v_pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 --others none > gpurun_out/v_bench_s20.json 2> gpurun_out/v_bench_s20.err
echo "rc=$? stdout lines: $(wc -l < gpurun_out/v_bench_s20.json)"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/v_bench_s20.json").read())
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"], "in flight", d["config"]["launches_in_flight"])
print("tune", d["timing"].get("launches_in_flight_autotune_ms_per_step"))
print("wire", {k: v for k, v in d.get("wire_stage", {}).items() if "ms" in k})
print("serial", d.get("one_launch_at_a_time"))
PY
