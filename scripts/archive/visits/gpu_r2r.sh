#!/bin/bash
# round 2, call r: GPU suite, then the bench line under the driver's flags (with the wire-stage leg)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r_pytest.txt
tail -3 gpurun_out/r_pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r_bench_s20.json 2> gpurun_out/r_bench_s20.err
tail -5 gpurun_out/r_bench_s20.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r_bench_s20.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "in flight", d["config"]["launches_in_flight"])
print("wire", json.dumps(d.get("wire_stage")))
print("serial", json.dumps(d.get("one_launch_at_a_time")))
print("cpu", json.dumps(d.get("cpu_baseline")))
PY
