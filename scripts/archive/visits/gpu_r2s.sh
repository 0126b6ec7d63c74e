#!/bin/bash
# round 2, call s: bench line hygiene (stdout = one JSON line) + wire-stage leg at the main leg's geometry
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 --others none > gpurun_out/s_bench.json 2> gpurun_out/s_bench.err
echo "rc=$? stdout lines: $(wc -l < gpurun_out/s_bench.json)"
tail -3 gpurun_out/s_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s_bench.json").read())
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "in flight", d["config"]["launches_in_flight"])
print("wire", json.dumps(d.get("wire_stage")))
print("others", list(d.get("other_workloads", {}).keys()))
PY
