#!/bin/bash
# BASELINE configs[4] (4K -> 400x120 half blocks) as the main workload: headline leg, one launch at a time, wire stage
TAG=${1:-k5}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --workload ${WL:-4k_400x120_halfblock} --steps 10 --warmup 3 --others none --no-cpu --no-d2h --input-sets ${SETS:-4} --streams ${STREAMS:-4} ${VARIANT:+--variant $VARIANT} > $OUT/k5.json 2> $OUT/k5.err; echo rc=$?; tail -3 $OUT/k5.err
python - <<PY
import json
d=json.load(open("$OUT/k5.json"))
print("value", round(d["value"]/1e6,3), "M  ms/step", round(d["ms_per_step"]*1e3,1), "us kernel", round(d["roofline"]["kernel_ms"]*1e3,1), "us frac", round(d["roofline"]["frac"],3), "variant", d["config"]["kernel_variant"], "S", d["config"]["launches_in_flight"])
s=d.get("one_launch_at_a_time"); print("serial", s and {k:(round(v*1e3,1) if "ms" in k else v) for k,v in s.items()})
print("wire", {k:(round(v*1e3,1) if isinstance(v,float) else v) for k,v in (d.get("wire_stage") or {}).items() if "ms" in k or "error" in k or "fused" in k})
PY
