#!/bin/bash
# quick visit: GPU suite, headline line (driver flags, no side legs), grid workload
TAG=${1:-q}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 --others "${OTHERS:-none}" --no-cpu --no-d2h > $OUT/bench.json 2> $OUT/bench.err; echo rc=$?
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value", round(d["value"]/1e6,2), "M  ms/step", round(d["ms_per_step"]*1e3,2), "us kernel", round(d["roofline"]["kernel_ms"]*1e3,2), "us frac", round(d["roofline"]["frac"],3), "variant", d["config"]["kernel_variant"], "S", d["config"]["launches_in_flight"])
s=d.get("one_launch_at_a_time"); print("serial", s and {k:(round(v*1e3,2) if "ms" in k else v) for k,v in s.items()})
print("wire", {k:v for k,v in (d.get("wire_stage") or {}).items() if "ms" in k})
for k,v in (d.get("other_workloads") or {}).items():
    if "frames_per_s" in v: print(k, round(v["frames_per_s"]/1e6,2), "M", round(v["kernel_ms"]*1e3,1), "us frac", round(v["roofline_frac"],3), "serial", v.get("one_launch_at_a_time",{}).get("kernel_ms"))
PY
timeout 600 python bench.py --workload grid9 --steps 40 > $OUT/grid.json 2> $OUT/grid.err; echo rc=$?
python - <<PY
import json
d=json.load(open("$OUT/grid.json"))
for a,x in d["grid9"].items(): print(a, {b:(round(c*1e3,2) if "ms" in b else c) for b,c in x.items() if b in ("ms_per_step","kernel_ms","roofline_frac","kernel_variant","bands_per_frame")})
PY
