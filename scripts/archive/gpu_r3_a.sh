#!/bin/bash
# round 3, visit A: GPU suite (incl. the two-rank comm.c test over the loopback transport, pack / publish_rows / direct
# grid tests) and the full default bench line with the new legs (with_d2h_packed, tick_e2e, grid9 direct).
TAG=${1:-r3a}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench.err
echo "bench rc=$? lines $(wc -l < $OUT/bench_driver_flags.json)"; tail -5 $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench_driver_flags.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
print("serial", d.get("one_launch_at_a_time"))
print("d2h", d.get("with_d2h")); print("d2h_packed", d.get("with_d2h_packed")); print("tick", json.dumps(d.get("tick_e2e"), indent=1))
for k,v in d["other_workloads"].items():
    if "frames_per_s" in v: print(k, round(v["frames_per_s"]/1e6,2), "M", round(v["kernel_ms"]*1e3,1), "us frac", round(v["roofline_frac"],3), v.get("with_d2h_packed",{}).get("frames_per_s"))
    else: print(k, json.dumps({a:{b:c for b,c in x.items() if b in ("frames_per_s","ms_per_step","kernel_ms","roofline_frac","kernel_variant","bands_per_frame")} for a,x in v.items()}, indent=1))
PY
