#!/bin/bash
# round 2, call ze: the grid path's tile resizes as ONE launch per tick: parity (RCCL world-1 grid tests) and the grid9 workload
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "grid or rccl or composite or tick" 2>&1 | grep -E "passed|failed|rror" | tee gpurun_out/ze_pytest.txt
timeout 600 python bench.py --workload grid9 --steps 20 --warmup 5 > gpurun_out/ze_grid9.json 2> gpurun_out/ze_grid9.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/ze_grid9.json").read())
print("grid9:", d["value"], "frames/s", d["ms_per_step"] * 1e3, "us/step, kernel", d["roofline"]["kernel_ms"] * 1e3, "frac", d["roofline"]["frac"])
for k, v in d["grid9"].items():
    print(k, {kk: v[kk] for kk in ("frames_per_s", "ms_per_step", "kernel_ms", "kernel_variant", "bands_per_frame")})
PY
