#!/bin/bash
# round 2, call zh: GPU suite on the final sources, then the two bench lines (driver flags; default) for profiles/
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/zh && export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tee gpurun_out/zh/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a gpurun_out/zh/pytest_gpu.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/zh/bench_driver_flags.json 2> gpurun_out/zh/bench_driver_flags.err
echo "driver flags: rc=$? lines $(wc -l < gpurun_out/zh/bench_driver_flags.json)"
timeout 900 python bench.py > gpurun_out/zh/bench_default.json 2> gpurun_out/zh/bench_default.err
echo "default: rc=$? lines $(wc -l < gpurun_out/zh/bench_default.json)"
python - <<'PY'
import json
for f in ("bench_driver_flags", "bench_default"):
    d = json.load(open("gpurun_out/zh/%s.json" % f))
    print(f, "value", d["value"], "ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"],
          "wire+", d["wire_stage"].get("extra_ms_fused"), "grid9", d["other_workloads"]["grid9_1080p_160x48_truecolor"]["256_targets"]["ms_per_step"])
PY
