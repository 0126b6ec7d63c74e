#!/bin/bash
# the end-to-end tick leg alone + the frame-table tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
K="frame_table or publish" bash scripts/gpu_pytest.sh ${1:-tick} | tail -3
python - <<PY
import json, sys, torch
sys.path.insert(0, ".")
import bench
from __graft_entry__ import load_package
pkg = load_package(); torch.cuda.set_device(0)
print(json.dumps(bench.tick_e2e(torch, pkg), indent=1))
PY
