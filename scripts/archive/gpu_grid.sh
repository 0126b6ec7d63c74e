#!/bin/bash
# the grid workload alone (BASELINE configs[3]) + the GPU tests that touch composites
TAG=${1:-grid}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "grid or composite or pack or publish_rows" 2>&1 | tail -4
timeout 600 python bench.py --workload grid9 --steps 40 > $OUT/grid.json 2> $OUT/grid.err; echo rc=$?
python - <<PY
import json
d=json.load(open("$OUT/grid.json"))
for a,x in d["grid9"].items(): print(a, {b:c for b,c in x.items() if b in ("frames_per_s","ms_per_step","kernel_ms","roofline_frac","kernel_variant","bands_per_frame")})
PY
