#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2l
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/pytest.log
timeout 600 python bench.py --workload grid9 --steps 20 > $OUT/grid9.json 2> $OUT/grid9.err; tail -3 $OUT/grid9.err; cut -c1-900 $OUT/grid9.json
