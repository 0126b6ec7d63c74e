#!/bin/bash
# reduced evidence visit: GPU suite, bench.py, rocprofv3 kernel trace of the same command + overlap summary (no PMC passes)
TAG=${1:-bp}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -q --timeout 200 -p no:cacheprovider 2>&1 | tail -2 | tee $OUT/pytest_gpu.log
timeout 300 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/bench.json; cut -c1-400 $OUT/bench.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-d2h --no-hot --others '' > $OUT/rocprof_run.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/trace_overlap.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) $OUT/trace_overlap.json
grep -o '"kernel_ms": [0-9.e-]*' $OUT/rocprof_run.log | head -1
rm -f $(find $OUT/prof -name "*kernel_trace.csv")  # large; the stats and the overlap summary are what is kept
