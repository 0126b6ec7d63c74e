#!/bin/bash
# One gpurun visit: GPU parity tests, smoke, bench, rocprofv3 kernel trace + PMC passes.
# Logs land in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -8 > $OUT/rocminfo.txt
nproc > $OUT/host.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/host.txt
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
echo "== bench"; timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/bench.json
echo "== rocprof kernel trace"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-d2h --no-hot --others '' > $OUT/rocprof_run.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do echo "-- $f"; head -6 $f | cut -c1-220; done
python scripts/trace_overlap.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) $OUT/trace_overlap.json
grep -o '"kernel_ms": [0-9.e-]*' $OUT/rocprof_run.log | head -1
echo "== PMC"
bash scripts/pmc_run.sh $TAG/pmc 1080p_80x24_truecolor 2>&1 | tail -20 | tee $OUT/pmc_summary.txt
