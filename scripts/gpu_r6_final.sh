#!/bin/bash
# round 6, the last visit: what the driver will run, on the round's last tree -- the GPU suite, smoke(), bench.py with the driver's flags
cd $GRAFT_REPO_ROOT; TAG=${1:-r6final}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log; grep -n "passed\|failed" $O/gputests.log | tail -n 2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 1 | tee $O/smoke.txt
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --extra $O/bench_extra.json ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt; tail -c 1500 $O/bench_stdout.txt; tail -n 4 $O/bench_stderr.txt
