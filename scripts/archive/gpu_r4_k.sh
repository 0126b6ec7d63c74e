#!/bin/bash
# round 4, visit K (session 2, fresh container): the suite, smoke() and the driver's bench command on the rebuilt tree
set -u
O=gpurun_out/r4k; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_stdout.txt 2> $O/bench_driver_stderr.txt; echo "bench rc=$?"
cp bench_extra.json $O/bench_extra_driver_flags.json 2>/dev/null
wc -c $O/bench_driver_stdout.txt; head -c 1500 $O/bench_driver_stdout.txt; echo
