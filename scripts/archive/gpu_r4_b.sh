#!/bin/bash
# round 4, visit B: new GPU tests (hand KATs, sampled-image ingest), the chunked wire experiment, then the profile visit
set -u
O=gpurun_out/r4b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python scripts/gpu_k5_chunked_wire.py 4k_400x120_halfblock 4 > $O/chunked_wire_k5_s4.txt 2> $O/chunked.err; cat $O/chunked_wire_k5_s4.txt
python scripts/gpu_k5_chunked_wire.py 4k_400x120_halfblock 1 > $O/chunked_wire_k5_s1.txt 2>> $O/chunked.err; cat $O/chunked_wire_k5_s1.txt
python scripts/gpu_k5_chunked_wire.py 4k_200x60_truecolor 4 > $O/chunked_wire_k3_s4.txt 2>> $O/chunked.err; cat $O/chunked_wire_k3_s4.txt
tail -3 $O/chunked.err
bash scripts/gpu_r4_profiles.sh r4b_prof 2>&1 | tail -60
