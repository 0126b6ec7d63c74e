#!/bin/bash
# round 6, visit G: length-first exact-length frames as a product instantiation: GPU tests (wire / pack / exact-length / the new one),
# then the A/B against render + pack pass on the product library.
TAG=${1:-r6g}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -k "length_first or exact_length or packed or wire or crc or stream or multibyte" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 900 python3 scripts/gpu_length_first_ab.py > $O/length_first.txt 2> $O/stderr.txt; cat $O/length_first.txt; tail -3 $O/stderr.txt
