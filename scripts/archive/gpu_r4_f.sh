#!/bin/bash
# round 4, visit f: the wave-level closing arithmetic of the frame checksum (exact-length instantiations + stand-alone kernel)
mkdir -p gpurun_out/r4f
python -m pytest tests/test_gpu_parity.py tests/test_crc_wire.py tests/test_hand_kats.py -q -m gpu -k "crc or pack or exact or wire or hand" > gpurun_out/r4f/pytest.txt 2>&1
tail -5 gpurun_out/r4f/pytest.txt
python scripts/gpu_exact_length_timing.py > gpurun_out/r4f/exact_length_timing.txt 2>&1
cat gpurun_out/r4f/exact_length_timing.txt
