#!/bin/bash
# round 2, call x: header + packet CRC written by the render launch (plan_render_packets): parity + cost
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 | tee gpurun_out/x_pytest.txt
timeout 300 python scripts/gpu_fused_crc.py 1080p_80x24_truecolor > gpurun_out/x_fused_crc.txt 2>&1
grep -v "c-only" gpurun_out/x_fused_crc.txt | grep "round 1\|geometry\|==" 
timeout 600 python scripts/gpu_soak.py --rounds 40 --seed 31 2>&1 | tail -2
