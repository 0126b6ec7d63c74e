#!/bin/bash
# round 2, call q: fused frame CRC (parity + cost + timeline)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/q_pytest.txt
timeout 300 python scripts/gpu_stream_timeline.py 1080p_80x24_truecolor 16 crc > gpurun_out/q_timeline.txt 2>&1
timeout 300 python scripts/gpu_fused_crc.py 1080p_80x24_truecolor 1080p_80x24_ansi256 > gpurun_out/q_fused_crc.txt 2>&1
tail -5 gpurun_out/q_pytest.txt; cat gpurun_out/q_timeline.txt gpurun_out/q_fused_crc.txt
