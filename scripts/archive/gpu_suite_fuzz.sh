#!/bin/bash
# full GPU parity suite + a short drop-in fuzz (used after a change to the exported entry points)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/suite
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/suite/pytest.log
timeout 400 python scripts/gpu_dropin_fuzz.py 77 ${1:-4000} 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/suite/fuzz.log
