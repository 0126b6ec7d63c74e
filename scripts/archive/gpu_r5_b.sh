#!/bin/bash
# round 5, visit B: the rows kernel's VALU diet, A/B on one box: round 4's library against the diet's steps (v2: cell records +
# scalar source rows + paired scans + ready-to-store decimal tables; v3: + the scalar-register diet with the flag tests hoisted;
# v4: + flag tests kept in the loop; v4w2: v4 at two waves per SIMD), then the GPU suite on the product library (= v4).
TAG=${1:-r5b}; O=gpurun_out/$TAG; mkdir -p $O
bash scripts/gpu_abn.sh $TAG "lib_r4.so lib_v2.so lib_v3.so lib_v4.so lib_v4w2.so" "sampled_400x240_halfblock 4k_400x120_halfblock 1080p_80x24_halfblock" 2
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -8
