#!/bin/bash
# round 5, visit C: v5 (samples turned into pixels mid-turn: no wait on fresh requests / the drain's stores) against v4 and round 4,
# then SQ counters of the rows kernel on the product library (= v5), sampled 400x240 half blocks
TAG=${1:-r5c}; O=gpurun_out/$TAG; mkdir -p $O
bash scripts/gpu_abn.sh $TAG "lib_r4.so lib_v4.so lib_v5.so" "sampled_400x240_halfblock 4k_400x120_halfblock" 2
bash scripts/gpu_pmc_rows.sh ${TAG}_pmc sampled_400x240_halfblock HEAD
