#!/bin/bash
# round-end evidence: gpu_check (tests, smoke, bench, rocprof, PMC), the N=2 control flow on one GPU (gloo),
# drop-in latency, all-workload kernel stats
cd $GRAFT_REPO_ROOT
bash scripts/gpu_check.sh ${1:-final}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-final}
echo "== N=2 control flow (two ranks on one GPU, gloo)"
ASCIICHAT_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/bench_n2_gloo.json
echo "== drop-in latency"
timeout 600 python scripts/gpu_tune.py --dropin 2>&1 | grep -v amdgpu.ids | tee $OUT/dropin_latency.txt
echo "== all workloads kernel stats"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_all -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-d2h --no-hot > $OUT/rocprof_all.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/prof_all -name "*kernel_stats.csv"); do head -12 $f | cut -c1-200; done
