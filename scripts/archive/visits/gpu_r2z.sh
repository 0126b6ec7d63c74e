#!/bin/bash
# round 2, call z: is the stream kernel instruction-bound on large frames?  SQ counters for 4K -> 200x60 truecolor (K3)
# next to the metric's workload; N = 2 control flow after the bench timing change
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/z && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/z
ASCIICHAT_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \
  bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu --no-d2h --no-hot --no-wire --others none > $OUT/bench_n2.json 2> $OUT/bench_n2.err
echo "N=2 flow: rc=$? lines $(wc -l < $OUT/bench_n2.json)"; python -c "import json; d=json.load(open('$OUT/bench_n2.json')); print(d['n_gpus'], d['value'], d['ms_per_step'])"
cd /tmp
for W in 4k_200x60_truecolor 1080p_80x24_truecolor; do
  BENCH="python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 40 --warmup 5 --regions 6 --no-cpu --no-d2h --no-hot --no-wire --others none --streams 4"
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/sq_$W -o p -- $BENCH > $OUT/sq_$W.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_I8 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq2_$W -o p -- $BENCH > $OUT/sq2_$W.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY | tee $OUT/summary.txt
import csv, glob, collections
for name in sorted(glob.glob("$OUT/sq*_*/")):
    for f in glob.glob(name + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for row in csv.DictReader(open(f)):
            kn = row["Kernel_Name"]
            if "render_stream_kernel" in kn:
                k = row["Counter_Name"]; acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        print(name.split("/")[-2])
        for k,(v,n) in sorted(acc.items()):
            print(f"   {k:28s} per-dispatch mean {v/n:16.1f}  (n={n})")
PY
rm -rf $OUT/sq_* $OUT/sq2_*
