#!/bin/bash
# round 2 evidence visit on the final build: GPU suite; N = 2 control flow (two gloo ranks sharing the GPU); rocprofv3
# kernel trace + stats of the bench command (4 launches in flight AND one at a time) and of the wire-stage comparison;
# PMC passes for HBM traffic and SQ counters (separate runs, counters only); the full default bench line.
TAG=${1:-r2final}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tee $OUT/pytest_gpu.txt
ASCIICHAT_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu --no-d2h --no-hot --no-wire --others none > $OUT/bench_n2_gloo_shared_gpu.json 2> $OUT/bench_n2.err
echo "N=2 flow: rc=$? stdout lines $(wc -l < $OUT/bench_n2_gloo_shared_gpu.json)" | tee -a $OUT/summary.txt
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --regions 8 --no-cpu --no-d2h --no-hot --no-wire --others none"
cd /tmp
for S in 4 1; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_s$S -o bench -- $BENCH --streams $S > $OUT/bench_under_rocprof_s$S.json 2> $OUT/rocprof_s$S.log
  python $GRAFT_REPO_ROOT/scripts/trace_overlap.py $(find $OUT/trace_s$S -name "*kernel_trace.csv" | head -1) $OUT/trace_overlap_s$S.json > /dev/null
  cp $(find $OUT/trace_s$S -name "*kernel_stats.csv" | head -1) $OUT/bench_s${S}_kernel_stats.csv
  python -c "import json,sys; d=json.load(open('$OUT/bench_under_rocprof_s$S.json')); print('streams', $S, 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'variant', d['config']['kernel_variant'])" | tee -a $OUT/summary.txt
  cat $OUT/trace_overlap_s$S.json | tee -a $OUT/summary.txt
  rm -rf $OUT/trace_s$S
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_wire -o wire -- python $GRAFT_REPO_ROOT/scripts/gpu_fused_crc.py 1080p_80x24_truecolor > $OUT/wire_under_rocprof.txt 2> $OUT/rocprof_wire.log
cp $(find $OUT/trace_wire -name "*kernel_stats.csv" | head -1) $OUT/wire_kernel_stats.csv; rm -rf $OUT/trace_wire
run() { # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- $BENCH --streams 4 > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $GRAFT_REPO_ROOT
python - <<PY | tee $OUT/pmc_summary.txt
import csv, glob, collections
print("# rocprofv3 --pmc passes (separate runs, counters only) over: bench.py --steps 100 --streams 4 (1080p -> 80x24 truecolor, 256 frames per launch)")
for name in ("sq1","sq2","fetch","write"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for row in csv.DictReader(open(f)):
            kn = row["Kernel_Name"]
            if "render_stream_kernel" in kn or "render_frames_kernel" in kn:
                k = (kn.split("(")[0].replace("void achip::",""), row["Counter_Name"]); acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for (kn,k),(v,n) in sorted(acc.items()):
            if n >= 50:
                print(f"{name:6s} {kn[:58]:58s} {k:24s} per-dispatch mean {v/n:16.1f}  (n={n})")
PY
rm -rf $OUT/sq1 $OUT/sq2 $OUT/fetch $OUT/write
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default bench: rc=$? stdout lines $(wc -l < $OUT/bench_default.json)" | tee -a $OUT/summary.txt
python -c "import json; d=json.load(open('$OUT/bench_default.json')); print('default: value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])" | tee -a $OUT/summary.txt
