#!/bin/bash
# round 3 evidence visit on the final build: GPU suite; N = 2 control flow through bench.py's OWN rank spawning (two gloo ranks
# sharing the GPU); rocprofv3 kernel trace + stats of the bench command (4 launches in flight AND one at a time), of the
# half-block workload and of all workloads; PMC passes for HBM traffic and SQ counters (separate runs, counters only); the
# full default bench line under the driver's flags and at 100-step regions.
TAG=${1:-r3final}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 > $OUT/host.txt; nproc >> $OUT/host.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tee $OUT/pytest_gpu.log
ASCIICHAT_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu --no-d2h --no-hot --no-wire --others none > $OUT/bench_n2_gloo_shared_gpu.json 2> $OUT/bench_n2.err
echo "N=2 flow (bench.py spawned its own ranks): rc=$? n_gpus=$(python -c "import json;print(json.load(open('$OUT/bench_n2_gloo_shared_gpu.json'))['n_gpus'])")" | tee -a $OUT/summary.txt
python bench.py --gpus 2 --steps 5 > /dev/null 2> $OUT/bench_n2_refused.err; echo "N=2 over RCCL on one device: rc=$? ($(tail -1 $OUT/bench_n2_refused.err))" | tee -a $OUT/summary.txt
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
# the half-block workload (rows kernel) under the kernel trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k5 -o k5 -- $BENCH --workload 4k_400x120_halfblock --steps 10 --regions 3 --input-sets 4 --streams 4 > $OUT/k5_under_rocprof.json 2> $OUT/rocprof_k5.log
cp $(find $OUT/trace_k5 -name "*kernel_stats.csv" | head -1) $OUT/k5_kernel_stats.csv
python $GRAFT_REPO_ROOT/scripts/trace_overlap.py $(find $OUT/trace_k5 -name "*kernel_trace.csv" | head -1) $OUT/trace_overlap_k5.json | tee -a $OUT/summary.txt; rm -rf $OUT/trace_k5
run() { # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- $BENCH --streams 4 > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
K5="$BENCH --workload 4k_400x120_halfblock --steps 10 --regions 3 --input-sets 4"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/k5fetch -o p -- $K5 --streams 4 > $OUT/k5fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/k5write -o p -- $K5 --streams 4 > $OUT/k5write.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY | tee $OUT/pmc_summary.txt
import csv, glob, collections
print("# rocprofv3 --pmc passes (separate runs, counters only) over: bench.py --steps 100 --streams 4 (1080p -> 80x24 truecolor, 256 frames per launch);")
print("# k5fetch / k5write: the same for --workload 4k_400x120_halfblock (rows kernel)")
for name in ("sq1","sq2","fetch","write","k5fetch","k5write"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for row in csv.DictReader(open(f)):
            kn = row["Kernel_Name"]
            if any(k in kn for k in ("render_stream_kernel", "render_frames_kernel", "render_rows_kernel")):
                k = (kn.split("(")[0].replace("void achip::",""), row["Counter_Name"]); acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for (kn,k),(v,n) in sorted(acc.items()):
            if n >= 20:
                print(f"{name:8s} {kn[:58]:58s} {k:24s} per-dispatch mean {v/n:16.1f}  (n={n})")
PY
rm -rf $OUT/sq1 $OUT/sq2 $OUT/fetch $OUT/write $OUT/k5fetch $OUT/k5write
timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
echo "driver-flags bench: rc=$? stdout lines $(wc -l < $OUT/bench_driver_flags.json)" | tee -a $OUT/summary.txt
python -c "import json; d=json.load(open('$OUT/bench_driver_flags.json')); print('driver flags: value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])" | tee -a $OUT/summary.txt
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default bench: rc=$? stdout lines $(wc -l < $OUT/bench_default.json)" | tee -a $OUT/summary.txt
python -c "import json; d=json.load(open('$OUT/bench_default.json')); print('default: value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])" | tee -a $OUT/summary.txt
