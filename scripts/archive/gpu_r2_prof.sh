#!/bin/bash
# round 2 evidence visit: rocprofv3 kernel trace + stats of the bench command (4 launches in flight AND one at a time),
# PMC passes for HBM traffic and SQ counters (separate runs, counters only), per-wave timeline, stream sweep.
TAG=${1:-r2prof}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --regions 8 --no-cpu --no-d2h --no-hot --others none"
cd /tmp
for S in 4 1; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_s$S -o bench -- $BENCH --streams $S > $OUT/rocprof_s$S.log 2>&1
  python $GRAFT_REPO_ROOT/scripts/trace_overlap.py $(find $OUT/trace_s$S -name "*kernel_trace.csv" | head -1) $OUT/trace_overlap_s$S.json > /dev/null
  cp $(find $OUT/trace_s$S -name "*kernel_stats.csv" | head -1) $OUT/bench_s${S}_kernel_stats.csv
  grep -o '{"metric.*' $OUT/rocprof_s$S.log > $OUT/bench_under_rocprof_s$S.json
  python -c "import json,sys; d=json.load(open('$OUT/bench_under_rocprof_s$S.json')); print('streams', $S, 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'variant', d['config']['kernel_variant'])" | tee -a $OUT/summary.txt
  cat $OUT/trace_overlap_s$S.json | tee -a $OUT/summary.txt
  rm -rf $OUT/trace_s$S
done
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
for v in 16 17; do timeout 120 python scripts/gpu_stream_timeline.py 1080p_80x24_truecolor $v 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stream_timeline.txt; done
OVERLAP_VARIANTS=16,17,1,4 OVERLAP_STREAMS=1,2,3,4,6 timeout 300 python scripts/gpu_overlap.py 1080p_80x24_truecolor 2>&1 | grep -v amdgpu.ids | tee $OUT/stream_sweep.txt
OVERLAP_VARIANTS=17,16,1,4 OVERLAP_STREAMS=1,4 timeout 300 python scripts/gpu_overlap.py 4k_200x60_truecolor 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stream_sweep.txt
OVERLAP_VARIANTS=17,16,1 OVERLAP_STREAMS=1,4 timeout 300 python scripts/gpu_overlap.py 1080p_80x24_ansi256 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stream_sweep.txt
timeout 120 python scripts/gpu_region_overhead.py 4 2>&1 | grep -v amdgpu.ids | tee $OUT/region_overhead.txt
