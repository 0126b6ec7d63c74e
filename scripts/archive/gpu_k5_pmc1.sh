#!/bin/bash
# one SQ instruction-count pass + the bench line of the half-block workload on the current build
TAG=${1:-k5pmc1}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --workload 4k_400x120_halfblock --steps 10 --warmup 3 --regions 3 --others none --no-cpu --no-d2h --no-hot --no-wire --input-sets 4 --streams 4"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -o p -- $BENCH ${VARIANT:+--variant $VARIANT} > $OUT/sq.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/sq/**/*counter_collection.csv", recursive=True):
    acc=collections.defaultdict(lambda:[0.0,0])
    for row in csv.DictReader(open(f)):
        kn=row["Kernel_Name"]
        if "render_rows_kernel" in kn or "render_frames_kernel" in kn:
            k=row["Counter_Name"]; acc[k][0]+=float(row["Counter_Value"]); acc[k][1]+=1
    for k,(v,n) in sorted(acc.items()): print(f"  {k:24s} per-dispatch mean {v/n/1e6:10.2f} M (n={n})")
PY
rm -rf $OUT/sq
bash scripts/gpu_k5.sh $TAG
