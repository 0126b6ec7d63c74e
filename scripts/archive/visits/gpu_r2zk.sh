#!/bin/bash
# round 2, call zk: HBM traffic of the large-frame workloads (separate --pmc passes, counters only): is anything fetched twice?
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/zk && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/zk
cd /tmp
for W in 4k_200x60_truecolor 4k_400x120_halfblock; do
  BENCH="python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 20 --warmup 5 --regions 5 --input-sets 4 --no-cpu --no-d2h --no-hot --no-wire --others none --streams 4"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${C}_$W -o p -- $BENCH > $OUT/${C}_$W.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python - <<PY | tee $OUT/summary.txt
import csv, glob, collections
for name in sorted(glob.glob("$OUT/*_4k_*/")):
    for f in glob.glob(name + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for row in csv.DictReader(open(f)):
            kn = row["Kernel_Name"]
            if "render_stream_kernel" in kn or "render_frames_kernel" in kn:
                k = (kn.split("(")[0].replace("void achip::","")[:60], row["Counter_Name"]); acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for (kn,k),(v,n) in sorted(acc.items()):
            if n >= 20: print(f"{name.split('/')[-2]:40s} {kn:60s} {k:12s} per-dispatch mean {v/n:14.1f} KiB (n={n})")
PY
rm -rf $OUT/FETCH_SIZE_* $OUT/WRITE_SIZE_*
