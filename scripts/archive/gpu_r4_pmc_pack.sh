#!/bin/bash
# round 4: HBM traffic of the exact-length launches (separate counter passes, counters only) over scripts/gpu_exact_length_timing.py
TAG=${1:-r4pmc}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/p_$c -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_exact_length_timing.py > $OUT/pmc_$c.log 2>&1 )
done
python - <<PY | tee $OUT/pmc_exact_length_summary.txt
import csv, glob, collections
print("# rocprofv3 --pmc passes (separate runs, counters only) over scripts/gpu_exact_length_timing.py: 256 x 1080p -> 80x24 truecolor per launch;")
print("# KiB per dispatch as the counters report them (FETCH_SIZE x2 on gfx950 per the guide); <..., PACK, PARTS>: PACK 0 = slab, 1 = exact-length frames, 2 = + frame checksum")
for name in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob("$OUT/p_%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for row in csv.DictReader(open(f)):
            kn = row["Kernel_Name"]
            if "achip::" in kn:
                k = (kn.split("(")[0].replace("void achip::",""), row["Counter_Name"]); acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for (kn,k),(v,n) in sorted(acc.items()):
            if n >= 20: print(f"{kn[:70]:70s} {k:12s} per-dispatch mean {v/n:14.1f} KiB (n={n})")
PY
rm -rf $OUT/p_FETCH_SIZE $OUT/p_WRITE_SIZE
