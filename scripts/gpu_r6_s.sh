#!/bin/bash
# round 6, visit S: full-frame sources in UNCACHED device memory (hipExtMallocWithFlags 3) against ordinary hipMalloc memory:
# does a 3-byte sample then cost a 32- / 64-byte request instead of a 128-byte line fill?  Times + FETCH_SIZE passes.
TAG=${1:-r6s}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for w in 1080p_80x24_truecolor 4k_200x60_truecolor; do NSETS=$([ $w = 4k_200x60_truecolor ] && echo 4 || echo 12) timeout 900 python3 scripts/gpu_uncached_sources.py $w 0 3 1 0 3 2>> $O/stderr.txt; done | tee $O/times.txt
NSETS=4 timeout 900 python3 scripts/gpu_uncached_sources.py 4k_400x120_halfblock 0 3 0 3 2>> $O/stderr.txt | tee -a $O/times.txt
for flag in 0 3; do
  ( cd /tmp && ONLY_FLAG=$flag REPS=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/p_$flag -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_uncached_sources.py 1080p_80x24_truecolor > $GRAFT_REPO_ROOT/$O/pmc_$flag.log 2>&1 )
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/p_$flag/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "achip::render" in row["Kernel_Name"]:
            k = (row["Kernel_Name"].split("(")[0][-60:], row["Counter_Name"]); acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
for (kn, c), (v, n) in sorted(acc.items()):
    print(f"flag $flag {kn:60s} {c:12s} per-dispatch mean {v/n:14.1f}  (n={n})")
PY
  rm -rf $O/p_$flag
done | tee $O/fetch.txt
tail -3 $O/stderr.txt
