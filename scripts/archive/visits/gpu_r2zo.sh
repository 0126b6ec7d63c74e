#!/bin/bash
# round 2, call zo: SQ / LDS counters of the stream kernel's CRC instantiation next to the plain one (counters only)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/zo; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/p -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_fused_crc.py 1080p_80x24_truecolor > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY | tee $OUT/summary.txt
import csv, glob, collections
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: [0.0,0])
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"]
        if "render_stream_kernel" in kn or "crc32c_frame_kernel" in kn:
            k = (kn.split("(")[0].replace("void achip::","")[:56], row["Counter_Name"]); acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    for (kn,k),(v,n) in sorted(acc.items()):
        if n >= 50: print(f"{kn:56s} {k:22s} per-dispatch mean {v/n:14.1f} (n={n})")
PY
rm -rf $OUT/p
