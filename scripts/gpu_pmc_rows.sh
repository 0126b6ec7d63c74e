#!/bin/bash
# SQ counter passes of the rows kernel (rocprofv3 --pmc with --kernel-trace only), per-dispatch means:
#   [KERNEL=render_stream_kernel] scripts/gpu_pmc_rows.sh <tag> <workload> [lib .so under ascii-chat_amd/ | HEAD]
TAG=${1:-pmcrows}; WL=${2:-sampled_400x240_halfblock}; LIB=${3:-HEAD}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
[ "$LIB" != HEAD ] && export ASCIICHAT_HIP_LIB=$GRAFT_REPO_ROOT/ascii-chat_amd/$LIB
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --regions 3 --no-cpu --no-d2h --no-hot --no-wire --others '' --batch-sweep '' --extra '' --workload $WL > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
run sq4 SQ_INSTS_CBRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_SENDMSG SQ_WAIT_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES
cd $GRAFT_REPO_ROOT
python - <<PY | tee $OUT/summary.txt
import csv, glob, collections
print("# $WL, library $LIB: per-dispatch means of ${KERNEL:-render_rows_kernel}")
for name in ("sq1","sq2","sq3","sq4"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True)
    if not fs:
        print(name, "no counters collected:", open("$OUT/%s.log" % name).read()[-300:].replace("\n", " | "))
    for f in fs:
        acc = collections.defaultdict(lambda: [0.0,0])
        for row in csv.DictReader(open(f)):
            if "${KERNEL:-render_rows_kernel}" in row["Kernel_Name"]:
                k = row["Counter_Name"]; acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for k,(v,n) in sorted(acc.items()):
            print(f"{name:4s} {k:28s} {v/n:16.1f}  (n={n})")
PY
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/sq4
