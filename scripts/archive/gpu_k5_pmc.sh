#!/bin/bash
# SQ counters of the half-block workload (BASELINE configs[4]) for the rows kernel (OR sink / byte sink) and the phase kernel
TAG=${1:-k5pmc}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --workload 4k_400x120_halfblock --steps 10 --warmup 3 --regions 3 --others none --no-cpu --no-d2h --no-hot --no-wire --input-sets 4 --streams 4"
run() { # name lib variant counters...
  local name=$1 lib=$2 variant=$3; shift 3
  ASCIICHAT_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- $BENCH --variant $variant > $OUT/$name.log 2>&1
}
L1=$GRAFT_REPO_ROOT/ascii-chat_amd/libasciichat_hip.so; L2=$GRAFT_REPO_ROOT/ascii-chat_amd/libasciichat_hip_noor.so
for cfg in "rows_or $L1 24" "rows_bytes $L2 24" "phase_or $L1 4"; do
  set -- $cfg
  run $1.sq1 $2 $3 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  run $1.sq2 $2 $3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
done
cd $GRAFT_REPO_ROOT
python - <<PY | tee $OUT/summary.txt
import csv, glob, collections, json
for cfg in ("rows_or","rows_bytes","phase_or"):
    for p in ("sq1","sq2"):
        try:
            d=json.loads(open("$OUT/%s.%s.log"%(cfg,p)).read().strip().splitlines()[-1]); print(cfg,p,"kernel_ms",round(d["roofline"]["kernel_ms"]*1e3,1),"variant",d["config"]["kernel_variant"])
        except Exception as e: print(cfg,p,"no json",e)
        for f in glob.glob("$OUT/%s.%s/**/*counter_collection.csv"%(cfg,p), recursive=True):
            acc=collections.defaultdict(lambda:[0.0,0])
            for row in csv.DictReader(open(f)):
                kn=row["Kernel_Name"]
                if "render_rows_kernel" in kn or "render_frames_kernel" in kn:
                    k=row["Counter_Name"]; acc[k][0]+=float(row["Counter_Value"]); acc[k][1]+=1
            for k,(v,n) in sorted(acc.items()): print(f"  {cfg:10s} {k:24s} per-dispatch mean {v/n:16.1f} (n={n})")
PY
rm -rf $OUT/*.sq1 $OUT/*.sq2
