#!/bin/bash
# rocprofv3 PMC passes over the bench workload (counters only; no trace domains beyond kernel-trace).
TAG=${1:-pmc}; shift
WL=${1:-1080p_80x24_truecolor}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-d2h --no-hot --others '' --workload $WL > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections, json
traffic = {}
for name in ("sq1","sq2","fetch","write"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for row in csv.DictReader(open(f)):
            if "render_frames_kernel" in row["Kernel_Name"]:
                k = row["Counter_Name"]; acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for k,(v,n) in sorted(acc.items()):
            print(f"{name:6s} {k:28s} per-dispatch mean {v/n:16.1f}  (n={n})")
            if k in ("FETCH_SIZE", "WRITE_SIZE"):
                traffic[k] = (v / n, n)
if len(traffic) == 2:  # what bench.py reads as profiles/pmc_traffic.json (copy it there together with the summary)
    import re
    m = re.search(r'"kernel_variant": (\d+)', open("$OUT/fetch.log").read())
    json.dump({"$WL": {"variant": int(m.group(1)) if m else 4, "batch": 256, "fetch_size_kb": round(traffic["FETCH_SIZE"][0], 1),
                       "write_size_kb": round(traffic["WRITE_SIZE"][0], 1),
                       "source": "profiles/r01_pmc_summary_$WL.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, "
                                 "per-dispatch mean over %d dispatches)" % traffic["FETCH_SIZE"][1]}},
              open("$OUT/pmc_traffic.json", "w"), indent=1)
PY
