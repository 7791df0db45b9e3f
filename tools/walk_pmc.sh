#!/bin/bash
# SQ counters of the walking median kernel (chr21), in separate passes
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "VALUBusy SALUBusy SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pw_$i -- python $R/bench.py --config c4 --chroms 20 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-sub --no-genome-files --e2e-bw-mbp 0 > $OUT/walk_sq$i.log 2>&1
done
python - <<PY
import csv, glob, json
per = {}
for f in glob.glob("/tmp/pw_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wt_walk_kernel" not in r.get("Kernel_Name", ""): continue
        per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {k: sum(v) / len(v) for k, v in per.items()}
json.dump(out, open("$OUT/walk_sq.json", "w"), indent=1)
for k, v in sorted(out.items()): print(k, "%.4g" % v)
PY
