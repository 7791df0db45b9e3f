#!/bin/bash
# Runs on the GPU box: SQ counter passes (instruction mix / busy cycles) of one bench configuration.
# usage: tools_sq.sh "<bench args>"   -> gpurun_out/sq/summary.json
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sq
ARGS="${1:---scale 0.01 --steps 3 --warmup 1 --no-cpu-baseline}"
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" \
           "VALUBusy SALUBusy MemUnitBusy MemUnitStalled LDSBankConflict" \
           "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/p_sq$i -- python $R/bench.py $ARGS > $OUT/run$i.log 2>&1
done
python - <<PY
import csv, glob, json, os
per = {}
for f in glob.glob("/tmp/p_sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "wt_reduce" not in name and "wt_delta" not in name: continue
        per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
s = {k: sum(v) / len(v) for k, v in per.items()}
json.dump(s, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(s, indent=1))
PY
for i in 1 2 3 4; do tail -2 $OUT/run$i.log | cut -c1-300; done
