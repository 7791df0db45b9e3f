#!/bin/bash
# Runs on the GPU box: SQ counter passes (instruction mix / busy cycles / instruction cache) of one bench
# configuration.   usage: tools/sq.sh "<bench args>" [tag]   -> gpurun_out/sq_<tag>/summary.json
R=$GRAFT_REPO_ROOT
TAG=${2:-run}
OUT=$R/gpurun_out/sq_$TAG
ARGS="${1:---scale 0.02 --steps 2 --warmup 1} --no-cpu-baseline --no-e2e"
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" \
           "VALUBusy SALUBusy MemUnitBusy MemUnitStalled LDSBankConflict" \
           "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVES" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
  i=$((i+1))
  rm -rf /tmp/p_sq$i
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
if s.get("SQ_WAVE_CYCLES"):
    s["derived_valu_insts_per_wave_cycle"] = s.get("SQ_INSTS_VALU", 0) / s["SQ_WAVE_CYCLES"]
    s["derived_wait_any_share"] = s.get("SQ_WAIT_ANY", 0) / s["SQ_WAVE_CYCLES"]
if s.get("SQC_ICACHE_REQ"):
    s["derived_icache_miss_rate"] = s.get("SQC_ICACHE_MISSES", 0) / s["SQC_ICACHE_REQ"]
s["bench_args"] = "$ARGS"
json.dump(s, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(s, indent=1))
PY
for i in 1; do tail -1 $OUT/run$i.log | cut -c1-200; done
