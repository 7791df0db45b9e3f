#!/bin/bash
# Evidence for MedianReduction by walking (chromosome 21, bench.py --config c4 --chroms 20): kernel stats of the walking
# and of the bitmap kernel, the SQ counters of the walking kernel (separate passes), and its phase cycles.
# Writes gpurun_out/walk_*; tools copy: profiles/r04_walk_*.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
B="--config c4 --chroms 20 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-sub --no-genome-files --e2e-bw-mbp 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw_stats -- python $R/bench.py $B > $OUT/walk_stats_run.log 2>&1
WTAMD_NO_WALK=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw_stats_old -- python $R/bench.py $B > $OUT/walk_stats_old_run.log 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "VALUBusy SALUBusy SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pw_$i -- python $R/bench.py --config c4 --chroms 20 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-sub --no-genome-files --e2e-bw-mbp 0 > $OUT/walk_sq$i.log 2>&1
done
if [ -f $R/wiggletools_amd/csrc/libwiggletools_amd_walkprof.so ]; then
  WTAMD_LIB=$R/wiggletools_amd/csrc/libwiggletools_amd_walkprof.so WTAMD_WALK_PROF=1 python $R/bench.py $B 2>&1 >/dev/null | grep wt_walk_profile | tail -1 > $OUT/walk_phases.txt
fi
python - <<PY
import csv, glob, json
out = {}
def stats(d):
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        return [r for r in csv.DictReader(open(f)) if "wt_" in r.get("Name", "")]
    return []
rows = stats("/tmp/pw_stats"); rows_old = stats("/tmp/pw_stats_old")
for name, rr in (("walk_kernel_stats.csv", rows), ("walk_kernel_stats_bitmap.csv", rows_old)):
    if rr:
        with open("$OUT/" + name, "w") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rr[0].keys())); w.writeheader()
            for r in rr: w.writerow(r)
per = {}
for f in glob.glob("/tmp/pw_[0-9]*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wt_walk_kernel" not in r.get("Kernel_Name", ""): continue
        per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
sq = {k: sum(v) / len(v) for k, v in per.items()}
line = json.loads([l for l in open("$OUT/walk_sq1.log") if l.startswith("{")][-1])
runs = line["output_runs"]
summ = {"workload": line["config"]["workload"], "output_runs_per_launch": runs, "counters_mean_per_launch": sq,
        "valu_instructions_per_output_run": sq.get("SQ_INSTS_VALU", 0) / runs, "salu_instructions_per_output_run": sq.get("SQ_INSTS_SALU", 0) / runs,
        "lds_instructions_per_output_run": sq.get("SQ_INSTS_LDS", 0) / runs,
        "wait_any_share": (sq.get("SQ_WAIT_ANY", 0) / sq["SQ_WAVE_CYCLES"]) if sq.get("SQ_WAVE_CYCLES") else None,
        "phases": open("$OUT/walk_phases.txt").read().strip() if glob.glob("$OUT/walk_phases.txt") else None}
json.dump(summ, open("$OUT/walk_sq.json", "w"), indent=1)
for r in rows + rows_old: print(r["Name"][:50], r["Calls"], r["AverageNs"])
print(json.dumps({k: summ[k] for k in summ if k != "counters_mean_per_launch"}, indent=1))
PY
