#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats + HBM PMC passes of bench.py; leaves only small
# summaries under gpurun_out/prof/ (the raw kernel traces are tens of MiB because the synthetic
# data generator launches thousands of torch kernels).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
ARGS="${BENCH_ARGS:---steps 2 --warmup 1 --no-cpu-baseline --no-e2e}"
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python $R/bench.py $ARGS > $OUT/stats_run.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -- python $R/bench.py $ARGS > $OUT/fetch_run.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -- python $R/bench.py $ARGS > $OUT/write_run.log 2>&1
python - <<PY
import csv, glob, json, os
out = "$OUT"
def one(pat):
    f = glob.glob(pat, recursive=True)
    return f[0] if f else None
ks = one("/tmp/p_stats/**/*kernel_stats.csv")
rows = list(csv.DictReader(open(ks))) if ks else []
keep = [r for r in rows if "wt_" in r.get("Name", "")]
with open(os.path.join(out, "kernel_stats_wt.csv"), "w") as fh:
    if rows:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader()
        for r in keep: w.writerow(r)
with open(os.path.join(out, "kernel_stats_top.csv"), "w") as fh:
    if rows:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader()
        for r in rows[:15]: w.writerow(r)
summary = {}
for tag, pat, ctr in (("fetch", "/tmp/p_fetch/**/*counter_collection.csv", "FETCH_SIZE"), ("write", "/tmp/p_write/**/*counter_collection.csv", "WRITE_SIZE")):
    f = one(pat)
    if not f: continue
    per = {}
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "wt_" not in name: continue
        if r.get("Counter_Name") != ctr: continue
        k = name.split("(")[0][:60]
        per.setdefault(k, []).append(float(r["Counter_Value"]))
    summary[tag] = {k: {"launches": len(v), "mean_counter_value": sum(v) / len(v)} for k, v in per.items()}
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary)[:1500])
for r in keep: print(r)
PY
tail -1 $OUT/stats_run.log | cut -c1-1200
