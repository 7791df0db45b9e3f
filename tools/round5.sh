#!/bin/bash
# Runs on the GPU box: round-5 evidence -> gpurun_out/round5/ (summaries are copied to profiles/ afterwards)
#   1. -m gpu tests (whole suite)
#   2. C2 default line: rocprofv3 kernel stats, HBM PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) -> traffic
#   3. c5 on chromosome 21: SQ instruction counters of the register-column MWU kernel (its erf is a table now) -> issue roofline
#   4. the default bench line
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/round5
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1
tail -1 $OUT/gpu_tests.log
cd /tmp && export TMPDIR=/tmp
BARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-sub"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python $R/bench.py $BARGS > $OUT/c2_stats_run.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -- python $R/bench.py $BARGS > $OUT/c2_fetch_run.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -- python $R/bench.py $BARGS > $OUT/c2_write_run.log 2>&1
i=0
if [ -z "$SKIP_C5" ]; then
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "VALUBusy SALUBusy SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/p_c5_$i -- python $R/bench.py --config c5 --chroms 20 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-sub > $OUT/c5_sq$i.log 2>&1
done
fi
python - <<PY
import csv, glob, json, os
out = "$OUT"
def one(pat):
    f = glob.glob(pat, recursive=True)
    return f[0] if f else None
def stats(dirn, name):
    ks = one(dirn + "/**/*kernel_stats.csv")
    rows = list(csv.DictReader(open(ks))) if ks else []
    keep = [r for r in rows if "wt_" in r.get("Name", "") or "copyBuffer" in r.get("Name", "")]
    with open(os.path.join(out, name), "w") as fh:
        if rows:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader()
            for r in keep: w.writerow(r)
    return keep
def pmc(dirn, ctrs, match):
    per = {}
    for f in glob.glob(dirn + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if match not in name or r.get("Counter_Name") not in ctrs: continue
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "mean": sum(v) / len(v), "sum": sum(v)} for k, v in per.items()}
c2 = stats("/tmp/p_stats", "c2_kernel_stats.csv")
for r in c2: print(r["Name"][:60], r["Calls"], r["AverageNs"])
summary = {"c2": {"fetch": pmc("/tmp/p_fetch", ("FETCH_SIZE",), "wt_delta_kernel"), "write": pmc("/tmp/p_write", ("WRITE_SIZE",), "wt_delta_kernel")}}
try:
    line = json.loads([l for l in open(os.path.join(out, "c2_fetch_run.log")) if l.startswith("{")][-1])
    launches = 24
    alg = line["roofline"]["algorithmic_bytes_per_launch"] / launches
    fetch = summary["c2"]["fetch"]["FETCH_SIZE"]["mean"] * 1024 * 2          # KiB; gfx950 reports half of the coalesced reads (MI355X guide; calibrated on wt_auc_kernel in round 1)
    write = summary["c2"]["write"]["WRITE_SIZE"]["mean"] * 1024
    dk = [r for r in c2 if "wt_delta_kernel" in r["Name"]]
    summary["traffic"] = {"kernel": "wt_delta_kernel<mean>", "hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
                          "algorithmic_bytes_per_launch_of_that_run": alg, "hbm_bytes_per_algorithmic_byte": (fetch + write) / alg,
                          "kernel_ms_rocprof_avg": float(dk[0]["AverageNs"]) / 1e6 if dk else None, "kernel_ms_bench_events": line["roofline"]["kernel_ms"] / launches,
                          "round": 5, "profile": "round 5 (tools/round5.sh: profiles/r05_pmc_sq_summary.json, profiles/r05_c2_kernel_stats.csv)"}
except Exception as e:
    summary["traffic_error"] = repr(e)
s = {}
for i in (1, 2):
    s.update(pmc("/tmp/p_c5_%d" % i, ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES", "VALUBusy", "SALUBusy", "SQ_WAIT_ANY", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"), "wt_reduce_kernel"))
try:
    line = json.loads([l for l in open(os.path.join(out, "c5_sq1.log")) if l.startswith("{")][-1])
    s["output_runs_per_launch"] = line["output_runs"]
    s["kernel_ms_bench_events"] = line["roofline"]["kernel_ms"]
except Exception as e:
    s["line_error"] = repr(e)
summary["c5"] = s
json.dump(summary, open(os.path.join(out, "pmc_sq_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
PY
cd $R
# the traffic file the bench line quotes: this round's
python - <<PY
import json
s = json.load(open("$OUT/pmc_sq_summary.json"))
if "traffic" in s: json.dump(s["traffic"], open("$R/profiles/traffic.json", "w"), indent=1)
PY
cp $R/profiles/traffic.json $OUT/traffic.json
(time python bench.py) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/bench_default.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("bench_seconds", d["bench_seconds"], "C2", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"))
print(json.dumps({k: v for k, v in d["config"].items() if k.startswith("e2e_") or k.endswith("_frac") or k.endswith("_step")}, indent=0))
print("bulk", d.get("value_e2e_bulk"), "levels", d.get("e2e_bigwig_levels"))
print("fresh", d["e2e_bigwig_genome"].get("fresh_process"))
print("cold", d["e2e_bigwig_genome"].get("cold"))
print("warm", d["e2e_bigwig_genome"].get("warm"))
PY
