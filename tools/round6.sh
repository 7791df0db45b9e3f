#!/bin/bash
# Runs on the GPU box: round-6 evidence -> gpurun_out/round6/ (summaries are copied to profiles/r06_* afterwards).
#   1. the -m gpu suite
#   2. per configuration (C2 mean, C3 var + stddev, C4 median, t-test): rocprofv3 kernel stats of the bench command, and the HBM traffic of
#      its dominant kernel from two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, --kernel-trace only) -> traffic.json keyed by the
#      bench's kernel label
#   3. the default bench line (what the driver runs)
# SKIP_TESTS=1 / SKIP_BENCH=1 / ONLY="c2 c3": parts of it
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/round6
mkdir -p $OUT
cd $R
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1
  tail -1 $OUT/gpu_tests.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-300 | tee $OUT/smoke.log
fi
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-sub"
declare -A ARGS
ARGS[c2]="--config c2"
ARGS[c3]="--config c3"
ARGS[c4]="--config c4 --chroms 0,20"
ARGS[ttest]="--config c5 --op ttest --chroms 0,20"
for cfg in ${ONLY:-c2 c3 c4 ttest}; do
  A="${ARGS[$cfg]} $COMMON"
  rm -rf /tmp/p_${cfg}_*
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_${cfg}_stats -- python $R/bench.py $A > $OUT/${cfg}_stats_run.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_${cfg}_fetch -- python $R/bench.py $A > $OUT/${cfg}_fetch_run.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_${cfg}_write -- python $R/bench.py $A > $OUT/${cfg}_write_run.log 2>&1
done
python - <<PY
import csv, glob, json, os
out = "$OUT"
cfgs = "${ONLY:-c2 c3 c4 ttest}".split()
match = {"c2": "wt_delta_kernel<2", "c3": "wt_delta_kernel<", "c4": "wt_walk_kernel", "ttest": "wt_delta_kernel<10"}
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
def pmc(dirn, ctr, m):
    v = []
    for f in glob.glob(dirn + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if m in r.get("Kernel_Name", "") and r.get("Counter_Name") == ctr:
                v.append(float(r["Counter_Value"]))
    return v
def line(path):
    try:
        return json.loads([l for l in open(path) if l.startswith("{")][-1])
    except Exception:
        return None
summary, kernels = {}, {}
for cfg in cfgs:
    st = stats("/tmp/p_%s_stats" % cfg, "%s_kernel_stats.csv" % cfg)
    for r in st[:6]: print(cfg, r["Name"][:70], r["Calls"], r["AverageNs"], r.get("Percentage"))
    ln = line(os.path.join(out, "%s_fetch_run.log" % cfg))
    ls = line(os.path.join(out, "%s_stats_run.log" % cfg))
    fe, wr = pmc("/tmp/p_%s_fetch" % cfg, "FETCH_SIZE", match[cfg]), pmc("/tmp/p_%s_write" % cfg, "WRITE_SIZE", match[cfg])
    s = {"fetch_launches": len(fe), "write_launches": len(wr)}
    try:
        label = ln["roofline"]["kernel"]
        # per launch: FETCH_SIZE is in KiB and, on gfx950, reports half of the coalesced reads (MI355X guide; calibrated on wt_auc_kernel in round 1); WRITE_SIZE in KiB
        fetch = sum(fe) / len(fe) * 1024 * 2
        write = sum(wr) / len(wr) * 1024
        # the bench's algorithmic bytes are per PASS; the launches of a pass: every profiled launch of the kernel is one (chromosome, reducer) item,
        # every chromosome equally often in the run, so mean bytes per launch x launches per pass = bytes per pass
        dk = [r for r in st if match[cfg] in r["Name"]]
        calls = sum(int(r["Calls"]) for r in dk)
        avg_ms = sum(float(r["AverageNs"]) * int(r["Calls"]) for r in dk) / max(calls, 1) / 1e6
        n_pass_launches = ln["config"].get("launches_per_pass")
        if not n_pass_launches:
            w = ln["config"]["workload"]
            n_pass_launches = int(w.split(" chrom(s)")[0].split(", ")[-1]) * len(ln["config"]["ops"])
        alg = ln["roofline"]["algorithmic_bytes_per_launch"] / n_pass_launches
        s.update({"kernel": label, "hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "launches_per_pass": n_pass_launches,
                  "algorithmic_bytes_per_launch_of_that_run": alg, "hbm_bytes_per_algorithmic_byte": (fetch + write) / alg,
                  "kernel_ms_rocprof_avg": avg_ms, "kernel_ms_bench_events_per_launch": (ls or ln)["roofline"]["kernel_ms"] / n_pass_launches,
                  "round": 6, "profile": "round 6 (tools/round6.sh: profiles/r06_%s_kernel_stats.csv, profiles/r06_pmc_summary.json)" % cfg})
        kernels[label] = s
    except Exception as e:
        s["error"] = repr(e)
    summary[cfg] = s
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:4000])
# the traffic file the bench line quotes: keyed by the bench's kernel label; kernels without a pass here keep what they had
tp = os.path.join("$R", "profiles", "traffic.json")
try:
    old = json.load(open(tp))
except Exception:
    old = {}
ks = dict(old.get("kernels") or ({old["kernel"]: old} if "kernel" in old else {}))
ks.update(kernels)
json.dump({"kernels": ks}, open(tp, "w"), indent=1)
json.dump({"kernels": ks}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
PY
# the file leg's kernels (chromosome 1 x 100 files): the inflate kernel's average per batch
if [ -z "$SKIP_BW" ]; then
  rm -rf /tmp/p_bw
  WTAMD_E2E_REPS=2 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bw -- python $R/tools/e2e_bw_only.py 248.9 100 > $OUT/bw_stats_run.log 2>&1
  f=$(find /tmp/p_bw -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/bw_kernel_stats.csv && head -4 $OUT/bw_kernel_stats.csv | cut -c1-160
fi
cd $R
if [ -z "$SKIP_BENCH" ]; then
  (time python bench.py --full-record $OUT/bench_default_full.json) > $OUT/bench_default.json 2> $OUT/bench_default.err
  tail -4 $OUT/bench_default.err | cut -c1-300
  tail -1 $OUT/bench_default.json | cut -c1-3000
fi
