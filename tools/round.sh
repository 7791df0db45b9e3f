#!/bin/bash
# Runs on the GPU box: the round's evidence in one go -> gpurun_out/round/
#   bench lines of every BASELINE configuration at full size, rocprofv3 kernel stats + HBM PMC passes of
#   the default line, SQ counter summaries of c3 / c4 / c5, the pipeline overlap trace.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/round
mkdir -p $OUT
cd $R
for c in c2 c3 c4 c5; do
  extra="--no-e2e"; [ $c = c2 ] && extra=""
  timeout 900 python bench.py --config $c --steps 3 --warmup 1 $extra > $OUT/bench_$c.log 2>&1
  tail -1 $OUT/bench_$c.log > $OUT/bench_$c.json
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$c.json")); r = d["roofline"]
    print("$c", "bp/s %.3e" % d["value"], "ms/step %.1f" % d["ms_per_step"], "frac %.3f" % r["frac"], r["kernel"], "cpu", d.get("cpu_baseline", {}).get("value"), (d.get("cpu_baseline", {}).get("many_core") or {}).get("value"))
except Exception as e:
    print("$c failed", e, open("$OUT/bench_$c.log").read()[-500:])
PY
done
