#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4g
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x > $OUT/gpu_tests.log 2>&1
tail -2 $OUT/gpu_tests.log
for v in k8 full; do
  timeout 300 python bench.py --chroms 20 --values $v --steps 3 --warmup 1 --no-sub --no-e2e --no-cpu-baseline --no-genome-files > $OUT/c2_$v.log 2>&1
  tail -1 $OUT/c2_$v.log > $OUT/c2_$v.json
  python - <<PY
import json
try:
    d = json.load(open("$OUT/c2_$v.json")); r = d["roofline"]
    print("%-6s ms/step %.3f kernel %.3f index %.3f frac %.3f with index %.3f patched %s of %s windows" % ("$v", d["ms_per_step"], r["kernel_ms"], r["index_kernel_ms"], r["frac"], r["frac_with_index"], d["config"]["patched_windows_per_step"], d["config"]["windows_per_step"]))
except Exception as e:
    print("$v failed", e, open("$OUT/c2_$v.log").read()[-800:])
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_full -- python $R/bench.py --chroms 20 --values full --steps 3 --warmup 1 --no-sub --no-e2e --no-cpu-baseline --no-genome-files > $OUT/stats_full.log 2>&1
f=$(find /tmp/p_full -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "wt_\|Name" $f | cut -c1-300 > $OUT/full_kernel_stats.csv
python - <<PY
import csv
for r in csv.DictReader(open("$OUT/full_kernel_stats.csv")):
    print("%-50s calls %4s avg %10.1f us  min %9.1f max %9.1f" % (r["Name"].split("(")[0][-50:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
