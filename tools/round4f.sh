#!/bin/bash
# GPU call F: delta / patch parity tests, the patched-window records, inflate kernel timing after the last trims
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4f
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_bwdev.py -q -m gpu -x > $OUT/gpu_tests.log 2>&1
tail -2 $OUT/gpu_tests.log
for v in k8 fullm full; do
  timeout 300 python bench.py --chroms 20 --values $v --steps 3 --warmup 1 --no-sub --no-e2e --no-cpu-baseline --no-genome-files > $OUT/c2_$v.log 2>&1
  tail -1 $OUT/c2_$v.log > $OUT/c2_$v.json
  python - <<PY
import json
try:
    d = json.load(open("$OUT/c2_$v.json")); r = d["roofline"]
    print("%-6s ms/step %.3f frac %.3f with index %.3f patched %s of %s windows" % ("$v", d["ms_per_step"], r["frac"], r["frac_with_index"], d["config"]["patched_windows_per_step"], d["config"]["windows_per_step"]))
except Exception as e:
    print("$v failed", e, open("$OUT/c2_$v.log").read()[-800:])
PY
done
export WTAMD_BENCH_BWDIR=/dev/shm/wtamd_r4f WTAMD_GENOME_ONLY=0,1
timeout 600 python tools/genome_files.py 0.5 > $OUT/files.json 2> $OUT/files.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bw -- python $R/tools/genome_files.py 0.5 > $OUT/stats_run.log 2>&1
f=$(find /tmp/p_bw -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "wt_\|copyBuffer\|Name" $f | cut -c1-300 > $OUT/bw_kernel_stats.csv
python - <<PY
import csv, json
d = json.loads(open("$OUT/files.json").read().strip().splitlines()[-1])
print("files chr1+2 x0.5: cold %.3e warm %.3e steady %.3e" % (d["bp_per_s"], d["warm_bp_per_s"], d["steady_bp_per_s"]))
for r in csv.DictReader(open("$OUT/bw_kernel_stats.csv")):
    print("%-40s calls %4s avg %10.1f us  min %9.1f max %9.1f  %5s %%" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
rm -rf /dev/shm/wtamd_r4f
