#!/bin/bash
# GPU call C: the bw tests, A/B of three inflate kernels on full batches (chromosomes 1 + 2 at half size), SQ counters.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4c
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_bwdev.py tests/test_bwreader.py tests/test_multidevice.py -q -m gpu > $OUT/gpu_tests.log 2>&1
tail -2 $OUT/gpu_tests.log
export WTAMD_BENCH_BWDIR=/dev/shm/wtamd_r4c
export WTAMD_GENOME_ONLY=0,1
SCALE=0.5
run() { name=$1; shift; env "$@" timeout 600 python tools/genome_files.py $SCALE > $OUT/ab_$name.json 2> $OUT/ab_$name.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/ab_$name.json").read().strip().splitlines()[-1])
    c, w = d["cold"], d["warm"]
    print("%-10s cold %.3e warm %.3e steady %.3e | warm: %d batches %d sections, decode %.1f ms (%.2f ms/batch), kernels %.1f, submit %.0f wait %.0f open %.3f s (cold %.3f) | cold submit %.0f dev afresh %.1f GB"
          % ("$name", d["bp_per_s"], d["warm_bp_per_s"], d.get("steady_bp_per_s") or 0, w["batches"], w["sections_inflated_on_device"], w["sum_device_decode_ms"], w["sum_device_decode_ms"] / max(w["batches"], 1),
             w["sum_kernel_ms"], w["host_submit_ms"], w["host_wait_ms"], w["open_seconds"], c["open_seconds"], c["host_submit_ms"], c["device_afresh"]["bytes"] / 1e9))
except Exception as e:
    print("$name failed:", e, open("$OUT/ab_$name.err").read()[-600:])
PY
}
L=$R/wiggletools_amd/csrc
run new   WTAMD_X=1
run v2a   WTAMD_LIB=$L/libwiggletools_amd_v2a.so
run r3    WTAMD_LIB=$L/libwiggletools_amd_r3.so
run new2  WTAMD_X=1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bw -- python $R/tools/genome_files.py $SCALE > $OUT/stats_run.log 2>&1
f=$(find /tmp/p_bw -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "wt_\|copyBuffer\|Name" $f | cut -c1-300 > $OUT/bw_kernel_stats.csv
python - <<PY
import csv
for r in csv.DictReader(open("$OUT/bw_kernel_stats.csv")):
    print("%-40s calls %4s avg %10.1f us  min %9.1f max %9.1f  %5s %%" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH" \
           "VALUBusy SALUBusy SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/b_sq$i -- python $R/tools/genome_files.py $SCALE > $OUT/bw_sq$i.log 2>&1
done
python - <<PY
import csv, glob, json
sq = {}
for f in glob.glob("/tmp/b_sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wt_bw_inflate" not in r.get("Kernel_Name", ""): continue
        if float(r.get("Grid_Size", 0) or 0) < 100000: continue
        sq.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
s = {k: sum(v) / len(v) for k, v in sq.items()}
if s.get("SQ_WAVE_CYCLES"):
    s["derived_wait_any_share"] = s.get("SQ_WAIT_ANY", 0) / s["SQ_WAVE_CYCLES"]
json.dump(s, open("$OUT/inflate_sq.json", "w"), indent=1)
print(json.dumps(s))
PY
rm -rf /dev/shm/wtamd_r4c
