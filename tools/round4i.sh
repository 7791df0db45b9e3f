#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4i
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_bwdev.py tests/test_bwreader.py tests/test_multidevice.py -q -m gpu > $OUT/gpu_tests.log 2>&1
tail -1 $OUT/gpu_tests.log
export WTAMD_BENCH_BWDIR=/dev/shm/wtamd_r4i
SCALE=${SCALE:-0.5}
run() { name=$1; shift; env "$@" timeout 600 python tools/genome_files.py $SCALE > $OUT/ab_$name.json 2> $OUT/ab_$name.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/ab_$name.json").read().strip().splitlines()[-1])
    c, w = d["cold"], d["warm"]
    print("%-10s cold %.3e (%.3f s) warm %.3e (%.3f s: %d batches, decode %.0f ms, open %.3f, submit %.0f, wait %.0f) steady %.3e | cold dev %.1f GB pinned %.1f GB"
          % ("$name", d["bp_per_s"], c["seconds"], d["warm_bp_per_s"], w["seconds"], w["batches"], w["sum_device_decode_ms"], w["open_seconds"], w["host_submit_ms"], w["host_wait_ms"],
             d.get("steady_bp_per_s") or 0, c["device_afresh"]["bytes"] / 1e9, c["pinned_afresh"]["bytes"] / 1e9))
except Exception as e:
    print("$name failed:", e, open("$OUT/ab_$name.err").read()[-600:])
PY
}
F=126976
run t100 WTAMD_X=1
run t094 WTAMD_BW_BATCH_SECTIONS=$((F*94/100))
run t088 WTAMD_BW_BATCH_SECTIONS=$((F*88/100))
run t100b WTAMD_X=1
run t080 WTAMD_BW_BATCH_SECTIONS=$((F*80/100))
rm -rf /dev/shm/wtamd_r4i
