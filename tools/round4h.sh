#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4h
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_bwdev.py tests/test_bwreader.py tests/test_multidevice.py -q -m gpu > $OUT/gpu_tests.log 2>&1
tail -1 $OUT/gpu_tests.log
export WTAMD_BENCH_BWDIR=/dev/shm/wtamd_r4h
SCALE=0.3
run() { name=$1; shift; env "$@" timeout 600 python tools/genome_files.py $SCALE > $OUT/ab_$name.json 2> $OUT/ab_$name.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/ab_$name.json").read().strip().splitlines()[-1])
    c, w = d["cold"], d["warm"]
    print("%-10s cold %.3e (%.3f s: open %.3f submit %.0f ms wait %.0f ms; pinned afresh %.1f GB in %d, device %.1f GB in %d) warm %.3e (%.3f s) steady %.3e"
          % ("$name", d["bp_per_s"], c["seconds"], c["open_seconds"], c["host_submit_ms"], c["host_wait_ms"], c["pinned_afresh"]["bytes"] / 1e9, c["pinned_afresh"]["buffers"],
             c["device_afresh"]["bytes"] / 1e9, c["device_afresh"]["buffers"], d["warm_bp_per_s"], w["seconds"], d.get("steady_bp_per_s") or 0))
except Exception as e:
    print("$name failed:", e, open("$OUT/ab_$name.err").read()[-600:])
PY
}
run write      WTAMD_X=1
run ahead      WTAMD_X=1
run no_ahead   WTAMD_PIN_AHEAD=0
run ahead2     WTAMD_X=1
run no_ahead2  WTAMD_PIN_AHEAD=0
rm -rf /dev/shm/wtamd_r4h
