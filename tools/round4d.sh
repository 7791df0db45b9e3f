#!/bin/bash
# GPU call D: two decode streams (tail filling) x hardware queues, on the 24-chromosome set at GRCh38 x 0.2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4d
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_bwdev.py tests/test_bwreader.py tests/test_multidevice.py tests/test_pipe.py -q -m gpu > $OUT/gpu_tests.log 2>&1
tail -2 $OUT/gpu_tests.log
export WTAMD_BENCH_BWDIR=/dev/shm/wtamd_r4d
SCALE=0.2
run() { name=$1; shift; env "$@" timeout 600 python tools/genome_files.py $SCALE > $OUT/ab_$name.json 2> $OUT/ab_$name.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/ab_$name.json").read().strip().splitlines()[-1])
    c, w = d["cold"], d["warm"]
    print("%-12s cold %.3e warm %.3e steady %.3e | warm: %.3f s, %d batches %d sections, decode %.1f ms, kernels %.1f, submit %.0f wait %.0f open %.3f s (cold %.3f) | cold submit %.0f dev afresh %.1f GB pinned %.1f GB"
          % ("$name", d["bp_per_s"], d["warm_bp_per_s"], d.get("steady_bp_per_s") or 0, w["seconds"], w["batches"], w["sections_inflated_on_device"], w["sum_device_decode_ms"],
             w["sum_kernel_ms"], w["host_submit_ms"], w["host_wait_ms"], w["open_seconds"], c["open_seconds"], c["host_submit_ms"], c["device_afresh"]["bytes"] / 1e9, c["pinned_afresh"]["bytes"] / 1e9))
except Exception as e:
    print("$name failed:", e, open("$OUT/ab_$name.err").read()[-600:])
PY
}
L=$R/wiggletools_amd/csrc
run two_q8     GPU_MAX_HW_QUEUES=8
run two_q4     WTAMD_X=1
run one_q8     GPU_MAX_HW_QUEUES=8 WTAMD_BW_DECODE_STREAMS=1
run one_q4     WTAMD_BW_DECODE_STREAMS=1
run two_q8_63k GPU_MAX_HW_QUEUES=8 WTAMD_BW_BATCH_SECTIONS=61504
run two_q8_d3  GPU_MAX_HW_QUEUES=8 WTAMD_PIPE_DEPTH=3
run r3         WTAMD_LIB=$L/libwiggletools_amd_r3.so
rm -rf /dev/shm/wtamd_r4d
