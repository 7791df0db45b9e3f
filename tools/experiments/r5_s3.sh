#!/bin/bash
# round 5, GPU session 3: the cold file leg after the fd-table / warm-up / page-locking changes.
#   1. GRCh38 x 0.25: fresh processes (tools/cli_cold.py) with traces; how long does dlopen of the HIP runtime alone take?
#   2. FULL scale (90 GB of files): in-process cold / warm with /proc/vmstat deltas + fresh process
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5s3
mkdir -p $OUT
cd $R
python - <<PY
import ctypes, time
t0 = time.perf_counter(); ctypes.CDLL("libamdhip64.so"); t1 = time.perf_counter()
print("dlopen(libamdhip64.so) alone: %.3f s" % (t1 - t0))
PY
python - <<PY
import ctypes, time
t0 = time.perf_counter(); ctypes.CDLL("$R/wiggletools_amd/csrc/libwiggletools_amd.so"); t1 = time.perf_counter()
print("dlopen(libwiggletools_amd.so) in a second new process: %.3f s" % (t1 - t0))
PY
export WTAMD_BENCH_BWDIR=/dev/shm/wtamd_r5
timeout 900 python tools/genome_files.py 0.25 > $OUT/q_write.json 2> $OUT/q_write.err
BP=$(python -c "
import json
print(json.loads(open('$OUT/q_write.json').read().strip().splitlines()[-1])['bp'])")
for k in 1 2 3; do
  T=""; [ $k = 1 ] && T="WTAMD_TRACE=1 WTAMD_TRACE_OPEN=1"
  env $T timeout 300 python tools/cli_cold.py /dev/shm/wtamd_r5 100 mean $BP > $OUT/q_cli$k.json 2> $OUT/q_cli$k.err
  cat $OUT/q_cli$k.json
done
grep -v "^\[bw_open\]\|^\[reader\] open\|\[pool\]" $OUT/q_cli1.err | head -14
grep "\[bw_open\]" $OUT/q_cli1.err | awk '$3>5' | wc -l
WTAMD_NO_WARMUP=1 timeout 300 python tools/cli_cold.py /dev/shm/wtamd_r5 100 mean $BP | tee $OUT/q_cli_nowarm.json
rm -rf /dev/shm/wtamd_r5
# full scale
export WTAMD_BENCH_VMSTAT=1 WTAMD_FRESH=1
timeout 1200 python tools/genome_files.py 1.0 > $OUT/full.json 2> $OUT/full.err
python - <<PY
import json
r = json.loads(open("$OUT/full.json").read().strip().splitlines()[-1])
print("FULL: write %.1f s; cold %.3f s = %.3g bp/s; warm %.3f s = %.3g; steady %.3g" % (r["files_written_s"], r["cold"]["seconds"], r["bp_per_s"], r["warm"]["seconds"], r["warm_bp_per_s"], r["steady_bp_per_s"]))
for k in ("cold", "warm"):
    o = r[k]
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in o.items() if a not in ("runs", "chromosomes_seen", "intervals_per_s")})
print("fresh", r.get("fresh_process"))
PY
free -g | head -2
rm -rf /dev/shm/wtamd_r5
