#!/bin/bash
# round 5, GPU session 2: page-locking by mmap + THP + hipHostRegister -- pipe tests, then the cold file leg (GRCh38 x 0.25):
# in-process cold / warm as before, and a FRESH process through tools/cli_cold.py with the open / feeder traces
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5s2
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_pipe.py tests/test_dropin.py tests/test_bwdev.py tests/test_integrator_doors.py -q -m gpu -x > $OUT/gpu_tests_pipe.log 2>&1
tail -3 $OUT/gpu_tests_pipe.log
export WTAMD_BENCH_BWDIR=/dev/shm/wtamd_r5
timeout 900 python tools/genome_files.py 0.25 > $OUT/files_write.json 2> $OUT/files_write.err
BP=$(python -c "
import json
r = json.loads(open('$OUT/files_write.json').read().strip().splitlines()[-1])
print(r['bp'])")
python - <<PY
import json
r = json.loads(open("$OUT/files_write.json").read().strip().splitlines()[-1])
print("first process: write %.1f s; cold %.3f s (open %.3f) warm %.3f s steady %.3g; cold pinned %s dev %s submit %.0f ms" % (r["files_written_s"], r["cold"]["seconds"], r["cold"]["open_seconds"], r["warm"]["seconds"], r["steady_bp_per_s"], r["cold"]["pinned_afresh"], r["cold"]["device_afresh"], r["cold"]["host_submit_ms"]))
PY
for k in 1 2; do
WTAMD_TRACE_POOL=1 timeout 600 python tools/genome_files.py 0.25 > $OUT/files_cold$k.json 2> $OUT/files_cold$k.err
python - <<PY
import json
r = json.loads(open("$OUT/files_cold$k.json").read().strip().splitlines()[-1])
print("torch process $k: cold %.3f s (open %.3f, readers %.3f) warm %.3f s steady %.3g; submit %.0f ms wait %.0f ms decode %.0f / %.0f ms" % (r["cold"]["seconds"], r["cold"]["open_seconds"], r["cold"]["open_readers_seconds"], r["warm"]["seconds"], r["steady_bp_per_s"], r["cold"]["host_submit_ms"], r["cold"]["host_wait_ms"], r["cold"]["sum_device_decode_ms"], r["warm"]["sum_device_decode_ms"]))
PY
done
grep "page-locked" $OUT/files_cold1.err | awk '{s+=$NF==""?0:$(NF-1); n+=$3} END {print "page-locked MB", n, "ms", s}'
grep "page-locked" $OUT/files_cold1.err | head -3
for k in 1 2 3; do
  T=""; [ $k = 1 ] && T="WTAMD_TRACE=1 WTAMD_TRACE_OPEN=1 WTAMD_TRACE_POOL=1"
  env $T timeout 300 python tools/cli_cold.py /dev/shm/wtamd_r5 100 mean $BP > $OUT/cli_cold$k.json 2> $OUT/cli_cold$k.err
  cat $OUT/cli_cold$k.json
done
grep -v "^\[feeder\]\|\[pool\]" $OUT/cli_cold1.err | head -30
grep "\[feeder\]" $OUT/cli_cold1.err | head -12
ls /proc/self/fd | wc -l; grep FDSize /proc/self/status; free -g | head -2; cat /sys/fs/cgroup/memory.max 2>/dev/null; nproc
rm -rf /dev/shm/wtamd_r5
