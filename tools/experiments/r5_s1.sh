#!/bin/bash
# round 5, GPU session 1: new GPU tests, allocation probe, the cold file leg with the pool trace (GRCh38 x 0.25)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5s1
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_bwdev.py tests/test_dropin.py -q -m gpu -k "padded or non_float or buffered_reader" > $OUT/gpu_tests_new.log 2>&1
tail -3 $OUT/gpu_tests_new.log
timeout 300 tools/probes/cold_probe > $OUT/cold_probe.txt 2>&1
cat $OUT/cold_probe.txt
export WTAMD_BENCH_BWDIR=/dev/shm/wtamd_r5 
timeout 900 python tools/genome_files.py 0.25 > $OUT/files_write.json 2> $OUT/files_write.err     # writes the files (and runs cold / warm once)
python - <<PY
import json
r = json.loads(open("$OUT/files_write.json").read().strip().splitlines()[-1])
print("first process: write %.1f s; cold %.3f s warm %.3f s steady %.3g; cold pinned %s dev %s submit %.0f ms" % (r["files_written_s"], r["cold"]["seconds"], r["warm"]["seconds"], r["steady_bp_per_s"], r["cold"]["pinned_afresh"], r["cold"]["device_afresh"], r["cold"]["host_submit_ms"]))
PY
for k in 1 2; do
WTAMD_TRACE_POOL=1 timeout 600 python tools/genome_files.py 0.25 > $OUT/files_cold$k.json 2> $OUT/files_cold$k.err
python - <<PY
import json
r = json.loads(open("$OUT/files_cold$k.json").read().strip().splitlines()[-1])
print("fresh process $k: cold %.3f s (open %.3f) warm %.3f s steady %.3g; cold pinned %s dev %s submit %.0f ms wait %.0f ms" % (r["cold"]["seconds"], r["cold"]["open_seconds"], r["warm"]["seconds"], r["steady_bp_per_s"], r["cold"]["pinned_afresh"], r["cold"]["device_afresh"], r["cold"]["host_submit_ms"], r["cold"]["host_wait_ms"]))
PY
done
grep -c "pool" $OUT/files_cold1.err
rm -rf /dev/shm/wtamd_r5
