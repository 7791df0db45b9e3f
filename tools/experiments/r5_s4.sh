#!/bin/bash
# round 5, GPU session 4: full-scale file leg with the files read through once before the cold run (page-cache first-read effect),
# the inflate kernel's occupancy by need, fresh process on the system runtime
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5s4
mkdir -p $OUT
cd $R
export WTAMD_BENCH_BWDIR=/dev/shm/wtamd_r5 WTAMD_BENCH_VMSTAT=1 WTAMD_FRESH=1
timeout 1500 python tools/genome_files.py 1.0 > $OUT/full.json 2> $OUT/full.err
python - <<PY
import json
r = json.loads(open("$OUT/full.json").read().strip().splitlines()[-1])
print("FULL: write %.1f s, read-through %.1f s; cold %.3f s = %.3g bp/s; warm %.3f s = %.3g; steady %.3g" % (r["files_written_s"], r["files_read_through_s"], r["cold"]["seconds"], r["bp_per_s"], r["warm"]["seconds"], r["warm_bp_per_s"], r["steady_bp_per_s"]))
for k in ("cold", "warm"):
    o = r[k]
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in o.items() if a not in ("runs", "chromosomes_seen", "intervals_per_s")})
print("fresh", r.get("fresh_process"))
PY
BP=$(python -c "
import json
print(json.loads(open('$OUT/full.json').read().strip().splitlines()[-1])['bp'])")
WTAMD_TRACE=1 timeout 300 python tools/cli_cold.py /dev/shm/wtamd_r5 100 mean $BP > $OUT/cli2.json 2> $OUT/cli2.err
cat $OUT/cli2.json
grep "bwdev\|opened\|multiplexer\|reducer" $OUT/cli2.err | head
# a third in-process pass: is the second read of the files slower than the third?
python - <<PY
import json, os, sys, time
sys.path.insert(0, "$R")
os.environ["WTAMD_BENCH_NO_READ_THROUGH"] = "1"
import torch, bench
torch.cuda.set_device(0)
r = bench.e2e_bigwig_genome("mean", 100, 16.0, 1.0, torch.device("cuda", 0))
print("third / fourth pass over the files (no read-through this time): cold %.3f s = %.3g bp/s, warm %.3f s = %.3g, steady %.3g" % (r["cold"]["seconds"], r["bp_per_s"], r["warm"]["seconds"], r["warm_bp_per_s"], r["steady_bp_per_s"]))
PY
rm -rf /dev/shm/wtamd_r5
