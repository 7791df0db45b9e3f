#!/bin/bash
# round 6: 128-run tiles for the launches with squares and the t-test (chosen per launch) against 256-run tiles (WTAMD_DELTA_U=4)
R=$GRAFT_REPO_ROOT; cd $R
OUT=$R/gpurun_out/ab15; mkdir -p $OUT
WTAMD_DELTA_U=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "var or ttest or two_sample or stddev or cv" 2>&1 | tail -1
B="python bench.py --no-cpu-baseline --no-e2e --no-sub --steps 3 --warmup 1"
for spec in "c5 20 16 ttest" "c3 0 16" "c2 20 16 var" "c2 20 200 stddev"; do
  set -- $spec
  for rep in 1 2; do
    for U in 4 auto; do
      if [ "$U" = "auto" ]; then unset WTAMD_DELTA_U; else export WTAMD_DELTA_U=$U; fi
      timeout 300 $B --config $1 --chroms $2 --mean-run $3 --op $4 > $OUT/b.json 2> $OUT/b.err
      python - $OUT/b.json "$spec" "U=$U" <<'PY' | tee -a $OUT/ab.txt
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line); r = j.get("roofline", {})
        print("%-8s %-18s step_ms %.4f kernel_ms %.4f frac %.4f %s" % (sys.argv[3], sys.argv[2], j.get("ms_per_step"), r.get("kernel_ms"), r.get("frac"), r.get("kernel")))
PY
    done
  done
done
