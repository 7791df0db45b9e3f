#!/bin/bash
# round 6: the tile size chosen per launch (default) against 256-run tiles everywhere (WTAMD_DELTA_U=4) and 128-run tiles everywhere (=2)
R=$GRAFT_REPO_ROOT; cd $R
OUT=$R/gpurun_out/ab14; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
WTAMD_DELTA_U=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "difference_array or exact or golden or config_sized or full_size or plans_agree" 2>&1 | tail -1
B="python bench.py --no-cpu-baseline --no-e2e --no-sub --steps 5 --warmup 2"
for spec in "c2 0 16" "c2 0 32" "c2 0 64" "c2 20 200" "c2 0 8"; do
  set -- $spec
  for rep in 1 2; do
    for U in 4 auto 2; do
      if [ "$U" = "auto" ]; then unset WTAMD_DELTA_U; else export WTAMD_DELTA_U=$U; fi
      timeout 300 $B --config $1 --chroms $2 --mean-run $3 > $OUT/b.json 2> $OUT/b.err
      python - $OUT/b.json "$spec" "U=$U" <<'PY' | tee -a $OUT/ab.txt
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line); r = j.get("roofline", {})
        print("%-8s %-14s step_ms %.4f kernel_ms %.4f frac %.4f" % (sys.argv[3], sys.argv[2], j.get("ms_per_step"), r.get("kernel_ms"), r.get("frac")))
PY
    done
  done
done
