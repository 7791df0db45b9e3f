#!/bin/bash
# C4 (median, 100 tracks) on chromosome 21: the library against a variant build ($2), alternating, one box; the walking tests first
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
if [ -z "$NOTEST" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "walk or median or mwu or wilcoxon" 2>&1 | tail -2 | tee $OUT/tests.txt; fi
B="python bench.py --no-cpu-baseline --no-e2e --no-sub --steps 3 --warmup 1 --config c4 --chroms 20"
for v in $2 "" $2 ""; do
  if [ -n "$v" ]; then export WTAMD_LIB=$R/wiggletools_amd/csrc/libwiggletools_amd_$v.so; else unset WTAMD_LIB; fi
  timeout 300 $B > $OUT/b.json 2> $OUT/b.err
  python - $OUT/b.json "${v:-new}" <<'PY' | tee -a $OUT/ab.txt
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line)
        r = j.get("roofline", {})
        print(sys.argv[2], "ms_per_step %.4f kernel_ms %.4f frac %.4f kernel %s" % (j.get("ms_per_step"), r.get("kernel_ms"), r.get("frac"), r.get("kernel")))
PY
done
