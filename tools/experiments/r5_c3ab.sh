#!/bin/bash
# C3 (var + stddev, 500 tracks) on chromosome 21 and chromosome 1: the library against a variant build ($2), alternating, one box
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
if [ -z "$NOTEST" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "var or delta or difference or config_sized or golden" 2>&1 | tail -2 | tee $OUT/tests.txt; fi
B="python bench.py --no-cpu-baseline --no-e2e --no-sub --steps 3 --warmup 1 --config c3"
for v in $2 "" $2 ""; do
  if [ -n "$v" ]; then export WTAMD_LIB=$R/wiggletools_amd/csrc/libwiggletools_amd_$v.so; else unset WTAMD_LIB; fi
  for ch in 20 0; do
    timeout 300 $B --chroms $ch > $OUT/b.json 2> $OUT/b.err
    python - $OUT/b.json "${v:-new} chrom_index $ch" <<'PY' | tee -a $OUT/ab.txt
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line)
        r = j.get("roofline", {})
        print(sys.argv[2], "ms_per_step %.4f kernel_ms %.4f frac %.4f" % (j.get("ms_per_step"), r.get("kernel_ms"), r.get("frac")))
PY
  done
done
