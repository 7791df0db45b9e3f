#!/bin/bash
# wt_delta_kernel's pass 2 with one resource taken out (-DWT_EXP=1: no LDS atomics; 2: no loads; 3: control), phase cycles + kernel time; results of 1 and 2 are wrong by design
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
shift
for v in "$@"; do
  L=$R/wiggletools_amd/csrc/libwiggletools_amd_$v.so
  for l in ${RUNS:-16 200}; do
    echo "== variant $v mean run $l" | tee -a $OUT/exp.txt
    WTAMD_LIB=$L timeout 200 python bench.py --no-cpu-baseline --no-e2e --no-sub --steps 3 --warmup 1 --config c2 --chroms 20 --mean-run $l > $OUT/b_${v}_$l.json 2> $OUT/b_${v}_$l.err
    grep "wt_profile" $OUT/b_${v}_$l.err | tail -1 | tee -a $OUT/exp.txt
    tail -3 $OUT/b_${v}_$l.err | grep -v wt_profile | tee -a $OUT/exp.txt
    python - $OUT/b_${v}_$l.json <<'PY' | tee -a $OUT/exp.txt
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line)
        r = j.get("roofline", {})
        print("ms_per_step", j.get("ms_per_step"), "kernel_ms", r.get("kernel_ms"), "frac", r.get("frac"))
PY
  done
done
