#!/bin/bash
# phase cycles of wt_delta_kernel against run length, every [wt_profile] line of every call kept and labelled (r5_prof.sh kept the last one only)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
L=$R/wiggletools_amd/csrc/libwiggletools_amd_$2.so
B="python bench.py --no-cpu-baseline --no-e2e --no-sub --steps 3 --warmup 1 --config c2 --chroms 20"
for l in 200 64 16 4 1; do
  echo "== mean run $l" | tee -a $OUT/prof_by_run.txt
  WTAMD_LIB=$L WTAMD_TRACE=1 timeout 300 $B --mean-run $l > $OUT/b_$l.json 2> $OUT/b_$l.err
  grep "wt_profile" $OUT/b_$l.err | tee -a $OUT/prof_by_run.txt
  python - $OUT/b_$l.json <<'PY' | tee -a $OUT/prof_by_run.txt
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line)
        r = j.get("roofline", {})
        print("ms_per_step", j.get("ms_per_step"), "kernel_ms", r.get("kernel_ms"), "frac", r.get("frac"), "achieved", r.get("achieved"), "kernel", r.get("kernel"))
PY
done
