#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
OUT=$R/gpurun_out/ab5; mkdir -p $OUT
lib() { if [ "$1" != "-" ]; then echo $R/wiggletools_amd/csrc/libwiggletools_amd_$1.so; else echo $R/wiggletools_amd/csrc/libwiggletools_amd.so; fi; }
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
WTAMD_LIB=$(lib nohead) timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "difference_array or exact or golden or config_sized" 2>&1 | tail -1
B="python bench.py --no-cpu-baseline --no-e2e --no-sub --steps 5 --warmup 2"
for spec in "c2 0 16" "c2 20 200" "c2 0 64" "c3 0 16" "c2 20 16 max"; do
  set -- $spec
  for rep in 1 2; do
    for L in wb nohead -; do
      WTAMD_LIB=$(lib $L) timeout 300 $B --config $1 --chroms $2 --mean-run $3 ${4:+--op $4} > $OUT/b.json 2> $OUT/b.err
      python - $OUT/b.json "$spec" "$L" <<'PY' | tee -a $OUT/ab.txt
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
