#!/bin/bash
# round 6, last sweep: compile-time shapes of the pass after the fixed phases shrank (everyone parks; 2 / 8 runs per lane and tile)
R=$GRAFT_REPO_ROOT; cd $R
OUT=$R/gpurun_out/ab13; mkdir -p $OUT
lib() { if [ "$1" != "-" ]; then echo $R/wiggletools_amd/csrc/libwiggletools_amd_$1.so; else echo $R/wiggletools_amd/csrc/libwiggletools_amd.so; fi; }
B="python bench.py --no-cpu-baseline --no-e2e --no-sub --steps 5 --warmup 2"
for spec in "c2 0 16" "c2 20 200" "c2 0 64"; do
  set -- $spec
  for rep in 1 2; do
    for L in - u3 u2; do
      [ -f $(lib $L) ] || continue
      WTAMD_LIB=$(lib $L) timeout 300 $B --config $1 --chroms $2 --mean-run $3 > $OUT/b.json 2> $OUT/b.err
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
