#!/bin/bash
# round 6, on the GPU box: A/B of two builds of the engine on ONE box, alternating (boxes differ by up to 10 %).
#   tools/r6_ab.sh <out> <libA> <libB> [specs...]     lib = "" for the product library, else libwiggletools_amd_<lib>.so
#   spec = "config chroms mean_run [op]"              TESTS=1: the difference-array parity tests on libB first
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; LA=$2; LB=$3; shift 3
mkdir -p $OUT
cd $R
lib() { if [ -n "$1" ] && [ "$1" != "-" ]; then echo $R/wiggletools_amd/csrc/libwiggletools_amd_$1.so; else echo $R/wiggletools_amd/csrc/libwiggletools_amd.so; fi; }
if [ -n "$TESTS" ]; then
  WTAMD_LIB=$(lib $LB) timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q ${TESTS_K:+-k "$TESTS_K"} 2>&1 | tail -3 | tee $OUT/tests.txt
fi
B="python bench.py --no-cpu-baseline --no-e2e --no-sub --steps ${STEPS:-5} --warmup 2"
[ $# -eq 0 ] && set -- "c2 20 16" "c2 0 16" "c2 20 200"
for spec in "$@"; do
  set -- $spec
  for rep in 1 2; do
    for L in "$LA" "$LB"; do
      WTAMD_LIB=$(lib $L) timeout 300 $B --config $1 --chroms $2 --mean-run $3 ${4:+--op $4} > $OUT/b.json 2> $OUT/b.err
      python - $OUT/b.json "$spec" "${L:--}" <<'PY' | tee -a $OUT/ab.txt
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line)
        r = j.get("roofline", {})
        print("%-8s %-22s step_ms %.4f kernel_ms %.4f frac %.4f  %s" % (sys.argv[3], sys.argv[2], j.get("ms_per_step"), r.get("kernel_ms"), r.get("frac"), r.get("kernel")))
PY
    done
  done
done
