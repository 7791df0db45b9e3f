#!/bin/bash
# quick check of a change to the difference-array kernels: the parity tests that touch them, then C2 / C3 kernel times at three run lengths
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
[ -n "$LIBV" ] && export WTAMD_LIB=$R/wiggletools_amd/csrc/libwiggletools_amd_$LIBV.so
if [ -z "$NOTEST" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/tests.txt; fi
B="python bench.py --no-cpu-baseline --no-e2e --no-sub --steps 5 --warmup 2"
for spec in "c2 20 16" "c2 20 200" "c2 20 64" "c2 0 16" "c3 20 16"; do
  set -- $spec
  timeout 300 $B --config $1 --chroms $2 --mean-run $3 > $OUT/b.json 2> $OUT/b.err
  python - $OUT/b.json "$spec" <<'PY' | tee -a $OUT/quick.txt
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line)
        r = j.get("roofline", {})
        print(sys.argv[2], "ms_per_step %.4f kernel_ms %.4f frac %.4f" % (j.get("ms_per_step"), r.get("kernel_ms"), r.get("frac")))
PY
done
