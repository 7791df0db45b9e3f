#!/bin/bash
# phase profiles of two -DWT_PROFILE -DWT_PROFILE_TAIL builds on one box: tools/experiments/r6_prof_ab.sh libA libB
R=$GRAFT_REPO_ROOT; cd $R
for L in "$@"; do
  for run in 16 200; do
    echo "== $L mean_run $run"
    WTAMD_LIB=$R/wiggletools_amd/csrc/libwiggletools_amd_$L.so timeout 600 python bench.py --config c2 --mean-run $run --chroms 20 --no-cpu-baseline --no-e2e --no-sub --steps 1 --warmup 1 2>&1 | grep -a "wt_profile" | tail -2 | cut -c1-300
  done
done
