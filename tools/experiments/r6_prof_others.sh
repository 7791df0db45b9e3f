#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
L=$R/wiggletools_amd/csrc/libwiggletools_amd_prof.so
for a in "--config c3" "--config c5 --op ttest" "--config c2 --op max" "--config c2 --tracks 500 --op sum"; do
  echo "== $a"
  WTAMD_LIB=$L timeout 600 python bench.py $a --chroms 20 --no-cpu-baseline --no-e2e --no-sub --steps 1 --warmup 1 2>&1 | grep -a "wt_profile" | tail -2 | cut -c1-300
done
