#!/bin/bash
# phase profile (-DWT_PROFILE build: cycles of wave 0 per phase) of a reducer on chromosome 21:  tools/r6_prof.sh "<bench args>" [lib]
R=$GRAFT_REPO_ROOT; cd $R
LIB=${2:-$R/wiggletools_amd/csrc/libwiggletools_amd_prof.so}
WTAMD_LIB=$LIB timeout 600 python bench.py $1 --chroms 20 --no-cpu-baseline --no-e2e --no-sub --steps 1 --warmup 1 2>&1 | grep -a "wt_profile\|^{" | tail -4 | cut -c1-400
