#!/bin/bash
# phase cycles of the difference-array kernels (a -DWT_PROFILE -DWT_PROFILE_TAIL build of the engine): C2 on chromosome 21 and 1, l = 200, C3
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
L=$R/wiggletools_amd/csrc/libwiggletools_amd_$2.so
B="python bench.py --no-cpu-baseline --no-e2e --no-sub --steps 2 --warmup 1"
WTAMD_LIB=$L timeout 300 $B --config c2 --chroms 20 2>&1 | grep "wt_profile" | tail -2 | tee $OUT/prof_c2_chr21.txt
WTAMD_LIB=$L timeout 300 $B --config c2 --chroms 0 2>&1 | grep "wt_profile" | tail -1 | tee $OUT/prof_c2_chr1.txt
WTAMD_LIB=$L timeout 300 $B --config c2 --chroms 20 --mean-run 200 2>&1 | grep "wt_profile" | tail -1 | tee $OUT/prof_c2_l200.txt
WTAMD_LIB=$L timeout 300 $B --config c2 --chroms 20 --mean-run 1 2>&1 | grep "wt_profile" | tail -1 | tee $OUT/prof_c2_l1.txt
WTAMD_LIB=$L timeout 300 $B --config c3 2>&1 | grep "wt_profile" | tail -2 | tee $OUT/prof_c3.txt
