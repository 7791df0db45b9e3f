#!/bin/bash
# round 5: A/B of engine variants on ONE box (csrc/build.py build_engine_variant; "-" = the default library):
# C2 whole genome, C2 at mean run 200 and 1 (chr 19-22), C3 (var + stddev, 500 tracks, chr 1), sum of 500 on chr 21
# usage: tools/r5_ab.sh <outdir> <variant> ...
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd $R
if [ -n "$AB_TESTS" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x -k "$AB_TESTS" > $OUT/gpu_tests.log 2>&1
  tail -3 $OUT/gpu_tests.log
fi
line() { python -c "
import sys,json
l=[x for x in open('$1') if x.startswith('{')]
if not l: print('$2 NO LINE'); sys.exit()
r=json.loads(l[-1]); f=r['roofline']
print('$2', 'step_ms %.2f' % r['ms_per_step'], 'kernel_ms %.2f' % f['kernel_ms'], 'index_ms %.2f' % f['index_kernel_ms'], 'frac %.4f' % f['frac'], f['kernel'], 'auc', r['auc_check'])"; }
for v in "$@"; do
  L=$R/wiggletools_amd/csrc/libwiggletools_amd.so
  [ "$v" != "-" ] && L=$R/wiggletools_amd/csrc/libwiggletools_amd_$v.so
  n=$v; [ "$v" = "-" ] && n=default
  B="python bench.py --no-cpu-baseline --no-e2e --no-sub --steps ${AB_STEPS:-3} --warmup 1"
  WTAMD_LIB=$L timeout 600 $B --config c2 > $OUT/${n}_c2.log 2>&1; line $OUT/${n}_c2.log "$n c2"
  WTAMD_LIB=$L timeout 600 $B --config c2 --mean-run 200 > $OUT/${n}_c2_l200.log 2>&1; line $OUT/${n}_c2_l200.log "$n c2/l200"
  WTAMD_LIB=$L timeout 600 $B --config c3 > $OUT/${n}_c3.log 2>&1; line $OUT/${n}_c3.log "$n c3"
  if [ -n "$AB_MORE" ]; then
    WTAMD_LIB=$L timeout 600 $B --config c2 --mean-run 1 --chroms 18,19,20,21,23 > $OUT/${n}_c2_l1.log 2>&1; line $OUT/${n}_c2_l1.log "$n c2/l1"
    WTAMD_LIB=$L timeout 600 $B --config c2 --op sum --tracks 500 --chroms 20 > $OUT/${n}_sum500.log 2>&1; line $OUT/${n}_sum500.log "$n sum/500"
    WTAMD_LIB=$L timeout 600 $B --config c2 --values fullm --chroms 20 > $OUT/${n}_fullm.log 2>&1; line $OUT/${n}_fullm.log "$n fullm"
  fi
done
