#!/bin/bash
# C5 (wilcoxon 50 v 50) on chromosome 21: engine variants on one box; also 40 v 60 and 64 v 64
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_setcmp_golden.py -q -m gpu -x -k "mwu or wilcoxon or two_sample or setcmp" > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
line() { python -c "
import sys,json
l=[x for x in open('$1') if x.startswith('{')]
if not l: print('$2 NO LINE'); sys.exit()
r=json.loads(l[-1]); f=r['roofline']
print('$2', 'step_ms %.2f' % r['ms_per_step'], 'kernel_ms %.2f' % f['kernel_ms'], 'frac %.4f' % f['frac'], f['kernel'], 'auc', r['auc_check'])"; }
for rep in 1 2; do
for v in "$@"; do
  L=$R/wiggletools_amd/csrc/libwiggletools_amd.so
  [ "$v" != "-" ] && L=$R/wiggletools_amd/csrc/libwiggletools_amd_$v.so
  n=$v; [ "$v" = "-" ] && n=default
  B="python bench.py --no-cpu-baseline --no-e2e --no-sub --steps 3 --warmup 1 --config c5 --chroms 20"
  WTAMD_LIB=$L timeout 300 $B > $OUT/${n}_c5_$rep.log 2>&1; line $OUT/${n}_c5_$rep.log "$n c5 50v50"
  WTAMD_LIB=$L timeout 300 $B --n-set0 40 > $OUT/${n}_c5b_$rep.log 2>&1; line $OUT/${n}_c5b_$rep.log "$n c5 40v60"
done
done
