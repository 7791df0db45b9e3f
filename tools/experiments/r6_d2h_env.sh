#!/bin/bash
# round 6: do the D2H copies of the file leg run as blit kernels or on the copy engines, and does it matter?  (GRCh38 x 0.25, 100 files)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export WTAMD_BENCH_BWDIR=/dev/shm/wt_r6_files
mkdir -p $WTAMD_BENCH_BWDIR
run() {
  echo "== $*"
  env "$@" timeout 900 python $R/tools/genome_files.py 0.25 100 mean 2>/dev/null | grep "^{" | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('cold %.3f s  warm %.3f s  warm bp/s %.4g  steady %.4g' % (j['cold']['seconds'], j['warm']['seconds'], j['warm_bp_per_s'], j['steady_bp_per_s']))"
}
run A=1
run A=1
run GPU_FORCE_BLIT_COPY_SIZE=0
run HSA_ENABLE_SDMA=0
run ROC_USE_FGS_KERNARG=0
run GPU_MAX_HW_QUEUES=8
run A=1
rm -rf $WTAMD_BENCH_BWDIR
