#!/bin/bash
# round 6: batches in flight on the files -> result leg (GRCh38 x SCALE, 100 files)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export WTAMD_BENCH_BWDIR=/dev/shm/wt_r6_files
mkdir -p $WTAMD_BENCH_BWDIR
run() {
  env "$@" timeout 900 python $R/tools/genome_files.py ${SCALE:-0.25} 100 mean 2>/dev/null | grep "^{" | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$*', 'cold %.3f s  warm %.3f s  warm bp/s %.4g  steady %.4g  host_wait %.0f ms' % (j['cold']['seconds'], j['warm']['seconds'], j['warm_bp_per_s'], j['steady_bp_per_s'], j['warm']['host_wait_ms']))"
}
run A=1 > /dev/null
for d in 2 3 4 2 3 4; do run WTAMD_PIPE_DEPTH=$d; done
rm -rf $WTAMD_BENCH_BWDIR
