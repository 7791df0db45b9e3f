#!/bin/bash
# the difference-array kernel's workgroup size against the data's density: 1024 lanes (one workgroup per CU, 8192-bp windows), 512 (two, 4096 bp), 256 (four, 2048 bp)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
line() { python -c "
import sys,json
l=[x for x in open('$1') if x.startswith('{')]
if not l: print('$2 NO LINE'); sys.exit()
r=json.loads(l[-1]); f=r['roofline']
print('$2', 'step_ms %.2f' % r['ms_per_step'], 'kernel_ms %.2f' % f['kernel_ms'], 'index_ms %.2f' % f['index_kernel_ms'], 'frac %.4f' % f['frac'], 'W', r['config']['window_bp'])"; }
B="python bench.py --no-cpu-baseline --no-e2e --no-sub --steps 2 --warmup 1 --config c2"
for T in 1024 512 256; do
  for mr in 200 64 16; do
    WTAMD_DELTA_T=$T timeout 600 $B --mean-run $mr --chroms 0,20 > $OUT/T${T}_l$mr.log 2>&1; line $OUT/T${T}_l$mr.log "T=$T l=$mr"
  done
done
