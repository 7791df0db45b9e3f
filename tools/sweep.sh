#!/bin/bash
# Runs on the GPU box: quick sweeps of experiment builds / knobs on one real chromosome (chr21-sized, id 20).
# usage: tools/sweep.sh "<config> <lib-suffix or -> [ENV=val ...]" ...
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  set -- $spec
  cfg=$1; lib=$2; shift 2
  L=$GRAFT_REPO_ROOT/wiggletools_amd/csrc/libwiggletools_amd.so
  [ "$lib" != "-" ] && L=$GRAFT_REPO_ROOT/wiggletools_amd/csrc/libwiggletools_amd_$lib.so
  env WTAMD_LIB=$L "$@" timeout 300 python bench.py --config $cfg --chroms 20 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$spec:', 'ms/step %.3f' % d['ms_per_step'], 'kernel_ms %.3f' % r['kernel_ms'], 'per31Mbp %.2f' % (d['ms_per_step']*31e6/46.7e6), 'auc', d.get('auc_check'))"
done
