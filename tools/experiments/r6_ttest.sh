#!/bin/bash
# round 6, on the GPU box: TTestReduction by difference arrays -- parity tests, then the bench record on chromosome 21 and the genome,
# the general kernel (WTAMD_NO_DELTA_TTEST=1) beside it
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6_ttest; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -q -m gpu -x -k "ttest or two_sample or kernel_resources" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
B="--config c5 --op ttest --no-cpu-baseline --no-e2e --no-sub --steps 2 --warmup 1"
for chroms in 20 ""; do
  for nd in "" 1; do
    tag="chr${chroms:-all}_${nd:-delta}"
    if [ -n "$nd" ]; then export WTAMD_NO_DELTA_TTEST=1; else unset WTAMD_NO_DELTA_TTEST; fi
    timeout 600 python bench.py $B ${chroms:+--chroms $chroms} --full-record $OUT/$tag.json 2> $OUT/$tag.err | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); f = r['roofline']
print('$tag', 'step_ms', round(r['ms_per_step'], 2), 'kernel_ms', round(f['kernel_ms'], 2), f['kernel'], 'frac', round(f['frac'], 4), 'auc', r.get('auc_check'), 'patched', r['config'].get('patched_windows_per_step'))"
  done
done
