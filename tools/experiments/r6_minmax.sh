#!/bin/bash
# round 6, on the GPU box: Max / Min by the segment tree -- parity, then the bench record on chromosome 21 beside the general kernel
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6_minmax; mkdir -p $OUT; cd $R
python -u -m pytest tests -q -m gpu -x -k "min_max or one_sample or config_sized or (plans_agree and max) or ttest or two_sample" 2>&1 | tail -4
B="--no-cpu-baseline --no-e2e --no-sub --steps 2 --warmup 1"
for op in max min; do
  for nd in "" 1; do
    for run in 16 200; do
      tag="${op}_l${run}_${nd:-tree}"
      if [ -n "$nd" ]; then export WTAMD_NO_DELTA_MINMAX=1; else unset WTAMD_NO_DELTA_MINMAX; fi
      timeout 600 python bench.py --config c2 --op $op --chroms 0,20 --mean-run $run $B --full-record $OUT/$tag.json 2> $OUT/$tag.err | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); f = r['roofline']
print('$tag', 'step_ms', round(r['ms_per_step'], 2), 'kernel_ms', round(f['kernel_ms'], 2), f['kernel'], 'frac', round(f['frac'], 4), 'auc', r.get('auc_check'))"
    done
  done
done
