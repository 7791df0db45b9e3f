#!/bin/bash
# round 6: the inflate kernel with / without the leading literal, on one box: parity, then the file leg (chromosome-sized, 100 files)
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_bwdev.py tests/test_bigwig.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
for L in nolead -; do
  if [ "$L" != "-" ]; then export WTAMD_LIB=$R/wiggletools_amd/csrc/libwiggletools_amd_$L.so; else unset WTAMD_LIB; fi
  WTAMD_E2E_REPS=2 timeout 600 python tools/e2e_bw_only.py ${MBP:-120} 100 2>/dev/null | grep "^{" | python -c "
import sys, json
for line in sys.stdin:
    j = json.loads(line)
    keys = [k for k in j if any(t in k for t in ('bp_per_s', 'seconds', 'decode', 'inflate', 'sections'))]
    print('$L', {k: (round(j[k], 4) if isinstance(j[k], float) else j[k]) for k in keys if not isinstance(j[k], (dict, list))})
"
done
done
