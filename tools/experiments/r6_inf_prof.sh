#!/bin/bash
# round 6: rocprofv3 kernel stats of the file leg (chromosome 1 x 100 files) for two builds of wt_bwdev.hip
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for L in "$@"; do
  if [ "$L" != "-" ]; then export WTAMD_LIB=$R/wiggletools_amd/csrc/libwiggletools_amd_$L.so; else unset WTAMD_LIB; fi
  rm -rf /tmp/pinf_$L
  WTAMD_E2E_REPS=2 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pinf_$L -- python $R/tools/e2e_bw_only.py ${MBP:-248.9} 100 > /tmp/pinf_$L.log 2>&1
  grep "^{" /tmp/pinf_$L.log | python -c "
import sys, json
for line in sys.stdin:
    j = json.loads(line); print('$L', 'bp_per_s %.4g warm %.4g steady %.4g' % (j.get('bp_per_s', 0), j.get('warm_bp_per_s', 0), j.get('steady_bp_per_s', 0)))"
  f=$(find /tmp/pinf_$L -name "*kernel_stats.csv" | head -1)
  python - "$f" "$L" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "wt_bw" in r["Name"] or "delta" in r["Name"]:
        print(sys.argv[2], r["Name"][:60], "calls", r["Calls"], "avg_ms %.3f" % (float(r["AverageNs"]) / 1e6), "pct", r["Percentage"])
PY
  mkdir -p $R/gpurun_out/inf; cp "$f" $R/gpurun_out/inf/kernel_stats_$L.csv
done
