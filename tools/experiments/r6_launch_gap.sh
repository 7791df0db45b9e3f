#!/bin/bash
# round 6: what the bench's events see around a wt_delta_kernel launch that rocprofv3's kernel duration does not
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pgap
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pgap -- python $R/bench.py --config c2 --chroms 0,10,20 --no-cpu-baseline --no-e2e --no-sub --steps 3 --warmup 1 > /tmp/pgap.log 2>&1
grep "^{" /tmp/pgap.log | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('bench events: kernel_ms per pass', j['roofline']['kernel_ms'], 'index', j['roofline']['index_kernel_ms'])"
python - <<'PY'
import csv, glob
kt = glob.glob('/tmp/pgap/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(kt))))
tot = 0
for i, (s, e, n) in enumerate(rows):
    if 'wt_delta_kernel' in n:
        prev = rows[i - 1]; nxt = rows[i + 1] if i + 1 < len(rows) else None
        print('delta %.3f ms | before: %s ended %.1f us earlier (ran %.1f us) | after: %s starts %.1f us later' % ((e - s) / 1e6, prev[2][:28], (s - prev[1]) / 1e3, (prev[1] - prev[0]) / 1e3, nxt[2][:28] if nxt else '-', (nxt[0] - e) / 1e3 if nxt else 0))
        tot += e - s
print('sum of delta kernel durations over all passes: %.3f ms' % (tot / 1e6))
PY
