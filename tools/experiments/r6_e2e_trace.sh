#!/bin/bash
# round 6: where the files -> result leg spends its time now (GRCh38 x SCALE, 100 files): kernel + memory-copy trace
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export WTAMD_BENCH_BWDIR=/dev/shm/wt_r6_files
mkdir -p $WTAMD_BENCH_BWDIR
timeout 900 python $R/tools/genome_files.py ${SCALE:-0.25} 100 mean 2>/dev/null | grep "^{" | python -c "
import sys, json
j = json.loads(sys.stdin.read())
for k in ('cold', 'warm'):
    w = j[k]; print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in w.items() if not isinstance(b, (dict, list))})"
rm -rf /tmp/pe2e
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/pe2e -- python $R/tools/genome_files.py ${SCALE:-0.25} 100 mean > /tmp/pe2e.log 2>&1
ls /tmp/pe2e/*/ | head
for f in $(find /tmp/pe2e -name "*_stats.csv"); do echo "== $f"; head -8 $f | cut -c1-200; done
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/pe2e/**/*memory_copy_trace.csv', recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    print(rows[0].keys())
    by = {}
    for r in rows:
        k = r.get('Direction') or r.get('Kind')
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        b = by.setdefault(k, [0, 0.0, 0]); b[0] += 1; b[1] += d; b[2] += int(r.get('Bytes') or r.get('Size') or 0)
    for k, (n, ms, by_) in by.items(): print(k, 'copies', n, 'ms', round(ms, 1), 'GB', round(by_ / 1e9, 2), 'GB/s while copying', round(by_ / 1e6 / max(ms, 1e-9), 1))
PY
rm -rf $WTAMD_BENCH_BWDIR
