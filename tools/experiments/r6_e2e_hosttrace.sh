#!/bin/bash
# round 6: the Feeder's own clock on the files -> result leg (GRCh38 x 0.25, 100 files): read-wait, submit, cadence
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export WTAMD_BENCH_BWDIR=/dev/shm/wt_r6_files
mkdir -p $WTAMD_BENCH_BWDIR
timeout 900 python $R/tools/genome_files.py 0.25 100 mean > /dev/null 2>&1
WTAMD_TRACE=1 timeout 900 python $R/tools/genome_files.py 0.25 100 mean > /tmp/ht.out 2> /tmp/ht.err
grep -c "bw read-wait" /tmp/ht.err
python - <<'PY'
import re
rw = []; plans = []
for l in open('/tmp/ht.err', errors='replace'):
    m = re.search(r'bw read-wait ([\d.]+) submit ([\d.]+) -> ([\d.]+)', l)
    if m: rw.append(tuple(float(x) for x in m.groups()))
    m = re.search(r'bw plan ([\d.]+) -> ([\d.]+)\s+\((\d+) sections, (\d+) bytes', l)
    if m: plans.append((float(m.group(1)), float(m.group(2)), int(m.group(3)), int(m.group(4))))
n = len(rw) // 2
w = rw[n:]          # the warm run
wait = [b - a for a, b, c in w]; sub = [c - b for a, b, c in w]
cad = [w[i + 1][1] - w[i][1] for i in range(len(w) - 1)]
print('warm run: %d submits; read-wait mean %.2f ms (median %.2f), submit mean %.2f ms, cadence mean %.2f ms (median %.2f)' % (len(w), sum(wait) / len(w), sorted(wait)[len(w) // 2], sum(sub) / len(w), sum(cad) / len(cad), sorted(cad)[len(cad) // 2]))
p = plans[len(plans) // 2:]
pl = [b - a for a, b, s, by in p]
print('plans: %d, mean %.2f ms, bytes per batch mean %.1f MB, sections mean %.0f' % (len(p), sum(pl) / len(p), sum(x[3] for x in p) / len(p) / 1e6, sum(x[2] for x in p) / len(p)))
# time from a plan's end (read started) to the submit that waited for it
PY
grep -a "^{" /tmp/ht.out | python -c "
import sys, json
j = json.loads(sys.stdin.read()); w = j['warm']; print({k: round(v, 3) if isinstance(v, float) else v for k, v in w.items() if not isinstance(v, (dict, list))})"
rm -rf $WTAMD_BENCH_BWDIR
