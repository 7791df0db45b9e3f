#!/bin/bash
# round 6: the GPU's timeline of the files -> result leg (GRCh38 x 0.25, 100 files): what lies between two inflate launches
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export WTAMD_BENCH_BWDIR=/dev/shm/wt_r6_files
mkdir -p $WTAMD_BENCH_BWDIR
timeout 900 python $R/tools/genome_files.py 0.25 100 mean > /dev/null 2>&1      # writes the files
rm -rf /tmp/pe2e
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/pe2e -- python $R/tools/genome_files.py 0.25 100 mean > /tmp/pe2e.log 2>&1
python - <<'PY'
import csv, glob
kt = glob.glob('/tmp/pe2e/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(kt)))
ev = []
for r in rows:
    n = r['Kernel_Name']
    short = 'inflate' if 'inflate' in n else 'count' if 'bw_count' in n else 'scatter' if 'bw_scatter' in n else 'scan' if 'bw_scan' in n else 'delta' if 'delta' in n else 'index' if 'index' in n else 'copyBuffer' if 'copyBuffer' in n else 'export' if 'export' in n else 'other'
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short))
mc = glob.glob('/tmp/pe2e/**/*memory_copy_trace.csv', recursive=True)
for f in mc:
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'H2D'))
ev.sort()
inf = [e for e in ev if e[2] == 'inflate']
# the second (warm) run: the last half of the inflate launches
half = inf[len(inf) // 2:]
t0, t1 = half[0][0], half[-1][1]
print('warm run: %d inflate launches over %.1f ms; inflate busy %.1f ms' % (len(half), (t1 - t0) / 1e6, sum(e[1] - e[0] for e in half) / 1e6))
gaps = [(half[i + 1][0] - half[i][1]) / 1e6 for i in range(len(half) - 1)]
print('gap between inflate launches: mean %.2f ms  median %.2f  max %.2f' % (sum(gaps) / len(gaps), sorted(gaps)[len(gaps) // 2], max(gaps)))
# what runs inside the gaps (non-inflate kernels on any stream), summed
busy = {}
for i in range(len(half) - 1):
    a, b = half[i][1], half[i + 1][0]
    for s, e, n in ev:
        if n in ('inflate',) or e <= a or s >= b: continue
        busy[n] = busy.get(n, 0) + (min(e, b) - max(s, a)) / 1e6
print('inside the gaps, per gap (ms):', {k: round(v / len(gaps), 3) for k, v in busy.items()})
h2d = [e for e in ev if e[2] == 'H2D' and e[0] >= t0 and e[1] <= t1]
print('H2D copies in the warm run: %d, mean %.2f ms, busy %.1f ms' % (len(h2d), sum(e[1] - e[0] for e in h2d) / 1e6 / max(len(h2d), 1), sum(e[1] - e[0] for e in h2d) / 1e6))
# does an inflate start right when its H2D ends?
import bisect
ends = sorted(e[1] for e in h2d)
lag = []
for s, e, n in half:
    i = bisect.bisect_right(ends, s) - 1
    if i >= 0: lag.append((s - ends[i]) / 1e6)
print('inflate start minus the latest H2D end before it: mean %.2f ms median %.2f' % (sum(lag) / len(lag), sorted(lag)[len(lag) // 2]))
PY
rm -rf $WTAMD_BENCH_BWDIR
