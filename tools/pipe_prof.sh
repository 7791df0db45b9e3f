#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel + memory-copy trace of the end-to-end (drop-in) leg,
# summarised into gpurun_out/pipe/overlap.json (copy / compute overlap of the streaming pipeline).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pipe
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/p_pipe -- python $R/tools/e2e_only.py ${1:-60} > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-1500
python $R/tools/pipe_overlap.py /tmp/p_pipe $OUT/overlap.json
python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/p_pipe/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("memcopy columns:", list(rows[0].keys()) if rows else None, "rows", len(rows))
    from collections import Counter
    print(Counter(r.get("Direction") for r in rows))
ev = []
for f in glob.glob("/tmp/p_pipe/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("kernel columns:", list(rows[0].keys()))
    for r in rows:
        if "wt_" in r["Kernel_Name"]:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-24:], r.get("Queue_Id"), r.get("Stream_Id")))
for f in glob.glob("/tmp/p_pipe/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?")[-14:] + " " + r.get("Bytes", r.get("Size", "?")), "-", r.get("Stream_Id")))
ev.sort()
g = [e for e in ev if "gather" in e[2]]
if g:
    t0 = g[len(g) // 2][0]
    sel = [e for e in ev if e[0] >= t0][:60]
    for e in sel:
        print("%10.3f %10.3f  %-28s q=%s s=%s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, e[2], e[3], e[4]))
PY
