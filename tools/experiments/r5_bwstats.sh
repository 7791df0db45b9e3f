#!/bin/bash
# kernel stats of the file leg (GRCh38 x 0.25, 100 files x 24 chromosomes): rocprofv3 --kernel-trace --stats
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5s12
mkdir -p $OUT
cd $R
export WTAMD_BENCH_BWDIR=/dev/shm/wtamd_r5
timeout 900 python tools/genome_files.py 0.25 > $OUT/files_plain.json 2> $OUT/files_plain.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/b_stats -- python $R/tools/genome_files.py 0.25 > $OUT/bw_stats_run.log 2>&1
python - <<PY
import csv, glob, os
for f in glob.glob("/tmp/b_stats/**/*kernel_stats.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "wt_" in r.get("Name", "") or "copyBuffer" in r.get("Name", "")]
    with open(os.path.join("$OUT", "bw_kernel_stats.csv"), "w") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader()
        for r in rows: w.writerow(r)
    for r in rows[:10]: print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
python - <<PY
import json
r = json.loads(open("$OUT/files_plain.json").read().strip().splitlines()[-1])
print("sections", r["sections"], "batches", r["warm"]["batches"], "warm decode ms", r["warm"]["sum_device_decode_ms"], "warm s", r["warm"]["seconds"])
PY
rm -rf /dev/shm/wtamd_r5
