#!/bin/bash
# HBM traffic of wt_delta_kernel<mean> on chromosome 1 (C2 shape): the library against a variant build ($2), same box; FETCH_SIZE and WRITE_SIZE in separate passes
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-sub --chroms 0"
for v in "" $2; do
  if [ -n "$v" ]; then export WTAMD_LIB=$R/wiggletools_amd/csrc/libwiggletools_amd_$v.so; else unset WTAMD_LIB; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p_$ctr
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/p_$ctr -- python $R/bench.py $BARGS > $OUT/run_${v:-new}_$ctr.log 2>&1
    python - /tmp/p_$ctr $ctr "${v:-new}" <<'PY' | tee -a $OUT/traffic_ab.txt
import csv, glob, sys
vals = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wt_delta_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == sys.argv[2]:
            vals.append(float(r["Counter_Value"]))
scale = 2 if sys.argv[2] == "FETCH_SIZE" else 1      # gfx950: FETCH_SIZE reports half of the coalesced reads (MI355X guide)
print(sys.argv[3], sys.argv[2], "launches", len(vals), "GB per launch %.3f" % (sum(vals) / max(1, len(vals)) * 1024 * scale / 1e9))
PY
  done
done
