#!/bin/bash
# Runs on the GPU box: round-3 evidence -> gpurun_out/round3/
#   * rocprofv3 kernel stats + HBM PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) of the default bench line (C2)
#   * the BigWig-files-to-result leg (100 files x chromosome 1): kernel stats, HBM PMC and SQ counters of the
#     per-lane inflate kernel (wt_bw_inflate_kernel)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/round3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-sub"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python $R/bench.py $BARGS > $OUT/c2_stats_run.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -- python $R/bench.py $BARGS > $OUT/c2_fetch_run.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -- python $R/bench.py $BARGS > $OUT/c2_write_run.log 2>&1
export WTAMD_BENCH_BWDIR=/dev/shm/wtamd_r3 WTAMD_BENCH_NO_HOSTDEC=1
MBP=${BW_MBP:-248.956422}
if [ -z "$SKIP_BW" ]; then
python $R/tools/e2e_bw_only.py $MBP > $OUT/bw_plain.json 2> $OUT/bw_plain.err        # writes the files, cold + warm figures
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/b_stats -- python $R/tools/e2e_bw_only.py $MBP > $OUT/bw_stats_run.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/b_fetch -- python $R/tools/e2e_bw_only.py $MBP > $OUT/bw_fetch_run.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/b_write -- python $R/tools/e2e_bw_only.py $MBP > $OUT/bw_write_run.log 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" \
           "VALUBusy SALUBusy MemUnitBusy MemUnitStalled LDSBankConflict" \
           "SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/b_sq$i -- python $R/tools/e2e_bw_only.py $MBP > $OUT/bw_sq$i.log 2>&1
done
fi      # SKIP_BW
python - <<PY
import csv, glob, json, os
out = "$OUT"
def one(pat):
    f = glob.glob(pat, recursive=True)
    return f[0] if f else None
def stats(dirn, name):
    ks = one(dirn + "/**/*kernel_stats.csv")
    rows = list(csv.DictReader(open(ks))) if ks else []
    keep = [r for r in rows if "wt_" in r.get("Name", "") or "copyBuffer" in r.get("Name", "")]
    with open(os.path.join(out, name), "w") as fh:
        if rows:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader()
            for r in keep: w.writerow(r)
    return keep
def pmc(dirn, ctr):
    f = one(dirn + "/**/*counter_collection.csv")
    per = {}
    if f:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if "wt_" not in name or r.get("Counter_Name") != ctr: continue
            per.setdefault(name.replace("(anonymous namespace)::", "").split("(")[0][:70], []).append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "mean": sum(v) / len(v), "max": max(v)} for k, v in per.items()}
summary = {"c2": {"fetch": pmc("/tmp/p_fetch", "FETCH_SIZE"), "write": pmc("/tmp/p_write", "WRITE_SIZE")},
           "bigwig": {"fetch": pmc("/tmp/b_fetch", "FETCH_SIZE"), "write": pmc("/tmp/b_write", "WRITE_SIZE")}}
for r in stats("/tmp/p_stats", "c2_kernel_stats.csv"): print(r)
for r in stats("/tmp/b_stats", "bw_kernel_stats.csv"): print(r)
sq = {}
for f in glob.glob("/tmp/b_sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wt_bw_inflate" not in r.get("Kernel_Name", ""): continue
        if float(r.get("Grid_Size", 0) or 0) < 60000 * 1: pass
        sq.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
s = {k: sum(v) / len(v) for k, v in sq.items()}
if s.get("SQ_WAVE_CYCLES"):
    s["derived_valu_insts_per_wave_cycle"] = s.get("SQ_INSTS_VALU", 0) / s["SQ_WAVE_CYCLES"]
    s["derived_wait_any_share"] = s.get("SQ_WAIT_ANY", 0) / s["SQ_WAVE_CYCLES"]
summary["inflate_sq"] = s
json.dump(summary, open(os.path.join(out, "pmc_sq_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
PY
[ -z "$SKIP_BW" ] && tail -c 600 $OUT/bw_plain.json
