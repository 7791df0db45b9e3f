#!/bin/bash
# round 5, GPU session 6: MWU walking v2 (every pair at its own pace, groups in value order): parity, chromosome 21, plans, whole genome, SQ counters
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5s6
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_setcmp_golden.py tests/test_dropin.py -q -m gpu -x -k "mwu or wilcoxon or two_sample or setcmp or multiset or survey or golden" > $OUT/gpu_tests_mwu.log 2>&1
tail -3 $OUT/gpu_tests_mwu.log
B="python bench.py --config c5 --chroms 20 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-sub"
pick() { python -c "
import json,sys
l=[x for x in open('$1') if x.startswith('{')][-1]; r=json.loads(l)
print('$2', 'ms_per_step %.2f' % r['ms_per_step'], 'kernel', r['roofline']['kernel'], 'kernel_ms %.2f' % r['roofline']['kernel_ms'], 'frac %.4f' % r['roofline']['frac'], 'runs', r['output_runs'], 'auc', r['auc_check'], 'W', r['config']['window_bp'], 'lds', r['config']['lds_bytes_per_workgroup'])"; }
timeout 300 $B > $OUT/c5_chr21_walk.log 2>&1; pick $OUT/c5_chr21_walk.log walk
WTAMD_NO_MWALK=1 timeout 300 $B > $OUT/c5_chr21_bitmap_table.log 2>&1; pick $OUT/c5_chr21_bitmap_table.log bitmap+table
WTAMD_WALK_S=8 timeout 300 $B > $OUT/c5_a.log 2>&1; pick $OUT/c5_a.log walk_S8
WTAMD_WALK_S=32 timeout 300 $B > $OUT/c5_b.log 2>&1; pick $OUT/c5_b.log walk_S32
WTAMD_WALK_T=128 WTAMD_WALK_S=32 timeout 300 $B > $OUT/c5_c.log 2>&1; pick $OUT/c5_c.log walk_T128_S32
WTAMD_WALK_T=128 WTAMD_WALK_S=16 timeout 300 $B > $OUT/c5_d.log 2>&1; pick $OUT/c5_d.log walk_T128_S16
WTAMD_WALK_T=64 WTAMD_WALK_S=32 timeout 300 $B > $OUT/c5_e.log 2>&1; pick $OUT/c5_e.log walk_T64_S32
timeout 600 python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-sub > $OUT/c5_genome_walk.log 2>&1; pick $OUT/c5_genome_walk.log genome_walk
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "VALUBusy SALUBusy SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/p_mw_$i -- python $R/bench.py --config c5 --chroms 20 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-sub > $OUT/mw_sq$i.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_mw_stats -- python $R/bench.py --config c5 --chroms 20 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-sub > $OUT/mw_stats.log 2>&1
python - <<PY
import csv, glob, json, os
out = "$OUT"
def pmc(dirn, match):
    per = {}
    for f in glob.glob(dirn + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if match not in r.get("Kernel_Name", ""): continue
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in per.items()}
s = {}
for i in (1, 2): s.update(pmc("/tmp/p_mw_%d" % i, "wt_mwalk_kernel"))
try:
    line = json.loads([l for l in open(os.path.join(out, "mw_sq1.log")) if l.startswith("{")][-1])
    s["output_runs_per_launch"] = line["output_runs"]
    s["valu_per_run"] = s["SQ_INSTS_VALU"]["mean"] / line["output_runs"]
    s["salu_per_run"] = s["SQ_INSTS_SALU"]["mean"] / line["output_runs"]
except Exception as e:
    s["error"] = repr(e)
for f in glob.glob("/tmp/p_mw_stats/**/*kernel_stats.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "wt_" in r.get("Name", "")]
    with open(os.path.join(out, "mw_kernel_stats.csv"), "w") as fh:
        if rows:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader()
            for r in rows: w.writerow(r)
    for r in rows[:2]: print(r["Name"][:70], r["Calls"], r["AverageNs"])
json.dump(s, open(os.path.join(out, "mw_sq.json"), "w"), indent=1)
print(json.dumps(s))
PY
