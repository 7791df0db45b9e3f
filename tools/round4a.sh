#!/bin/bash
# GPU call A of round 4: box probe, -m gpu tests, inflate-kernel A/B on a whole-genome file set, first bench line.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4a
mkdir -p $OUT
cd $R
{ nproc; free -g; df -h /dev/shm /tmp; cat /sys/fs/cgroup/cpu.max 2>/dev/null; rocminfo | grep -m3 "Compute Unit\|Marketing"; } > $OUT/box.txt 2>&1
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1
tail -3 $OUT/gpu_tests.log
# ---- A/B of the inflate kernel over 100 files x 24 chromosomes at GRCh38 x 0.1 (309 Mbp); files written once
export WTAMD_BENCH_BWDIR=/dev/shm/wtamd_r4a
SCALE=${SCALE:-0.1}
run() { name=$1; shift; env "$@" timeout 600 python tools/genome_files.py $SCALE > $OUT/ab_$name.json 2> $OUT/ab_$name.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/ab_$name.json").read().strip().splitlines()[-1])
    c, w = d["cold"], d["warm"]
    print("%-14s cold %.3e warm %.3e steady %.3e | warm: %d batches, decode %.1f ms (%.2f ms/batch), kernels %.1f, submit %.0f wait %.0f open %.3f s | cold submit %.0f dev afresh %.1f GB | written %.0f s"
          % ("$name", d["bp_per_s"], d["warm_bp_per_s"], d.get("steady_bp_per_s") or 0, w["batches"], w["sum_device_decode_ms"], w["sum_device_decode_ms"] / max(w["batches"], 1),
             w["sum_kernel_ms"], w["host_submit_ms"], w["host_wait_ms"], w["open_seconds"], c["host_submit_ms"], c["device_afresh"]["bytes"] / 1e9, d["files_written_s"]))
except Exception as e:
    print("$name failed:", e, open("$OUT/ab_$name.err").read()[-600:])
PY
}
L=$R/wiggletools_amd/csrc
run new_ring8      WTAMD_X=1
run r3             WTAMD_LIB=$L/libwiggletools_amd_r3.so
run new_ring8_63k  WTAMD_BW_BATCH_SECTIONS=61504
run new_ring64     WTAMD_INFLATE_RING=64
run new_round2     WTAMD_LIB=$L/libwiggletools_amd_round2.so
run new_round8     WTAMD_LIB=$L/libwiggletools_amd_round8.so
run new_ring8_254k WTAMD_BW_BATCH_SECTIONS=246000
# kernel stats of the default variant
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bw -- python $R/tools/genome_files.py $SCALE > $OUT/stats_run.log 2>&1
f=$(find /tmp/p_bw -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "wt_\|copyBuffer\|Name" $f | cut -c1-260 > $OUT/bw_kernel_stats.csv
cat $OUT/bw_kernel_stats.csv | cut -c1-200 | head -12
rm -rf /dev/shm/wtamd_r4a
cd $R
# ---- first bench line: C2 resident + e2e legs + whole-genome files (no sub-records, no CPU baseline yet)
unset WTAMD_BENCH_BWDIR
timeout 1200 python bench.py --steps 3 --warmup 1 --no-sub --no-cpu-baseline > $OUT/bench_nosub.log 2>&1
tail -1 $OUT/bench_nosub.log > $OUT/bench_nosub.json
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_nosub.json"))
    print("C2 value %.3e ms/step %.1f frac %.3f | bulk %.3e | genome files:" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("value_e2e_bulk") or 0), json.dumps(d["config"].get("north_star_files_to_result")))
    g = d.get("e2e_bigwig_genome", {})
    print({k: g.get(k) for k in ("genome_scale", "bp", "file_bytes", "files_written_s", "generate_s", "error")})
    print("cold", g.get("cold")); print("warm", g.get("warm")); print("bench_seconds", d.get("bench_seconds"))
except Exception as e:
    print("bench failed", e, open("$OUT/bench_nosub.log").read()[-1500:])
PY
