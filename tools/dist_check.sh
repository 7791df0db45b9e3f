#!/bin/bash
# The N > 1 path of bench.py on ONE GPU: 2 ranks over gloo sharing device 0 -- rendezvous, the store work queue, the
# all_reduce of the scalars, the all_gather + ordered merge of the Pearson moments -- against the 1-rank line.
# Also the drop-in layer's WTAMD_DEVICES=2 on the whole-genome file leg (two pipes on the one GPU).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/dist
mkdir -p $OUT
cd $R
A="--moments --steps 2 --warmup 1 --no-sub --no-e2e --no-cpu-baseline --no-genome-files --scale 0.05"
timeout 600 python bench.py $A > $OUT/one_rank.log 2>&1; tail -1 $OUT/one_rank.log > $OUT/one_rank.json
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --dist-backend gloo $A > $OUT/two_ranks_gloo.log 2>&1
grep '^{' $OUT/two_ranks_gloo.log | tail -1 > $OUT/two_ranks_gloo.json
python - <<PY
import json
a = json.load(open("$OUT/one_rank.json")); b = json.load(open("$OUT/two_ranks_gloo.json"))
ok = True
def same(k, rel):
    global ok
    x, y = a[k], b[k]
    good = abs(x - y) <= rel * max(1.0, abs(x), abs(y))
    ok = ok and good
    print("%-22s 1 rank %.12g   2 ranks %.12g   %s" % (k, x, y, "ok" if good else "DIFFERENT"))
same("auc_check", 1e-12); same("output_runs", 0.0); same("pearson_tracks_0_1", 1e-9)
q = b.get("work_queue_check")
print("work queue:", q)
ok = ok and bool(q) and q["every_chromosome_exactly_once"] and min(q["chromosomes_per_rank"]) > 0
print("n_gpus", b["n_gpus"], "value", b["value"], "(two ranks on one GPU: not a scaling figure)")
json.dump({"one_rank": {k: a[k] for k in ("auc_check", "output_runs", "pearson_tracks_0_1", "value")},
           "two_ranks_gloo": {k: b[k] for k in ("auc_check", "output_runs", "pearson_tracks_0_1", "value", "work_queue_check", "n_gpus")}, "agree": ok},
          open("$OUT/dist_check.json", "w"), indent=1)
print("DIST CHECK", "PASSED" if ok else "FAILED")
PY
# two pipes inside the drop-in layer on the file leg
export WTAMD_BENCH_BWDIR=/dev/shm/wtamd_dist
WTAMD_DEVICES=1 timeout 300 python tools/genome_files.py 0.05 > $OUT/files_one_pipe.json 2> $OUT/files_one_pipe.err
WTAMD_DEVICES=2 timeout 300 python tools/genome_files.py 0.05 > $OUT/files_two_pipes.json 2> $OUT/files_two_pipes.err
python - <<PY
import json
a = json.loads(open("$OUT/files_one_pipe.json").read().strip().splitlines()[-1]); b = json.loads(open("$OUT/files_two_pipes.json").read().strip().splitlines()[-1])
print("file leg, runs: one pipe %d, two pipes %d; chromosomes %d / %d -> %s" % (a["warm"]["runs"], b["warm"]["runs"], a["warm"]["chromosomes_seen"], b["warm"]["chromosomes_seen"],
      "ok" if a["warm"]["runs"] == b["warm"]["runs"] and b["warm"]["chromosomes_seen"] == 24 else "DIFFERENT"))
PY
rm -rf /dev/shm/wtamd_dist
