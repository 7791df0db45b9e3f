"""RCCL readiness on a ONE-GPU box (`-m gpu`): `bench.py --force-dist --dist-backend nccl` launched by torch.distributed.run with a
world of one rank -- the NCCL (= RCCL on ROCm) communicator is created, the chromosome work queue runs through the rendezvous
store's counter, and the scalar collectives of BASELINE.json's config 5 (all_reduce of the AUC sums, all_gather of
the Pearson moments) execute on the device.  Same figures as the plain one-rank run.  The world-2 logic is covered on CPU over
gloo (tests/test_dist_gloo.py); no scaling number comes from here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--gpus", "1", "--config", "c5", "--chroms", "20", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e", "--no-sub"]


def _line(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-2000:]
    compact = json.loads(lines[-1])             # the driver's line: the last one of stdout, small
    assert len(lines[-1]) < 6000 and "roofline" in compact
    # the full record (every figure at full precision) goes to stderr and to --full-record
    full = [l for l in r.stderr.splitlines() if l.startswith("bench full record: ")]
    assert full, r.stderr[-2000:]
    return json.loads(full[-1][len("bench full record: "):])


@pytest.mark.gpu
@pytest.mark.timeout(1300)
def test_bench_c5_through_rccl_world_of_one():
    port = str(29600 + os.getpid() % 300)
    dist = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                  "--master-port", port, "bench.py"] + COMMON + ["--force-dist", "--dist-backend", "nccl"])
    plain = _line([sys.executable, "bench.py"] + COMMON + ["--moments"])
    c = dist["collectives"]
    assert c and c["backend"] == "nccl" and c["world"] == 1 and len(c["ran"]) == 5
    q = dist["work_queue_check"]
    assert q["every_chromosome_exactly_once"] and q["chromosomes_per_rank"] == [1] and "store" in q["how"]
    assert plain["collectives"] is None
    assert dist["output_runs"] == plain["output_runs"] > 1e7
    assert dist["auc_check"] == plain["auc_check"]
    assert dist["pearson_tracks_0_1"] is not None and abs(dist["pearson_tracks_0_1"] - plain["pearson_tracks_0_1"]) < 1e-12
    assert dist["roofline"]["kernel"].startswith("wt_")
