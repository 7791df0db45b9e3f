"""N>1 path on CPU: world_size-2 gloo.  Each rank takes its shard (run-start ranges from
wiggletools_amd.shard.plan_shards), evaluates it, and the scalar gather runs through
torch.distributed.  No GPU here, so the per-shard evaluation is done by the kernel-logic
emulator (tests/emu) -- the checker stands in for the device, the thing under test is the
sharding / ordering / collective logic that bench.py and multi-GPU callers use."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, seed, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu import emu
    from oracle import oracle as O
    from wiggletools_amd import shard
    from wiggletools_amd.runlists import synth
    t = synth(6, [9000, 300, 5000], mean_run=5, seed=seed, gap_prob=0.2)
    ranges = shard.plan_shards(shard.chrom_extents(t), world)
    got, info = emu.reduce(t, "mean", ranges=ranges[rank])
    auc_local = O.auc(got[1], got[2], got[3])
    bp_local = float((got[2] - got[1]).sum())
    auc, bp, runs = shard.allreduce_scalars([auc_local, bp_local, float(len(got[0]))])
    gathered = [None] * world
    dist.all_gather_object(gathered, tuple(np.asarray(x) for x in got))
    if rank == 0:
        q.put((auc, bp, runs, gathered))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("seed", [1, 2])
def test_two_rank_sharded_mean_matches_unsharded(oracle, seed):
    import torch.multiprocessing as mp
    from wiggletools_amd import shard
    from wiggletools_amd.runlists import synth
    from helpers import assert_runs_equal
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, seed, q)) for r in range(2)]
    for p in procs:
        p.start()
    auc, bp, runs, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    t = synth(6, [9000, 300, 5000], mean_run=5, seed=seed, gap_prob=0.2)
    exp = oracle.reduce(t.as_dict(), "mean")
    cat = shard.concat_runs(gathered, t.n_chrom)
    assert_runs_equal(cat, exp, 0.0, "sharded concat")
    assert runs == len(exp[0])
    assert bp == float((exp[2] - exp[1]).sum())
    ref_auc = oracle.auc(exp[1], exp[2], exp[3])
    assert abs(auc - ref_auc) <= 1e-9 * abs(ref_auc)


def test_plan_shards_covers_every_position_once():
    from wiggletools_amd import shard
    ext = [(1, 1001), None, (50, 60), (7, 100007)]
    for world in (1, 2, 3, 8):
        ranges = shard.plan_shards(ext, world)
        for c, e in enumerate(ext):
            if e is None:
                continue
            owned = sorted((r[c] for r in ranges if r[c][0] < r[c][1]), key=lambda x: x[0])
            assert owned[0][0] <= e[0] and owned[-1][1] >= e[1]
            for a, b in zip(owned, owned[1:]):
                assert a[1] == b[0]
