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


def _moments_sequential(tile, inplay, start, finish, dx, dy):
    """The reference's run-by-run update of {count, sum_X, sum_Y, T_XX, T_XY, T_YY}
    (statistics.c:442-456, weights = run lengths) over a 2-track Multiplexer tile -- test helper
    standing in for the device kernel (wt_pearson_kernel) in the GPU-less world-2 test."""
    n = sx = sy = txx = txy = tyy = 0.0
    for r in range(len(start)):
        X = tile[r, 0] if inplay[r, 0] else dx
        Y = tile[r, 1] if inplay[r, 1] else dy
        L = float(finish[r] - start[r])
        if n > 0:
            nn = n + L
            omx, nmx, omy, nmy, ratio = sx / n, sx / nn, sy / n, sy / nn, n / nn
            txy += (nmx * omy + ratio * X * Y - nmx * Y - nmy * X) * L
            txx += (nmx * (omx - 2 * X) + ratio * X * X) * L
            tyy += (nmy * (omy - 2 * Y) + ratio * Y * Y) * L
        n += L; sx += X * L; sy += Y * L
    return np.array([n, sx, sy, txx, txy, tyy])


def _pearson_worker(rank, world, port, q):
    """Chromosomes dealt out by the shared work queue (a counter in the rendezvous store, as
    bench.py --shard genome does); per-chromosome Pearson moments; all_gather; ordered merge."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    store = dist.distributed_c10d._get_default_store()
    from emu import emu
    from wiggletools_amd import shard
    from wiggletools_amd.runlists import RunLists, synth
    t = synth(2, [4000, 2500, 300, 3500, 900], mean_run=7, seed=4, gap_prob=0.1)
    N = 2
    table = np.zeros((t.n_chrom, 6))
    mine = []
    while True:
        c = store.add("wt_queue_0", 1) - 1          # host-side work queue: the next chromosome
        if c >= t.n_chrom:
            break
        mine.append(c)
        a, b = int(t.seg_off[c * N]), int(t.seg_off[c * N + 2])
        one = RunLists(1, 2, t.seg_off[c * N:c * N + 3] - t.seg_off[c * N], t.start[a:b], t.finish[a:b], t.value[a:b], t.defaults)
        (chrom, s, f, tile, ip), _ = emu.reduce(one, "sum", multiplex=True)
        table[c] = _moments_sequential(tile, ip, s, f, t.defaults[0], t.defaults[1])
    full = shard.allgather_moments(table)
    owners = [None] * world
    dist.all_gather_object(owners, mine)
    if rank == 0:
        q.put((full, owners))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pearson_gather_and_ordered_merge(oracle):
    """The genome-wide Pearson of a 2-track set sharded by chromosome over 2 ranks: 6 doubles per
    shard through all_gather, merged pairwise in genome order == the unsharded sequential result."""
    import torch.multiprocessing as mp
    from wiggletools_amd import shard
    from wiggletools_amd.runlists import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pearson_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, owners = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(owners[0] + owners[1]) == [0, 1, 2, 3, 4]             # every chromosome exactly once
    t = synth(2, [4000, 2500, 300, 3500, 900], mean_run=7, seed=4, gap_prob=0.1)
    exp = oracle.pearson(t.as_dict())
    got = shard.pearson_from_moments([full[c] for c in range(t.n_chrom)])
    assert abs(got - exp) <= 1e-9 * abs(exp), (got, exp)
    # the order matters only to rounding; a wrong merge formula would not survive a shuffle either
    shuffled = shard.pearson_from_moments([full[c] for c in (3, 0, 4, 2, 1)])
    assert abs(shuffled - exp) <= 1e-9 * abs(exp)


def _wilcoxon_worker(rank, world, port, q):
    """BASELINE config C5 as specified: wilcoxon 50 v 50, chromosomes dealt by the shared work queue, per-position
    output concatenated in chromosome order, plus the scalar `AUC wilcoxon ...` through an all_reduce."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    store = dist.distributed_c10d._get_default_store()
    from emu import emu
    from oracle import oracle as O
    from wiggletools_amd import shard
    from wiggletools_amd.runlists import RunLists, synth
    N = 100
    t = synth(N, [700, 450, 120, 600], mean_run=6, seed=12, gap_prob=0.05, value_levels=9)
    order = sorted(range(t.n_chrom), key=lambda c: -(int(t.seg_off[(c + 1) * N]) - int(t.seg_off[c * N])))   # largest first
    mine, auc_local, runs_local = {}, 0.0, 0.0
    while True:
        k = store.add("wt_queue_c5", 1) - 1
        if k >= len(order):
            break
        c = order[k]
        a, b = int(t.seg_off[c * N]), int(t.seg_off[(c + 1) * N])
        one = RunLists(1, N, t.seg_off[c * N:(c + 1) * N + 1] - t.seg_off[c * N], t.start[a:b], t.finish[a:b], t.value[a:b], t.defaults)
        got, _ = emu.reduce(one, "mwu", n_set0=50)
        mine[c] = tuple(np.asarray(x) for x in got)
        auc_local += O.auc(got[1], got[2], got[3])
        runs_local += len(got[0])
    auc, runs = shard.allreduce_scalars([auc_local, runs_local])
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        q.put((auc, runs, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_wilcoxon_50v50_sharded_with_auc_gather(oracle):
    import torch.multiprocessing as mp
    from wiggletools_amd.runlists import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wilcoxon_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    auc, runs, gathered = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    t = synth(100, [700, 450, 120, 600], mean_run=6, seed=12, gap_prob=0.05, value_levels=9)
    c, s, f, v = oracle.reduce(t.as_dict(), "mwu", n_set0=50)
    owned = {}
    for part in gathered:
        assert not (set(part) & set(owned))            # every chromosome exactly once
        owned.update(part)
    assert sorted(owned) == list(range(t.n_chrom))
    # per-position output concatenated in chromosome order == the unsharded run list, bit for bit
    S = np.concatenate([owned[k][1] for k in range(t.n_chrom)])
    F = np.concatenate([owned[k][2] for k in range(t.n_chrom)])
    V = np.concatenate([owned[k][3] for k in range(t.n_chrom)])
    C = np.concatenate([np.full(len(owned[k][1]), k) for k in range(t.n_chrom)])
    assert np.array_equal(C, c) and np.array_equal(S, s) and np.array_equal(F, f)
    assert np.array_equal(V.view(np.uint64), v.view(np.uint64))
    assert runs == len(s)
    ref = oracle.auc(s, f, v)
    assert abs(auc - ref) <= 1e-9 * abs(ref)
