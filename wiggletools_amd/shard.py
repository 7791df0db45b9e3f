"""Multi-GPU sharding of the hot path.

Every reducer is stateless across positions and the Multiplexer never looks across
chromosomes (reference multiplexer.c:45,56-73), so the genome is cut into `world`
contiguous pieces of run STARTS -- whole chromosomes or ranges inside a chromosome --
one piece per rank / GPU.  No collective is needed on the data path: per-position
outputs concatenate in rank order.  Only genome-wide scalars (AUC, covered bp, run
counts) are combined, with one all_reduce of a few doubles (RCCL over xGMI on the
GPU node; gloo in the CPU tests).

The reference does the same by region with `seek` + one LSF job per 30 Mbp
(reference python/wiggletools/parallelWiggleTools.py:66-68,103-113).
"""
import numpy as np

INT32_MAX = 2 ** 31 - 1


def chrom_extents(rl):
    """Per chromosome (first start, last finish) over all tracks of a RunLists, or None if empty."""
    out = []
    N = rl.n_tracks
    for c in range(rl.n_chrom):
        lo, hi = None, None
        for i in range(N):
            a, b = rl.seg_off[c * N + i], rl.seg_off[c * N + i + 1]
            if b > a:
                s, f = int(rl.start[a]), int(rl.finish[b - 1])
                lo = s if lo is None else min(lo, s)
                hi = f if hi is None else max(hi, f)
        out.append(None if lo is None else (lo, hi))
    return out


def plan_shards(extents, world):
    """Cuts the genome (list of per-chromosome (lo, hi) or None) into `world` pieces of about
    equal span.  Returns ranges[rank][chrom] = (lo, hi) run-start bounds; a chromosome a rank
    does not touch gets the empty range (0, 0)."""
    spans = [0 if e is None else e[1] - e[0] for e in extents]
    total = sum(spans)
    ranges = [[(0, 0)] * len(extents) for _ in range(world)]
    if total == 0:
        return ranges
    # genome coordinate g in [0,total): cut points at k*total/world
    cuts = [(total * k) // world for k in range(world + 1)]
    base = 0
    for c, e in enumerate(extents):
        if e is None:
            continue
        lo, hi = e
        for r in range(world):
            a = max(cuts[r], base)
            b = min(cuts[r + 1], base + spans[c])
            if a < b:
                r_lo = lo + (a - base)
                r_hi = lo + (b - base)
                # first / last piece of a chromosome are open-ended
                ranges[r][c] = (-INT32_MAX if a == base else r_lo, INT32_MAX if b == base + spans[c] else r_hi)
        base += spans[c]
    return ranges


def concat_runs(pieces, n_chrom):
    """pieces[rank] = (chrom, start, finish, value) -> genome-ordered concatenation."""
    cat = [[], [], [], []]
    for c in range(n_chrom):
        for g in pieces:
            m = g[0] == c
            for k in range(4):
                cat[k].append(np.asarray(g[k])[m])
    return tuple(np.concatenate(x) if x else np.zeros(0) for x in cat)


def allreduce_scalars(values, group=None):
    """Sum of a few genome-wide float64 scalars over all ranks (the only collective of the path)."""
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [float(x) for x in t.tolist()]


def merge_moments(a, b):
    """a (+) b for two Pearson moment vectors {n, sum_X, sum_Y, T_XX, T_XY, T_YY}, b after a in genome
    order: the pairwise form of the reference's sequential update (statistics.c:442-456)."""
    a = np.array(a, np.float64)
    b = np.asarray(b, np.float64)
    if b[0] == 0:
        return a
    if a[0] == 0:
        return b.copy()
    n = a[0] + b[0]
    dx, dy = b[1] / b[0] - a[1] / a[0], b[2] / b[0] - a[2] / a[0]
    w = a[0] * b[0] / n
    return np.array([n, a[1] + b[1], a[2] + b[2], a[3] + b[3] + dx * dx * w, a[4] + b[4] + dx * dy * w,
                     a[5] + b[5] + dy * dy * w])


def pearson_from_moments(rows):
    """Pearson correlation from per-shard moment vectors listed in genome order (NaN when undefined,
    statistics.c:421-423)."""
    m = np.zeros(6)
    for r in rows:
        m = merge_moments(m, r)
    den = m[3] * m[5]
    return float(m[4] / np.sqrt(den)) if den else float("nan")


def allgather_moments(mine, group=None):
    """all_gather of one [n_items, 6] moment table per rank (rows a rank did not compute are zero)
    -> their sum, i.e. the full table (each row is computed by exactly one rank)."""
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.as_tensor(np.asarray(mine, np.float64), device=dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    return torch.stack(out).sum(0).cpu().numpy()
