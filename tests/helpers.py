"""Shared test helpers: random run-list cases and comparison utilities."""
import numpy as np

from wiggletools_amd.runlists import RunLists, synth

STREAM_OPS = ["sum", "product", "mean", "var", "stddev", "entropy", "cv", "min", "max"]
ALL_MULTIPLEX_OPS = STREAM_OPS + ["median"]


def random_case(seed, n_tracks=None, n_chrom=None, max_len=400, dtype=np.float64):
    """Small adversarial case: gaps, ties, NaNs, non-zero/NaN defaults, empty tracks,
    shared breakpoints, multiple chromosomes."""
    rng = np.random.default_rng(seed)
    n_tracks = n_tracks or int(rng.integers(1, 9))
    n_chrom = n_chrom or int(rng.integers(1, 4))
    clens = [int(rng.integers(1, max_len)) for _ in range(n_chrom)]
    mean_run = float(rng.choice([1, 2, 5, 20, 80]))
    gap = float(rng.choice([0.0, 0.02, 0.3, 0.7]))
    levels = int(rng.choice([2, 5, 800]))
    nanp = float(rng.choice([0.0, 0.0, 0.05]))
    t = synth(n_tracks, clens, mean_run=mean_run, gap_prob=gap, seed=int(rng.integers(1 << 30)),
              dtype=dtype, nan_prob=nanp, value_levels=levels,
              first_start=int(rng.choice([1, 1, 7, 1000])))
    mode = rng.integers(0, 4)
    if mode == 1:
        t.defaults[:] = rng.integers(-3, 4, n_tracks)
    elif mode == 2:
        t.defaults[:] = rng.random(n_tracks) * 10
    elif mode == 3 and n_tracks > 1:
        t.defaults[int(rng.integers(0, n_tracks))] = np.nan
    if rng.random() < 0.3 and n_tracks > 1:
        # make one track completely empty
        keep = [i for i in range(n_tracks) if i != 0]
        sub = t.subset(keep)
        empty = RunLists(t.n_chrom, 1, np.zeros(t.n_chrom + 1, np.int64), [], [], np.zeros(0, dtype),
                         [t.defaults[0]])
        t = merge_tracks([empty, sub])
    return t


def merge_tracks(parts):
    """Concatenate RunLists along the track axis (same chromosomes)."""
    n_chrom = parts[0].n_chrom
    seg_off = [0]
    S, F, V, D = [], [], [], []
    dt = parts[0].value.dtype
    for c in range(n_chrom):
        for p in parts:
            for i in range(p.n_tracks):
                lo, hi = p.seg_off[c * p.n_tracks + i], p.seg_off[c * p.n_tracks + i + 1]
                S.append(p.start[lo:hi]); F.append(p.finish[lo:hi]); V.append(p.value[lo:hi].astype(dt))
                seg_off.append(seg_off[-1] + int(hi - lo))
    for p in parts:
        D.extend(p.defaults.tolist())
    return RunLists(n_chrom, sum(p.n_tracks for p in parts), seg_off,
                    np.concatenate(S) if S else [], np.concatenate(F) if F else [],
                    np.concatenate(V) if V else np.zeros(0, dt), D)


def assert_runs_equal(a, b, rtol=0.0, what=""):
    """a, b = (chrom, start, finish, value). Coordinates bit-exact; values within rtol
    (rtol=0 -> bit-exact incl. NaN positions)."""
    assert len(a[0]) == len(b[0]), "%s: run count %d != %d" % (what, len(a[0]), len(b[0]))
    for k, name in enumerate(("chrom", "start", "finish")):
        assert np.array_equal(a[k], b[k]), "%s: %s differ" % (what, name)
    va, vb = np.asarray(a[3], np.float64), np.asarray(b[3], np.float64)
    assert np.array_equal(np.isnan(va), np.isnan(vb)), "%s: NaN pattern differs" % what
    m = ~np.isnan(va)
    if rtol == 0.0:
        assert np.array_equal(va[m], vb[m]), "%s: values differ (max abs %g)" % (
            what, np.max(np.abs(va[m] - vb[m])) if m.any() else 0)
    else:
        fin = m & np.isfinite(va) & np.isfinite(vb)
        assert np.array_equal(va[m & ~fin], vb[m & ~fin]), "%s: inf pattern differs" % what
        err = np.abs(va[fin] - vb[fin]) / np.maximum(np.abs(vb[fin]), 1e-300)
        ok = (err <= rtol) | (np.abs(va[fin] - vb[fin]) <= 1e-300)
        assert ok.all(), "%s: max rel err %g > %g" % (what, err.max(), rtol)
