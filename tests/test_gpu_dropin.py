"""The DROP-IN layer on a real GPU: the reference's own C API (newMultiplexer, MeanReduction,
newMultiset, MWUReduction, seek, popMultiplexer ...) exported by libwiggletools_amd.so, driven by
the very harness that drives the compiled reference (oracle/ref_harness.c), and compared with
the oracle.  Array-backed child WiggleIterators are popped one interval at a time, exactly as
the reference's readers would be."""
import numpy as np
import pytest

from helpers import ALL_MULTIPLEX_OPS, assert_runs_equal, random_case

pytestmark = pytest.mark.gpu

EXACT = {"sum", "product", "mean", "min", "max", "median"}


def _tol(op):
    return 0.0 if op in EXACT else (1e-9 if op in ("ttest", "mwu") else 1e-12)


@pytest.fixture(scope="module")
def H(oracle):
    import torch
    assert torch.cuda.is_available()
    from wiggletools_amd import _lib
    return oracle.Harness(_lib.LIB_PATH, "amd")


@pytest.mark.parametrize("seed", range(8))
def test_dropin_reducers(oracle, H, seed):
    t = random_case(7000 + seed, max_len=6000)
    d = t.as_dict()
    for strict in (0, 1):
        for op in ALL_MULTIPLEX_OPS:
            exp = oracle.reduce(d, op, flags=strict)
            got = H.reduce(d, op, flags=strict)
            assert_runs_equal(got, exp, _tol(op), "seed %d op %s strict %d" % (seed, op, strict))


@pytest.mark.parametrize("seed", range(4))
def test_dropin_multiplexer_fields(oracle, H, seed):
    """popMultiplexer keeps chrom/start/finish/values[]/inplay[] coherent (multiplexer.h:21-36)."""
    t = random_case(7100 + seed, max_len=5000)
    d = t.as_dict()
    for strict in (0, 1):
        exp = oracle.multiplex(d, flags=strict)
        got = H.multiplex(d, flags=strict)
        assert len(got[0]) == len(exp[0])
        for a, b in zip(got, exp):
            assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("seed", range(4))
def test_dropin_two_sample(oracle, H, seed):
    rng = np.random.default_rng(seed)
    t = random_case(7200 + seed, n_tracks=int(rng.integers(6, 11)), max_len=4000)
    d = t.as_dict()
    n1 = int(rng.integers(3, t.n_tracks - 2))
    for flags in (0, 1, 2, 3):
        for op in ("ttest", "mwu"):
            exp = oracle.reduce(d, op, flags=flags, n_set0=n1)
            got = H.reduce(d, op, flags=flags, n_set0=n1)
            assert_runs_equal(got, exp, _tol(op), "seed %d op %s flags %d" % (seed, op, flags))


@pytest.mark.parametrize("seed", range(3))
def test_dropin_multiset_stepping(oracle, H, seed):
    """Raw popMultiset over two Multiplexers (fields read by reference callers)."""
    if not oracle.have_ref():
        pytest.skip("compiled reference not available")
    t = random_case(7300 + seed, n_tracks=6, max_len=3000)
    d = t.as_dict()
    for flags in (0, 3):
        exp = oracle.ref_multiset(d, 3, flags)
        got = H.multiset(d, 3, flags)
        assert len(got[0]) == len(exp[0])
        for a, b in zip(got, exp):
            assert np.array_equal(a, b, equal_nan=True)


def _clip(t, chrom, start, finish):
    """Tracks as the reference's readers deliver them after seek(chrom, start, finish)
    (e.g. bigWiggleReader.c:125-145): only that chromosome, intervals clipped to the region."""
    from wiggletools_amd.runlists import RunLists
    tracks = []
    for i in range(t.n_tracks):
        per_c = []
        for c in range(t.n_chrom):
            lo, hi = t.seg_off[c * t.n_tracks + i], t.seg_off[c * t.n_tracks + i + 1]
            rows = []
            if c == chrom:
                for g in range(lo, hi):
                    s, f = int(t.start[g]), int(t.finish[g])
                    if f <= start or s >= finish:
                        continue
                    rows.append((max(s, start), min(f, finish), float(t.value[g])))
            per_c.append(rows)
        tracks.append(per_c)
    return RunLists.from_lists(tracks, t.defaults)


def test_dropin_seek(oracle, H):
    """seek(chrom, start, finish) on a reducer (reducers.c:25-29) == the reducer over the tracks
    clipped to the region, which is what the CLI's `seek` yields (readers are held until the
    first seek, commandParser.c:615-624).  NOT reproduced on purpose: seeking a Multiplexer that
    was already primed with data leaves stale inplay[]/values[] in the reference
    (seekCoreMultiplexer, multiplexer.c:130-141, resets the heaps but not those arrays)."""
    t = random_case(7400, n_tracks=5, n_chrom=2, max_len=8000)
    d = t.as_dict()
    for (c, s, f) in ((0, 100, 3000), (1, 1, 50), (0, 2500, 2600), (1, 10, 4000)):
        for op in ("mean", "median"):
            got = H.reduce_seek(d, op, c, s, f)
            exp = oracle.reduce(_clip(t, c, s, f).as_dict(), op)
            assert_runs_equal(got, exp, 0.0, "seek %s %s" % ((c, s, f), op))


def test_dropin_batches_cross_seams(oracle, H):
    """Long tracks: several geometric batches, intervals crossing every cut."""
    from wiggletools_amd.runlists import synth
    t = synth(8, [300000, 5000], mean_run=40, seed=9, gap_prob=0.1)
    d = t.as_dict()
    for op in ("mean", "max"):
        exp = oracle.reduce(d, op)
        got = H.reduce(d, op)
        assert_runs_equal(got, exp, 0.0, op)


def test_dropin_ctor_defaults(oracle, H):
    for dv in (np.zeros(3), np.array([1.0, 2.0, 3.5]), np.array([1.0, np.nan, 2.0])):
        for op in ALL_MULTIPLEX_OPS:
            a, b = H.reducer_default(op, dv), oracle.reducer_default(op, dv)
            assert (np.isnan(a) and np.isnan(b)) or a == b, (op, dv)
