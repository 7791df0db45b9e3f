"""Parity of the HIP path (through the C ABI) with the oracle, on a real MI355X.

Coordinates: bit-exact.  Values: the north-star tolerance is 1e-6 relative; the
kernels visit tracks in the reference's order in f64, so sums/means/min/max/
median are expected bit-exact and are asserted so; var/stddev/CV/ttest/MWU go
through device sqrt/div/erf/lgamma and are held to 1e-12 / 1e-9.
"""
import numpy as np
import pytest

from helpers import ALL_MULTIPLEX_OPS, assert_runs_equal, random_case

pytestmark = pytest.mark.gpu

EXACT = {"sum", "product", "mean", "min", "max", "median"}


@pytest.fixture(scope="module")
def engine():
    import torch
    assert torch.cuda.is_available()
    from wiggletools_amd import engine as E
    return E


def _tol(op):
    if op in EXACT:
        return 0.0
    return 1e-9 if op in ("ttest", "mwu") else 1e-12


@pytest.mark.parametrize("seed", range(24))
def test_gpu_one_sample_ops_fuzz(oracle, engine, seed):
    t = random_case(seed, dtype=np.float64 if seed % 2 else np.float32)
    d = t.as_dict()
    ts = engine.TrackSet.from_runlists(t)
    for strict in (0, 1):
        for op in ALL_MULTIPLEX_OPS:
            exp = oracle.reduce(d, op, flags=strict)
            got = ts.reduce_host(op, flags=strict)
            assert_runs_equal(got, exp, _tol(op), "seed %d op %s strict %d" % (seed, op, strict))
    ts.close()


@pytest.mark.parametrize("seed", range(12))
def test_gpu_two_sample_ops_fuzz(oracle, engine, seed):
    rng = np.random.default_rng(seed)
    t = random_case(500 + seed, n_tracks=int(rng.integers(6, 12)), dtype=np.float64 if seed % 2 else np.float32)
    d = t.as_dict()
    n1 = int(rng.integers(3, t.n_tracks - 2))
    ts = engine.TrackSet.from_runlists(t)
    for flags in (0, 1, 2, 3):
        for op in ("ttest", "mwu"):
            exp = oracle.reduce(d, op, flags=flags, n_set0=n1)
            got = ts.reduce_host(op, flags=flags, n_set0=n1)
            assert_runs_equal(got, exp, _tol(op), "seed %d op %s flags %d" % (seed, op, flags))
    ts.close()


@pytest.mark.parametrize("seed", range(6))
def test_gpu_multiplexer_tile(oracle, engine, seed):
    t = random_case(900 + seed)
    ts = engine.TrackSet.from_runlists(t)
    for strict in (0, 1):
        exp = oracle.multiplex(t.as_dict(), flags=strict)
        got = ts.multiplex_host(flags=strict)
        assert len(got[0]) == len(exp[0])
        for a, b in zip(got, exp):
            assert np.array_equal(a, b, equal_nan=True)
    ts.close()


@pytest.mark.parametrize("n_tracks,ops", [(100, ["mean", "sum", "max", "median"]), (500, ["var", "stddev"]),
                                          (100, ["cv", "min", "product"])])
def test_gpu_config_sized_tracks(oracle, engine, n_tracks, ops):
    """BASELINE configs C2/C3/C4 track counts on a window-crossing multi-chromosome batch."""
    from wiggletools_amd.runlists import synth
    t = synth(n_tracks, [60000, 20000, 7], mean_run=16, seed=n_tracks)
    ts = engine.TrackSet.from_runlists(t)
    for op in ops:
        exp = oracle.reduce(t.as_dict(), op)
        got = ts.reduce_host(op)
        assert_runs_equal(got, exp, _tol(op), "N %d op %s" % (n_tracks, op))
    st = ts.stats()
    assert st["covered_bp"] == int((exp[2] - exp[1]).sum())
    ts.close()


def test_gpu_many_tracks_chunked(oracle, engine):
    """More tracks than one workgroup's LDS holds: bitmaps rebuilt per chunk and pass (streaming ops,
    Multiplexer tile), median / MWU columns in a global slab per workgroup."""
    from wiggletools_amd.runlists import synth
    t = synth(700, [9000, 2500, 7], mean_run=16, seed=700)
    d = t.as_dict()
    ts = engine.TrackSet.from_runlists(t)
    for op, kw in (("mean", {}), ("var", {}), ("max", {}), ("median", {}),
                   ("mwu", dict(n_set0=300)), ("ttest", dict(n_set0=300))):
        exp = oracle.reduce(d, op, **kw)
        got = ts.reduce_host(op, **kw)
        assert_runs_equal(got, exp, _tol(op), "N 700 op %s" % op)
    exp = oracle.multiplex(d)
    got = ts.multiplex_host()
    for a, b in zip(got, exp):
        assert np.array_equal(a, b, equal_nan=True)
    ts.close()


@pytest.mark.parametrize("seed", range(6))
def test_gpu_exact_difference_array_path(oracle, engine, seed, monkeypatch):
    """Sum / Mean over float tracks with zero defaults run through the O(intervals) difference-array
    kernel (8 positions per lane: W = 4096) and stay bit-identical; data it cannot prove exact
    (wide dynamic range, NaN) is redone by the general kernel."""
    from wiggletools_amd.runlists import synth
    monkeypatch.setenv("WTAMD_DELTA_MIN_TRACKS", "1")
    monkeypatch.setenv("WTAMD_DELTA_T", "512" if seed % 3 else "256")
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 40))
    t = synth(n, [int(rng.integers(1, 60000)), 9000, 3], mean_run=float(rng.choice([1, 4, 16, 300])),
              gap_prob=float(rng.choice([0.0, 0.1, 0.5])), seed=seed, first_start=int(rng.choice([1, 77, 5000])))
    if seed % 2:
        t.value[:] = ((rng.random(len(t.value)) - 0.3) * 1000).astype(np.float32)     # full mantissas, signs
    d = t.as_dict()
    ts = engine.TrackSet.from_runlists(t)
    for strict in (0, 1):
        for op in ("sum", "mean"):
            exp = oracle.reduce(d, op, flags=strict)
            got = ts.reduce_host(op, flags=strict)
            if seed % 2 == 0:       # k/8 values: every window is provably exact
                assert ts.stats()["window_bp"] == (4096 if seed % 3 else 2048) and ts.stats()["kernel"] == 1, \
                    "difference-array path not taken"
            assert_runs_equal(got, exp, 0.0, "seed %d op %s strict %d" % (seed, op, strict))
    ts.close()
    # one window not provably exact -> its values come from the general kernel (patch), same answer;
    # later reductions of the same track set reuse the verdict (no host round trip before the patch)
    t.value[len(t.value) // 2] = np.nan if seed % 2 else np.float32(1e-35)
    ts = engine.TrackSet.from_runlists(t)
    n_win = None
    for op, strict in (("sum", 0), ("mean", 0), ("mean", 1), ("sum", 1)):
        exp = oracle.reduce(t.as_dict(), op, flags=strict)
        got = ts.reduce_host(op, flags=strict)
        st = ts.stats()
        n_win = st["n_windows"]
        if n_win >= 4 and seed % 2 == 0:        # (odd seeds: random magnitudes may add inexact windows of their own)
            assert st["kernel"] == 1 and st["patched_windows"] >= 1, st
        assert_runs_equal(got, exp, 0.0, "patched seed %d op %s strict %d" % (seed, op, strict))
    ts.close()
    # NaNs all over: patched or redone entirely, depending on how many windows they hit
    t.value[::50] = np.nan
    ts = engine.TrackSet.from_runlists(t)
    for op in ("sum", "mean"):
        exp = oracle.reduce(t.as_dict(), op)
        got = ts.reduce_host(op)
        assert_runs_equal(got, exp, 0.0, "fallback seed %d op %s" % (seed, op))
    ts.close()


def test_gpu_difference_array_patch_is_asynchronous_after_the_verdict(oracle, engine, monkeypatch):
    """Device path: once a completed launch has established which windows need the general kernel,
    later Sum / Mean launches run difference-array + patch kernels back to back on the stream."""
    import torch
    from wiggletools_amd.runlists import synth
    monkeypatch.setenv("WTAMD_DELTA_MIN_TRACKS", "1")
    t = synth(24, [200000, 30000], mean_run=16, gap_prob=0.05, seed=99)
    t.value[1000] = np.nan
    t.value[len(t.value) - 500] = np.inf
    t.value[len(t.value) // 3] = np.float32(3e-38)
    dev = torch.device("cuda", 0)
    ts = engine.TrackSet.from_device(t.n_chrom, t.n_tracks, t.seg_off, torch.from_numpy(t.start).to(dev),
                                     torch.from_numpy(t.finish).to(dev), torch.from_numpy(t.value).to(dev), t.defaults)
    out = ts.alloc_runs()
    stream = torch.cuda.current_stream().cuda_stream
    n = ts.reduce("sum", out, stream=stream, sync=True)          # establishes the verdict
    assert ts.stats()["kernel"] == 1 and ts.stats()["patched_windows"] == 3
    exp = oracle.reduce(t.as_dict(), "sum")
    out.n = n
    assert_runs_equal(out.to_host(), exp, 0.0, "probe launch")
    for op, strict in (("mean", 0), ("sum", 1)):
        out.value.fill_(-1.0)
        ts.reduce(op, out, flags=strict, stream=stream, sync=False)
        torch.cuda.synchronize()
        exp = oracle.reduce(t.as_dict(), op, flags=strict)
        out.n = len(exp[0])
        cro = out.chrom_run_off.cpu().numpy()
        assert int(cro[-1]) == len(exp[0])
        assert_runs_equal(out.to_host(), exp, 0.0, "async %s strict %d" % (op, strict))
    ts.close()


@pytest.mark.parametrize("direction", ["falling", "rising", "mixed"])
def test_gpu_difference_array_speculative_unit(oracle, engine, direction):
    """Magnitudes drifting along the genome: windows whose exponents leave the workgroup's guessed
    unit are redone with their own; whichever workgroup got which window, same bits."""
    from wiggletools_amd.runlists import synth
    t = synth(30, [300000, 50000], mean_run=16, gap_prob=0.1, seed=11)
    rng = np.random.default_rng(5)
    v = rng.random(len(t.value)) + 1.0
    pos = t.start.astype(np.int64)
    if direction == "mixed":
        shift = rng.integers(-8, 9, size=80)[(pos // 4096) % 80]
    else:
        shift = (1 if direction == "rising" else -1) * (pos // 8192)
    t.value[:] = np.ldexp(v, shift).astype(np.float32)
    ts = engine.TrackSet.from_runlists(t)
    for op in ("sum", "mean"):
        exp = oracle.reduce(t.as_dict(), op)
        got = ts.reduce_host(op)
        assert ts.stats()["kernel"] == 1
        assert_runs_equal(got, exp, 0.0, "%s %s" % (direction, op))
    ts.close()


def test_gpu_wilcoxon_50_vs_50(oracle, engine):
    """BASELINE config C5 shape (n1 = n2 = 50: mu = 1250, sigma = sqrt(21041))."""
    from wiggletools_amd.runlists import synth
    t = synth(100, [30000, 5000], mean_run=16, seed=50)
    ts = engine.TrackSet.from_runlists(t)
    for op in ("mwu", "ttest"):
        exp = oracle.reduce(t.as_dict(), op, n_set0=50)
        got = ts.reduce_host(op, n_set0=50)
        assert_runs_equal(got, exp, _tol(op), op)
    ts.close()


def test_gpu_golden_reference_fixtures(engine):
    """The reference's own fixtures through the HIP path == outputs of the compiled reference."""
    import json
    import os
    from wiggletools_amd.textio import load_runlists
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cases = json.load(open(os.path.join(G, "reference_fixtures.json")))["cases"]
    cache = {}
    for case in cases:
        key = tuple(case["files"])
        if key not in cache:
            rl = load_runlists([os.path.join(G, f) for f in case["files"]])
            cache[key] = (rl, engine.TrackSet.from_runlists(rl))
        rl, ts = cache[key]
        got = ts.reduce_host(case["op"], flags=case["strict"])
        names_got = [rl.chrom_names[c] for c in got[0]]
        assert names_got == [case["chrom_names"][c] for c in case["chrom"]]
        exp_v = np.array([np.nan if v is None else v for v in case["value"]], np.float64)
        z = np.zeros(len(exp_v))
        assert_runs_equal((z, got[1], got[2], got[3]),
                          (z, np.array(case["start"]), np.array(case["finish"]), exp_v), _tol(case["op"]),
                          "%s %s" % (case["set"], case["op"]))


def test_gpu_device_resident_path_and_auc(oracle, engine):
    """Zero-copy device tensors in, device run list out, AUC on device."""
    import torch
    from wiggletools_amd.runlists import synth
    t = synth(20, [40000, 3000], mean_run=8, seed=77)
    dev = torch.device("cuda", 0)
    ts = engine.TrackSet.from_device(t.n_chrom, t.n_tracks, t.seg_off, torch.from_numpy(t.start).to(dev),
                                     torch.from_numpy(t.finish).to(dev), torch.from_numpy(t.value).to(dev),
                                     t.defaults)
    out = ts.alloc_runs()
    stream = torch.cuda.current_stream().cuda_stream
    n = ts.reduce("mean", out, stream=stream, sync=True)
    exp = oracle.reduce(t.as_dict(), "mean")
    assert n == len(exp[0])
    assert_runs_equal(out.to_host(), exp, 0.0, "device path")
    auc = out.auc()
    ref = oracle.auc(exp[1], exp[2], exp[3])
    assert abs(auc - ref) <= 1e-9 * abs(ref)
    # meanI (statistics.c:62-100): length-weighted mean over the non-NaN runs
    m = ~np.isnan(exp[3])
    span = float((exp[2][m] - exp[1][m]).sum())
    assert abs(out.mean() - ref / span) <= 1e-9 * abs(ref / span)
    # idempotence: a second pass over the same resident tracks gives the same run list
    n2 = ts.reduce("mean", out, stream=stream, sync=True)
    assert n2 == n
    assert_runs_equal(out.to_host(), exp, 0.0, "second pass")
    ts.close()


def test_gpu_capacity_error_is_reported(engine):
    from wiggletools_amd import _lib
    from wiggletools_amd.runlists import synth
    t = synth(4, [5000], mean_run=4, seed=5)
    ts = engine.TrackSet.from_runlists(t)
    out = ts.alloc_runs(capacity=10)
    with pytest.raises(_lib.WtamdError):
        ts.reduce("sum", out, sync=True)
    ts.close()


def test_gpu_precondition_messages(engine):
    """Error text of the reference ctors (setComparisons.c:123-128, 374-377)."""
    from wiggletools_amd import _lib
    from wiggletools_amd.runlists import synth
    t = synth(4, [500], mean_run=4, seed=5)
    ts = engine.TrackSet.from_runlists(t)
    with pytest.raises(_lib.WtamdError, match="t-test function only works"):
        ts.reduce_host("ttest", n_set0=2)
    with pytest.raises(_lib.WtamdError, match="Mann-Whitney U function only works"):
        ts.reduce_host("mwu", n_set0=0)
    ts.close()


def _device_runs(engine, c, s, f, v, n_chrom):
    import torch
    dev = torch.device("cuda", 0)
    cro = np.zeros(n_chrom + 1, np.int64)
    np.cumsum(np.bincount(c, minlength=n_chrom), out=cro[1:])
    r = engine.DeviceRuns(torch.from_numpy(np.ascontiguousarray(s, np.int32)).to(dev),
                          torch.from_numpy(np.ascontiguousarray(f, np.int32)).to(dev),
                          torch.from_numpy(np.ascontiguousarray(v, np.float64)).to(dev),
                          torch.from_numpy(cro).to(dev))
    r.n = len(s)
    return r


@pytest.mark.parametrize("seed", range(8))
def test_gpu_run_compression_matches_reference_rule(oracle, engine, seed):
    """Device CompressionWiggleIterator (unaryOps.c:235-253) on reducer outputs and on adversarial
    slowly drifting values (|dv| < 1e-6 per run but > 1e-6 from the group leader)."""
    rng = np.random.default_rng(seed)
    t = random_case(5000 + seed, max_len=3000)
    c, s, f, v = oracle.reduce(t.as_dict(), ["mean", "max", "stddev", "min"][seed % 4])
    if seed % 2 and len(v):
        # drift: consecutive differences of 0.4e-6 .. 0.9e-6, plus exact repeats and NaN runs
        drift = np.cumsum(rng.choice([0.0, 4e-7, 7e-7, 9e-7, -6e-7], len(v)))
        v = np.where(rng.random(len(v)) < 0.05, np.nan, np.round(v) + drift)
    exp = oracle.compress(c, s, f, v)
    r = _device_runs(engine, c, s, f, v, t.n_chrom)
    out = r.compress()
    got = out.to_host()
    assert_runs_equal(got, exp, 0.0, "compress seed %d" % seed)


def test_gpu_compress_long_constant_stretch(oracle, engine):
    n = 300000
    s = np.arange(1, n + 1, dtype=np.int32)
    f = s + 1
    v = np.zeros(n)
    v[100000:100010] = 1.0
    v[200000] = np.nan
    c = np.zeros(n, np.int32)
    exp = oracle.compress(c, s, f, v)
    got = _device_runs(engine, c, s, f, v, 1).compress().to_host()
    assert_runs_equal(got, exp, 0.0, "long stretch")
    assert len(got[0]) == 5


def test_gpu_bigwig_config_c1(oracle, engine):
    """BASELINE config C1: `mean test/fixedStep.bw test/variableStep.bw` -- the two reference
    fixtures decoded by the library's BigWig section decoder, reduced on the GPU; expected
    values are the ones of SURVEY 8c (the reference prints them as fixedStep chr1 start=1)."""
    import os
    from wiggletools_amd import bigwig
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    t = bigwig.load_runlists([os.path.join(G, "fixedStep.bw"), os.path.join(G, "variableStep.bw")])
    ts = engine.TrackSet.from_runlists(t)
    c, s, f, v = ts.reduce_host("mean")
    assert [t.chrom_names[x] for x in c] == ["chr1"] * 10
    assert s.tolist() == list(range(1, 11)) and f.tolist() == list(range(2, 12))
    assert v.tolist() == [0.5, 1.5, 1, 3, 2, 4.5, 3, 6, 4, 4.5]
    ts.close()


@pytest.mark.parametrize("seed", range(6))
def test_gpu_pearson_integrator(oracle, engine, seed):
    """PearsonIntegrator over a 2-track Multiplexer (statistics.c:414-465) on device: slices of runs
    merged with the reference's own update formula -- agreement to rounding with the oracle's
    sequential restatement (which equals the compiled reference bit for bit, tests/test_oracle_vs_ref.py)."""
    from wiggletools_amd.runlists import synth
    rng = np.random.default_rng(seed)
    t = synth(2, [int(rng.integers(50, 400000)), 3000], mean_run=float(rng.choice([1, 16, 200])),
              gap_prob=float(rng.choice([0.0, 0.2])), seed=100 + seed)
    if seed % 2:
        t.defaults[:] = [0.5, -2.0]
    exp = oracle.pearson(t.as_dict())
    ts = engine.TrackSet.from_runlists(t)
    got = ts.pearson()
    ts.close()
    assert abs(got - exp) <= 1e-9 * max(1.0, abs(exp)), (got, exp)


def test_gpu_pearson_golden_and_degenerate(oracle, engine):
    import os
    from wiggletools_amd.textio import load_runlists
    G = os.path.join(os.path.dirname(__file__), "golden")
    t = load_runlists([os.path.join(G, "fixedStep.wig"), os.path.join(G, "variableStep.wig")])
    ts = engine.TrackSet.from_runlists(t)
    assert abs(ts.pearson() - (-0.028968)) < 5e-7          # reference test/expected/pearson.txt
    ts.close()
    from wiggletools_amd.runlists import RunLists
    t = RunLists.from_lists([[[(1, 10, 2.0)]], [[(1, 10, 3.0)]]])       # constant tracks: T_XX * T_YY == 0 -> NaN
    ts = engine.TrackSet.from_runlists(t)
    assert np.isnan(ts.pearson())
    ts.close()


def test_gpu_pearson_moments_per_chromosome_merge(oracle, engine):
    """Multi-GPU readiness of the Pearson gather: the 6 moments of every chromosome on its own
    (wtamd_pearson_moments), merged in genome order on the host (wtamd_pearson_merge / _finish and
    their Python mirror in shard.py) == Pearson over the whole genome in one go == the oracle."""
    import ctypes as C
    from wiggletools_amd import _lib, shard
    from wiggletools_amd.runlists import RunLists, synth
    t = synth(2, [90000, 41000, 700, 66000], mean_run=9, gap_prob=0.1, seed=77)
    exp = oracle.pearson(t.as_dict())
    rows = []
    for c in range(t.n_chrom):
        a, b = int(t.seg_off[2 * c]), int(t.seg_off[2 * c + 2])
        one = RunLists(1, 2, t.seg_off[2 * c:2 * c + 3] - t.seg_off[2 * c], t.start[a:b], t.finish[a:b], t.value[a:b], t.defaults)
        ts = engine.TrackSet.from_runlists(one)
        rows.append(ts.pearson_moments())
        ts.close()
    assert abs(shard.pearson_from_moments(rows) - exp) <= 1e-9 * abs(exp)
    acc = np.zeros(6)
    L = _lib.lib()
    for r in rows:
        L.wtamd_pearson_merge(acc.ctypes.data, np.ascontiguousarray(r).ctypes.data)
    assert abs(L.wtamd_pearson_finish(acc.ctypes.data) - exp) <= 1e-9 * abs(exp)
    assert acc[0] == sum(r[0] for r in rows)


@pytest.mark.parametrize("seed", range(8))
def test_gpu_difference_array_var_family(oracle, engine, seed):
    """Var / StdDev / Entropy / CV over float tracks with zero defaults run the difference-array
    kernel with exact integer sum and sum of squares (wt_delta.h, WT_DELTA_QSHIFT); the oracle does the
    reference's two sequential f64 passes (reducers.c:428-479, 511-563, 672-725).  1e-12 relative,
    coordinates bit-exact; strict and not; NaN / Inf / huge dynamic range windows patched."""
    from wiggletools_amd.runlists import synth
    rng = np.random.default_rng(seed)
    n = int(rng.choice([8, 16, 33, 100, 500]))
    t = synth(n, [int(rng.integers(2000, 60000)), 900], mean_run=float(rng.choice([1, 3, 16, 60])), seed=seed,
              gap_prob=float(rng.choice([0, 0.05, 0.5])), dtype=np.float32, value_levels=int(rng.choice([2, 800])))
    if seed % 4 == 1:
        # the poisoned cases: 49 windows of 4096 bp, so that the few bad ones are PATCHED (many would redo the launch)
        t = synth(16, [200000, 900], mean_run=16.0, seed=seed, gap_prob=0.05, dtype=np.float32, value_levels=800)
    if seed % 3 == 0:
        t.value[:] = (t.value * rng.choice([1e-3, 1.0, 37.5], len(t.value))).astype(np.float32)
    if seed % 4 == 1:
        t.value[7] = np.nan
        t.value[len(t.value) // 2] = np.inf
        t.value[len(t.value) // 3] = 2.0 ** -120
        t.value[len(t.value) // 3 + 1] = 2.0 ** 100
    d = t.as_dict()
    ts = engine.TrackSet.from_runlists(t)
    for op in ("var", "stddev", "cv", "entropy"):
        for strict in (0, 1):
            got = ts.reduce_host(op, flags=strict)
            st = ts.stats()
            # (25 binades of dynamic range at 500 tracks exceed the exactness bound in most windows:
            #  the whole launch is then redone by the general kernel -- also a path worth covering)
            assert st["kernel"] == 1 or seed % 3 == 0, (op, st)
            if seed % 4 == 1 and st["kernel"] == 1:
                assert st["patched_windows"] > 0
            assert_runs_equal(got, oracle.reduce(d, op, flags=strict), 1e-12, "%s strict %d" % (op, strict))
    ts.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_gpu_difference_array_ttest(oracle, engine, seed):
    """TTestReduction by difference arrays (round 6: wt_delta_kernel<ttest>, wt_delta_scan3_tt): per set the exact integer sum
    and sum of squares of the tracks in play (setComparisons.c:60-81), the reference's arithmetic from there (:88-117).
    1e-9 against the oracle (coarse values and full mantissas); coordinates and emitted runs exact under the four strictness flags;
    NaN / Inf / wide exponent range / cancelling variance: those windows come from the general kernel (wt_patch_kernel)."""
    from wiggletools_amd.runlists import synth
    rng = np.random.default_rng(7100 + seed)
    n = int(rng.choice([8, 16, 33, 100, 200, 900]))
    n1 = int(rng.integers(3, n - 2)) if seed % 2 else n // 2
    t = synth(n, [int(rng.integers(3000, 90000)), 900], mean_run=float(rng.choice([1, 3, 16, 60, 3000])), seed=seed,
              gap_prob=float(rng.choice([0, 0.05, 0.5])), dtype=np.float32, value_levels=800)
    if seed % 3 == 0:
        t.value[:] = (t.value * rng.choice([0.3, 1.0, 37.5], len(t.value))).astype(np.float32)
    if seed % 4 == 1:
        t = synth(40, [200000, 900], mean_run=16.0, seed=seed, gap_prob=0.05, dtype=np.float32, value_levels=800)     # 98 windows: a few bad ones are patched
        t.value[7] = np.nan
        t.value[len(t.value) // 2] = np.inf
        t.value[len(t.value) // 3] = 2.0 ** -120
        t.value[len(t.value) // 3 + 1] = 2.0 ** 100
        n, n1 = 40, 17
    d = t.as_dict()
    ts = engine.TrackSet.from_runlists(t)
    for flags in (0, 1, 2, 3):
        got = ts.reduce_host("ttest", flags=flags, n_set0=n1)
        st = ts.stats()
        assert st["kernel"] == 1 and st["window_bp"] == 2048, st
        if seed % 4 == 1:
            assert st["patched_windows"] > 0, st
        # 1e-9 on the device where the sums are exact on both sides (coarse values: what differs is the device's log / exp; the
        # emulator's test holds those cases to tolerance 0).  Full mantissas: the reference's own sum of squares rounds n times,
        # the exact sums once, and a p-value deep in the tail magnifies a relative change of t by ~t^2 (900 tracks, t ~ 6:
        # 1.7e-9 measured) -- 1e-7 there; BASELINE.json's bound for float statistics is 1e-6.
        assert_runs_equal(got, oracle.reduce(d, "ttest", flags=flags, n_set0=n1), 1e-7 if seed % 3 == 0 else 1e-9,
                          "ttest seed %d flags %d n %d n1 %d %s" % (seed, flags, n, n1, st))
    ts.close()


@pytest.mark.gpu
def test_gpu_difference_array_ttest_vs_general_kernel(oracle, engine, monkeypatch):
    """The two routes of TTestReduction on the same 100 tracks x 300 kbp (50 v 50, the shape of the bench record): coordinates
    equal, values equal bit for bit (k/8 values: neither route's sums round); and the second launch -- verdict known, no
    host round trip -- gives the same runs as the first."""
    from wiggletools_amd.runlists import synth
    t = synth(100, [300000, 5000], mean_run=16.0, seed=4, gap_prob=0.02, dtype=np.float32, value_levels=800)
    ts = engine.TrackSet.from_runlists(t)
    a = ts.reduce_host("ttest", n_set0=50)
    assert ts.stats()["kernel"] == 1
    a2 = ts.reduce_host("ttest", n_set0=50)
    monkeypatch.setenv("WTAMD_NO_DELTA_TTEST", "1")
    ts2 = engine.TrackSet.from_runlists(t)
    b = ts2.reduce_host("ttest", n_set0=50)
    assert ts2.stats()["kernel"] == 0
    for x, y, z in zip(a, b, a2):
        assert np.array_equal(x, y, equal_nan=True) and np.array_equal(x, z, equal_nan=True)
    assert len(a[0]) > 250000
    exp = oracle.reduce(t.as_dict(), "ttest", n_set0=50)
    assert_runs_equal(a, exp, 1e-9, "ttest 50 v 50")
    ts.close(); ts2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_gpu_difference_array_min_max(oracle, engine, seed):
    """MaxReduction / MinReduction by range updates of a segment tree (round 6: wt_delta_kernel<max | min>, wt_delta_apply_mm) against the
    oracle, bit for bit: negative values, full mantissas, runs from one position to longer than a window, strict and not, more tracks
    than lanes; a NaN / a -0.0 in a window: that window comes from the general kernel (wt_patch_kernel), signs of zeros included."""
    from wiggletools_amd.runlists import synth
    rng = np.random.default_rng(9100 + seed)
    n = int(rng.choice([4, 9, 33, 100, 300, 1500]))
    t = synth(n, [int(rng.integers(3000, 90000)), 900], mean_run=float(rng.choice([1, 3, 16, 60, 5000, 20000])), seed=seed,
              gap_prob=float(rng.choice([0, 0.05, 0.5])), dtype=np.float32, value_levels=int(rng.choice([2, 800])))
    t.value[:] = ((t.value - rng.choice([0, 3, 50])) * rng.choice([1.0, 1e-3, 7.7], len(t.value))).astype(np.float32)
    t.value[t.value == 0] = 0.0
    if seed % 4 == 1:
        t = synth(40, [600000, 900], mean_run=16.0, seed=seed, gap_prob=0.05, dtype=np.float32, value_levels=800)     # 74 windows: a few bad ones are patched
        t.value[:] = (t.value - 40).astype(np.float32)
        t.value[7] = np.nan
        t.value[len(t.value) // 2] = np.inf
        t.value[len(t.value) // 3] = -0.0
    d = t.as_dict()
    ts = engine.TrackSet.from_runlists(t)
    for op in ("max", "min"):
        for flags in (0, 1):
            got = ts.reduce_host(op, flags=flags)
            st = ts.stats()
            assert st["kernel"] == 1 and st["window_bp"] == 8192, st
            if seed % 4 == 1:
                assert st["patched_windows"] > 0, st
            exp = oracle.reduce(d, op, flags=flags)
            assert_runs_equal(got, exp, 0.0, "%s seed %d flags %d %s" % (op, seed, flags, st))
            assert np.array_equal(np.signbit(got[3]), np.signbit(exp[3]))
    ts.close()


def test_gpu_input_contract_validation(engine):
    """wtamd_trackset_validate: zero-length, inverted and overlapping runs are counted; a run list
    may start a new (chrom, track) segment below the previous segment's last finish."""
    from wiggletools_amd.runlists import RunLists
    ok = RunLists.from_lists([[[(5, 9, 1.0), (9, 12, 2.0)], [(1, 3, 1.0)]], [[(2, 4, 1.0)], []]])
    ts = engine.TrackSet.from_runlists(ok)
    assert ts.validate() == (0, -1)
    ts.close()
    bad = RunLists.from_lists([[[(5, 9, 1.0), (9, 9, 2.0), (9, 12, 3.0), (11, 15, 4.0)]], [[(7, 6, 1.0)]]])
    ts = engine.TrackSet.from_runlists(bad)
    assert ts.validate() == (3, 1)          # zero-length, overlapping, inverted
    ts.close()


@pytest.mark.parametrize("op,param", [("scale", -2.5), ("offset", 3.25), ("ln", 0.0), ("log", 2.0), ("exp", 0.0),
                                      ("expb", 2.0), ("pow", 2.0), ("pow", -1.0), ("abs", 0.0),
                                      ("gt", 12.5), ("gte", 12.5), ("lt", 12.5), ("lte", -3.0)])
def test_gpu_map_ops(oracle, engine, op, param):
    """`map`-able unary operators on device (wtamd_runs_map) vs the oracle's restatement (itself pinned
    bit for bit on the compiled reference): dropped runs and segment offsets exact, values to 1e-12
    (device libm), defaults exact; then `sum map <op>` end to end."""
    from wiggletools_amd.runlists import RunLists
    for seed in range(4):
        t = random_case(8100 + seed, n_tracks=5, max_len=9000, dtype=np.float64 if seed % 2 else np.float32)
        rng = np.random.default_rng(seed)
        t.value[:] = (t.value * rng.choice([1.0, -1.0, 0.0], size=len(t.value), p=[0.6, 0.3, 0.1])).astype(t.value.dtype)
        got = engine.map_runlists(t, op, param)
        out, keep = oracle.map_values(op, param, t.value)
        k = keep != 0
        n_seg = t.n_chrom * t.n_tracks
        exp_seg = np.concatenate([[0], np.cumsum([int(k[t.seg_off[q]:t.seg_off[q + 1]].sum()) for q in range(n_seg)])])
        assert np.array_equal(got.seg_off, exp_seg), (op, seed)
        assert np.array_equal(got.start, t.start[k]) and np.array_equal(got.finish, t.finish[k])
        a, b = got.value, out[k]
        assert np.array_equal(np.isnan(a), np.isnan(b))
        m = ~np.isnan(a) & np.isfinite(b)
        assert np.array_equal(a[~np.isnan(a) & ~np.isfinite(b)], b[~np.isnan(a) & ~np.isfinite(b)])
        assert np.all(np.abs(a[m] - b[m]) <= 1e-12 * np.maximum(np.abs(b[m]), 1e-300)), (op, seed)
        expd = np.array([oracle.map_default(op, param, x) for x in t.defaults])
        assert np.array_equal(got.defaults, expd, equal_nan=True)
        # end to end: sum over the mapped tracks (oracle over the oracle-mapped run lists)
        ref_t = RunLists(t.n_chrom, t.n_tracks, exp_seg, t.start[k], t.finish[k], out[k], expd)
        ts = engine.TrackSet.from_runlists(got)
        res = ts.reduce_host("sum")
        ts.close()
        assert_runs_equal(res, oracle.reduce(ref_t.as_dict(), "sum"), 1e-12, "sum map %s seed %d" % (op, seed))


def test_gpu_empty_and_minimal_inputs(oracle, engine, monkeypatch):
    """No runs at all, one 1-bp run, one run at the top of the coordinate range, through both kernels."""
    from wiggletools_amd.runlists import RunLists
    monkeypatch.setenv("WTAMD_DELTA_MIN_TRACKS", "1")
    cases = [
        RunLists.from_lists([[[], []], [[], []]]),
        RunLists.from_lists([[[(5, 6, 1.5)], []]]),
        RunLists.from_lists([[[(2 ** 31 - 70000, 2 ** 31 - 65600, 2.0)], [(2 ** 31 - 69000, 2 ** 31 - 65537, -1.0)]]]),
    ]
    for t in cases:
        t = RunLists(t.n_chrom, t.n_tracks, t.seg_off, t.start, t.finish, t.value.astype(np.float32), t.defaults)
        ts = engine.TrackSet.from_runlists(t)
        for op in ("sum", "mean", "max", "median"):
            for strict in (0, 1):
                exp = oracle.reduce(t.as_dict(), op, flags=strict)
                got = ts.reduce_host(op, flags=strict)
                assert_runs_equal(got, exp, 0.0, "%s strict %d n=%d" % (op, strict, len(t.start)))
        ts.close()
    # above the supported coordinate range: refused at creation, not silently mishandled
    t = RunLists.from_lists([[[(2 ** 31 - 10, 2 ** 31 - 2, 2.0)]]])
    with pytest.raises(Exception, match="above the supported maximum"):
        engine.TrackSet.from_runlists(t)


def test_gpu_region_like_tracks_spanning_many_windows(oracle, engine, monkeypatch):
    """BED-region-like tracks (runs of tens of kbp, each spanning many alignment windows, and one
    run covering the whole chromosome) multiplexed with dense signal tracks, both kernels."""
    from helpers import merge_tracks
    from wiggletools_amd.runlists import RunLists, synth
    monkeypatch.setenv("WTAMD_DELTA_MIN_TRACKS", "1")
    clens = [400000, 90000]
    dense = synth(5, clens, mean_run=8, gap_prob=0.1, seed=5)
    wide = synth(3, clens, mean_run=40000, gap_prob=0.3, seed=6)
    whole = RunLists.from_lists([[[(1, clens[0] + 1, 3.0)], [(1, clens[1] + 1, 0.5)]]])
    whole = RunLists(whole.n_chrom, 1, whole.seg_off, whole.start, whole.finish, whole.value.astype(np.float32), whole.defaults)
    t = merge_tracks([wide, dense, whole])
    d = t.as_dict()
    ts = engine.TrackSet.from_runlists(t)
    assert ts.validate() == (0, -1)
    for op, kw in (("sum", {}), ("mean", dict(flags=1)), ("max", {}), ("var", {}), ("median", {}), ("mwu", dict(n_set0=4))):
        exp = oracle.reduce(d, op, **kw)
        got = ts.reduce_host(op, **kw)
        assert_runs_equal(got, exp, _tol(op), "regions op %s" % op)
    exp = oracle.multiplex(d)
    got = ts.multiplex_host()
    for a, b in zip(got, exp):
        assert np.array_equal(a, b, equal_nan=True)
    ts.close()


def test_gpu_many_small_chromosomes(oracle, engine, monkeypatch):
    """A scaffold-level assembly: hundreds of short chromosomes, several of them empty in some or all
    tracks (one window each, window tables and the index walk across empty segments)."""
    from wiggletools_amd.runlists import RunLists, synth
    monkeypatch.setenv("WTAMD_DELTA_MIN_TRACKS", "1")
    rng = np.random.default_rng(12)
    clens = [int(x) for x in rng.integers(1, 3000, 400)]
    t = synth(6, clens, mean_run=40, gap_prob=0.5, seed=13)
    # empty every 7th chromosome entirely and every 5th in track 2
    keep = np.ones(len(t.start), bool)
    for c in range(t.n_chrom):
        for i in range(t.n_tracks):
            if c % 7 == 3 or (c % 5 == 1 and i == 2):
                keep[t.seg_off[c * t.n_tracks + i]:t.seg_off[c * t.n_tracks + i + 1]] = False
    seg = np.concatenate([[0], np.cumsum([int(keep[t.seg_off[q]:t.seg_off[q + 1]].sum()) for q in range(t.n_chrom * t.n_tracks)])])
    t = RunLists(t.n_chrom, t.n_tracks, seg, t.start[keep], t.finish[keep], t.value[keep], t.defaults)
    d = t.as_dict()
    ts = engine.TrackSet.from_runlists(t)
    for op, kw in (("sum", {}), ("mean", dict(flags=1)), ("min", {}), ("stddev", {}), ("median", {}), ("ttest", dict(n_set0=3))):
        exp = oracle.reduce(d, op, **kw)
        got = ts.reduce_host(op, **kw)
        assert_runs_equal(got, exp, _tol(op), "scaffolds op %s" % op)
    ts.close()


def test_gpu_searched_window_index_equals_scanned():
    """The window index is built by search (coarse binary search + interpolation, csrc/wt_engine.hip
    wt_index_search_kernel); WTAMD_INDEX_CHECK=1 also builds it by the scan over every finish[] and fails the
    reduction if any entry differs.  The switch is read once per process: a child runs the cases."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, sys
sys.path.insert(0, "tests")
from helpers import random_case
from wiggletools_amd import engine as E
for seed in range(40):
    rng = np.random.default_rng(9000 + seed)
    t = random_case(9000 + seed, n_tracks=int(rng.integers(1, 40)), dtype=np.float32)
    ts = E.TrackSet.from_runlists(t)
    for op in ("mean", "max"):
        ts.reduce_host(op)
    ts.close()
print("index-check-ok")
'''
    env = dict(os.environ, WTAMD_INDEX_CHECK="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "index-check-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("seed", range(10))
def test_gpu_difference_array_nonzero_defaults(oracle, engine, seed, monkeypatch):
    """Non-zero defaults that are floats stay on the difference-array kernel (wt_delta_kernel<OP, DF = true>):
    bit-identical Sum / Mean; a default far outside the data's exponent range goes to the patch / general kernel
    and is still bit-identical; defaults that are not floats never take the path."""
    from wiggletools_amd.runlists import synth
    monkeypatch.setenv("WTAMD_DELTA_MIN_TRACKS", "1")
    rng = np.random.default_rng(4200 + seed)
    n = int(rng.integers(2, 40)) if seed % 3 else int(rng.integers(520, 700))       # (one chunk of the default 1024-lane plan;
    #  more tracks than lanes: test_gpu_difference_array_more_tracks_than_lanes)
    t = synth(n, [int(rng.integers(3000, 60000)), 9000, 3], mean_run=float(rng.choice([1, 4, 16, 300])),
              gap_prob=float(rng.choice([0.0, 0.1, 0.5])), seed=seed, first_start=int(rng.choice([1, 77, 5000])))
    t.value[:] = (rng.integers(-800, 800, len(t.value)) / 8.0).astype(np.float32)
    t.defaults[:] = np.where(rng.random(n) < 0.6, rng.choice(np.array([1.5, -2.25, 0.125, 3.0, 1024.0, 7.0]), n), 0.0)
    t.defaults[0] = 2.5
    d = t.as_dict()
    ts = engine.TrackSet.from_runlists(t)
    for strict in (0, 1):
        for op in ("sum", "mean"):
            got = ts.reduce_host(op, flags=strict)
            st = ts.stats()
            assert st["kernel"] == 1 and st["patched_windows"] == 0, st
            assert_runs_equal(got, oracle.reduce(d, op, flags=strict), 0.0, "seed %d op %s strict %d" % (seed, op, strict))
    ts.close()
    t.defaults[1 % n] = float(np.float32(1e-30))           # no window can be exact with this term
    ts = engine.TrackSet.from_runlists(t)
    for op in ("sum", "mean"):
        assert_runs_equal(ts.reduce_host(op), oracle.reduce(t.as_dict(), op), 0.0, "far default, seed %d op %s" % (seed, op))
    ts.close()
    t.defaults[1 % n] = 0.1                                # not a float
    ts = engine.TrackSet.from_runlists(t)
    got = ts.reduce_host("mean")
    assert ts.stats()["kernel"] == 0
    assert_runs_equal(got, oracle.reduce(t.as_dict(), "mean"), 0.0, "non-float default, seed %d" % seed)
    ts.close()


@pytest.mark.parametrize("case", [(1100, None, 0), (1500, None, 1), (700, "256", 0), (700, "256", 1), (1300, "512", 1)])
def test_gpu_difference_array_more_tracks_than_lanes(oracle, engine, case, monkeypatch):
    """Sum / Mean on the difference-array kernel with MORE TRACKS THAN LANES -- the multi-chunk walk of
    wt_delta_kernel<sum / mean> (csrc/wt_engine.hip, `N > T`): at the default 1024-lane plan (N = 1100, 1500) and with
    WTAMD_DELTA_T = 256 / 512, zero and non-zero (float) defaults, strict and not, against the oracle at tolerance 0.
    (Round 3 moved the default plan from 512 to 1024 lanes; the 520-700-track cases of the tests above stopped chunking.)"""
    from wiggletools_amd.runlists import synth
    n, T, with_defaults = case
    monkeypatch.setenv("WTAMD_DELTA_MIN_TRACKS", "1")
    if T:
        monkeypatch.setenv("WTAMD_DELTA_T", T)
    else:
        monkeypatch.delenv("WTAMD_DELTA_T", raising=False)
    rng = np.random.default_rng(n + 7 * with_defaults)
    t = synth(n, [20000, 3], mean_run=40, gap_prob=0.1, seed=n)
    t.value[:] = (rng.integers(-800, 800, len(t.value)) / 8.0).astype(np.float32)
    if with_defaults:
        t.defaults[:] = np.where(rng.random(n) < 0.5, rng.choice(np.array([1.5, -2.25, 0.125, 3.0]), n), 0.0)
        t.defaults[n - 1] = 2.5         # a default in the LAST chunk
    d = t.as_dict()
    ts = engine.TrackSet.from_runlists(t)
    lanes = int(T) if T else 1024
    for strict in (0, 1):
        for op in ("sum", "mean"):
            got = ts.reduce_host(op, flags=strict)
            st = ts.stats()
            assert st["kernel"] == 1 and st["window_bp"] == 8 * lanes and st["patched_windows"] == 0, st
            assert n > lanes
            assert_runs_equal(got, oracle.reduce(d, op, flags=strict), 0.0, "N %d T %s op %s strict %d" % (n, T, op, strict))
    ts.close()


@pytest.mark.parametrize("T", ["64", "128"])
def test_gpu_difference_array_smallest_workgroups(oracle, engine, T, monkeypatch):
    """The difference-array kernels on one and two wavefronts per workgroup (WTAMD_DELTA_T = 64 / 128: 512- / 1024-bp windows): round 6's
    per-wavefront run counts and sub-range ranks (epfx[0 .. 16), epfx[32 .. 48): a 512-bp window has 9 words of its own) and the staging's
    spare entries; with a NaN and an Inf in the data the reduction falls back as a whole; Sum / Mean / Max / Min against the oracle at tolerance 0."""
    from wiggletools_amd.runlists import synth
    monkeypatch.setenv("WTAMD_DELTA_T", T)
    monkeypatch.setenv("WTAMD_DELTA_MIN_TRACKS", "1")
    rng = np.random.default_rng(int(T))
    for bad in (False, True):
        t = synth(30, [40000, 700], mean_run=6.0, seed=int(T) + bad, gap_prob=0.05, dtype=np.float32, value_levels=800)
        t.value[:] = (t.value - 30).astype(np.float32)
        if bad:
            t.value[11] = np.nan
            t.value[len(t.value) // 2] = np.inf
        ts = engine.TrackSet.from_runlists(t)
        for op in ("sum", "mean", "max", "min"):
            for flags in (0, 1):
                got = ts.reduce_host(op, flags=flags)
                st = ts.stats()
                if not bad:
                    assert st["kernel"] == 1 and st["window_bp"] == 8 * int(T) and st["patched_windows"] == 0, st
                # (windows to patch: the general kernel's windows are wider than these, so the whole reduction is the general kernel's --
                #  wt_launch_patch: "no general plan compatible with the difference-array windows"; the result is the oracle's either way)
                assert_runs_equal(got, oracle.reduce(t.as_dict(), op, flags=flags), 0.0, "T %s op %s flags %d bad %s" % (T, op, flags, bad))
        ts.close()


def test_gpu_c2_shape_long_runs(oracle, engine):
    """BASELINE config C2's shape (mean, 100 float tracks, 2 % gaps, values k / 8) at MEAN RUN 200 bp: an 8192-bp window
    of the default plan holds ~40 runs per track instead of 512 -- per-window fixed costs, window-base runs spanning
    several windows -- against the oracle at tolerance 0."""
    from wiggletools_amd.runlists import synth
    t = synth(100, [300000, 70000], mean_run=200, gap_prob=0.02, seed=2024)
    d = t.as_dict()
    ts = engine.TrackSet.from_runlists(t)
    for strict in (0, 1):
        for op in ("mean", "sum"):
            got = ts.reduce_host(op, flags=strict)
            st = ts.stats()
            assert st["kernel"] == 1 and st["window_bp"] == 8192, st
            assert_runs_equal(got, oracle.reduce(d, op, flags=strict), 0.0, "run 200 op %s strict %d" % (op, strict))
    ts.close()


def test_gpu_difference_array_every_run_crosses_a_window_edge(oracle, engine):
    """Runs LONGER than a window, 300 tracks: every run of every window crosses an edge (spans w0, ends beyond w1, or
    both), a window's flat space is one or two runs per track, and lane l of a wavefront holds flat indices l, l + 64,
    l + 128, l + 192 of a tile -- several parked runs per lane, i.e. the in-loop flush of wt_delta_apply_or_park, not
    only the one at the end of the pass.  Sum / mean (DF and not) and the squares, against the oracle."""
    from wiggletools_amd.runlists import RunLists, synth
    t = synth(300, [120000, 30000], mean_run=11000, gap_prob=0.05, seed=77)
    d = t.as_dict()
    ts = engine.TrackSet.from_runlists(t)
    for op, tol in (("sum", 0.0), ("mean", 0.0), ("var", 1e-12), ("stddev", 1e-12)):
        for strict in (0, 1):
            got = ts.reduce_host(op, flags=strict)
            assert ts.stats()["kernel"] == 1, ts.stats()
            assert_runs_equal(got, oracle.reduce(d, op, flags=strict), tol, "long runs op %s strict %d" % (op, strict))
    ts.close()
    # non-zero defaults (the DF instantiation parks the default's bits with the run)
    t2 = RunLists(t.n_chrom, t.n_tracks, t.seg_off, t.start, t.finish, t.value, np.where(np.arange(t.n_tracks) % 3 == 0, 1.5, 0.0))
    ts = engine.TrackSet.from_runlists(t2)
    for op in ("sum", "mean"):
        got = ts.reduce_host(op)
        assert ts.stats()["kernel"] == 1, ts.stats()
        assert_runs_equal(got, oracle.reduce(t2.as_dict(), op), 0.0, "long runs, defaults, op %s" % op)
    ts.close()


@pytest.mark.gpu
def test_gpu_mwu_kernel_by_the_values(oracle, engine):
    """Round 6: MWUReduction's kernel follows the VALUES (csrc/wt_engine.hip wt_mwu_few_ties: 4096 of them sampled once per track set).
    Values that are nearly all distinct -- equal values at one position are rare -- walk (csrc/wt_mwalk.h, kernel 3: 28.5 against 35.8 ms
    on chromosome 21); the generator's few levels stay on the register columns (kernel 0: 35.8 against 43.8 ms).  Either way the
    oracle's bits."""
    from wiggletools_amd.runlists import synth
    rng = np.random.default_rng(77)
    for distinct, want in ((True, 3), (False, 0)):
        t = synth(60, [90000, 1500], mean_run=12.0, seed=3, gap_prob=0.05, dtype=np.float32, value_levels=800)
        if distinct:
            t.value[:] = rng.normal(0, 5, len(t.value)).astype(np.float32)
        ts = engine.TrackSet.from_runlists(t)
        got = ts.reduce_host("mwu", n_set0=25)
        assert ts.stats()["kernel"] == want, (distinct, ts.stats())
        exp = oracle.reduce(t.as_dict(), "mwu", n_set0=25)
        assert_runs_equal(got, exp, 0.0, "distinct values %s" % distinct)
        ts.close()


@pytest.mark.parametrize("seed", range(12))
def test_gpu_mwu_walk_paths(oracle, engine, seed, monkeypatch):
    """MWUReduction by walking (csrc/wt_mwalk.h) on the device: the default plan, short stretches, slots too few for the data
    (overflow list), no overflow list at all (fallback: sorted events, in rounds) -- against the oracle's literal scan
    (setComparisons.c:293-366) and against the bitmap kernel (the default; the walking kernel is selected by WTAMD_MWALK=1: it
    measured slower), tolerance 0 (the erf table is the host's);
    set sizes 1 v 1 ... 64 v 64, value levels from "everything ties" (more tie groups than the lanes keep: enumeration) to a
    few groups, NaN, non-zero defaults, both strict flags."""
    from wiggletools_amd.runlists import synth
    rng = np.random.default_rng(8000 + seed)
    n1, n2 = [(1, 1), (2, 3), (8, 8), (17, 33), (50, 50), (64, 64), (3, 50), (40, 9)][seed % 8]
    n = n1 + n2
    defaults = rng.integers(-3, 4, n).astype(np.float64) / 4.0 if rng.random() < 0.4 else None
    t = synth(n, [int(rng.integers(3000, 40000)), int(rng.integers(1, 900))], mean_run=float(rng.choice([1, 2, 5, 16, 70])), seed=seed,
              gap_prob=float(rng.choice([0, 0.05, 0.5])), dtype=np.float32, value_levels=int(rng.choice([2, 5, 14, 30, 800])),
              nan_prob=float(rng.choice([0, 0, 0.001])), defaults=defaults)
    env = [dict(), dict(WTAMD_WALK_S="8"), dict(WTAMD_WALK_S="32"), dict(WTAMD_WALK_CAPP="2"),
           dict(WTAMD_WALK_CAPP="2", WTAMD_WALK_OV="0"), dict(WTAMD_WALK_CAPP="4", WTAMD_WALK_OV="0", WTAMD_WALK_S="4")][seed % 6]
    flags = int(rng.choice([0, 0, 1, 2, 3]))
    d = t.as_dict()
    exp = oracle.reduce(d, "mwu", flags=flags, n_set0=n1)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    import subprocess, sys, json, os, tempfile
    # (the engine reads WTAMD_MWALK once per process: the walking kernel runs in a process of its own)
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "t.npz"), seg_off=t.seg_off, start=t.start, finish=t.finish, value=t.value, defaults=t.defaults)
        code = ("import sys, numpy as np; sys.path.insert(0, %r); from wiggletools_amd import engine; from wiggletools_amd.runlists import RunLists;"
                "z = np.load(%r); t = RunLists(%d, %d, z['seg_off'], z['start'], z['finish'], z['value'], z['defaults']);"
                "ts = engine.TrackSet.from_runlists(t); c, s, f, v = ts.reduce_host('mwu', flags=%d, n_set0=%d); assert ts.stats()['kernel'] == 3, ts.stats();"
                "np.savez(%r, c=c, s=s, f=f, v=v)"
                % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(td, "t.npz"), t.n_chrom, t.n_tracks, flags, n1, os.path.join(td, "o.npz")))
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, WTAMD_MWALK="1"), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        z = np.load(os.path.join(td, "o.npz"))
        got = (z["c"], z["s"], z["f"], z["v"])
    assert_runs_equal(got, exp, 0.0, "walking %s" % (env,))
    ts = engine.TrackSet.from_runlists(t)
    old = ts.reduce_host("mwu", flags=flags, n_set0=n1)
    assert ts.stats()["kernel"] == 0
    assert_runs_equal(old, exp, 0.0, "bitmap kernel")
    ts.close()


@pytest.mark.parametrize("seed", range(18))
def test_gpu_median_walk_paths(oracle, engine, seed, monkeypatch):
    """MedianReduction by walking (csrc/wt_walk.h) on the device: the default plan, small workgroups / stretches, slots too
    few for the data (overflow list) and no overflow list at all (every window falls back to the sorted events, in rounds)
    -- against the oracle (reducers.c:780-813) and against the bitmap kernel (WTAMD_NO_WALK), tolerance 0; gaps, ties,
    NaN, non-zero defaults, the strict predicate."""
    from wiggletools_amd.runlists import synth
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.choice([1, 3, 8, 33, 64, 100, 128]))
    defaults = rng.integers(-3, 4, n).astype(np.float64) / 4.0 if rng.random() < 0.4 else None
    t = synth(n, [int(rng.integers(3000, 60000)), int(rng.integers(1, 900))], mean_run=float(rng.choice([1, 2, 5, 16, 70])), seed=seed,
              gap_prob=float(rng.choice([0, 0.05, 0.5])), dtype=np.float32, value_levels=int(rng.choice([2, 5, 800])),
              nan_prob=float(rng.choice([0, 0, 0.001])), defaults=defaults)
    env = [dict(), dict(WTAMD_WALK_T="128"), dict(WTAMD_WALK_T="64", WTAMD_WALK_S="8"), dict(WTAMD_WALK_CAPP="2"),
           dict(WTAMD_WALK_CAPP="2", WTAMD_WALK_OV="0"), dict(WTAMD_WALK_CAPP="2", WTAMD_WALK_OV="0", WTAMD_WALK_T="64", WTAMD_WALK_S="4")][seed % 6]
    env = dict(env, WTAMD_WALK_PAIR=str((seed // 3) % 2))       # one lane per stretch / two (the default)
    flags = int(rng.choice([0, 0, 1]))
    d = t.as_dict()
    exp = oracle.reduce(d, "median", flags=flags)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ts = engine.TrackSet.from_runlists(t)
    got = ts.reduce_host("median", flags=flags)
    assert ts.stats()["kernel"] == 2, "the walking kernel did not run"
    assert_runs_equal(got, exp, 0.0, "walking %s" % (env,))
    ts.close()
    monkeypatch.setenv("WTAMD_NO_WALK", "1")
    ts = engine.TrackSet.from_runlists(t)
    old = ts.reduce_host("median", flags=flags)
    assert ts.stats()["kernel"] == 0
    assert_runs_equal(old, exp, 0.0, "bitmap kernel")
    ts.close()
