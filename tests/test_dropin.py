"""The DROP-IN layer: the reference's own C API (newMultiplexer, MeanReduction, newMultiset,
MWUReduction, seek, popMultiplexer, seekMultiset ...) exported by the product, driven by the very
harness that drives the compiled reference (oracle/ref_harness.c) and compared with the oracle and
with the compiled reference.  Array-backed child WiggleIterators are popped one interval at a
time, exactly as the reference's readers would be.

Two backends run the same cases:
  * "amd"  (-m gpu): wiggletools_amd/csrc/libwiggletools_amd.so on a real MI355X -- the parity tests proper;
  * "emu"  (CPU):    the product's csrc/wt_iter_abi.cpp linked against the emulated pipeline
                     (tests/emu/wt_pipe_emu.cpp), so the host logic of the layer -- batch cuts,
                     carried intervals, sentinels, take-over, seek, slot bookkeeping -- is checked
                     in the GPU-less container too.
"""
import numpy as np
import pytest

from helpers import ALL_MULTIPLEX_OPS, assert_runs_equal, random_case

EXACT = {"sum", "product", "mean", "min", "max", "median"}


def _tol(op):
    return 0.0 if op in EXACT else (1e-9 if op in ("ttest", "mwu") else 1e-12)


_harness = {}


def _get(oracle, backend):
    if backend not in _harness:
        if backend == "amd":
            import torch
            assert torch.cuda.is_available()
            from wiggletools_amd import _lib
            _harness[backend] = oracle.Harness(_lib.LIB_PATH, "amd")
        else:
            from emu import build as emu_build
            _harness[backend] = oracle.Harness(emu_build.build_dropin(), "emu")
    return _harness[backend]


@pytest.fixture(params=["emu", pytest.param("amd", marks=pytest.mark.gpu)])
def H(request, oracle):
    return _get(oracle, request.param)


@pytest.fixture
def tiny_batches(monkeypatch):
    """Cuts a batch every few bp / intervals and starts the staging at 3 entries: every seam rule
    (interval reaching the cut, finishing exactly at it, sentinel, empty batch) fires many times."""
    monkeypatch.setenv("WTAMD_MIN_SPAN", "7")
    monkeypatch.setenv("WTAMD_BATCH_INTERVALS", "20")
    monkeypatch.setenv("WTEMU_PIPE_CAP", "3")


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


def clip(t, chrom, start, finish):
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


def _regions(rng, t, k):
    for _ in range(k):
        c = int(rng.integers(0, t.n_chrom))
        s = int(rng.integers(1, 2500))
        yield c, s, s + int(rng.integers(1, 1500))


@pytest.mark.parametrize("seed", range(5))
def test_dropin_seek_vs_reference(oracle, H, seed):
    """`seek chr s f <reducer>` as the CLI issues it (children held until the seek,
    commandParser.c:615-624; then reducers.c:25-29 -> multiplexer.c:130-141) against the COMPILED
    REFERENCE doing the same, and against the oracle over the tracks clipped to the region."""
    if not oracle.have_ref():
        pytest.skip("compiled reference not available")
    R = oracle.ref_harness()
    rng = np.random.default_rng(seed)
    t = random_case(7400 + seed, n_tracks=5, n_chrom=2, max_len=3000)
    d = t.as_dict()
    for (c, s, f) in _regions(rng, t, 4):
        for op in ("mean", "median", "max"):
            for strict in (0, 1):
                ref = R.reduce_seek_held(d, op, c, s, f, flags=strict)
                got = H.reduce_seek_held(d, op, c, s, f, flags=strict)
                assert_runs_equal(got, ref, 0.0, "seek %s %s strict %d vs reference" % ((c, s, f), op, strict))
                exp = oracle.reduce(clip(t, c, s, f).as_dict(), op, flags=strict)
                assert_runs_equal(got, exp, 0.0, "seek %s %s strict %d vs oracle" % ((c, s, f), op, strict))


@pytest.mark.parametrize("seed", range(3))
def test_dropin_seek_multiset_vs_reference(oracle, H, seed):
    """seekMultiset (multiSet.c:103-113) over two Multiplexers of held children."""
    if not oracle.have_ref():
        pytest.skip("compiled reference not available")
    R = oracle.ref_harness()
    rng = np.random.default_rng(100 + seed)
    t = random_case(7450 + seed, n_tracks=6, n_chrom=2, max_len=3000)
    d = t.as_dict()
    for (c, s, f) in _regions(rng, t, 4):
        for flags in (0, 3):
            exp = R.multiset_seek_held(d, 3, c, s, f, flags)
            got = H.multiset_seek_held(d, 3, c, s, f, flags)
            assert len(got[0]) == len(exp[0]), (c, s, f, flags)
            for a, b in zip(got, exp):
                assert np.array_equal(a, b, equal_nan=True)


def test_dropin_seek_two_sample(oracle, H):
    """SetComparisonSeek / MWUSeek (setComparisons.c:25-29,253-257): the two-sample reducers after a
    held seek == the oracle over the clipped tracks (setComparisons.c cannot be compiled here)."""
    t = random_case(7460, n_tracks=8, n_chrom=2, max_len=3000)
    d = t.as_dict()
    rng = np.random.default_rng(3)
    for (c, s, f) in _regions(rng, t, 3):
        for op in ("ttest", "mwu"):
            got = H.reduce_seek_held(d, op, c, s, f, n_set0=4)
            exp = oracle.reduce(clip(t, c, s, f).as_dict(), op, n_set0=4)
            assert_runs_equal(got, exp, _tol(op), "seek %s %s" % ((c, s, f), op))


def test_dropin_seek_primed(oracle, H):
    """seek on a reducer whose children already delivered data == the reducer over the clipped
    tracks.  (The REFERENCE is deliberately not consulted here: seeking a Multiplexer that was
    already primed leaves stale inplay[]/values[] behind, multiplexer.c:130-141 resets the heaps
    but not those arrays; the CLI never does that.)"""
    t = random_case(7400, n_tracks=5, n_chrom=2, max_len=8000)
    d = t.as_dict()
    for (c, s, f) in ((0, 100, 3000), (1, 1, 50), (0, 2500, 2600), (1, 10, 4000)):
        for op in ("mean", "median"):
            got = H.reduce_seek(d, op, c, s, f)
            exp = oracle.reduce(clip(t, c, s, f).as_dict(), op)
            assert_runs_equal(got, exp, 0.0, "seek %s %s" % ((c, s, f), op))


def test_dropin_batches_cross_seams(oracle, H):
    """Long tracks: several batches, intervals crossing every cut."""
    from wiggletools_amd.runlists import synth
    t = synth(8, [300000, 5000], mean_run=40, seed=9, gap_prob=0.1)
    d = t.as_dict()
    for op in ("mean", "max"):
        exp = oracle.reduce(d, op)
        got = H.reduce(d, op)
        assert_runs_equal(got, exp, 0.0, op)


@pytest.mark.parametrize("seed", range(30))
def test_dropin_tiny_batches(oracle, H, tiny_batches, seed):
    """A cut every ~7 bp: gapped tracks whose intervals finish exactly at a cut (the run that starts
    there must not be lost), cross it, or start right after it."""
    t = random_case(9000 + seed, max_len=600)
    d = t.as_dict()
    for strict in (0, 1):
        for op in ("mean", "median", "var"):
            exp = oracle.reduce(d, op, flags=strict)
            got = H.reduce(d, op, flags=strict)
            assert_runs_equal(got, exp, _tol(op), "seed %d %s strict %d" % (seed, op, strict))
    exp = oracle.multiplex(d)
    got = H.multiplex(d)
    assert len(got[0]) == len(exp[0])
    for a, b in zip(got, exp):
        assert np.array_equal(a, b, equal_nan=True)


def test_dropin_cut_at_finish_example(oracle, H, monkeypatch):
    """The advisor's example: A=[1,100), B=[10,50), cut at 50: the run [50,100) starts where B
    finishes, exactly on the cut."""
    from wiggletools_amd.runlists import RunLists
    monkeypatch.setenv("WTAMD_MIN_SPAN", "49")       # first batch = run starts in [1, 50)
    monkeypatch.setenv("WTAMD_BATCH_INTERVALS", "1")
    t = RunLists.from_lists([[[(1, 100, 2.0)]], [[(10, 50, 4.0)]]])
    got = H.reduce(t.as_dict(), "sum")
    assert got[1].tolist() == [1, 10, 50] and got[2].tolist() == [10, 50, 100]
    assert got[3].tolist() == [2.0, 6.0, 2.0]


def test_dropin_float64_switch(oracle, H, tiny_batches):
    """Values are staged as float32 until one is not float32-exact; from then on float64."""
    from wiggletools_amd.runlists import synth
    t = synth(4, [3000], mean_run=6, seed=5, dtype=np.float64)
    t.value[len(t.value) // 2] = 0.1            # not float32-exact, in the middle of a track
    t.value[-3] = 1e-300
    d = t.as_dict()
    for op in ("sum", "mean", "median", "stddev"):
        assert_runs_equal(H.reduce(d, op), oracle.reduce(d, op), _tol(op), op)


def test_dropin_ctor_defaults(oracle, H):
    for dv in (np.zeros(3), np.array([1.0, 2.0, 3.5]), np.array([1.0, np.nan, 2.0])):
        for op in ALL_MULTIPLEX_OPS:
            a, b = H.reducer_default(op, dv), oracle.reducer_default(op, dv)
            assert (np.isnan(a) and np.isnan(b)) or a == b, (op, dv)


@pytest.fixture(params=[(1, 0), (1, 1), (0, 1)], ids=["bulk-children", "bulk-children+blocks", "blocks"])
def doors(request, H):
    """The bulk doors of the drop-in layer: children that are the library's own wtamd_ArrayReader
    (whole SoA blocks instead of one pop per interval; big blocks go to HBM unstaged) and / or the
    reducer's runs taken through wtamd_iterator_next_block (mixed with a few plain pops)."""
    H.set_modes(*request.param)
    yield request.param
    H.set_modes(0, 0)


@pytest.mark.parametrize("env", [{}, {"WTAMD_MIN_SPAN": "7", "WTAMD_BATCH_INTERVALS": "20", "WTEMU_PIPE_CAP": "3"},
                                 {"WTAMD_MIN_SPAN": "300", "WTAMD_BATCH_INTERVALS": "2000"}],
                         ids=["default-batches", "tiny-batches", "small-batches"])
def test_dropin_bulk_doors(oracle, H, doors, env, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for seed in range(12):
        t = random_case(9300 + seed, max_len=3000)
        d = t.as_dict()
        for strict in (0, 1):
            for op in ("mean", "median", "var", "max"):
                assert_runs_equal(H.reduce(d, op, flags=strict), oracle.reduce(d, op, flags=strict), _tol(op),
                                  "doors %s seed %d %s strict %d" % (doors, seed, op, strict))
        if t.n_chrom > 1:       # seek on array readers: one chromosome, intervals clipped to the window
            got = H.reduce_seek(d, "mean", 1, 50, 900)
            assert_runs_equal(got, oracle.reduce(clip(t, 1, 50, 900).as_dict(), "mean"), 0.0, "seek, doors %s" % (doors,))
    if doors[0]:
        t = random_case(9400, n_tracks=7, max_len=3000)
        d = t.as_dict()
        for op in ("ttest", "mwu"):
            assert_runs_equal(H.reduce(d, op, n_set0=3), oracle.reduce(d, op, n_set0=3), _tol(op), op)
        exp, got = oracle.multiplex(d), H.multiplex(d)
        assert len(got[0]) == len(exp[0])
        for a, b in zip(got, exp):
            assert np.array_equal(a, b, equal_nan=True)


def test_dropin_bulk_long_tracks(oracle, H):
    """Blocks big enough to bypass the staging (>= 64 intervals per track and batch)."""
    from wiggletools_amd.runlists import synth
    H.set_modes(1, 1)
    try:
        t = synth(6, [200000, 30000], mean_run=12, seed=21, gap_prob=0.05)
        d = t.as_dict()
        for op in ("mean", "min"):
            assert_runs_equal(H.reduce(d, op), oracle.reduce(d, op), 0.0, op)
    finally:
        H.set_modes(0, 0)


def test_dropin_bulk_past_the_staging(oracle, H, monkeypatch):
    """Unstaged blocks push the batch's interval count far past the (3-entry) staging arrays; the
    few intervals that are staged after them (window edges, short blocks) must grow the staging
    first."""
    from wiggletools_amd.runlists import synth
    monkeypatch.setenv("WTEMU_PIPE_CAP", "3")
    H.set_modes(1, 1)
    try:
        t = synth(5, [60000, 900], mean_run=10, seed=23, gap_prob=0.05)
        d = t.as_dict()
        assert_runs_equal(H.reduce(d, "mean"), oracle.reduce(d, "mean"), 0.0, "mean")
        got = H.reduce_seek(d, "max", 0, 20000, 55000)
        assert_runs_equal(got, oracle.reduce(clip(t, 0, 20000, 55000).as_dict(), "max"), 0.0, "seek")
    finally:
        H.set_modes(0, 0)


def test_dropin_mixed_children_and_float64_switch(oracle, H, monkeypatch):
    """Even tracks are the library's bulk-capable array readers (float32), odd tracks foreign lazy
    iterators, one of which delivers a value that is not float32-exact in the middle of a batch:
    the bulk blocks are then staged (block copy) and the whole batch switches to float64."""
    from wiggletools_amd.runlists import synth
    monkeypatch.setenv("WTEMU_PIPE_CAP", "3")
    H.set_modes(2, 0)
    try:
        t = synth(6, [50000, 4000], mean_run=8, seed=29, gap_prob=0.05, dtype=np.float64)
        d = t.as_dict()
        for op in ("mean", "median"):
            assert_runs_equal(H.reduce(d, op), oracle.reduce(d, op), 0.0, op + " (float32-exact)")
        N = t.n_tracks
        a, b = int(t.seg_off[1]), int(t.seg_off[2])         # track 1 (foreign), chromosome 0
        t.value[(a + b) // 2] = 0.1
        d = t.as_dict()
        for op in ("sum", "median", "stddev"):
            assert_runs_equal(H.reduce(d, op), oracle.reduce(d, op), _tol(op), op + " (float64 switch)")
    finally:
        H.set_modes(0, 0)


@pytest.mark.parametrize("seed", range(8))
def test_dropin_parallel_drain(oracle, H, tiny_batches, monkeypatch, seed):
    """Foreign children dealt to worker threads (a child always to the same one), batches laid out in
    track order afterwards: same runs as the single-threaded drain -- seams, carried intervals,
    sentinels, the float64 switch, two-sample sets, seek."""
    monkeypatch.setenv("WTAMD_DRAIN_THREADS", "3")
    t = random_case(9500 + seed, n_tracks=int(4 + seed), max_len=900)
    if seed % 2:
        t.value[:: 7] = 0.1 + t.value[:: 7]         # not float32-exact: the batch that meets one switches to float64
    d = t.as_dict()
    for strict in (0, 1):
        for op in ("mean", "max", "stddev", "median"):
            assert_runs_equal(H.reduce(d, op, flags=strict), oracle.reduce(d, op, flags=strict), _tol(op),
                              "threads seed %d %s strict %d" % (seed, op, strict))
    n0 = t.n_tracks // 2
    for op in ("ttest", "mwu"):
        if n0 >= 3 and t.n_tracks - n0 >= 3:
            assert_runs_equal(H.reduce(d, op, n_set0=n0), oracle.reduce(d, op, n_set0=n0), _tol(op), "threads %s" % op)


MAPS = [("scale", -2.5), ("offset", 3.25), ("ln", 0.0), ("log", 2.0), ("exp", 0.0), ("expb", 2.0), ("pow", 2.0),
        ("abs", 0.0), ("gt", 2.5), ("gte", 2.5), ("lt", 2.5), ("lte", -1.0)]


@pytest.mark.parametrize("mop,param", MAPS)
def test_dropin_map_vs_reference_operator_iterators(oracle, H, mop, param, tiny_batches):
    """`<reducer> map <op> tracks...`: the COMPILED REFERENCE with its own operator iterators in front of
    its Multiplexer (ScaleWiggleIterator, NaturalLogWiggleIterator, ..., HighPassFilterWiggleIterator;
    lt / lte as commandParser.c:185-199 builds them) against this library with wtamd_MapIterator, whose
    chains run inside the pipeline.  Same harness, same children.  Emulated pipe: bit for bit (one libm);
    product: 1e-12 for the transcendental operators (device libm), exact otherwise."""
    R = oracle.ref_harness()
    if R is None:
        pytest.skip("compiled reference not available")
    exact = mop in ("scale", "offset", "abs", "gt", "gte", "lt", "lte")
    for seed in range(3):
        t = random_case(9800 + seed, n_tracks=int(3 + 2 * seed), max_len=700, dtype=np.float32)
        rng = np.random.default_rng(seed)
        t.value[:] = (t.value * rng.choice([1.0, -1.0, 0.0], size=len(t.value), p=[0.6, 0.3, 0.1])).astype(np.float32)
        t.value[:] = np.where(np.abs(t.value) > 60, t.value / 16, t.value)
        d = t.as_dict()
        try:
            R.set_map(mop, param); H.set_map(mop, param)
            for op in ("sum", "mean", "stddev", "median", "max"):
                for strict in (0, 1):
                    exp = R.reduce(d, op, flags=strict)
                    got = H.reduce(d, op, flags=strict)
                    tol = 0.0 if (exact and op in ("sum", "mean", "median", "max")) else 1e-12
                    assert_runs_equal(got, exp, tol, "map %s %s seed %d strict %d" % (mop, op, seed, strict))
        finally:
            R.set_map(None); H.set_map(None)


@pytest.mark.parametrize("mop,param", [("scale", 2.0), ("ln", 0.0), ("gte", 1.0)])
def test_dropin_seek_with_mapped_children_vs_reference(oracle, H, mop, param):
    """`seek chr s f <reducer> map <op> ...` with held children: the reference forwards the seek through its
    operator iterators to the readers (unaryOps.c UnaryWiggleIteratorSeek), this library seeks the raw
    children it unwrapped and maps the region's batches on device."""
    R = oracle.ref_harness()
    if R is None:
        pytest.skip("compiled reference not available")
    rng = np.random.default_rng(77)
    t = random_case(7480, n_tracks=5, n_chrom=2, max_len=3000, dtype=np.float32)
    t.value[:] = (t.value * rng.choice([1.0, -1.0, 0.0], size=len(t.value), p=[0.6, 0.3, 0.1])).astype(np.float32)
    d = t.as_dict()
    try:
        R.set_map(mop, param); H.set_map(mop, param)
        for (c, s, f) in _regions(rng, t, 4):
            for op in ("mean", "max"):
                for strict in (0, 1):
                    ref = R.reduce_seek_held(d, op, c, s, f, flags=strict)
                    got = H.reduce_seek_held(d, op, c, s, f, flags=strict)
                    assert_runs_equal(got, ref, 0.0 if mop != "ln" else 1e-12, "seek %s map %s %s strict %d" % ((c, s, f), mop, op, strict))
    finally:
        R.set_map(None); H.set_map(None)


@pytest.mark.parametrize("tiny", [False, True])
def test_dropin_children_reusing_one_name_buffer(oracle, H, tiny, monkeypatch):
    """Foreign children that rewrite ONE chromosome-name buffer (same pointer, new content) and are SPARSE: the next
    chromosome's first start lies beyond the previous chromosome's last finish, so no coordinate goes backwards at the
    transition.  The reference compares names by content on every pop (multiplexer.c:56); round 3 re-compared only on
    a coordinate regression and merged such a track's next chromosome into the previous one (the advisor's finding)."""
    from wiggletools_amd.runlists import RunLists
    if tiny:
        monkeypatch.setenv("WTAMD_MIN_SPAN", "16")
        monkeypatch.setenv("WTAMD_BATCH_INTERVALS", "24")
    rng = np.random.default_rng(77)
    tracks = []
    for i in range(18):             # (>= 16 foreign children: the parallel drain's workers compare names too)
        per_c, base = [], 1
        for c in range(4):
            rows, pos = [], base + int(rng.integers(0, 40))
            for _ in range(int(rng.integers(1, 6))):
                ln = int(rng.integers(1, 30))
                rows.append((pos, pos + ln, float(rng.integers(1, 9))))
                pos += ln + int(rng.integers(0, 25))
            per_c.append(rows)
            base = pos + 5          # the next chromosome starts beyond everything seen so far
        tracks.append(per_c)
    t = RunLists.from_lists(tracks, np.zeros(18))
    d = t.as_dict()
    H.set_modes(3, 0)
    try:
        for op in ("mean", "max"):
            for strict in (0, 1):
                assert_runs_equal(H.reduce(d, op, flags=strict), oracle.reduce(d, op, flags=strict), _tol(op), "name buffer %s" % op)
    finally:
        H.set_modes(0, 0)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("threads", [None, "1", "3"])
def test_dropin_buffered_reader_non_float_values(oracle, H, threads, monkeypatch):
    """Buffered-reader children (child mode 4) whose values are NOT float32-exact (0.1 * k as double): wt_buf_peek offers
    nothing for such an element and the track must fall back to one pop at a time -- the batch turns float64.  Round 4's
    SERIAL drain (the default below 16 tracks) skipped to the next track instead: the child never advanced and the Feeder
    span for ever (the advisor's finding, reproduced with 5 tracks; the parallel drain was right).  One track turns
    non-float only half-way through, so the switch happens inside a chromosome and inside a block."""
    from wiggletools_amd.runlists import synth, RunLists
    monkeypatch.setenv("WTAMD_MIN_SPAN", "4000")
    monkeypatch.setenv("WTAMD_BATCH_INTERVALS", "5000")
    if threads:
        monkeypatch.setenv("WTAMD_DRAIN_THREADS", threads)
    t = synth(5, [60000, 3000], mean_run=5, seed=1404, gap_prob=0.1, dtype=np.float64)
    v = t.value.copy()
    N = t.n_tracks
    for c in range(t.n_chrom):
        for i in range(N):
            lo, hi = int(t.seg_off[c * N + i]), int(t.seg_off[c * N + i + 1])
            if i in (0, 3):
                v[lo:hi] = v[lo:hi] * 8 * 0.1                   # 0.1 * k
            elif i == 2 and c == 0:
                v[(lo + hi) // 2:hi] = v[(lo + hi) // 2:hi] * 8 * 0.1
    t = RunLists(t.n_chrom, N, t.seg_off, t.start, t.finish, v, t.defaults)
    d = t.as_dict()
    H.set_modes(4, 0)
    try:
        for op in ("mean", "max"):
            assert_runs_equal(H.reduce(d, op), oracle.reduce(d, op), _tol(op), "non-float buffered children %s" % op)
    finally:
        H.set_modes(0, 0)


@pytest.mark.parametrize("tiny,threads", [(False, None), (True, None), (False, "3"), (True, "2")])
def test_dropin_buffered_reader_children(oracle, H, tiny, threads, monkeypatch):
    """Children built on src/bufferedReader.h the way the reference's binary-file readers are (a reader thread pushing
    into 10 000-entry blocks, pop = BufferedReaderPop; oracle/ref_harness.c child mode 4) -- over the COMPILED REFERENCE's
    bufferedReader.o + Multiplexer and over this library's drop-in for both (csrc/wt_bufreader.h), where the Multiplexer
    takes the buffer's blocks whole: same runs, and the oracle's.  Tracks long enough for several blocks and chromosome
    changes inside a block; seek (kill + free + relaunch on the region) with held readers, as the CLI issues it."""
    from wiggletools_amd.runlists import synth
    if tiny:
        monkeypatch.setenv("WTAMD_MIN_SPAN", "4000")
        monkeypatch.setenv("WTAMD_BATCH_INTERVALS", "5000")
    if threads:         # the children dealt to worker threads (the default from 16 children on): the blocks go over inside the workers
        monkeypatch.setenv("WTAMD_DRAIN_THREADS", threads)
    t = synth(5, [220000, 9000, 130000], mean_run=7, seed=404, gap_prob=0.2)      # ~ 25 000 intervals per track and chromosome
    d = t.as_dict()
    ref = oracle.ref_harness() if oracle.have_ref() else None
    for L in [x for x in (H, ref) if x is not None]:
        L.set_modes(4, 0)
    import ctypes as C
    lib = C.CDLL(H.lib_path)
    lib.wtamd_bufreader_bulk_entries.restype = C.c_longlong
    before = lib.wtamd_bufreader_bulk_entries()
    try:
        for op in ("mean", "max", "median"):
            exp = oracle.reduce(d, op)
            got = H.reduce(d, op)
            assert_runs_equal(got, exp, _tol(op), "buffered children %s" % op)
            if ref is not None:
                assert_runs_equal(ref.reduce(d, op), exp, _tol(op), "reference over its own bufferedReader, %s" % op)
        # the blocks went over whole: (almost) every entry of the three passes left through the bulk door
        assert lib.wtamd_bufreader_bulk_entries() - before > 0.9 * 3 * len(t.start)
        for (c, s, f) in ((0, 5000, 150000), (2, 1, 70000), (1, 100, 200)):
            got = H.reduce_seek_held(d, "mean", c, s, f)
            exp = oracle.reduce(clip(t, c, s, f).as_dict(), "mean")
            assert_runs_equal(got, exp, 0.0, "buffered children, held seek %s" % ((c, s, f),))
            if ref is not None:
                assert_runs_equal(ref.reduce_seek_held(d, "mean", c, s, f), exp, 0.0, "reference, held seek %s" % ((c, s, f),))
        # the library's own array reader on the same protocol (wtamd_BufferedArrayReader; the bench's `e2e.buffered` leg)
        H.set_modes(5, 0)
        before = lib.wtamd_bufreader_bulk_entries()
        assert_runs_equal(H.reduce(d, "mean"), oracle.reduce(d, "mean"), 0.0, "wtamd_BufferedArrayReader children")
        assert lib.wtamd_bufreader_bulk_entries() - before > 0.9 * len(t.start)
        for (c, s, f) in ((0, 5000, 150000), (1, 100, 200)):
            assert_runs_equal(H.reduce_seek(d, "mean", c, s, f), oracle.reduce(clip(t, c, s, f).as_dict(), "mean"), 0.0,
                              "wtamd_BufferedArrayReader, seek %s" % ((c, s, f),))
    finally:
        for L in [x for x in (H, ref) if x is not None]:
            L.set_modes(0, 0)
