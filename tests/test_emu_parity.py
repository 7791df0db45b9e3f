"""Kernel LOGIC vs oracle on CPU: the HIP kernel source (wt_core.h) compiled for the
host and executed phase by phase (tests/emu).  The same cases run on the real GPU
in tests/test_gpu_parity.py."""
import numpy as np
import pytest

from helpers import ALL_MULTIPLEX_OPS, assert_runs_equal, random_case
from emu import emu
from wiggletools_amd.runlists import synth

# values: sum/mean/min/max/median are bit-exact by construction (same op order);
# var/stddev/cv go through sqrt/div in the same order as well -> bit-exact on CPU.
GEOMS = [(None, None), (4, 64), (1, 64), (4, 128), (1, 128)]   # (positions per lane, workgroup size)


@pytest.mark.parametrize("seed", range(40))
def test_emu_one_sample_ops(oracle, seed):
    t = random_case(seed, dtype=np.float64 if seed % 2 else np.float32)
    d = t.as_dict()
    ppt, T = GEOMS[seed % len(GEOMS)]
    for strict in (0, 1):
        for op in ALL_MULTIPLEX_OPS:
            exp = oracle.reduce(d, op, flags=strict)
            got, info = emu.reduce(t, op, flags=strict, ppt=ppt, T=T)
            assert_runs_equal(got, exp, 0.0, "seed %d op %s strict %d %s" % (seed, op, strict, info))
            assert info["covered_bp"] == int((exp[2] - exp[1]).sum())


@pytest.mark.parametrize("seed", range(30))
def test_emu_two_sample_ops(oracle, seed):
    rng = np.random.default_rng(seed)
    t = random_case(500 + seed, n_tracks=int(rng.integers(6, 12)), dtype=np.float64 if seed % 2 else np.float32)
    d = t.as_dict()
    n1 = int(rng.integers(3, t.n_tracks - 2))
    ppt, T = GEOMS[seed % len(GEOMS)]
    for flags in (0, 1, 2, 3):
        for op in ("ttest", "mwu"):
            exp = oracle.reduce(d, op, flags=flags, n_set0=n1)
            got, info = emu.reduce(t, op, flags=flags, n_set0=n1, ppt=ppt, T=T)
            assert_runs_equal(got, exp, 1e-12 if op == "ttest" else 0.0,
                              "seed %d op %s flags %d %s" % (seed, op, flags, info))


@pytest.mark.parametrize("seed", range(10))
def test_emu_multiplex_tile(oracle, seed):
    t = random_case(900 + seed)
    d = t.as_dict()
    ppt, T = GEOMS[seed % len(GEOMS)]
    for strict in (0, 1):
        exp = oracle.multiplex(d, flags=strict)
        got, info = emu.reduce(t, "sum", flags=strict, ppt=ppt, T=T, multiplex=True)
        assert len(got[0]) == len(exp[0])
        for a, b in zip(got, exp):
            assert np.array_equal(a, b, equal_nan=True)


def test_emu_long_intervals_span_many_windows(oracle):
    """One interval much longer than the window; breakpoints exactly on window edges."""
    from wiggletools_amd.runlists import RunLists
    tracks = [
        [[(1, 1000, 2.0)]],
        [[(65, 129, 1.0), (129, 500, 3.0), (900, 1200, 4.0)]],
        [[(64, 65, 5.0), (128, 129, 6.0), (193, 257, 7.0)]],
    ]
    t = RunLists.from_lists(tracks)
    for ppt, T in ((1, 64), (4, 64), (1, 128)):
        for op in ("sum", "mean", "max", "median"):
            exp = oracle.reduce(t.as_dict(), op)
            got, info = emu.reduce(t, op, ppt=ppt, T=T)
            assert_runs_equal(got, exp, 0.0, "ppt %d T %d op %s" % (ppt, T, op))


def test_emu_empty_and_single(oracle):
    from wiggletools_amd.runlists import RunLists
    t = RunLists.from_lists([[[], []], [[], []]])
    got, info = emu.reduce(t, "mean")
    assert len(got[0]) == 0
    t = RunLists.from_lists([[[(5, 6, 1.5)], []]])
    got, info = emu.reduce(t, "mean")
    assert got[1].tolist() == [5] and got[2].tolist() == [6] and got[3].tolist() == [1.5]


@pytest.mark.parametrize("seed", range(12))
def test_emu_run_start_ranges_concatenate(oracle, seed):
    """Cutting every chromosome into run-start ranges (batches / shards) and concatenating
    the pieces reproduces the unsharded output: a run spanning a cut belongs to the piece
    holding its start and keeps its true finish."""
    rng = np.random.default_rng(seed)
    t = random_case(4000 + seed, max_len=9000)
    d = t.as_dict()
    INT_MAX = 2 ** 31 - 1
    for op in ("mean", "median", "max"):
        exp = oracle.reduce(d, op)
        # arbitrary cut points per chromosome
        origin = [int(rng.integers(1, 3000)) for _ in range(t.n_chrom)]
        step = int(rng.integers(500, 3000))
        pieces = []
        for part in range(4):
            ranges = []
            for c in range(t.n_chrom):
                lo = origin[c] + step * (part - 1) if part > 0 else -INT_MAX
                hi = origin[c] + step * part if part < 3 else INT_MAX
                ranges.append((max(lo, -INT_MAX), hi))
            got, info = emu.reduce(t, op, ranges=ranges, ppt=[None, 4, 1][seed % 3], T=[None, 64, 128][seed % 3])
            pieces.append(got)
        # interleave per chromosome
        cat = [[], [], [], []]
        for c in range(t.n_chrom):
            for g in pieces:
                m = g[0] == c
                for k in range(4):
                    cat[k].append(g[k][m])
        cat = tuple(np.concatenate(x) for x in cat)
        assert_runs_equal(cat, exp, 0.0, "seed %d op %s" % (seed, op))


@pytest.mark.parametrize("seed", range(16))
def test_emu_chunked_tracks(oracle, seed):
    """Tracks visited in chunks (bitmaps rebuilt per chunk and pass): same results as all-resident."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 14))
    t = random_case(1300 + seed, n_tracks=n, dtype=np.float64 if seed % 2 else np.float32)
    d = t.as_dict()
    chunk = int(rng.integers(1, n))
    ppt, T = GEOMS[seed % len(GEOMS)]
    n1 = int(rng.integers(2, n - 1))
    for flags in (0, 1):
        for op in ALL_MULTIPLEX_OPS:
            exp = oracle.reduce(d, op, flags=flags)
            got, info = emu.reduce(t, op, flags=flags, ppt=ppt, T=T, chunk=chunk)
            assert info["n_chunks"] == -(-n // chunk)
            assert_runs_equal(got, exp, 0.0, "seed %d op %s flags %d %s" % (seed, op, flags, info))
        exp = oracle.multiplex(d, flags=flags)
        got, info = emu.reduce(t, "sum", flags=flags, ppt=ppt, T=T, multiplex=True, chunk=chunk)
        for a, b in zip(got, exp):
            assert np.array_equal(a, b, equal_nan=True)
    for flags in (0, 1, 2, 3):
        for op in ("ttest", "mwu"):
            exp = oracle.reduce(d, op, flags=flags, n_set0=n1)
            got, info = emu.reduce(t, op, flags=flags, n_set0=n1, ppt=ppt, T=T, chunk=chunk)
            assert_runs_equal(got, exp, 1e-12 if op == "ttest" else 0.0,
                              "seed %d op %s flags %d %s" % (seed, op, flags, info))


@pytest.mark.parametrize("seed", range(8))
def test_emu_global_scratch_columns(oracle, seed):
    """median / MWU with the per-lane columns in a global slab (very many tracks), chunked bitmaps."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 14))
    t = random_case(1700 + seed, n_tracks=n, dtype=np.float64 if seed % 2 else np.float32)
    d = t.as_dict()
    chunk = int(rng.integers(1, n + 1)) if seed % 3 else None
    n1 = int(rng.integers(2, n - 1))
    for flags in (0, 1):
        exp = oracle.reduce(d, "median", flags=flags)
        got, info = emu.reduce(t, "median", flags=flags, chunk=chunk, global_scratch=1)
        assert info["scratch_slab"] > 0
        assert_runs_equal(got, exp, 0.0, "seed %d median flags %d %s" % (seed, flags, info))
    for flags in (0, 1, 2, 3):
        exp = oracle.reduce(d, "mwu", flags=flags, n_set0=n1)
        got, info = emu.reduce(t, "mwu", flags=flags, n_set0=n1, chunk=chunk, global_scratch=1)
        assert info["scratch_slab"] > 0
        assert_runs_equal(got, exp, 0.0, "seed %d mwu flags %d %s" % (seed, flags, info))


def test_emu_many_tracks_plans(oracle):
    """Track counts beyond one workgroup's LDS: chunked bitmaps / global columns chosen automatically."""
    t = random_case(77, n_tracks=700)
    d = t.as_dict()
    for op, kw in (("mean", {}), ("var", {}), ("median", {}), ("mwu", dict(n_set0=300)), ("ttest", dict(n_set0=300))):
        exp = oracle.reduce(d, op, **kw)
        got, info = emu.reduce(t, op, **kw)
        if op in ("median", "mwu"):
            assert info["scratch_slab"] > 0
        else:
            assert info["n_chunks"] > 1
        assert_runs_equal(got, exp, 1e-12 if op == "ttest" else 0.0, "%s %s" % (op, info))


# ---- exact difference-array path (wt_delta.h) ----
def _delta_case(seed, n_tracks, clens, mean_run, value_fn, gap=0.1, first_start=1):
    from wiggletools_amd.runlists import synth
    t = synth(n_tracks, clens, mean_run=mean_run, gap_prob=gap, seed=seed, dtype=np.float32, first_start=first_start)
    rng = np.random.default_rng(seed + 1)
    t.value[:] = value_fn(rng, len(t.value)).astype(np.float32)
    return t


@pytest.mark.parametrize("seed", range(4))
def test_emu_delta_8192bp_windows(oracle, seed):
    """1024-lane workgroups (8192-bp windows, 128 bitmap words): the layout the difference-array kernels
    can be launched with since round 2 (WTAMD_DELTA_T=1024)."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(2, 20))
    t = _delta_case(200 + seed, n, [int(rng.integers(9000, 40000)), 700], float(rng.choice([1, 3, 16])),
                    lambda r, k: r.integers(-800, 800, k) / 8.0, gap=float(rng.choice([0.0, 0.1])), first_start=int(rng.choice([1, 8193])))
    d = t.as_dict()
    for strict in (0, 1):
        for op in ("sum", "mean"):
            got, info = emu.reduce(t, op, flags=strict, delta_T=1024)
            assert info["delta"] == 1 and info["W"] == 8192, info
            assert_runs_equal(got, oracle.reduce(d, op, flags=strict), 0.0, "seed %d %s strict %d" % (seed, op, strict))


def test_emu_delta_every_run_crosses_a_window_edge(oracle):
    """Runs longer than a window, 300 tracks: every run of every window is one wt_delta_apply_or_park PARKS, and a lane
    meets several of them per pass (flat indices l, l + 64, ... of a tile): the flush inside the loop, not only the one
    at the end.  Both workgroup sizes, sum / mean / the squares, zero and non-zero defaults."""
    from wiggletools_amd.runlists import RunLists, synth
    t = synth(300, [70000, 9000], mean_run=11000, gap_prob=0.05, seed=77)
    d = t.as_dict()
    for T in (1024, 512):
        for op, tol in (("sum", 0.0), ("mean", 0.0), ("var", 1e-12), ("stddev", 1e-12)):
            if T == 1024 and op in ("var", "stddev"):
                continue
            got, info = emu.reduce(t, op, flags=1, delta_T=T)
            assert info["delta"] == 1, info
            assert_runs_equal(got, oracle.reduce(d, op, flags=1), tol, "long runs T %d op %s" % (T, op))
    t2 = RunLists(t.n_chrom, t.n_tracks, t.seg_off, t.start, t.finish, t.value, np.where(np.arange(t.n_tracks) % 3 == 0, 1.5, 0.0))
    got, info = emu.reduce(t2, "mean", delta_T=1024)
    assert info["delta"] == 1, info
    assert_runs_equal(got, oracle.reduce(t2.as_dict(), "mean"), 0.0, "long runs, defaults")


@pytest.mark.parametrize("seed", range(16))
def test_emu_delta_sum_mean_exact(oracle, seed):
    """Sum / Mean of float tracks with zero defaults through the difference-array path: bit-identical."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 12))
    clens = [int(rng.integers(1, 3000)) for _ in range(int(rng.integers(1, 4)))]
    kinds = [
        lambda r, k: r.integers(-800, 800, k) / 8.0,                      # signed, ties, zeros
        lambda r, k: r.random(k) * 1000.0,                                # full 24-bit mantissas
        lambda r, k: np.ldexp(r.integers(1, 1 << 24, k).astype(np.float64), r.integers(-30, -10, k)),  # span < 29 - lg n
        lambda r, k: np.where(r.random(k) < 0.3, 0.0, -r.random(k)),      # many zeros, negatives
    ]
    t = _delta_case(seed, n, clens, float(rng.choice([1, 3, 16, 200])), kinds[seed % 4],
                    gap=float(rng.choice([0.0, 0.1, 0.6])), first_start=int(rng.choice([1, 5, 4097])))
    d = t.as_dict()
    for T in (64, 128):
        for strict in (0, 1):
            for op in ("sum", "mean"):
                exp = oracle.reduce(d, op, flags=strict)
                got, info = emu.reduce(t, op, flags=strict, delta_T=T)
                assert info["delta"] == 1 and info["delta_bad"] == 0, info
                assert info["W"] == 8 * T
                assert_runs_equal(got, exp, 0.0, "seed %d op %s strict %d %s" % (seed, op, strict, info))
                assert info["covered_bp"] == int((exp[2] - exp[1]).sum())


def test_emu_delta_more_tracks_than_lanes(oracle):
    """Track ranges are scanned in chunks of T tracks: 150 tracks on a 64-lane workgroup."""
    t = _delta_case(3, 150, [1200, 90], 5, lambda r, k: r.integers(0, 2000, k) / 16.0, gap=0.2)
    for strict in (0, 1):
        for op in ("sum", "mean"):
            exp = oracle.reduce(t.as_dict(), op, flags=strict)
            got, info = emu.reduce(t, op, flags=strict, delta_T=64)
            assert info["delta"] == 1 and info["T"] == 64
            assert_runs_equal(got, exp, 0.0, "%s strict %d %s" % (op, strict, info))


def test_emu_delta_dense_window_beyond_tile_table(oracle):
    """A window holding more intervals than the tile table covers (binary-search fallback)."""
    t = _delta_case(8, 150, [5000], 1, lambda r, k: r.integers(0, 64, k) / 4.0, gap=0.05)
    exp = oracle.reduce(t.as_dict(), "mean")
    got, info = emu.reduce(t, "mean", delta_T=512)
    assert info["delta"] == 1 and info["T"] == 512 and info["n_intervals"] > 2048 * 256
    assert_runs_equal(got, exp, 0.0, "dense %s" % info)


@pytest.mark.parametrize("direction", ["falling", "rising"])
def test_emu_delta_speculative_unit_is_redone(oracle, direction):
    """The single-pass flavour scales with the smallest exponent the workgroup has met so far; a
    window holding a smaller one (falling magnitudes), or too wide a span relative to that guess
    although narrow in itself (rising magnitudes), is redone with its own unit -- same bits."""
    t = _delta_case(21, 7, [12000], 6, lambda r, k: r.random(k) + 1.0, gap=0.1)
    pos = t.start.astype(np.int64)
    step = 3 if direction == "rising" else -3
    t.value[:] = np.ldexp(t.value.astype(np.float64), step * (pos // 512)).astype(np.float32)   # 2^(+-3) per window
    for op in ("sum", "mean"):
        exp = oracle.reduce(t.as_dict(), op)
        got, info = emu.reduce(t, op, delta_T=64)
        assert info["delta"] == 1 and info["delta_bad"] == 0 and info["delta_redo"] > 0, info
        assert_runs_equal(got, exp, 0.0, "%s %s %s" % (direction, op, info))


def test_emu_delta_inexact_windows_are_patched(oracle):
    """Windows whose values span too many binades, or hold NaN / Inf, keep their coordinates from the
    difference-array kernel and get their values from the general kernel (few such windows), or
    the general kernel redoes everything (many)."""
    for kind in ("wide", "nan", "inf", "denormal", "many"):
        t = _delta_case(5, 6, [5000, 300], 8, lambda r, k: r.random(k) + 0.5)
        if kind == "wide":
            t.value[3] = np.float32(1e-30)
        elif kind == "nan":
            t.value[7] = np.nan
            t.value[len(t.value) // 2] = np.nan
        elif kind == "inf":
            t.value[7] = np.inf
        elif kind == "many":
            t.value[::40] = np.nan                                             # NaN in every window
        else:
            t.value[:] = (t.value * np.float32(1e-42)).astype(np.float32)     # all denormal: still exact
        for op in ("sum", "mean"):
            for strict in (0, 1):
                exp = oracle.reduce(t.as_dict(), op, flags=strict)
                # general plan of 256-bp windows under 512-bp difference-array windows: two per patch
                got, info = emu.reduce(t, op, flags=strict, delta_T=64, ppt=4, T=64)
                if kind == "denormal":
                    assert info["delta"] == 1 and info["delta_bad"] == 0
                elif kind == "many":
                    assert info["delta"] == 0 and info["delta_bad"] > 0 and info["patched"] == 0, info
                else:
                    assert info["delta"] == 1 and info["delta_bad"] > 0 and info["patched"] == info["delta_bad"], info
                assert_runs_equal(got, exp, 0.0, "%s %s %s" % (kind, op, info))
                assert info["covered_bp"] == int((exp[2] - exp[1]).sum())


def test_emu_delta_not_used_when_ineligible(oracle):
    t = _delta_case(9, 5, [700], 8, lambda r, k: r.random(k))
    t.defaults[2] = 0.1                                    # a non-zero default value that is not a float
    got, info = emu.reduce(t, "sum")
    assert info["delta"] == 0 and info["delta_bad"] == 0
    assert_runs_equal(got, oracle.reduce(t.as_dict(), "sum"), 0.0, "defaults")
    t.defaults[2] = np.nan
    got, info = emu.reduce(t, "sum")
    assert info["delta"] == 0
    assert_runs_equal(got, oracle.reduce(t.as_dict(), "sum"), 0.0, "NaN default")
    # the var family's squares count the tracks in play only: zero defaults or the general kernel
    t9 = _delta_case(10, 9, [700], 8, lambda r, k: r.random(k))
    t9.defaults[2] = 1.5
    got, info = emu.reduce(t9, "stddev")
    assert info["delta"] == 0
    assert_runs_equal(got, oracle.reduce(t9.as_dict(), "stddev"), 1e-12, "stddev with a default")
    from wiggletools_amd.runlists import synth
    t64 = synth(4, [700], mean_run=8, seed=3, dtype=np.float64)
    got, info = emu.reduce(t64, "mean")
    assert info["delta"] == 0
    assert_runs_equal(got, oracle.reduce(t64.as_dict(), "mean"), 0.0, "f64 tracks")


@pytest.mark.parametrize("seed", range(24))
def test_emu_delta_nonzero_defaults_exact(oracle, seed):
    """Round 3: non-zero defaults that are floats stay on the difference-array path (Sum / Mean): an absent
    track is a term of the sum like any other (reducers.c:294-307, 375-401), so the window's base holds the sum
    of the defaults and an interval adds value - default.  Tolerance 0, like the zero-default path; covers
    negative and denormal defaults, more tracks than lanes, strict and not, and a default so far from the data
    that the windows must be patched."""
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(4, 30)) if seed % 4 else int(rng.integers(65, 150))       # > 64 lanes: chunks of tracks
    t = _delta_case(300 + seed, n, [int(rng.integers(600, 5000)), 300], float(rng.choice([1, 3, 16])),
                    lambda r, k: r.integers(-800, 800, k) / 8.0, gap=float(rng.choice([0.0, 0.1, 0.5])))
    which = rng.random(n) < 0.6
    vals = rng.choice(np.array([1.5, -2.25, 0.125, 3.0, 1024.0, -0.0, 7.0], dtype=np.float32), n)
    t.defaults[:] = np.where(which, vals, 0.0)
    if seed % 6 == 1:                                      # everything denormal (exponent 1 without the hidden bit): still exact
        t.value[:] = (t.value * np.float32(1e-42)).astype(np.float32)
        t.defaults[:] = (t.defaults.astype(np.float32) * np.float32(1e-42)).astype(np.float64)
    if not np.any(t.defaults != 0):
        t.defaults[0] = 2.5
    far = seed % 6 == 5
    if far:
        t.defaults[1] = float(np.float32(1e-30))           # outside any window's exact range with these values
    d = t.as_dict()
    for strict in (0, 1):
        for op in ("sum", "mean"):
            got, info = emu.reduce(t, op, flags=strict, delta_T=64, ppt=4, T=64)
            if far:
                assert info["delta_bad"] > 0, info         # patched windows or everything redone: still exact
            else:
                assert info["delta"] == 1 and info["delta_bad"] == 0, info
            assert_runs_equal(got, oracle.reduce(d, op, flags=strict), 0.0, "seed %d %s strict %d %s" % (seed, op, strict, info))


@pytest.mark.parametrize("seed", range(6))
def test_emu_delta_ranges_concatenate(oracle, seed):
    """Run-start ranges (batches / shards) through the difference-array path tile the full result."""
    rng = np.random.default_rng(seed)
    t = _delta_case(40 + seed, 5, [1500, 900], 6, lambda r, k: r.integers(0, 64, k) / 4.0, first_start=int(rng.choice([1, 300])))
    full = oracle.reduce(t.as_dict(), "mean")
    cuts = sorted(int(x) for x in rng.integers(1, 1900, 3))
    edges = [-(2 ** 31 - 1)] + cuts + [2 ** 31 - 1]
    parts = []
    for a, b in zip(edges[:-1], edges[1:]):
        got, info = emu.reduce(t, "mean", delta_T=64, ranges=[(a, b)] * t.n_chrom)
        assert info["delta"] == 1
        parts.append(got)
    cat = [[], [], [], []]
    for c in range(t.n_chrom):
        for g in parts:
            m = g[0] == c
            for k in range(4):
                cat[k].append(g[k][m])
    cat = tuple(np.concatenate(x) for x in cat)
    assert_runs_equal(cat, full, 0.0, "ranges")


def test_emu_delta_patch_with_run_start_ranges(oracle):
    """Batches / shards (run-start ranges) whose windows need patching: the pieces still tile the full result."""
    t = _delta_case(61, 6, [30000, 12000], 6, lambda r, k: r.integers(1, 64, k) / 4.0, first_start=300)
    t.value[50] = np.nan
    t.value[len(t.value) // 2] = np.float32(1e-33)
    full = oracle.reduce(t.as_dict(), "sum")
    edges = [-(2 ** 31 - 1), 7500, 16100, 2 ** 31 - 1]
    parts, patched = [], 0
    for a, b in zip(edges[:-1], edges[1:]):
        got, info = emu.reduce(t, "sum", delta_T=64, ppt=4, T=64, ranges=[(a, b)] * t.n_chrom)
        assert info["delta"] == 1, info
        patched += info["patched"]
        parts.append(got)
    assert patched >= 2
    cat = [[], [], [], []]
    for c in range(t.n_chrom):
        for g in parts:
            m = g[0] == c
            for k in range(4):
                cat[k].append(g[k][m])
    cat = tuple(np.concatenate(x) for x in cat)
    assert_runs_equal(cat, full, 0.0, "ranges + patches")


# ---- assorted edge cases (both kernels where they apply) ----
def _f32(t):
    from wiggletools_amd.runlists import RunLists
    return RunLists(t.n_chrom, t.n_tracks, t.seg_off, t.start, t.finish, t.value.astype(np.float32), t.defaults)


def test_emu_edge_all_nan_track_and_all_nan_data(oracle):
    t = _delta_case(71, 4, [3000], 6, lambda r, k: r.random(k) + 1)
    t.value[t.seg_off[1]:t.seg_off[2]] = np.nan          # one track entirely NaN
    for op in ("sum", "mean", "max", "median"):
        got, info = emu.reduce(t, op, delta_T=64, ppt=4, T=64)
        assert_runs_equal(got, oracle.reduce(t.as_dict(), op), 0.0, "%s %s" % (op, info))
    t.value[:] = np.nan
    got, info = emu.reduce(t, "sum", delta_T=64, ppt=4, T=64)
    assert info["delta"] == 0 and info["delta_bad"] > 0
    assert_runs_equal(got, oracle.reduce(t.as_dict(), "sum"), 0.0, "all NaN")


def test_emu_edge_strict_with_an_absent_track(oracle):
    """strict: a track without runs on a chromosome means no run is ever emitted there."""
    from wiggletools_amd.runlists import RunLists
    t = _f32(RunLists.from_lists([[[(1, 50, 1.0), (60, 90, 2.0)], [(5, 20, 1.0)]],
                                  [[(10, 70, 3.0)], []],
                                  [[(1, 100, 0.5)], [(1, 30, 4.0)]]]))
    for op in ("sum", "mean", "min"):
        for strict in (0, 1):
            got, info = emu.reduce(t, op, flags=strict, delta_T=64)
            exp = oracle.reduce(t.as_dict(), op, flags=strict)
            assert_runs_equal(got, exp, 0.0, "%s strict %d" % (op, strict))
            if strict:
                assert not (exp[0] == 1).any()


def test_emu_edge_ranges_that_exclude_everything(oracle):
    t = _delta_case(72, 3, [2000, 500], 6, lambda r, k: r.integers(1, 9, k) / 2.0)
    for op in ("sum", "max"):
        got, info = emu.reduce(t, op, delta_T=64, ranges=[(5000, 6000), (-10, 0)])
        assert len(got[0]) == 0
        got, info = emu.reduce(t, op, delta_T=64, ranges=[(5000, 6000), (1, 100)])      # only chromosome 1, partly
        exp = oracle.reduce(t.as_dict(), op)
        m = (exp[0] == 1) & (exp[1] < 100)
        assert_runs_equal(got, tuple(x[m] for x in exp), 0.0, op)


def test_emu_edge_two_sample_extremes(oracle):
    t = random_case(9100, n_tracks=9, dtype=np.float32)
    d = t.as_dict()
    for n1 in (1, 8):
        for flags in (0, 3):
            got, info = emu.reduce(t, "mwu", flags=flags, n_set0=n1)
            assert_runs_equal(got, oracle.reduce(d, "mwu", flags=flags, n_set0=n1), 0.0, "mwu n1 %d" % n1)
    for n1 in (3, 6):
        got, info = emu.reduce(t, "ttest", n_set0=n1)
        assert_runs_equal(got, oracle.reduce(d, "ttest", n_set0=n1), 1e-12, "ttest n1 %d" % n1)


def test_emu_edge_very_many_tracks_small_workgroup(oracle):
    """2000 tracks: 32 chunks of 64 tracks in the difference-array kernel, chunked bitmaps in the general one."""
    t = _delta_case(73, 2000, [700], 30, lambda r, k: r.integers(0, 16, k) / 2.0, gap=0.3)
    d = t.as_dict()
    got, info = emu.reduce(t, "sum", delta_T=64)
    assert info["delta"] == 1
    assert_runs_equal(got, oracle.reduce(d, "sum"), 0.0, "delta 2000 tracks")
    got, info = emu.reduce(t, "product", ppt=4, T=64)
    assert info["n_chunks"] > 1
    assert_runs_equal(got, oracle.reduce(d, "product"), 0.0, "general 2000 tracks")
    got, info = emu.reduce(t, "max", delta_T=64)            # (round 6: max / min ride the difference-array kernel's passes)
    assert info["delta"] == 1
    assert_runs_equal(got, oracle.reduce(d, "max"), 0.0, "segment tree, 2000 tracks")


def test_emu_edge_data_only_in_late_tracks_and_late_chromosomes(oracle):
    from wiggletools_amd.runlists import RunLists
    t = _f32(RunLists.from_lists([[[], [], []],
                                  [[], [], [(7, 9, 1.0)]],
                                  [[], [(1000, 1001, 2.0), (1001, 5000, 3.0)], [(8, 12, 4.0)]]]))
    for op in ("sum", "mean", "max", "median", "product"):
        got, info = emu.reduce(t, op, delta_T=64, ppt=4, T=64)
        assert_runs_equal(got, oracle.reduce(t.as_dict(), op), 0.0, op)


def test_emu_delta_everybody_parks(oracle):
    """wt_delta.h with -DWT_DELTA_PARK=2 (not the default: only the launches with squares park runs that cross a window edge;
    for Sum / Mean it measured a wash with two more spilled registers): Sum / Mean with zero and non-zero defaults through
    wt_delta_apply_or_park and its two flushes, against the oracle at tolerance 0."""
    from wiggletools_amd.runlists import RunLists, synth
    emu.use_variant("park2", ["-DWT_DELTA_PARK=2"])
    try:
        t = synth(300, [70000, 9000], mean_run=11000, gap_prob=0.05, seed=78)      # every run crosses an edge: flushes inside the loop
        t2 = RunLists(t.n_chrom, t.n_tracks, t.seg_off, t.start, t.finish, t.value, np.where(np.arange(t.n_tracks) % 3 == 0, 1.5, 0.0))
        u = synth(60, [50000], mean_run=9, gap_prob=0.1, seed=79)                   # ordinary signal: one flush per wavefront and window
        for case, name in ((t, "long"), (t2, "long, defaults"), (u, "dense")):
            for op in ("sum", "mean"):
                for flags in (0, 1):
                    got, info = emu.reduce(case, op, flags=flags, delta_T=1024)
                    assert info["delta"] == 1, info
                    assert_runs_equal(got, oracle.reduce(case.as_dict(), op, flags=flags), 0.0, "park2 %s %s strict %d" % (name, op, flags))
    finally:
        emu.use_variant(None)


def test_emu_delta_small_tiles(oracle):
    """wt_delta.h with 2 runs per lane and tile (-DWT_DELTA_U=2: what the device takes for Sum / Mean launches whose windows hold few
    tiles per wavefront, wt_launch_delta; the emulator's default build has 4): the tile -> track table, the fetch's walk across track
    boundaries and the partial last tile at the other tile size -- Sum / Mean / Max / Var with zero and non-zero defaults, many tracks with few
    runs each (a tile spans several tracks) and few tracks with many, against the oracle at tolerance 0 (1e-12 for Var)."""
    from wiggletools_amd.runlists import RunLists, synth
    emu.use_variant("u2", ["-DWT_DELTA_U=2"])
    try:
        a = synth(300, [30000, 900], mean_run=400, gap_prob=0.05, seed=91)         # ~75 runs per track and window: tiles span tracks
        b = synth(12, [60000], mean_run=3, gap_prob=0.02, seed=92)                   # many tiles per track
        b2 = RunLists(b.n_chrom, b.n_tracks, b.seg_off, b.start, b.finish, b.value, np.where(np.arange(b.n_tracks) % 2 == 0, 0.75, 0.0))
        for case, name in ((a, "sparse"), (b, "dense"), (b2, "dense, defaults")):
            for op in ("sum", "mean", "max", "var"):
                if name.endswith("defaults") and op in ("max", "var"):
                    continue
                for T in (1024, 64):
                    got, info = emu.reduce(case, op, delta_T=T)
                    assert info["delta"] == 1, info
                    assert_runs_equal(got, oracle.reduce(case.as_dict(), op), 1e-12 if op == "var" else 0.0, "u2 %s %s T %d" % (name, op, T))
    finally:
        emu.use_variant(None)


def test_emu_delta_squares_workgroup_sizes(oracle, monkeypatch):
    """The launches with squares: the passes over the runs are shared by ALL the workgroup's wavefronts, the scans are run by
    the first W / 8 = 512 lanes -- 768 lanes by default (round 5), any multiple of 64 from 512 to 768 through
    WTAMD_DELTA_SQ_T, the old 512-lane layout through WTAMD_DELTA_T; same results, same 4096-bp windows."""
    from wiggletools_amd.runlists import synth
    t = synth(40, [30000, 5000], mean_run=6, gap_prob=0.05, seed=31)
    d = t.as_dict()
    exp = {op: oracle.reduce(d, op) for op in ("var", "stddev", "cv")}
    for env, want_T in ((None, 768), ("512", 512), ("640", 640), ("768", 768), ("1024", 768), ("700", 768)):
        if env is None:
            monkeypatch.delenv("WTAMD_DELTA_SQ_T", raising=False)
        else:
            monkeypatch.setenv("WTAMD_DELTA_SQ_T", env)
        for op in ("var", "stddev", "cv"):
            got, info = emu.reduce(t, op)
            assert (info["delta"], info["W"], info["T"]) == (1, 4096, want_T), (env, info)
            assert_runs_equal(got, exp[op], 1e-12, "squares, WTAMD_DELTA_SQ_T=%s, op %s" % (env, op))
    monkeypatch.delenv("WTAMD_DELTA_SQ_T", raising=False)
    got, info = emu.reduce(t, "var", delta_T=512)
    assert (info["W"], info["T"]) == (4096, 512), info
    assert_runs_equal(got, exp["var"], 1e-12, "squares, 512 lanes")
    got, info = emu.reduce(t, "var", delta_T=256)
    assert (info["W"], info["T"]) == (2048, 256), info
    assert_runs_equal(got, exp["var"], 1e-12, "squares, 256 lanes")


def test_plan_policy_snapshot():
    """The execution plans the measurements in DESIGN.md were taken with (MI355X, round 1)."""
    from wiggletools_amd.runlists import synth
    def plan(n, op, dtype=np.float32, **kw):
        t = synth(n, [300], mean_run=8, seed=1, dtype=dtype)
        return emu.reduce(t, op, **kw)[1]
    p = plan(100, "mean")
    assert (p["delta"], p["W"], p["T"]) == (1, 8192, 1024)               # difference-array kernel (round 3: 1024 lanes, 8192-bp windows)
    p = plan(100, "max")
    assert (p["delta"], p["W"], p["T"]) == (1, 8192, 1024)               # round 6: max / min by range updates of a segment tree in the difference-array kernel
    p = plan(100, "product")
    assert (p["W"], p["T"], p["n_chunks"]) == (2048, 512, 1)             # general kernel: bitmaps of 100 tracks: half the LDS
    p = plan(500, "var")
    assert (p["delta"], p["W"], p["T"]) == (1, 4096, 768)                # difference arrays with exact squares: 4096-bp windows; round 5: 768 lanes for the passes (168 registers each), the first 512 run the scans
    assert p["lds"] <= 160 * 1024
    p = plan(500, "var", no_delta=1)
    assert (p["W"], p["T"]) == (2048, 512) and p["n_chunks"] == 5        # general kernel: chunks of <= 112 tracks
    p = plan(100, "median")
    assert (p["walk"], p["W"], p["T"]) == (1, 2048, 256)                                              # round 4: walking, two lanes per stretch: 128 stretches x 16 positions, two workgroups per CU
    p = plan(100, "median", walk_pair=0)
    assert (p["walk"], p["W"], p["T"]) == (1, 8192, 256)                                              # ... one lane per stretch: 256 lanes x 32 positions, one workgroup per CU
    p = plan(100, "median", no_walk=1)
    assert (p["W"], p["T"], p["lds"] < 32 * 1024) == (512, 256, True) and p["scratch_slab"] == 0     # round 2: value column in REGISTERS, LDS = bitmaps only, 2 positions per lane
    p = plan(100, "mwu", n_set0=50)
    assert (p["walk"], p["W"], p["T"]) == (0, 512, 256) and p["lds"] < 80 * 1024       # register columns + the sorted set 0 (50 x 4 B per lane) in LDS, 2 positions per lane
    p = plan(100, "mwu", n_set0=50, mwalk=1)
    assert (p["walk"], p["W"], p["T"]) == (1, 2048, 256) and p["lds"] < 80 * 1024    # round 5, opt-in (measured slower): walking (wt_mwalk.h), the two lanes of a stretch hold one set each
    p = plan(100, "mwu", n_set0=90)
    assert (p["W"], p["T"]) == (256, 512)                                # a set above 64 tracks: LDS columns, two lanes per run (round 1 plan)
    p = plan(200, "median")
    assert (p["W"], p["T"]) == (128, 128) or p["scratch_slab"] >= 0      # more than 128 tracks: LDS / global columns
    p = plan(100, "median", dtype=np.float64)
    assert p["lds"] > 100 * 1024                                         # f64 values: one f64 column per lane in LDS
    p = plan(20, "mwu", n_set0=10)
    assert (p["W"], p["T"]) == (512, 256)
    p = plan(100, "mean", dtype=np.float64)
    assert p["delta"] == 0 and p["W"] == 2048                            # f64 tracks: general kernel
    p = plan(1000, "median")
    assert p["scratch_slab"] > 0                                         # columns in a global slab


@pytest.mark.parametrize("seed", range(12))
def test_emu_delta_var_family(oracle, seed):
    """Var / StdDev / Entropy / CV over float tracks with zero defaults: difference arrays with exact
    integer sum and sum of squares (wt_delta.h, WT_DELTA_QSHIFT) against the oracle's two sequential
    f64 passes (reducers.c:428-479, 511-563, 672-725), 1e-12; coordinates bit-exact."""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([8, 9, 16, 33, 100, 200]))
    t = synth(n, [int(rng.integers(200, 5000)), 300], mean_run=float(rng.choice([1, 3, 16, 60])), seed=seed,
              gap_prob=float(rng.choice([0, 0.05, 0.5])), dtype=np.float32, value_levels=int(rng.choice([2, 800])))
    if seed % 3 == 0:       # several binades of dynamic range, values that are not multiples of 1/8
        t.value[:] = (t.value * rng.choice([1e-3, 1.0, 37.5], len(t.value))).astype(np.float32)
    d = t.as_dict()
    for op in ("var", "stddev", "cv", "entropy"):
        for strict in (0, 1):
            got, info = emu.reduce(t, op, flags=strict)
            assert info["delta"] == 1, (op, info)
            assert_runs_equal(got, oracle.reduce(d, op, flags=strict), 1e-12, "%s strict %d" % (op, strict))


def test_emu_delta_var_family_patched_windows(oracle):
    """NaN, Inf and a dynamic range beyond the exactness bound in a few windows: those windows are
    recomputed by the general kernel's two passes (wt_patch_kernel), the rest stays on the exact path."""
    t = synth(12, [200000], mean_run=9, seed=5, dtype=np.float32)      # (49 windows of 4096 bp: a few bad ones are patched, many would redo the launch)
    v = t.value
    v[100] = np.nan
    v[len(v) // 2] = np.inf
    v[len(v) // 3] = 2.0 ** -120
    v[len(v) // 3 + 1] = 2.0 ** 100
    d = t.as_dict()
    for op in ("var", "stddev", "cv"):
        got, info = emu.reduce(t, op)
        assert info["patched"] > 0 and info["delta"] == 1, info
        assert_runs_equal(got, oracle.reduce(d, op), 1e-12, op)
    # fewer than 8 tracks, non-zero defaults, f64 values: general kernel
    t8 = synth(7, [3000], mean_run=5, seed=1, dtype=np.float32)
    assert emu.reduce(t8, "var")[1]["delta"] == 0
    t9 = synth(9, [3000], mean_run=5, seed=1, dtype=np.float32)
    t9.defaults[2] = 1.0
    assert emu.reduce(t9, "stddev")[1]["delta"] == 0


def test_emu_delta_more_tiles_than_the_first_track_table(oracle):
    """A window whose flat interval space has more than WT_DELTA_TF (2048) tiles of 256: the tile's first track comes from
    a binary search over tpfx[] instead of the tfirst[] table (csrc/wt_delta.h wt_delta_fetch).  Needs a wide window
    and dense tracks: 4096 bp x 140 tracks at one interval per position = 573 000 intervals = 2240 tiles."""
    t = _delta_case(77, 140, [9000], 1, lambda r, k: r.integers(-64, 64, k) / 4.0, gap=0.0)
    d = t.as_dict()
    for op in ("sum", "mean"):
        got, info = emu.reduce(t, op, delta_T=512)
        assert info["delta"] == 1 and info["W"] == 4096 and info["delta_bad"] == 0, info
        assert_runs_equal(got, oracle.reduce(d, op), 0.0, op)


@pytest.mark.parametrize("seed", range(24))
def test_emu_median_walk_fuzz(oracle, seed):
    """MedianReduction by walking (csrc/wt_walk.h): a lane carries its column of current values over consecutive positions,
    events in, the order statistic moved by sweeps.  Against the oracle (reducers.c:780-813) at tolerance 0 and against the
    bitmap kernel (WTAMD_NO_WALK), over lane counts, stretch lengths, slabs too small for a window (several rounds), gaps,
    ties (2 value levels), NaN, non-zero defaults, the strict predicate and ranges."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 3, 7, 8, 31, 32, 33, 64, 65, 100, 128]))
    lens = [int(rng.integers(40, 6000)), int(rng.integers(1, 400))]
    defaults = None
    if rng.random() < 0.4:
        defaults = rng.integers(-3, 4, n).astype(np.float64) / 4.0
    t = synth(n, lens, mean_run=float(rng.choice([1, 2, 5, 16, 70])), seed=seed, gap_prob=float(rng.choice([0, 0.05, 0.5, 0.9])),
              dtype=np.float32, value_levels=int(rng.choice([2, 5, 800])), nan_prob=float(rng.choice([0, 0, 0.002])), defaults=defaults,
              first_start=int(rng.choice([1, 1, 777])))
    T = int(rng.choice([64, 128, 256]))
    if n > T:
        T = 128
    S = int(rng.choice([4, 8, 16, 32]))
    # event slots per position / entries of the overflow list: the defaults (16, 2048) hold ordinary windows; small ones
    # send events through the overflow list, none at all sends every window with a full position to the fallback (its
    # events sorted into the slab by a second pass, in rounds when they do not fit at once)
    capp, ov = [(None, None), (None, None), (2, None), (1, 5), (2, 0), (1, 0)][int(rng.integers(0, 6))]
    pair = int(seed % 2)            # one lane per stretch / two lanes (half the column each, values exchanged inside the pair)
    if pair and capp == 1:
        capp = 2                    # (a slot per lane of the pair at least)
    flags = int(rng.choice([0, 0, 1]))
    ranges = None
    if rng.random() < 0.3:
        ranges = [(int(rng.integers(1, L // 2 + 2)), int(rng.integers(L // 2 + 1, L + 60))) for L in lens]
    got, info = emu.reduce(t, "median", flags=flags, walk_T=T, walk_S=S, walk_capp=capp, walk_ov=ov, walk_pair=pair, ranges=ranges)
    assert info["walk"] == 1
    assert (info["T"], info["W"]) == (T, (T >> pair) * S) or n * T * 4 + T * S * 4 > 100 * 1024 or (T >> pair) * S < 64      # (the columns of 256 lanes do not fit: fewer positions, fewer lanes)
    exp = oracle.reduce(t.as_dict(), "median", flags=flags) if ranges is None else None
    old, info2 = emu.reduce(t, "median", flags=flags, no_walk=1, ranges=ranges)
    assert info2["walk"] == 0
    assert_runs_equal(got, old, 0.0, "walking vs bitmap kernel")
    if exp is not None:
        assert_runs_equal(got, exp, 0.0, "walking vs oracle")
    if ov == 0 and n >= 8 and t.n_intervals > 50 * n:
        assert info["walk_fallback"] > 0


@pytest.mark.parametrize("seed", range(36))
def test_emu_mwu_walk_fuzz(oracle, seed):
    """MWUReduction by walking (csrc/wt_mwalk.h): a pair of lanes carries the two sets' columns over consecutive positions and keeps
    S = #{y < x} and the tie groups (value, c0, c1, r0) up to date event by event; the reference's leaking tie state machine
    (setComparisons.c:328-359) runs over the groups.  Against the oracle's literal scan at tolerance 0 -- value included: the
    table of 2 erf(-k / 2 sigma) is the host's -- and against the bitmap kernel (the default: the walking kernel measured slower on
    MI355X and is selected by WTAMD_MWALK=1), over set sizes (1 v 1 ...
    64 v 64), value levels (2: everything ties, more groups than the lanes keep -> enumeration; 14-40: around the slots'
    capacity; 800: a few groups), NaN, non-zero defaults, both strict flags, stretch lengths, slots too few for a position
    (overflow list) and none at all (fallback: events sorted into the slab, in rounds), ranges."""
    rng = np.random.default_rng(7000 + seed)
    n1, n2 = [(1, 1), (1, 3), (2, 2), (3, 1), (5, 7), (8, 8), (17, 33), (50, 50), (64, 64), (3, 50), (40, 9), (30, 30)][seed % 12]
    n = n1 + n2
    lens = [int(rng.integers(40, 2500)), int(rng.integers(1, 300))]
    defaults = None
    if rng.random() < 0.4:
        defaults = rng.integers(-3, 4, n).astype(np.float64) / 4.0
    t = synth(n, lens, mean_run=float(rng.choice([1, 2, 5, 16, 70])), seed=seed, gap_prob=float(rng.choice([0, 0.05, 0.5, 0.9])),
              dtype=np.float32, value_levels=int(rng.choice([2, 5, 14, 25, 40, 800])), nan_prob=float(rng.choice([0, 0, 0.002])), defaults=defaults,
              first_start=int(rng.choice([1, 1, 777])))
    S = int(rng.choice([4, 8, 16, 32]))
    capp, ov = [(None, None), (None, None), (2, None), (2, 5), (2, 0), (4, 0)][int(rng.integers(0, 6))]
    flags = int(rng.choice([0, 0, 1, 2, 3]))
    ranges = None
    if rng.random() < 0.3:
        ranges = [(int(rng.integers(1, L // 2 + 2)), int(rng.integers(L // 2 + 1, L + 60))) for L in lens]
    got, info = emu.reduce(t, "mwu", flags=flags, n_set0=n1, walk_S=S, walk_capp=capp, walk_ov=ov, ranges=ranges, mwalk=1)
    assert info["walk"] == 1
    old, info2 = emu.reduce(t, "mwu", flags=flags, n_set0=n1, ranges=ranges)
    assert info2["walk"] == 0
    assert_runs_equal(got, old, 0.0, "walking vs bitmap kernel")
    if ranges is None:
        assert_runs_equal(got, oracle.reduce(t.as_dict(), "mwu", flags=flags, n_set0=n1), 0.0, "walking vs oracle")
    if ov == 0 and n >= 8 and t.n_intervals > 50 * n:
        assert info["walk_fallback"] > 0


def test_emu_mwu_walk_tie_structures(oracle):
    """Hand-made columns for the tie state machine: a single set-0 entry tied with set 1 (the state LEAKS into the following
    groups), several entries in one group (reset at the last one), leaked state running over tie-free entries to the end of
    set 0, previousTies overshooting ties (never reset) -- each as a track set whose values change one track at a time, so
    that every configuration is reached by EVENTS from the previous one, not by the stretch's initialisation."""
    from wiggletools_amd.runlists import RunLists
    rng = np.random.default_rng(11)
    n1, n2, L = 6, 7, 900
    tracks = []
    for i in range(n1 + n2):
        rows, pos = [], 1
        while pos < L:
            ln = int(rng.integers(1, 9))
            rows.append((pos, min(pos + ln, L), float(rng.integers(0, 4))))      # four values: ties everywhere
            pos += ln
        tracks.append([rows])
    t = _f32(RunLists.from_lists(tracks))
    for S in (4, 16, 32):
        got, info = emu.reduce(t, "mwu", n_set0=n1, walk_S=S, mwalk=1)
        assert info["walk"] == 1
        assert_runs_equal(got, oracle.reduce(t.as_dict(), "mwu", n_set0=n1), 0.0, "tie structures, S = %d" % S)


@pytest.mark.parametrize("pair", [0, 1])
def test_emu_median_walk_dense_and_sparse(oracle, pair):
    """The extremes: every track a run per base pair (the most events a window can hold), and tracks with one run each;
    one lane per stretch and two."""
    from wiggletools_amd.runlists import RunLists
    rng = np.random.default_rng(5)
    n, L = 9, 700
    per_bp = [[(p, p + 1, float(rng.integers(0, 50)) / 4.0) for p in range(1, L)] for _ in range(n)]
    t = _f32(RunLists.from_lists([[r] for r in per_bp]))         # (track, chromosome)
    got, info = emu.reduce(t, "median", walk_T=64, walk_S=4, walk_capp=2, walk_ov=0, walk_pair=pair)
    assert info["walk"] == 1 and info["walk_fallback"] == info["n_windows"] and info["walk_rounds"] > info["n_windows"]      # sorted, in rounds
    assert_runs_equal(got, oracle.reduce(t.as_dict(), "median"), 0.0, "one run per bp, fallback")
    # a few positions with more events than slots: those go through the overflow list, no fallback
    burst = [[(100, 100 + 7 * (i + 1), float(i)), (400 + i // 3, 500, float(-i))] for i in range(n)]
    t2 = _f32(RunLists.from_lists([[r] for r in burst]))
    got, info = emu.reduce(t2, "median", walk_T=64, walk_S=4, walk_capp=4, walk_pair=pair)
    assert info["walk"] == 1 and info["walk_fallback"] == 0
    assert_runs_equal(got, oracle.reduce(t2.as_dict(), "median"), 0.0, "nine starts at one position, overflow list")
    one = [[(int(rng.integers(1, 3000)), 0, float(i))] for i in range(n)]
    one = [[(s, s + int(rng.integers(1, 4000)), v)] for [(s, _, v)] in one]
    t = _f32(RunLists.from_lists([[r] for r in one]))
    for flags in (0, 1):
        got, info = emu.reduce(t, "median", flags=flags, walk_pair=pair)
        assert info["walk"] == 1
        assert_runs_equal(got, oracle.reduce(t.as_dict(), "median", flags=flags), 0.0, "one run per track")


# ---- TTestReduction by difference arrays (round 6: wt_delta.h, wt_delta_scan3_tt) ----
@pytest.mark.parametrize("seed", range(16))
def test_emu_delta_ttest(oracle, seed):
    """Welch's t-test over two sets of float tracks: per set the exact integer sum and sum of squares of the tracks IN PLAY
    (setComparisons.c:60-81), the reference's own arithmetic from there on (:88-117).  Values on a coarse grid (the
    reference's sums do not round): bit for bit; full mantissas: 1e-9 (the reference's sum of squares carries its own
    rounding); coordinates and the set of emitted runs (both sets in play, :48-54; the four strictness flags) exact."""
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([8, 9, 16, 33, 100, 200]))
    n1 = int(rng.integers(3, n - 2))
    levels = int(rng.choice([2, 800]))
    t = synth(n, [int(rng.integers(200, 7000)), 300], mean_run=float(rng.choice([1, 3, 16, 60])), seed=seed,
              gap_prob=float(rng.choice([0, 0.05, 0.5])), dtype=np.float32, value_levels=levels)
    coarse = seed % 3 != 0
    if not coarse:          # several binades of dynamic range, values that are not multiples of 1/8
        t.value[:] = (t.value * rng.choice([1e-3, 1.0, 37.5], len(t.value))).astype(np.float32)
    d = t.as_dict()
    for flags in (0, 1, 2, 3):
        got, info = emu.reduce(t, "ttest", flags=flags, n_set0=n1)
        # (delta 0 with bad windows: more than a quarter of these few windows had to be patched -- exponent range or cancelling
        #  variance in a set of three or four tracks -- and the general kernel redid the launch)
        assert (info["delta"] == 1 and info["W"] == 2048 and info["T"] == 768) or info["delta_bad"] > 0, info
        exp = oracle.reduce(d, "ttest", flags=flags, n_set0=n1)
        assert_runs_equal(got, exp, 0.0 if coarse else 1e-9,
                          "ttest seed %d flags %d n1 %d %s" % (seed, flags, n1, info))


def test_emu_delta_ttest_long_runs_and_edges(oracle):
    """Runs longer than a window (every run is parked, both sets' bases carry the sums across the edge), a breakpoint exactly
    on a window edge in one set only, a set that leaves play for a while."""
    from wiggletools_amd.runlists import RunLists
    t = synth(24, [30000, 5000], mean_run=5000, gap_prob=0.2, seed=91, dtype=np.float32, value_levels=40)
    d = t.as_dict()
    for n1 in (3, 12, 21):
        for flags in (0, 3):
            got, info = emu.reduce(t, "ttest", flags=flags, n_set0=n1)
            assert info["delta"] == 1, info
            assert_runs_equal(got, oracle.reduce(d, "ttest", flags=flags, n_set0=n1), 0.0, "long runs n1 %d flags %d" % (n1, flags))
    # set 1 (tracks 4..7) only covers [2049, 4097): no run before, breakpoints on both window edges
    tracks = []         # [track][chromosome] -> [(start, finish, value)]
    for i in range(8):
        if i < 4:
            s = list(range(1, 6001, 10 + i))
            tracks.append([[(a, b, (j * (i + 3)) % 17 / 4.0 + 3 * i) for j, (a, b) in enumerate(zip(s, s[1:] + [6001]))]])
        else:
            s = list(range(2049, 4097, 64 * (i - 3)))
            tracks.append([[(a, b, (j * (i + 1)) % 5 / 2.0 + 3 * i) for j, (a, b) in enumerate(zip(s, s[1:] + [4097]))]])
    t2 = RunLists.from_lists(tracks, dtype=np.float32)
    got, info = emu.reduce(t2, "ttest", n_set0=4)
    exp = oracle.reduce(t2.as_dict(), "ttest", n_set0=4)
    assert info["delta"] == 1 and len(exp[0]) > 100, info
    assert exp[1].min() == 2049 and exp[2].max() == 4097
    assert_runs_equal(got, exp, 0.0, "window edges")


def test_emu_delta_ttest_cancelling_variance_is_patched(oracle):
    """Both sets fully in play with nearly equal values: var = meanSq - mean^2 cancels, the reference's result is made of its
    own rounding errors -- such windows (wt_ttest_stat: var * 2^10 < meanSq at an emitted position) are recorded and the
    general kernel, which adds in the reference's order, rewrites them: bit-identical with the oracle there too.  NaN / Inf /
    too wide an exponent range: the same route."""
    t = synth(10, [60000], mean_run=40, gap_prob=0.0, seed=17, dtype=np.float32, value_levels=800)
    d0 = t.as_dict()
    got, info = emu.reduce(t, "ttest", n_set0=5)
    assert info["delta"] == 1 and info["delta_bad"] == 0, info
    # windows 3 and 11 (2048 bp each): every value 1000.1 +- a few ulp
    v = t.value
    rng = np.random.default_rng(3)
    for w in (3, 11):
        m = (t.start >= 1 + w * 2048 - 200) & (t.start < 1 + (w + 1) * 2048)
        v[m] = (np.float32(1000.1) + rng.integers(-2, 3, int(m.sum())).astype(np.float32) * np.float32(6.1e-5)).astype(np.float32)
    v[len(v) // 2] = np.nan
    d = t.as_dict()
    got, info = emu.reduce(t, "ttest", n_set0=5)
    assert info["delta"] == 1 and 3 <= info["delta_bad"] <= 7 and info["patched"] == info["delta_bad"], info
    exp = oracle.reduce(d, "ttest", n_set0=5)
    assert_runs_equal(got, exp, 1e-9, "patched")
    # ... and inside the patched windows the values are the general kernel's, i.e. the oracle's bits
    inside = ((exp[1] >= 1 + 3 * 2048) & (exp[1] < 1 + 4 * 2048)) | ((exp[1] >= 1 + 11 * 2048) & (exp[1] < 1 + 12 * 2048))
    assert inside.sum() > 50
    assert np.array_equal(got[3][inside], exp[3][inside], equal_nan=True)


def test_emu_delta_ttest_not_used_when_ineligible(oracle):
    t7 = synth(7, [3000], mean_run=5, seed=1, dtype=np.float32)
    assert emu.reduce(t7, "ttest", n_set0=3)[1]["W"] != 2048 or emu.reduce(t7, "ttest", n_set0=3)[1]["T"] != 768
    t64 = synth(12, [3000], mean_run=5, seed=1, dtype=np.float64)
    got, info = emu.reduce(t64, "ttest", n_set0=6)
    assert info["delta"] == 0 and info["delta_bad"] == 0
    # defaults play no part in the t-test (setComparisons.c:69-81): non-zero defaults stay on the difference arrays
    t9 = synth(40, [30000], mean_run=5, seed=1, dtype=np.float32, value_levels=800)
    t9.defaults[2] = 1.5
    t9.defaults[30] = -7.0
    got, info = emu.reduce(t9, "ttest", n_set0=14)
    assert info["delta"] == 1, info
    assert_runs_equal(got, oracle.reduce(t9.as_dict(), "ttest", n_set0=14), 0.0, "defaults ignored")


# ---- MaxReduction / MinReduction by range updates of a segment tree (round 6: wt_delta.h, wt_delta_apply_mm) ----
@pytest.mark.parametrize("seed", range(12))
def test_emu_delta_min_max(oracle, seed):
    """Float tracks with zero defaults: a run is a range update (atomic max / min of order-preserving keys on its canonical nodes of
    the window's segment tree), a position's value the max over its ancestors, `max(.., 0)` where a track is not in play
    (reducers.c:125-168, 192-235).  Bit for bit: negative values, all tracks in play or not, runs of one position and runs longer than
    a window, strict and not, more tracks than lanes."""
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([4, 9, 33, 100, 300, 1500]))
    t = synth(n, [int(rng.integers(200, 30000)), 300], mean_run=float(rng.choice([1, 3, 16, 60, 5000, 20000])), seed=seed,
              gap_prob=float(rng.choice([0, 0.05, 0.5])), dtype=np.float32, value_levels=int(rng.choice([2, 800])))
    t.value[:] = ((t.value - rng.choice([0, 3, 50])) * rng.choice([1.0, 1e-3, 7.7], len(t.value))).astype(np.float32)
    t.value[t.value == 0] = 0.0         # (no -0.0: that is the next test's subject)
    d = t.as_dict()
    for op in ("max", "min"):
        for flags in (0, 1):
            got, info = emu.reduce(t, op, flags=flags, delta_T=int(rng.choice([64, 256, 1024])))
            assert info["delta"] == 1 and info["delta_bad"] == 0, info
            assert_runs_equal(got, oracle.reduce(d, op, flags=flags), 0.0, "%s seed %d flags %d %s" % (op, seed, flags, info))


def test_emu_delta_min_max_patched_and_ineligible(oracle):
    """A NaN (the reference answers NaN) or a -0.0 (the one value whose outcome depends on the reference's track order) sends its window to
    the general kernel; Inf is an ordinary value; non-zero defaults and float64 values keep the whole reduction on the general kernel."""
    t = synth(12, [400000], mean_run=9, seed=5, dtype=np.float32)      # 49 windows of 8192 bp
    v = t.value
    v[100] = np.nan
    v[len(v) // 2] = np.inf
    v[len(v) // 3] = -np.inf
    v[len(v) // 4] = -0.0
    d = t.as_dict()
    for op in ("max", "min"):
        got, info = emu.reduce(t, op)
        assert info["delta"] == 1 and 2 <= info["delta_bad"] <= 8 and info["patched"] == info["delta_bad"], info
        exp = oracle.reduce(d, op)
        assert_runs_equal(got, exp, 0.0, op)
        assert np.isnan(exp[3]).any() and np.isinf(exp[3]).any()
        assert np.array_equal(np.signbit(got[3]), np.signbit(exp[3]))       # ... the sign of every zero included
    t9 = synth(9, [3000], mean_run=5, seed=1, dtype=np.float32)
    t9.defaults[2] = 1.0
    assert emu.reduce(t9, "max")[1]["delta"] == 0
    t64 = synth(9, [3000], mean_run=5, seed=1, dtype=np.float64)
    assert emu.reduce(t64, "min")[1]["delta"] == 0
