"""Pins oracle/wt_oracle.c against the COMPILED REFERENCE (oracle/_ref), bit for bit.

Skipped when oracle/_ref is unavailable (it is built from /root/reference in the
build container and travels to the GPU box as a prebuilt .so).
"""
import numpy as np
import pytest

from helpers import ALL_MULTIPLEX_OPS, assert_runs_equal, random_case


@pytest.fixture(scope="module")
def O(oracle):
    if not oracle.have_ref():
        pytest.skip("compiled reference not available")
    return oracle


@pytest.mark.parametrize("seed", range(60))
def test_reducers_match_reference(O, seed):
    t = random_case(seed)
    d = t.as_dict()
    for strict in (0, 1):
        for op in ALL_MULTIPLEX_OPS:
            a = O.reduce(d, op, flags=strict)
            b = O.ref_reduce(d, op, flags=strict)
            assert_runs_equal(a, b, 0.0, "seed %d op %s strict %d" % (seed, op, strict))


@pytest.mark.parametrize("seed", range(20))
def test_multiplexer_tile_matches_reference(O, seed):
    t = random_case(1000 + seed)
    d = t.as_dict()
    for strict in (0, 1):
        a = O.multiplex(d, flags=strict)
        b = O.ref_multiplex(d, flags=strict)
        assert len(a[0]) == len(b[0])
        for x, y in zip(a, b):
            assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("seed", range(30))
def test_multiset_alignment_matches_reference(O, seed):
    """Two-sample run structure (multiSet.c) == closed form used for ttest/MWU."""
    t = random_case(2000 + seed, n_tracks=int(np.random.default_rng(seed).integers(2, 9)))
    d = t.as_dict()
    n1 = max(1, t.n_tracks // 2)
    for flags in (0, 1, 2, 3):
        rc, rs, rf, rtile, rip = O.ref_multiset(d, n1, flags)
        # the oracle's two-sample sweep is exercised through a reducer: use MWU run coordinates
        c, s, f, v = O.reduce(d, "mwu", flags=flags, n_set0=n1)
        assert np.array_equal(c, rc) and np.array_equal(s, rs) and np.array_equal(f, rf), \
            "seed %d flags %d" % (seed, flags)


def test_reducer_ctor_defaults_match_reference(O):
    rng = np.random.default_rng(7)
    cases = [np.zeros(3), np.array([1.0, 2.0, 3.5]), np.array([0.1, 0.2, 0.7, 1e-3]),
             np.array([1.0, np.nan, 2.0]), np.array([5.0]), rng.random(17) * 100 - 50,
             np.array([0.0, 0.0, 1.0, 2.0])]
    for d in cases:
        for op in ALL_MULTIPLEX_OPS:
            a = O.reducer_default(op, d)
            b = O.ref_reducer_default(op, d)
            assert (np.isnan(a) and np.isnan(b)) or a == b, (op, d, a, b)


@pytest.mark.parametrize("seed", range(10))
def test_auc_and_compression_match_reference(O, seed):
    t = random_case(3000 + seed)
    d = t.as_dict()
    for op in ("mean", "max", "stddev"):
        c, s, f, v = O.reduce(d, op)
        assert O.auc(s, f, v) == O.ref_auc_of_reduce(d, op)
        a = O.compress(c, s, f, v)
        b = O.ref_reduce(d, op, compressed=True)
        assert_runs_equal(a, b, 0.0, "compress %s" % op)


@pytest.mark.parametrize("seed", range(10))
def test_oracle_pearson_matches_compiled_reference(oracle, seed):
    """statistics.c:414-465 restated (oracle/wt_oracle.c:wto_pearson) vs the compiled reference."""
    rng = np.random.default_rng(seed)
    t = random_case(7000 + seed, n_tracks=2, dtype=np.float64 if seed % 2 else np.float32)
    if np.isnan(t.value).any() or np.isnan(t.defaults).any():
        t.value[np.isnan(t.value)] = 1.5
        t.defaults[np.isnan(t.defaults)] = 0.0
    d = t.as_dict()
    exp = oracle.ref_pearson(d)
    got = oracle.pearson(d)
    assert (np.isnan(exp) and np.isnan(got)) or got == exp, (got, exp)


MAP_CASES = [("scale", -2.5), ("offset", 3.25), ("ln", 0.0), ("log", 2.0), ("log", 10.0), ("exp", 0.0), ("expb", 2.0),
             ("pow", 2.0), ("pow", -1.0), ("pow", 0.5), ("abs", 0.0),
             ("gt", 0.0), ("gt", 12.5), ("gte", 12.5), ("lt", 12.5), ("lte", 12.5), ("lte", -3.0)]


@pytest.mark.parametrize("op,param", MAP_CASES)
def test_oracle_map_ops_match_compiled_reference(oracle, op, param):
    """unaryOps.c:650-949 restated value by value (oracle/wt_oracle.c:wto_map / wto_map_default) vs the
    compiled reference's operator iterators: values bit for bit (same libm), dropped runs, defaults."""
    for seed in range(6):
        t = random_case(8000 + seed, dtype=np.float64 if seed % 2 else np.float32)
        rng = np.random.default_rng(seed)
        t.value[:] = (t.value * rng.choice([1.0, -1.0, 0.0], size=len(t.value), p=[0.6, 0.3, 0.1])).astype(t.value.dtype)
        d = t.as_dict()
        for i in range(t.n_tracks):
            rc, rs, rf, rv, rd = oracle.ref_map(d, i, op, param)
            # the same track through the restatement
            idx = np.concatenate([np.arange(t.seg_off[c * t.n_tracks + i], t.seg_off[c * t.n_tracks + i + 1])
                                  for c in range(t.n_chrom)]).astype(np.int64)
            out, keep = oracle.map_values(op, param, t.value[idx])
            k = keep != 0
            assert len(rs) == int(k.sum()), (op, param, seed, i)
            assert np.array_equal(rs, t.start[idx][k]) and np.array_equal(rf, t.finish[idx][k])
            assert np.array_equal(rv, out[k], equal_nan=True), (op, param, seed, i)
            od = oracle.map_default(op, param, t.defaults[i])
            assert (np.isnan(od) and np.isnan(rd)) or od == rd, (op, param, t.defaults[i], od, rd)
