"""wtamd_MapIterator + wtamd_pipe_set_map: the reference's `map`-able operator iterators (unaryOps.c:650-949,
:386-419) in front of the drop-in Multiplexer, run as per-track operator chains ON DEVICE inside the
streaming pipeline -- `sum map ln a b c`, `mean scale 2 a  ln b  c` (commandParser.c:115-211).
Expected: the oracle's operator restatement (pinned bit for bit on the compiled reference's operator
iterators, tests/test_oracle_vs_ref.py) applied per track, then the oracle's reducer.

CPU: host layer over the emulated pipeline (bit-exact: same libm); `-m gpu`: the product (transcendental
operators to 1e-12 relative -- device libm; the others exact)."""
import ctypes as C

import numpy as np
import pytest

from helpers import random_case
from test_bwreader import WI, _blocks, _pops
from wiggletools_amd.runlists import RunLists

MAP_OPS = {"scale": 0, "offset": 1, "ln": 2, "log": 3, "exp": 4, "expb": 5, "pow": 6, "abs": 7, "gt": 8, "gte": 9, "lt": 10,
           "lte": 11}
REDUCERS = {"sum": "SumReduction", "mean": "MeanReduction", "var": "VarianceReduction", "median": "MedianReduction",
            "max": "MaxReduction", "stddev": "StdDevReduction"}


def _bind(L):
    L.wtamd_ArrayReader.restype = C.c_void_p
    L.wtamd_ArrayReader.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
    L.wtamd_MapIterator.restype = C.c_void_p
    L.wtamd_MapIterator.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.newMultiplexer.restype = C.c_void_p
    L.newMultiplexer.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char]
    for name in REDUCERS.values():
        getattr(L, name).restype = C.c_void_p
        getattr(L, name).argtypes = [C.c_void_p]
    L.wtamd_iterator_next_block.restype = C.c_int64
    L.wtamd_iterator_next_block.argtypes = [C.c_void_p, C.POINTER(C.c_char_p)] + [C.POINTER(C.c_void_p)] * 3
    L.pop.argtypes = [C.c_void_p]
    L.pop.restype = None
    return L


@pytest.fixture(scope="module")
def emu_lib():
    from emu.build import build_dropin
    return _bind(C.CDLL(build_dropin()))


@pytest.fixture(scope="module")
def amd_lib():
    from wiggletools_amd import _lib
    return _bind(_lib.lib())


_KEEP = []


def _track_arrays(t, i):
    """Track i of RunLists t as the per-chromosome-contiguous arrays wtamd_ArrayReader takes."""
    seg, S, F, V = [0], [], [], []
    for c in range(t.n_chrom):
        a, b = t.seg_off[c * t.n_tracks + i], t.seg_off[c * t.n_tracks + i + 1]
        S.append(t.start[a:b]); F.append(t.finish[a:b]); V.append(t.value[a:b])
        seg.append(seg[-1] + int(b - a))
    cat = lambda xs, dt: np.ascontiguousarray(np.concatenate(xs) if xs else [], dt)
    return np.array(seg, np.int64), cat(S, np.int32), cat(F, np.int32), cat(V, np.float32)


def _readers(L, t, chains):
    names = (C.c_char_p * t.n_chrom)(*[n.encode() for n in t.chrom_names])
    its = []
    for i in range(t.n_tracks):
        seg, s, f, v = _track_arrays(t, i)
        _KEEP.append((names, seg, s, f, v))
        wi = L.wtamd_ArrayReader(t.n_chrom, names, seg.ctypes.data, s.ctypes.data, f.ctypes.data, v.ctypes.data, float(t.defaults[i]))
        for op, param in chains[i]:
            wi = L.wtamd_MapIterator(wi, MAP_OPS[op], float(param))
        its.append(wi)
    return its


def _expected_tracks(oracle, t, chains):
    """Oracle: the chains applied per track -> RunLists with f64 values and mapped defaults."""
    seg_off, S, F, V = [0], [], [], []
    for c in range(t.n_chrom):
        for i in range(t.n_tracks):
            a, b = t.seg_off[c * t.n_tracks + i], t.seg_off[c * t.n_tracks + i + 1]
            s, f, v = t.start[a:b], t.finish[a:b], t.value[a:b].astype(np.float64)
            for op, param in chains[i]:
                v, keep = oracle.map_values(op, param, v)
                k = keep != 0
                s, f, v = s[k], f[k], v[k]
            S.append(s); F.append(f); V.append(v)
            seg_off.append(seg_off[-1] + len(s))
    d = []
    for i in range(t.n_tracks):
        x = float(t.defaults[i])
        for op, param in chains[i]:
            x = oracle.map_default(op, param, x)
        d.append(x)
    return RunLists(t.n_chrom, t.n_tracks, seg_off, np.concatenate(S), np.concatenate(F), np.concatenate(V), d,
                    chrom_names=t.chrom_names)


def _close(got, exp, rtol):
    assert len(got) == len(exp), (len(got), len(exp))
    for g, e in zip(got, exp):
        assert g[:3] == e[:3], (g, e)
        if np.isnan(e[3]) or np.isnan(g[3]):
            assert np.isnan(e[3]) and np.isnan(g[3]), (g, e)
        elif not np.isfinite(e[3]):
            assert g[3] == e[3]
        else:
            assert abs(g[3] - e[3]) <= rtol * max(abs(e[3]), 1e-300), (g, e)


CHAINS = [
    lambda n: [[("ln", 0)]] * n,                                            # sum map ln ...
    lambda n: [[("scale", -2.5)]] * n,
    lambda n: [[("gt", 12.5)]] * n,                                         # drops runs, values become 1
    lambda n: [[("abs", 0), ("log", 2.0)]] * n,                             # log 2 abs x
    lambda n: [[("offset", 3.25), ("pow", 2.0), ("lte", 100.0)]] * n,
    lambda n: [[("exp", 0)] if i % 3 == 0 else [] if i % 3 == 1 else [("scale", 0.5), ("ln", 0)] for i in range(n)],   # mixed
    lambda n: [[("expb", 2.0), ("offset", -1.0), ("abs", 0), ("gte", 0.25)] for i in range(n)],
]


def _case(seed):
    t = random_case(9300 + seed, n_tracks=int(3 + seed % 5), max_len=6000, dtype=np.float32)
    rng = np.random.default_rng(seed)
    t.value[:] = (t.value * rng.choice([1.0, -1.0, 0.0], size=len(t.value), p=[0.6, 0.3, 0.1])).astype(np.float32)
    t.value[:] = np.where(np.abs(t.value) > 60, t.value / 16, t.value)      # keep exp() finite and varied
    return t


def _run(L, oracle, rtol, seeds=range(6)):
    for seed in seeds:
        t = _case(seed)
        for ci, mk in enumerate(CHAINS):
            chains = mk(t.n_tracks)
            exp_t = _expected_tracks(oracle, t, chains)
            for op in ("sum", "var", "median") if (seed + ci) % 2 else ("mean", "max", "stddev"):
                for strict in (False, True):
                    its = (C.c_void_p * t.n_tracks)(*_readers(L, t, chains))
                    m = L.newMultiplexer(its, t.n_tracks, b"\x01" if strict else b"\x00")
                    got = _blocks(L, getattr(L, REDUCERS[op])(m))
                    c, s, f, v = oracle.reduce(exp_t.as_dict(), op, flags=oracle.STRICT_SET0 if strict else 0)
                    exp = [(t.chrom_names[a], int(b), int(d), float(x)) for a, b, d, x in zip(c, s, f, v)]
                    _close(got, exp, rtol)


def _run_pop_protocol(L, oracle, rtol):
    """The operator iterator popped by a foreign consumer: per-interval protocol, dropped runs skipped."""
    t = _case(3)
    for mk in CHAINS[:5]:
        chains = mk(t.n_tracks)
        exp_t = _expected_tracks(oracle, t, chains)
        for i, wi in enumerate(_readers(L, t, chains)):
            w = C.cast(wi, C.POINTER(WI)).contents
            x = float(t.defaults[i])
            for op, param in chains[i]:
                x = oracle.map_default(op, param, x)
            assert w.default_value == x or (np.isnan(x) and np.isnan(w.default_value))
            exp = []
            for c in range(t.n_chrom):
                a, b = exp_t.seg_off[c * t.n_tracks + i], exp_t.seg_off[c * t.n_tracks + i + 1]
                exp += [(t.chrom_names[c], int(s), int(f), float(v)) for s, f, v in
                        zip(exp_t.start[a:b], exp_t.finish[a:b], exp_t.value[a:b])]
            _close(_pops(L, wi), exp, rtol)


def test_map_chains_in_pipeline_emu(emu_lib, oracle):
    _run(emu_lib, oracle, 0.0)


def test_map_iterator_pop_protocol_emu(emu_lib, oracle):
    _run_pop_protocol(emu_lib, oracle, 0.0)


def test_map_chains_small_batches_emu(emu_lib, oracle, monkeypatch):
    monkeypatch.setenv("WTAMD_BATCH_INTERVALS", "300")
    monkeypatch.setenv("WTAMD_BATCH_RUNS", "200")
    _run(emu_lib, oracle, 0.0, seeds=range(2))


def test_map_chains_foreign_children_emu(emu_lib, oracle, monkeypatch):
    """Children popped one interval at a time (WTAMD_NO_BULK): raw values are staged, the chain still runs in the pipe."""
    monkeypatch.setenv("WTAMD_NO_BULK", "1")
    _run(emu_lib, oracle, 0.0, seeds=range(2))


@pytest.mark.gpu
def test_map_chains_in_pipeline_gpu(amd_lib, oracle):
    _run(amd_lib, oracle, 1e-12)


@pytest.mark.gpu
def test_map_chains_small_batches_gpu(amd_lib, oracle, monkeypatch):
    monkeypatch.setenv("WTAMD_BATCH_INTERVALS", "300")
    monkeypatch.setenv("WTAMD_BATCH_RUNS", "200")
    _run(amd_lib, oracle, 1e-12, seeds=range(2))


@pytest.mark.gpu
def test_map_iterator_pop_protocol_gpu(amd_lib, oracle):
    _run_pop_protocol(amd_lib, oracle, 1e-12)


def _nested(L, oracle, rtol, monkeypatch):
    """Reducers of this library as CHILDREN of another Multiplexer (`sum mean a b : mean c d : ...`), popped by
    the worker threads of the outer one: every inner reducer drives its own pipeline from the thread that
    pops it (HIP's current device is per thread: the workers inherit the caller's)."""
    monkeypatch.setenv("WTAMD_DRAIN_THREADS", "3")
    t = random_case(9700, n_tracks=9, n_chrom=2, max_len=5000, dtype=np.float32)
    t.defaults[:] = 0
    groups = [list(range(0, 3)), list(range(3, 6)), list(range(6, 9))]
    inner, inner_exp = [], []
    for g in groups:
        sub = t.subset(g)
        its = (C.c_void_p * len(g))(*_readers(L, sub, [[]] * len(g)))
        inner.append(L.MeanReduction(L.newMultiplexer(its, len(g), b"\x00")))
        inner_exp.append(oracle.reduce(sub.as_dict(), "mean"))
    outer = (C.c_void_p * 3)(*inner)
    got = _blocks(L, L.SumReduction(L.newMultiplexer(outer, 3, b"\x00")))
    # expected: the three mean run lists as tracks (default 0: reducers.c:416-420 of zero defaults), summed
    seg_off, S, F, V = [0], [], [], []
    for c in range(t.n_chrom):
        for ch, s, f, v in inner_exp:
            k = ch == c
            S.append(s[k]); F.append(f[k]); V.append(v[k])
            seg_off.append(seg_off[-1] + int(k.sum()))
    mid = RunLists(t.n_chrom, 3, seg_off, np.concatenate(S), np.concatenate(F), np.concatenate(V), [0.0] * 3,
                   chrom_names=t.chrom_names)
    c, s, f, v = oracle.reduce(mid.as_dict(), "sum")
    exp = [(t.chrom_names[a], int(b), int(d), float(x)) for a, b, d, x in zip(c, s, f, v)]
    _close(got, exp, rtol)


def test_nested_reducers_under_worker_threads_emu(emu_lib, oracle, monkeypatch):
    _nested(emu_lib, oracle, 1e-15, monkeypatch)


@pytest.mark.gpu
def test_nested_reducers_under_worker_threads_gpu(amd_lib, oracle, monkeypatch):
    _nested(amd_lib, oracle, 1e-15, monkeypatch)


def _two_sample(L, oracle, rtol):
    """ttest / wilcoxon over two sets of mapped tracks (`ttest map ln a b c : map ln d e f`): two Multiplexers
    under a Multiset, every child a wtamd_MapIterator handle."""
    L.newMultiset.restype = C.c_void_p
    L.newMultiset.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    for name in ("TTestReduction", "MWUReduction"):
        getattr(L, name).restype = C.c_void_p
        getattr(L, name).argtypes = [C.c_void_p]
    for seed in (23, 24, 28):               # (_case: 3 + seed % 5 tracks -> 6, 7, 6)
        t = _case(seed)
        n0 = t.n_tracks // 2
        for mk in (CHAINS[0], CHAINS[1], CHAINS[5]):
            chains = mk(t.n_tracks)
            exp_t = _expected_tracks(oracle, t, chains)
            for red, op in (("TTestReduction", "ttest"), ("MWUReduction", "mwu")):
                its = _readers(L, t, chains)
                a = (C.c_void_p * n0)(*its[:n0])
                b = (C.c_void_p * (t.n_tracks - n0))(*its[n0:])
                ms = (C.c_void_p * 2)(L.newMultiplexer(a, n0, b"\x00"), L.newMultiplexer(b, t.n_tracks - n0, b"\x00"))
                _KEEP.append(ms)
                got = _blocks(L, getattr(L, red)(L.newMultiset(ms, 2)))
                c, s, f, v = oracle.reduce(exp_t.as_dict(), op, n_set0=n0)
                exp = [(t.chrom_names[x], int(y), int(z), float(w)) for x, y, z, w in zip(c, s, f, v)]
                _close(got, exp, rtol)


def test_two_sample_over_mapped_sets_emu(emu_lib, oracle):
    _two_sample(emu_lib, oracle, 1e-12)


@pytest.mark.gpu
def test_two_sample_over_mapped_sets_gpu(amd_lib, oracle):
    _two_sample(amd_lib, oracle, 1e-9)
