"""wtamd_BigWiggleReader: BigWig files -> Multiplexer -> reducer through the drop-in API, the path
`wiggletools mean a.bw b.bw ...` takes in the reference (commandParser.c -> SmartReader ->
bigWiggleReader.c -> newMultiplexer -> MeanReduction).  Expected values: the oracle over the run
lists the section decoder returns for the same files (tests/test_bigwig.py pins that decoder against
the reference's own fixedStep.bw / variableStep.bw fixtures and against the writer's input).

CPU: the host layer over the emulated pipeline (tests/emu); `-m gpu`: the product library."""
import ctypes as C
import os

import numpy as np
import pytest

from bw_writer import write_bigwig
from wiggletools_amd import bigwig

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class WI(C.Structure):      # reference src/wiggleIterator.h:21-35
    _fields_ = [("chrom", C.c_char_p), ("start", C.c_int), ("finish", C.c_int), ("value", C.c_double),
                ("valuePtr", C.c_void_p), ("done", C.c_bool), ("strand", C.c_int), ("data", C.c_void_p),
                ("pop", C.c_void_p), ("seek", C.c_void_p), ("overlaps", C.c_bool), ("default_value", C.c_double),
                ("append", C.c_void_p)]


def _bind(L):
    L.wtamd_BigWiggleReader.restype = C.c_void_p
    L.wtamd_BigWiggleReader.argtypes = [C.c_char_p, C.c_int]
    L.newMultiplexer.restype = C.c_void_p
    L.newMultiplexer.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char]
    for name in ("MeanReduction", "SumReduction", "MedianReduction", "MaxReduction", "VarianceReduction"):
        getattr(L, name).restype = C.c_void_p
        getattr(L, name).argtypes = [C.c_void_p]
    L.wtamd_iterator_next_block.restype = C.c_int64
    L.wtamd_iterator_next_block.argtypes = [C.c_void_p, C.POINTER(C.c_char_p)] + [C.POINTER(C.c_void_p)] * 3
    L.pop.argtypes = [C.c_void_p]
    L.pop.restype = None
    L.seek.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    L.seek.restype = None
    L.destroyWiggleIterator.argtypes = [C.c_void_p]
    L.destroyWiggleIterator.restype = None
    return L


@pytest.fixture(scope="module")
def emu_lib():
    from emu.build import build_dropin
    return _bind(C.CDLL(build_dropin()))


@pytest.fixture(scope="module")
def amd_lib():
    from wiggletools_amd import _lib
    return _bind(_lib.lib())


def _pops(L, wi, limit=10 ** 7):
    w = C.cast(wi, C.POINTER(WI)).contents
    out = []
    while not w.done and len(out) < limit:
        out.append((w.chrom.decode(), w.start, w.finish, w.value))
        L.pop(wi)
    return out


def _blocks(L, wi):
    chrom, s, f, v = C.c_char_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    out = []
    while True:
        n = L.wtamd_iterator_next_block(wi, C.byref(chrom), C.byref(s), C.byref(f), C.byref(v))
        assert n >= 0
        if n == 0:
            return out
        sa = np.ctypeslib.as_array(C.cast(s, C.POINTER(C.c_int32)), shape=(n,))
        fa = np.ctypeslib.as_array(C.cast(f, C.POINTER(C.c_int32)), shape=(n,))
        va = np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_double)), shape=(n,))
        out += [(chrom.value.decode(), int(a), int(b), float(x)) for a, b, x in zip(sa, fa, va)]


def _reduce(L, paths, op, box=1, strict=False):
    its = (C.c_void_p * len(paths))(*[L.wtamd_BigWiggleReader(p.encode(), box) for p in paths])
    m = L.newMultiplexer(its, len(paths), b"\x01" if strict else b"\x00")
    return getattr(L, op)(m), its


def _expected(oracle, paths, op, box=True, strict=False):
    t = bigwig.load_runlists(paths, box=box)
    c, s, f, v = oracle.reduce(t.as_dict(), op, flags=oracle.STRICT_SET0 if strict else 0)
    names = t.chrom_names
    return [(names[ci], int(a), int(b), float(x)) for ci, a, b, x in zip(c, s, f, v)]


def _same(got, exp):
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert g[:3] == e[:3]
        assert g[3] == e[3] or (np.isnan(g[3]) and np.isnan(e[3])), (g, e)


def _write_set(tmp_path, n_tracks, seed, chroms=None, density=0.8, block=97, pad=0):
    rng = np.random.default_rng(seed)
    chroms = chroms or {"chr1": 60000, "chr10": 25001, "chr2": 41000, "chrM": 900}
    paths = []
    for t in range(n_tracks):
        data = {}
        mine = dict(chroms)
        if t % 3 == 2:
            mine.pop("chr10")               # a file without one of the chromosomes
        for c, Ln in mine.items():
            pos, recs = int(rng.integers(0, 50)), []
            while pos < Ln - 60:
                ln = int(rng.integers(1, 40))
                if rng.random() < density:
                    recs.append((pos, pos + ln, float(np.float32(rng.integers(0, 64) / 4))))
                pos += ln
            data[c] = recs
        p = str(tmp_path / ("t%d.bw" % t))
        write_bigwig(p, mine, data, items_per_block=block, compress=(t % 2 == 0), mix_types=True, pad=pad)
        paths.append(p)
    return paths


def _run_all(L, oracle, tmp_path):
    # C1 of BASELINE.json: the reference's own fixtures (test/test.py:28,52)
    paths = [os.path.join(G, "fixedStep.bw"), os.path.join(G, "variableStep.bw")]
    wi, keep = _reduce(L, paths, "MeanReduction")
    got = _pops(L, wi)
    assert [g[1] for g in got] == list(range(1, 11))
    assert [g[3] for g in got] == [0.5, 1.5, 1, 3, 2, 4.5, 3, 6, 4, 4.5]
    # synthetic files: several chromosomes, a file lacking one, all section types
    paths = _write_set(tmp_path, 7, seed=11)
    for op, name in (("MeanReduction", "mean"), ("MedianReduction", "median"), ("MaxReduction", "max"),
                     ("VarianceReduction", "var")):
        for box in (1, 0):
            wi, keep = _reduce(L, paths, op, box=box)
            _same(_blocks(L, wi), _expected(oracle, paths, name, box=bool(box)))
    wi, keep = _reduce(L, paths, "SumReduction", strict=True)
    _same(_pops(L, wi), _expected(oracle, paths, "sum", strict=True))


def _run_single_reader(L, tmp_path):
    """The reader alone, one pop at a time, and after seek (bigWiggleReader.c:125-145)."""
    paths = _write_set(tmp_path, 1, seed=5)
    t = bigwig.load_runlists(paths, box=True)
    exp = []
    for ci, name in enumerate(t.chrom_names):
        a, b = t.seg_off[ci], t.seg_off[ci + 1]       # one track
        exp += [(name, int(s), int(f), float(v)) for s, f, v in zip(t.start[a:b], t.finish[a:b], t.value[a:b])]
    wi = L.wtamd_BigWiggleReader(paths[0].encode(), 1)
    assert _pops(L, wi) == exp
    # after seek the reference issues ONE region query (bigWiggleReader.c:91-92,125-145): intervals are boxed
    # into [lo, hi) only (:42-44), NOT cut at the 10 000-bp stretch edges of a whole-chromosome read
    u = bigwig.load_runlists(paths, box=False)
    raw = []
    for ci, name in enumerate(u.chrom_names):
        a, b = u.seg_off[ci], u.seg_off[ci + 1]
        raw += [(name, int(s), int(f), float(v)) for s, f, v in zip(u.start[a:b], u.finish[a:b], u.value[a:b])]
    assert len(raw) < len(exp)              # some interval does cross a stretch edge
    wi = L.wtamd_BigWiggleReader(paths[0].encode(), 1)
    for chrom, lo, hi in (("chr2", 10007, 23456), ("chr1", 1, 500), ("chr2", 40000, 50000), ("chrZ", 1, 10),
                          ("chr1", 15000, 45000)):
        L.seek(wi, chrom.encode(), lo, hi)
        want = [(c, max(s, lo), min(f, hi), v) for c, s, f, v in raw if c == chrom and f > lo and s < hi]
        assert _pops(L, wi) == want
    # the reference destroys iterators with free(data) (wiggleIterator.c:52-55): the handle is free()-able,
    # the reader behind it (and its idle producer thread) stays
    wi = L.wtamd_BigWiggleReader(paths[0].encode(), 1)
    assert len(_pops(L, wi, limit=100)) == 100
    L.destroyWiggleIterator(wi)


def test_bigwig_reader_pop_and_seek_emu(emu_lib, tmp_path):
    _run_single_reader(emu_lib, tmp_path)


def test_bigwig_files_to_reducers_emu(emu_lib, oracle, tmp_path):
    _run_all(emu_lib, oracle, tmp_path)


def test_bigwig_small_batches_emu(emu_lib, oracle, tmp_path, monkeypatch):
    """Batch seams inside chromosomes while the producer thread recycles its buffers."""
    monkeypatch.setenv("WTAMD_BATCH_INTERVALS", "700")
    monkeypatch.setenv("WTAMD_BATCH_RUNS", "300")
    paths = _write_set(tmp_path, 5, seed=23)
    wi, keep = _reduce(emu_lib, paths, "MeanReduction")
    _same(_blocks(emu_lib, wi), _expected(oracle, paths, "mean"))


@pytest.mark.gpu
def test_bigwig_reader_pop_and_seek_gpu(amd_lib, tmp_path):
    _run_single_reader(amd_lib, tmp_path)


@pytest.mark.gpu
def test_bigwig_files_to_reducers_gpu(amd_lib, oracle, tmp_path):
    _run_all(amd_lib, oracle, tmp_path)


def _close_case(L, tmp_path):
    """wtamd_BigWiggleReader_close: after the priming block only (no producer thread yet) and after several parts
    (thread running); anything else is refused."""
    import threading
    L.wtamd_BigWiggleReader_close.argtypes = [C.c_void_p]
    L.wtamd_BigWiggleReader_close.restype = C.c_int
    paths = _write_set(tmp_path, 2, seed=77, block=13)
    before = threading.active_count()
    wi = L.wtamd_BigWiggleReader(paths[0].encode(), 1)
    assert L.wtamd_BigWiggleReader_close(wi) == 0
    assert C.cast(wi, C.POINTER(WI)).contents.done
    assert L.wtamd_BigWiggleReader_close(wi) != 0                  # already closed
    wi = L.wtamd_BigWiggleReader(paths[1].encode(), 1)
    got = _pops(L, wi, limit=500)                                   # several 13-item blocks: the producer was started
    assert len(got) == 500
    assert L.wtamd_BigWiggleReader_close(wi) == 0
    assert threading.active_count() == before                       # (python's view; the native thread was joined)
    its = (C.c_void_p * 1)(L.wtamd_BigWiggleReader(paths[0].encode(), 1))
    m = L.newMultiplexer(its, 1, b"\x00")
    assert L.wtamd_BigWiggleReader_close(L.MeanReduction(m)) != 0   # not a BigWig reader


def test_bigwig_reader_close_emu(emu_lib, tmp_path):
    _close_case(emu_lib, tmp_path)


@pytest.mark.gpu
def test_bigwig_reader_close_gpu(amd_lib, tmp_path):
    _close_case(amd_lib, tmp_path)


def test_array_writer_round_trip(tmp_path):
    """wiggletools_amd/bwwrite.py (bench files): two-level index, chromosomes without data."""
    from wiggletools_amd import bwwrite
    rng = np.random.default_rng(1)
    n = 300000
    ln, gap = rng.integers(1, 30, n), rng.integers(0, 5, n)
    s = np.cumsum(ln + gap) - ln
    e = s + ln
    v = rng.integers(0, 100, n).astype(np.float32) / 4
    p = str(tmp_path / "a.bw")
    assert bwwrite.write_arrays(p, {"chr1": int(e[-1]) + 5, "chrA": 10}, {"chr1": (s, e, v)}) > 256
    bw = bigwig.BigWig(p)
    a, b, c = bw.read("chr1", box=False)
    assert np.array_equal(a, s + 1) and np.array_equal(b, e + 1) and np.array_equal(c, v)
    assert bw.read("chrA")[0].size == 0
