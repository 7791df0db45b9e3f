"""BigWig section decoder (csrc/wt_bigwig.cpp).  Pins: the reference's own fixtures
fixedStep.bw == fixedStep.wig and variableStep.bw == variableStep.wig (reference
test/test.py:28,52), plus synthetic files written by tests/bw_writer.py (all three section
types, several blocks, several chromosomes, uncompressed sections, 10 kb boxing)."""
import os

import numpy as np
import pytest

from bw_writer import write_bigwig
from wiggletools_amd import bigwig, textio

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("stem", ["fixedStep", "variableStep"])
def test_reference_fixture_bw_equals_wig(stem):
    bw = bigwig.load_runlists([os.path.join(G, stem + ".bw")])
    wig = textio.load_runlists([os.path.join(G, stem + ".wig")])
    assert bw.chrom_names == wig.chrom_names == ["chr1"]
    assert np.array_equal(bw.start, wig.start) and np.array_equal(bw.finish, wig.finish)
    assert np.array_equal(bw.value.astype(np.float64), wig.value)


def test_not_a_bigwig(tmp_path):
    p = tmp_path / "x.bw"
    p.write_bytes(b"not a bigwig at all" * 10)
    with pytest.raises(ValueError, match="not in BigWig format"):
        bigwig.BigWig(str(p))


@pytest.mark.parametrize("compress", [True, False])
def test_synthetic_all_section_types(tmp_path, compress):
    rng = np.random.default_rng(3)
    chroms = {"chr1": 50000, "chr10": 30001, "chrX": 12345}
    data = {}
    for c, L in chroms.items():
        pos, recs = 0, []
        while pos < L - 40:
            ln = int(rng.integers(1, 30))
            if rng.random() < 0.9:
                recs.append((pos, pos + ln, float(np.float32(rng.integers(0, 800) / 8))))
            pos += ln
        data[c] = recs
    path = str(tmp_path / "s.bw")
    write_bigwig(path, chroms, data, items_per_block=97, compress=compress, mix_types=True)
    bw = bigwig.BigWig(path)
    assert bw.chroms == chroms
    for c in chroms:
        s, f, v = bw.read(c, box=False)
        exp = data[c]
        assert s.tolist() == [a + 1 for a, _, _ in exp]
        assert f.tolist() == [b + 1 for _, b, _ in exp]
        assert v.tolist() == [x for _, _, x in exp]


def test_boxing_matches_reference_reader_rule(tmp_path):
    """Runs are cut at 1+10000k (bigWiggleReader.c:42-44,73-83); the stretch loop stops at
    `start < length` (:76), so a chromosome of length 1 (mod 10000) loses its last base."""
    chroms = {"chrA": 30001, "chrB": 25000}
    data = {"chrA": [(5, 9995, 1.0), (9995, 10005, 2.0), (19990, 30001, 3.0)],
            "chrB": [(0, 25000, 7.0)]}
    path = str(tmp_path / "b.bw")
    write_bigwig(path, chroms, data)
    bw = bigwig.BigWig(path)
    s, f, v = bw.read("chrA", box=True)
    assert list(zip(s.tolist(), f.tolist(), v.tolist())) == [
        (6, 9996, 1.0), (9996, 10001, 2.0), (10001, 10006, 2.0),
        (19991, 20001, 3.0), (20001, 30001, 3.0)]          # base 30001 (1-based) is never visited
    s, f, v = bw.read("chrB", box=True)
    assert list(zip(s.tolist(), f.tolist())) == [(1, 10001), (10001, 20001), (20001, 25001)]
    s, f, v = bw.read("chrA", box=False)
    assert list(zip(s.tolist(), f.tolist())) == [(6, 9996), (9996, 10006), (19991, 30002)]


def test_bigwig_tracks_feed_the_oracle_like_wig_tracks(oracle):
    """BASELINE config C1 plumbing on CPU: mean fixedStep.bw variableStep.bw == expected text of
    SURVEY 8c (the GPU run of the same is in tests/test_gpu_parity.py)."""
    t = bigwig.load_runlists([os.path.join(G, "fixedStep.bw"), os.path.join(G, "variableStep.bw")])
    c, s, f, v = oracle.reduce(t.as_dict(), "mean")
    assert s.tolist() == list(range(1, 11))
    assert v.tolist() == [0.5, 1.5, 1, 3, 2, 4.5, 3, 6, 4, 4.5]


def test_read_part_streams_a_chromosome_in_pieces(tmp_path):
    """wtamd_bw_read_part (the reader iterator's producer calls it): successive parts concatenate to the
    whole chromosome, `last` ends it, a too-small capacity returns the needed size without moving the
    cursor, and a window skips the data blocks outside it."""
    import ctypes as C
    from wiggletools_amd import _lib, bwwrite
    rng = np.random.default_rng(4)
    n = 20000
    ln, gap = rng.integers(1, 30, n), rng.integers(0, 3, n)
    s = np.cumsum(ln + gap) - ln
    e = s + ln
    v = rng.integers(0, 100, n).astype(np.float32) / 4
    p = str(tmp_path / "p.bw")
    bwwrite.write_arrays(p, {"chr1": int(e[-1]) + 5, "chr2": 50}, {"chr1": (s, e, v), "chr2": (np.array([3]), np.array([9]), np.array([1.0], np.float32))},
                         items_per_block=256)
    bw = bigwig.BigWig(p)
    whole = bw.read("chr1", box=True)
    L = _lib.lib()
    L.wtamd_bw_read_part.restype = C.c_int64
    L.wtamd_bw_read_part.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.c_int, C.c_int32, C.c_int32, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    cap = 1 << 16
    S, F, V = np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float32)

    def parts(lo0, hi0, blocks):
        cur, last, out, calls = C.c_int64(0), C.c_int(0), [], 0
        while not last.value:
            k = L.wtamd_bw_read_part(bw._h, b"chr1", 1, C.byref(cur), blocks, lo0, hi0, cap, S.ctypes.data, F.ctypes.data, V.ctypes.data, C.byref(last))
            assert k >= 0
            out.append((S[:k].copy(), F[:k].copy(), V[:k].copy()))
            calls += 1
        return [np.concatenate([o[q] for o in out]) for q in range(3)], calls

    (a, b, c), calls = parts(0, 2 ** 31 - 1, 7)
    assert calls > 5
    assert np.array_equal(a, whole[0]) and np.array_equal(b, whole[1]) and np.array_equal(c, whole[2])
    # capacity too small: the needed size comes back, the cursor stays, the retry delivers
    cur, last = C.c_int64(0), C.c_int(0)
    k = L.wtamd_bw_read_part(bw._h, b"chr1", 1, C.byref(cur), 4, 0, 2 ** 31 - 1, 10, S.ctypes.data, F.ctypes.data, V.ctypes.data, C.byref(last))
    assert k > 10 and cur.value == 0
    k2 = L.wtamd_bw_read_part(bw._h, b"chr1", 1, C.byref(cur), 4, 0, 2 ** 31 - 1, cap, S.ctypes.data, F.ctypes.data, V.ctypes.data, C.byref(last))
    assert k2 == k and cur.value > 0 and np.array_equal(S[:k2], whole[0][:k2])
    # a window: only the blocks overlapping it are decoded; every run overlapping it is there
    lo, hi = int(s[n // 2]), int(s[n // 2 + 900])
    (a, b, c), _ = parts(lo, hi, 64)
    inside = (whole[1] - 1 > lo) & (whole[0] - 1 < hi)
    assert a.size < whole[0].size // 4
    assert set(zip(whole[0][inside].tolist(), whole[1][inside].tolist())) <= set(zip(a.tolist(), b.tolist()))
    bw.close()
