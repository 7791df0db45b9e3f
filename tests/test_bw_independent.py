"""Second opinion on the BigWig decoder: tests/bw_indep_reader.py -- written from the
format description, sharing nothing with the library's decoder or with the writers used elsewhere in the tests --
against (a) the reference's own fixtures test/fixedStep.bw / variableStep.bw and their .wig twins (test/test.py:28,52),
which carry ZOOM LEVELS and a total-summary block the decoders must step over, and (b) the library's host decoder
(csrc/wt_bigwig.cpp) and device-path emulation on synthetic files of every section type, compressed and raw, with a
chromosome without data, a two-level R-tree, and index leaves whose extents overlap."""
import os
import struct

import numpy as np
import pytest

from bw_indep_reader import IndependentBigWig, parse_wig
from bw_writer import write_bigwig
from wiggletools_amd import bigwig, bwwrite

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _lib_unboxed(path, chrom):
    s, f, v = bigwig.BigWig(path).read(chrom, box=False)
    return s.astype(np.int64) - 1, f.astype(np.int64) - 1, v      # back to 0-based half-open


def _same(a, b):
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))


@pytest.mark.parametrize("name", ["fixedStep", "variableStep"])
def test_reference_fixtures_three_ways(name):
    """The reference's .bw == its .wig (what test/test.py asserts through the CLI), read by the independent reader
    and by the library; the files have zoom levels and a summary block."""
    bw = IndependentBigWig(os.path.join(G, name + ".bw"))
    assert bw.n_zoom >= 1 and bw.summary_off > 0 and bw.uncompress_buf > 0
    wig = parse_wig(os.path.join(G, name + ".wig"))
    assert set(bw.chroms) == set(wig)
    for chrom in wig:
        got = bw.intervals(chrom)
        _same(got, wig[chrom])
        _same(_lib_unboxed(os.path.join(G, name + ".bw"), chrom), wig[chrom])
    # zoom data is a separate R-tree of summary records: neither decoder may mistake it for sections
    lib = bigwig.BigWig(os.path.join(G, name + ".bw"))
    assert sum(len(lib.read(c, box=False)[0]) for c in lib.chroms) == sum(len(wig[c][0]) for c in wig)


def _random_records(rng, length, density=0.8, max_len=40):
    pos, recs = int(rng.integers(0, 50)), []
    while pos < length - max_len - 20:
        ln = int(rng.integers(1, max_len))
        if rng.random() < density:
            recs.append((pos, pos + ln, float(np.float32(rng.integers(-64, 64) / 4))))
        pos += ln
    return recs


@pytest.mark.parametrize("compress", [True, False])
@pytest.mark.parametrize("mix", [True, False])
def test_library_decoder_vs_independent_reader(tmp_path, compress, mix):
    """All three section types (mix), zlib and raw, a chromosome that has no data, > 256 sections (two-level R-tree)."""
    rng = np.random.default_rng(17 + 2 * compress + mix)
    chroms = {"chr1": 90000, "chr10": 31001, "chrEmpty": 5000, "chrM": 900}
    data = {c: _random_records(rng, n) for c, n in chroms.items() if c != "chrEmpty"}
    p = str(tmp_path / "a.bw")
    write_bigwig(p, chroms, data, items_per_block=13, compress=compress, mix_types=mix)
    bw = IndependentBigWig(p)
    assert bw.n_sections > 256 and (bw.uncompress_buf > 0) == compress
    for c in chroms:
        want = data.get(c, [])
        got = bw.intervals(c)
        assert [(int(a), int(b), float(v)) for a, b, v in zip(*got)] == want
        _same(_lib_unboxed(p, c), got)
    # region queries: the R-tree pruning of the independent reader against the full list
    s, e, v = bw.intervals("chr1")
    for lo, hi in ((0, 10), (12345, 23456), (89000, 90000), (40000, 40001)):
        qs, qe, qv = bw.intervals("chr1", lo, hi)
        keep = (e > lo) & (s < hi)
        assert np.array_equal(qs, s[keep]) and np.array_equal(qe, e[keep])


def test_array_writer_files_vs_independent_reader(tmp_path):
    """wiggletools_amd/bwwrite.py (the bench's files: 1024-item bedGraph sections, zlib level 1)."""
    rng = np.random.default_rng(3)
    n = 50000
    ln, gap = rng.integers(1, 30, n), (rng.random(n) < 0.05) * rng.integers(1, 100, n)
    e = np.cumsum(ln + gap)
    s = e - ln
    v = (rng.integers(0, 800, n) / 8).astype(np.float32)
    p = str(tmp_path / "b.bw")
    bwwrite.write_arrays(p, {"chr1": int(e[-1]) + 5, "chrA": 10}, {"chr1": (s, e, v)})
    bw = IndependentBigWig(p)
    got = bw.intervals("chr1")
    _same(got, (s.astype(np.int64), e.astype(np.int64), v))
    _same(_lib_unboxed(p, "chr1"), got)
    assert bw.intervals("chrA")[0].size == 0 and bw.intervals("nope")[0].size == 0


def test_overlapping_index_leaves(tmp_path):
    """An index leaf whose extents reach into the next leaf (legal): every item still comes back exactly once, from
    both readers, and a region query inside the overlap finds the items of both sections."""
    recs = [(10 + 7 * k, 14 + 7 * k, float(k % 5)) for k in range(400)]
    p = str(tmp_path / "overlap.bw")
    write_bigwig(p, {"chr1": 60000}, {"chr1": recs}, items_per_block=100)
    raw = bytearray(open(p, "rb").read())
    idx = struct.unpack_from("<Q", raw, 24)[0]
    struct.pack_into("<I", raw, idx + 48 + 4 + 12, recs[150][1])       # first leaf's end -> inside the second section
    open(p, "wb").write(bytes(raw))
    bw = IndependentBigWig(p)
    got = bw.intervals("chr1")
    assert [(int(a), int(b), float(v)) for a, b, v in zip(*got)] == recs
    _same(_lib_unboxed(p, "chr1"), got)
    lo, hi = recs[120][0], recs[130][1]
    qs, _, _ = bw.intervals("chr1", lo, hi)
    assert list(qs) == [r[0] for r in recs[120:131]]


def test_native_fileset_writer_three_ways(tmp_path):
    """csrc/wt_bwwrite.cpp through wiggletools_amd.bwwrite.FileSet -- the writer of bench.py's whole-genome inputs: several
    files, several chromosomes (one of them without data), chromosome-wise calls with the tracks dealt to threads, and
    MORE THAN 65 536 SECTIONS in one file (a three-level R-tree; bwwrite.write_arrays stopped at two).  Read back by the
    library's host decoder and by the independent reader: the input, value bits included."""
    rng = np.random.default_rng(12)
    chroms = {"chr1": 4_000_000, "chr10": 900_000, "chr2": 50, "chrX": 2_000_000}
    n_tracks = 3
    paths = [str(tmp_path / ("fs%d.bw" % t)) for t in range(n_tracks)]
    fs = bwwrite.FileSet(paths, chroms, items_per_block=5, level=1, threads=3)
    truth = {}
    for name in sorted(chroms, key=lambda x: x.encode()):
        if name == "chr2":
            continue                            # a chromosome of the tree that holds no data
        seg, S, F, V = [0], [], [], []
        for t in range(n_tracks):
            n = 340_000 if (name == "chr1" and t == 0) else int(rng.integers(2000, 6000))
            ln = rng.integers(1, 9, n)
            gap = (rng.random(n) < 0.1) * rng.integers(1, 30, n)
            f = np.cumsum(ln + gap) + 1
            s = f - ln
            keep = f < chroms[name]
            s, f = s[keep].astype(np.int32), f[keep].astype(np.int32)
            v = (rng.integers(-80, 80, int(keep.sum())) / 8).astype(np.float32)
            S.append(s); F.append(f); V.append(v)
            seg.append(seg[-1] + len(s))
            truth[(t, name)] = (s.astype(np.int64) - 1, f.astype(np.int64) - 1, v)
        fs.add_chrom(name, seg, np.concatenate(S), np.concatenate(F), np.concatenate(V))
    sections = fs.close()
    assert sections[0] > 65536
    for t, path in enumerate(paths):
        ind = IndependentBigWig(path)
        assert set(ind.chroms) == set(chroms)
        for name in chroms:
            if name == "chr2":
                assert len(ind.intervals(name)[0]) == 0
                continue
            _same(ind.intervals(name), truth[(t, name)])
            _same(_lib_unboxed(path, name), truth[(t, name)])
