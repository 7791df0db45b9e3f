"""BigWig sections decoded ON THE DEVICE (csrc/wt_inflate.h, csrc/wt_bwdev_core.h, csrc/wt_bwdev.hip; the Feeder's
file-byte batches in csrc/wt_iter_abi.cpp) -- what replaces the host-side zlib inflate the reference gets from
libBigWig (src/bigWiggleReader.c:52-83).

CPU (`-m "not gpu"`): the kernels' own per-lane inflate state machine and per-item arithmetic, compiled for the host
(tests/emu/wt_pipe_emu.cpp), against zlib itself, against the library's host decoder and against the oracle over the
run lists -- incl. the reference's fixtures test/fixedStep.bw / variableStep.bw (== their .wig, test/test.py:28,52).
`-m gpu`: the same through the product library (HIP kernels)."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

from bw_writer import write_bigwig
from test_bwreader import G, WI, _bind, _blocks, _expected, _pops, _reduce, _same, _write_set
from wiggletools_amd import bigwig
from wiggletools_amd.pipe import PipeStats


@pytest.fixture(scope="module")
def emu_lib():
    from emu.build import build_dropin
    L = _bind(C.CDLL(build_dropin()))
    L.wtemu_inflate.restype = C.c_longlong
    L.wtemu_inflate.argtypes = [C.c_char_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int]
    return L


@pytest.fixture(scope="module")
def amd_lib():
    from wiggletools_amd import _lib
    return _bind(_lib.lib())


def _stats(L, wi):
    st = PipeStats()
    L.wtamd_iterator_pipe_stats.argtypes = [C.c_void_p, C.POINTER(PipeStats)]
    assert L.wtamd_iterator_pipe_stats(wi, C.byref(st)) == 0
    return st


def _streams():
    rng = np.random.default_rng(5)
    for trial in range(240):
        kind = trial % 5
        n = int(rng.integers(0, 20)) if trial < 10 else int(rng.integers(1, 14000)) if trial % 40 else int(rng.integers(70000, 140000))
        if kind == 0:
            raw = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        elif kind == 1:
            raw = (b"abcabcabd" * (n // 9 + 1))[:n]
        elif kind == 2:         # bedGraph-like records
            k = n // 12 + 1
            ln = rng.integers(1, 30, k)
            e = np.cumsum(ln + (rng.random(k) < 0.02) * rng.integers(0, 100, k)).astype(np.uint32)
            rec = np.empty((k, 3), np.uint32)
            rec[:, 0] = e - ln.astype(np.uint32); rec[:, 1] = e
            rec[:, 2] = (rng.integers(0, 800, k) / 8).astype(np.float32).view(np.uint32)
            raw = rec.tobytes()[:n]
        elif kind == 3:
            raw = b"x" * n
        else:
            raw = rng.integers(0, 4, n, dtype=np.uint8).tobytes()
        level = trial % 10
        strategy = (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED)[(trial // 10) % 5]
        wbits = -15 if (trial // 7) % 3 == 0 else 15
        co = zlib.compressobj(level, zlib.DEFLATED, wbits, 1 + trial % 9, strategy)
        if trial % 3 == 0 and n > 10:     # several blocks
            comp = co.compress(raw[:n // 2]) + co.flush(zlib.Z_FULL_FLUSH if trial % 2 else zlib.Z_SYNC_FLUSH) + co.compress(raw[n // 2:]) + co.flush()
        else:
            comp = co.compress(raw) + co.flush()
        yield trial, raw, comp, wbits < 0


def test_lane_inflate_equals_zlib(emu_lib):
    """csrc/wt_inflate.h (the state machine one GPU lane runs per section) == zlib on stored / fixed / dynamic blocks
    of every level and strategy, raw and zlib-wrapped, several blocks, long matches and distances > the LDS ring."""
    for trial, raw, comp, is_raw in _streams():
        out = np.zeros(len(raw) + 8, np.uint8)
        got = emu_lib.wtemu_inflate(comp, len(comp), out.ctypes.data, len(raw), int(is_raw))
        assert got == len(raw), (trial, got, len(raw))
        assert out[:len(raw)].tobytes() == raw, trial
        if len(raw) > 8:            # too little room: a clean error, never an overrun
            assert emu_lib.wtemu_inflate(comp, len(comp), out.ctypes.data, len(raw) - 5, int(is_raw)) == -6
        if len(comp) > 16:          # truncated stream
            assert emu_lib.wtemu_inflate(comp, len(comp) - 9, out.ctypes.data, len(raw), int(is_raw)) < 0
    rng = np.random.default_rng(1)
    for trial, raw, comp, is_raw in _streams():     # corrupted streams: any verdict, no crash
        bad = bytearray(comp)
        for _ in range(3):
            if bad:
                bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        out = np.zeros(len(raw) + 8, np.uint8)
        emu_lib.wtemu_inflate(bytes(bad), len(bad), out.ctypes.data, len(raw), int(is_raw))


def test_lane_inflate_literal_in_front_of_a_match(emu_lib):
    """Round 6: a step emits a leading literal and the symbol behind it (csrc/wt_inflate.h WT_INF_LEAD).  The cases where the literal
    matters to what follows: a match at distance 1 .. 4 right behind it (its source IS the literal, or the partial dword it completes),
    at every alignment of the output position, lengths across the 7-byte first put, runs of literals, a literal before the end of a
    block, and output space that ends between the literal and the symbol behind it."""
    cases = []
    for pre in range(0, 9):
        for period in (1, 2, 3, 4):
            for rep in (3, 4, 7, 8, 9, 15, 16, 17, 40, 258, 259, 600):
                unit = bytes([65 + k for k in range(period)])
                head = bytes([200 + (k * 7) % 50 for k in range(pre)])
                cases.append(head + unit + unit * rep + b"Z" + unit[:1] * 5 + b"qrs")
    cases += [b"a", b"ab", b"abc", b"aaaa", b"abababab", bytes(range(256)) * 3, b"\0" * 1000 + b"\1" + b"\0" * 1000]
    for i, raw in enumerate(cases):
        for level, strategy in ((1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_RLE), (6, zlib.Z_FIXED)):
            co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, strategy)
            comp = co.compress(raw) + co.flush()
            out = np.zeros(len(raw) + 8, np.uint8)
            got = emu_lib.wtemu_inflate(comp, len(comp), out.ctypes.data, len(raw), 0)
            assert got == len(raw) and out[:len(raw)].tobytes() == raw, (i, level, strategy, got, len(raw))
            for short in (1, 2, 3):     # no room for the last bytes: a clean error whichever symbol of a step hits the end
                if len(raw) > short:
                    assert emu_lib.wtemu_inflate(comp, len(comp), out.ctypes.data, len(raw) - short, 0) == -6, (i, level, short)


def _device_vs_host(L, oracle, tmp_path, monkeypatch, must_be_device=True):
    # C1 of BASELINE.json through the device decoder: the reference's own fixtures (type 3 and type 2 sections)
    paths = [os.path.join(G, "fixedStep.bw"), os.path.join(G, "variableStep.bw")]
    wi, keep = _reduce(L, paths, "MeanReduction")
    got = _pops(L, wi)
    assert [g[1] for g in got] == list(range(1, 11))
    assert [g[3] for g in got] == [0.5, 1.5, 1, 3, 2, 4.5, 3, 6, 4, 4.5]
    if must_be_device:
        assert _stats(L, wi).bw_sections > 0
    # synthetic files: several chromosomes, a file lacking one, all section types, compressed and raw
    paths = _write_set(tmp_path, 7, seed=31)
    for op, name in (("MeanReduction", "mean"), ("MedianReduction", "median"), ("VarianceReduction", "var")):
        for box in (1, 0):
            for strict in (False, True):
                monkeypatch.setenv("WTAMD_BW_DEVICE", "1")
                wi, keep = _reduce(L, paths, op, box=box, strict=strict)
                dev = _blocks(L, wi)
                st = _stats(L, wi)
                assert st.bw_sections > 0 and st.intervals > 0
                monkeypatch.setenv("WTAMD_BW_DEVICE", "0")
                wi, keep = _reduce(L, paths, op, box=box, strict=strict)
                host = _blocks(L, wi)
                assert _stats(L, wi).bw_sections == 0
                assert dev == host or (_same(dev, host) is None)
                _same(dev, _expected(oracle, paths, name, box=bool(box), strict=strict))
    monkeypatch.delenv("WTAMD_BW_DEVICE")


def test_device_decode_equals_host_decoder_emu(emu_lib, oracle, tmp_path, monkeypatch):
    _device_vs_host(emu_lib, oracle, tmp_path, monkeypatch)


def test_device_decode_batch_seams_emu(emu_lib, oracle, tmp_path, monkeypatch):
    """Tiny batches: sections straddle every cut, are decoded by both neighbours and must yield each run once."""
    paths = _write_set(tmp_path, 5, seed=23, block=37)
    exp = _expected(oracle, paths, "mean")
    for nbytes, nruns, span in (("600", "300", None), ("3000", "1500", None), ("1", "64", "16")):
        monkeypatch.setenv("WTAMD_BW_BATCH_BYTES", nbytes)
        monkeypatch.setenv("WTAMD_BATCH_RUNS", nruns)
        if span:
            monkeypatch.setenv("WTAMD_MIN_SPAN", span)
        wi, keep = _reduce(emu_lib, paths, "MeanReduction")
        _same(_blocks(emu_lib, wi), exp)
        st = _stats(emu_lib, wi)
        assert st.bw_sections > 0 and st.batches > 20


def _seek_cases(L, oracle, tmp_path, monkeypatch):
    """seek on a reducer over device-decoded files == the host decoder == the oracle over the clipped, UNBOXED runs
    (one region query after seek: bigWiggleReader.c:91-92,125-145)."""
    paths = _write_set(tmp_path, 4, seed=3, block=53)
    L.seek.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    for chrom, lo, hi in (("chr1", 9000, 31234), ("chr2", 1, 700), ("chr10", 20000, 99999), ("chrQ", 5, 50), ("chr1", 59000, 61000)):
        res = {}
        for dev in ("1", "0"):
            monkeypatch.setenv("WTAMD_BW_DEVICE", dev)
            wi, keep = _reduce(L, paths, "SumReduction")
            L.seek(wi, chrom.encode(), lo, hi)
            res[dev] = _pops(L, wi)
            assert (_stats(L, wi).bw_sections > 0) == (dev == "1") or not res[dev]
        assert res["1"] == res["0"], (chrom, lo, hi)
        # expected: oracle over the unboxed runs clipped to the window
        t = bigwig.load_runlists(paths, box=False)
        if chrom in t.chrom_names:
            d = t.as_dict()
            ci = t.chrom_names.index(chrom)
            N = t.n_tracks
            seg, S, F, V = [0], [], [], []
            for i in range(N):
                a, b = t.seg_off[ci * N + i], t.seg_off[ci * N + i + 1]
                s, f, v = t.start[a:b], t.finish[a:b], t.value[a:b]
                m = (f > lo) & (s < hi)
                S.append(np.maximum(s[m], lo)); F.append(np.minimum(f[m], hi)); V.append(v[m])
                seg.append(seg[-1] + int(m.sum()))
            d = dict(d, n_chrom=1, seg_off=np.array(seg, np.int64), start=np.concatenate(S).astype(np.int32),
                     finish=np.concatenate(F).astype(np.int32), value=np.concatenate(V).astype(np.float32))
            c, s, f, v = oracle.reduce(d, "sum")
            assert [(a, b, x) for _, a, b, x in res["1"]] == [(int(a), int(b), float(x)) for a, b, x in zip(s, f, v)]
        else:
            assert res["1"] == []
    monkeypatch.delenv("WTAMD_BW_DEVICE")


def test_device_decode_seek_emu(emu_lib, oracle, tmp_path, monkeypatch):
    _seek_cases(emu_lib, oracle, tmp_path, monkeypatch)


def test_overlapping_leaves_fall_back_to_host_decoder_emu(emu_lib, oracle, tmp_path):
    """A file whose index leaves overlap (legal, if unusual) is not handed to the device: the Feeder keeps the host
    decoder for the whole track set -- same results, bw_sections == 0."""
    paths = _write_set(tmp_path, 3, seed=9)
    # second file: two leaves whose extents overlap although their items do not
    recs = [(10 + 7 * k, 14 + 7 * k, float(k % 5)) for k in range(400)]
    p = str(tmp_path / "overlap.bw")
    write_bigwig(p, {"chr1": 60000}, {"chr1": recs}, items_per_block=100)
    raw = bytearray(open(p, "rb").read())
    # widen the first leaf's end beyond the second leaf's start (R-tree leaf item: chromIx, start, chromIx, end, off, size)
    import struct
    idx = struct.unpack_from("<Q", raw, 24)[0]
    first_item = idx + 48 + 4
    struct.pack_into("<I", raw, first_item + 12, recs[150][1])
    open(p, "wb").write(bytes(raw))
    paths.append(p)
    wi, keep = _reduce(emu_lib, paths, "MeanReduction")
    got = _blocks(emu_lib, wi)
    assert _stats(emu_lib, wi).bw_sections == 0
    _same(got, _expected(oracle, paths, "mean"))


def test_corrupt_section_fails_loudly_emu(tmp_path):
    """A damaged zlib stream: the batch fails with the engine's message and exit(1) (the host decoder's behaviour too)."""
    import subprocess, sys
    rng = np.random.default_rng(2)
    recs = [(20 * k, 20 * k + 7, float(rng.integers(0, 9))) for k in range(3000)]
    p = str(tmp_path / "bad.bw")
    write_bigwig(p, {"chr1": 80000}, {"chr1": recs}, items_per_block=256)
    raw = bytearray(open(p, "rb").read())
    import struct
    data_off = struct.unpack_from("<Q", raw, 16)[0]
    for q in range(40, 60):
        raw[data_off + 8 + q] ^= 0x5A
    open(p, "wb").write(bytes(raw))
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from emu.build import build_dropin; L = C.CDLL(build_dropin());"
            "L.wtamd_BigWiggleReader.restype = C.c_void_p; L.wtamd_BigWiggleReader.argtypes = [C.c_char_p, C.c_int];"
            "L.newMultiplexer.restype = C.c_void_p; L.newMultiplexer.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char];"
            "L.MeanReduction.restype = C.c_void_p; L.MeanReduction.argtypes = [C.c_void_p];"
            "its = (C.c_void_p * 1)(L.wtamd_BigWiggleReader(%r, 1)); m = L.newMultiplexer(its, 1, b'\\0'); L.MeanReduction(m); print('survived')"
            % (os.path.dirname(os.path.abspath(__file__)), p.encode()))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 1 and "survived" not in r.stdout


@pytest.mark.gpu
def test_device_decode_equals_host_decoder_gpu(amd_lib, oracle, tmp_path, monkeypatch):
    _device_vs_host(amd_lib, oracle, tmp_path, monkeypatch)


@pytest.mark.gpu
def test_device_decode_seek_gpu(amd_lib, oracle, tmp_path, monkeypatch):
    _seek_cases(amd_lib, oracle, tmp_path, monkeypatch)


def _one_track(t, L):
    """A dense synthetic track (1-based runs): mean run 16 bp, 2 % gaps, values k/8."""
    rng = np.random.default_rng(100 + t)
    n = L // 12
    ln = rng.geometric(1 / 16.0, n).astype(np.int64)
    gap = (rng.random(n) < 0.02) * rng.integers(1, 200, n)
    f = np.cumsum(ln + gap)
    s = f - ln
    keep = f < L
    return (s[keep] + 1).astype(np.int32), (f[keep] + 1).astype(np.int32), (rng.integers(0, 800, n)[keep] / 8).astype(np.float32)


@pytest.mark.gpu
def test_device_decode_larger_files_gpu(amd_lib, oracle, tmp_path, monkeypatch):
    """Bench-style files (wiggletools_amd/bwwrite.py: 1024-item bedGraph sections, zlib level 1), 12 tracks x 3 Mbp:
    thousands of sections per batch through the lane-per-section inflate; == host decoder run for run."""
    from wiggletools_amd import bwwrite
    n_tracks, L = 12, 3_000_000
    paths = []
    for t in range(n_tracks):
        s, f, v = _one_track(t, L)
        p = str(tmp_path / ("big%d.bw" % t))
        bwwrite.write_arrays(p, {"chr1": L + 10}, {"chr1": (s - 1, f - 1, v)})
        paths.append(p)
    res = {}
    for dev in ("1", "0"):
        monkeypatch.setenv("WTAMD_BW_DEVICE", dev)
        wi, keep = _reduce(amd_lib, paths, "MeanReduction")
        res[dev] = _blocks(amd_lib, wi)
        st = _stats(amd_lib, wi)
        assert (st.bw_sections > 0) == (dev == "1")
    assert len(res["1"]) > 1_000_000
    assert res["1"] == res["0"]


def _genome_fileset(tmp_path, n_tracks, chroms, seed, items=64):
    """Multi-chromosome files from the native writer (wiggletools_amd.bwwrite.FileSet), dense tracks of mean run 16 bp;
    every third file lacks the second chromosome."""
    from wiggletools_amd import bwwrite
    rng = np.random.default_rng(seed)
    paths = [str(tmp_path / ("g%d.bw" % t)) for t in range(n_tracks)]
    fs = bwwrite.FileSet(paths, chroms, items_per_block=items, level=1, threads=4)
    names = sorted(chroms, key=lambda x: x.encode())
    for ci, name in enumerate(names):
        seg, S, F, V = [0], [], [], []
        for t in range(n_tracks):
            L = chroms[name]
            if ci == 1 and t % 3 == 2:
                seg.append(seg[-1])
                continue
            n = L // 12
            ln = rng.geometric(1 / 16.0, n).astype(np.int64)
            gap = (rng.random(n) < 0.02) * rng.integers(1, 200, n)
            f = np.cumsum(ln + gap) + 1 + int(rng.integers(0, 50))
            s = f - ln
            keep = f < L
            S.append(s[keep].astype(np.int32)); F.append(f[keep].astype(np.int32))
            V.append((rng.integers(0, 800, n)[keep] / 8).astype(np.float32))
            seg.append(seg[-1] + int(keep.sum()))
        fs.add_chrom(name, seg, np.concatenate(S), np.concatenate(F), np.concatenate(V))
    fs.close()
    return paths


def _multichrom(L, oracle, tmp_path, monkeypatch, n_tracks, scale, batch_sections, min_span=None):
    """Files that each hold SEVERAL chromosomes (the walk of reference src/bigWiggleReader.c:91-101 in strcmp order,
    stretches :52-83) through the device decoder, batches of a few hundred sections so that every chromosome takes several
    and every chromosome transition is a batch seam: == the host decoder == the oracle, run for run."""
    chroms = {"chr1": 60 * scale, "chr10": 25 * scale + 1, "chr2": 41 * scale, "chrM": 900, "chrX": 33 * scale}
    paths = _genome_fileset(tmp_path, n_tracks, chroms, seed=41)
    monkeypatch.setenv("WTAMD_BW_BATCH_SECTIONS", str(batch_sections))
    if min_span:
        monkeypatch.setenv("WTAMD_MIN_SPAN", str(min_span))        # (also the first span: 65 536 bp otherwise)
    res = {}
    for dev in ("1", "0"):
        monkeypatch.setenv("WTAMD_BW_DEVICE", dev)
        wi, keep = _reduce(L, paths, "MeanReduction")
        res[dev] = _blocks(L, wi)
        st = _stats(L, wi)
        assert (st.bw_sections > 0) == (dev == "1")
        if dev == "1":
            assert st.batches >= 3 * len(chroms) - 4, st.batches       # several batches per (large) chromosome
    monkeypatch.delenv("WTAMD_BW_DEVICE")
    assert [r[0] for r in res["1"][:1]] == ["chr1"] and {r[0] for r in res["1"]} == set(chroms)
    assert res["1"] == res["0"]
    _same(res["1"], _expected(oracle, paths, "mean"))


def test_device_decode_multichrom_emu(emu_lib, oracle, tmp_path, monkeypatch):
    _multichrom(emu_lib, oracle, tmp_path, monkeypatch, n_tracks=5, scale=1000, batch_sections=40, min_span=1000)


@pytest.mark.gpu
def test_device_decode_multichrom_gpu(amd_lib, oracle, tmp_path, monkeypatch):
    _multichrom(amd_lib, oracle, tmp_path, monkeypatch, n_tracks=12, scale=20000, batch_sections=1200, min_span=30000)


def _shrink_leaf(path, leaf_no, by):
    """Rewrites index leaf `leaf_no` of a (single-leaf-node or two-level) R-tree so that it claims to end `by` bases
    early: its last items then lie beyond the extents the index states -- something libBigWig, hence the reference
    (src/bigWiggleReader.c:52-83), never checks."""
    import struct
    raw = bytearray(open(path, "rb").read())
    idx = struct.unpack_from("<Q", raw, 24)[0]
    node = idx + 48
    is_leaf, _, cnt = struct.unpack_from("<BBH", raw, node)
    if not is_leaf:
        node = struct.unpack_from("<Q", raw, node + 4 + 16)[0]       # first child
        is_leaf, _, cnt = struct.unpack_from("<BBH", raw, node)
    assert is_leaf and leaf_no < cnt
    item = node + 4 + 32 * leaf_no
    end = struct.unpack_from("<I", raw, item + 12)[0]
    struct.pack_into("<I", raw, item + 12, end - by)
    open(path, "wb").write(bytes(raw))


def _fallback_case(L, oracle, tmp_path, monkeypatch, capfd):
    """A file whose items reach beyond their index leaf's stated extents works with the reference; the device decoder
    rejects the batch (error bit 4) and the drop-in layer goes on with the HOST decoder from that batch on -- mid-stream,
    after batches that were decoded on the device -- instead of exiting (the round-3 advisor's finding).  The result is
    the oracle's, run for run; WTAMD_BW_NO_FALLBACK=1 restores the loud failure."""
    paths = _write_set(tmp_path, 4, seed=19, block=41)
    victim = paths[1]
    _shrink_leaf(victim, 30, 3)
    monkeypatch.setenv("WTAMD_BW_BATCH_SECTIONS", "60")
    monkeypatch.setenv("WTAMD_MIN_SPAN", "1500")
    for op, name in (("MeanReduction", "mean"), ("MaxReduction", "max")):
        wi, keep = _reduce(L, paths, op)
        got = _blocks(L, wi)
        st = _stats(L, wi)
        _same(got, _expected(oracle, paths, name))
        assert "continuing with the host decoder" in capfd.readouterr().err
    # seek into the damaged region: same switch, the window's clipping kept
    wi, keep = _reduce(L, paths, "SumReduction")
    L.seek(wi, b"chr1", 100, 50000)
    got = _pops(L, wi)
    monkeypatch.setenv("WTAMD_BW_DEVICE", "0")
    wi, keep = _reduce(L, paths, "SumReduction")
    L.seek(wi, b"chr1", 100, 50000)
    assert got == _pops(L, wi) and len(got) > 100
    monkeypatch.delenv("WTAMD_BW_DEVICE")


def test_device_decode_error_falls_back_to_host_emu(emu_lib, oracle, tmp_path, monkeypatch, capfd):
    _fallback_case(emu_lib, oracle, tmp_path, monkeypatch, capfd)


@pytest.mark.gpu
def test_device_decode_error_falls_back_to_host_gpu(amd_lib, oracle, tmp_path, monkeypatch, capfd):
    _fallback_case(amd_lib, oracle, tmp_path, monkeypatch, capfd)


def _padded_leaves_case(L, oracle, tmp_path, monkeypatch, capfd):
    """Index leaves whose size includes padding BEHIND the zlib stream (zlib / libBigWig stop at the end of the stream and
    never look at it).  The device route finds the Adler-32 trailer where the final block ended (the inflate kernel's
    consumed-byte count), not at the end of the leaf: such files stay on the device decoder -- round 4 rejected every
    section of them and fell back to the host decoder for the rest of the run (the advisor's finding)."""
    paths = _write_set(tmp_path, 4, seed=23, block=53, pad=5)
    monkeypatch.setenv("WTAMD_BW_BATCH_SECTIONS", "80")
    monkeypatch.setenv("WTAMD_MIN_SPAN", "1500")
    for op, name in (("MeanReduction", "mean"), ("MaxReduction", "max")):
        wi, keep = _reduce(L, paths, op)
        got = _blocks(L, wi)
        st = _stats(L, wi)
        _same(got, _expected(oracle, paths, name))
        assert st.bw_sections > 0
        assert "continuing with the host decoder" not in capfd.readouterr().err


def test_padded_index_leaves_stay_on_device_emu(emu_lib, oracle, tmp_path, monkeypatch, capfd):
    _padded_leaves_case(emu_lib, oracle, tmp_path, monkeypatch, capfd)


@pytest.mark.gpu
def test_padded_index_leaves_stay_on_device_gpu(amd_lib, oracle, tmp_path, monkeypatch, capfd):
    _padded_leaves_case(amd_lib, oracle, tmp_path, monkeypatch, capfd)


def test_adler32_mismatch_is_an_inflate_error_emu(emu_lib, tmp_path):
    """A payload that still inflates but fails the zlib stream's Adler-32 trailer is rejected by the device route (the
    count kernel's check) and then by the host decoder's zlib, as libBigWig's uncompress() would: exit(1), no result."""
    import subprocess, sys, struct
    recs = [(20 * k, 20 * k + 7, 1.0) for k in range(3000)]
    p = str(tmp_path / "adler.bw")
    write_bigwig(p, {"chr1": 80000}, {"chr1": recs}, items_per_block=256)
    raw = bytearray(open(p, "rb").read())
    idx = struct.unpack_from("<Q", raw, 24)[0]
    off, size = struct.unpack_from("<QQ", raw, idx + 48 + 4 + 32 * 3 + 16)     # fourth section: offset, size
    raw[off + size - 1] ^= 0x01                                               # last byte of its Adler-32
    open(p, "wb").write(bytes(raw))
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from emu.build import build_dropin; L = C.CDLL(build_dropin());"
            "L.wtamd_BigWiggleReader.restype = C.c_void_p; L.wtamd_BigWiggleReader.argtypes = [C.c_char_p, C.c_int];"
            "L.newMultiplexer.restype = C.c_void_p; L.newMultiplexer.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char];"
            "L.MeanReduction.restype = C.c_void_p; L.MeanReduction.argtypes = [C.c_void_p]; L.wtamd_drain.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p];"
            "its = (C.c_void_p * 1)(L.wtamd_BigWiggleReader(%r, 1)); m = L.newMultiplexer(its, 1, b'\\0'); r = L.MeanReduction(m);"
            "a = C.c_int64(); b = C.c_double(); L.wtamd_drain(r, C.byref(a), C.byref(b)); print('survived')"
            % (os.path.dirname(os.path.abspath(__file__)), p.encode()))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 1 and "survived" not in r.stdout, (r.returncode, r.stdout, r.stderr[-500:])
