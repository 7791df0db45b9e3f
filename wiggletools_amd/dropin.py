"""Python mirror of the DROP-IN layer of the C ABI (include/wiggletools_amd.h): the reference's own
constructors (newMultiplexer, MeanReduction, ..., reference src/wiggletools.h:80-103) plus the
library's bulk doors (wtamd_ArrayReader, wtamd_iterator_next_block, wtamd_drain).  Used by
bench.py's end-to-end leg and by tests; everything here is a thin ctypes call.
"""
import ctypes as C

import numpy as np

from . import _lib

REDUCERS = {"sum": "SumReduction", "product": "ProductReduction", "mult": "ProductReduction", "mean": "MeanReduction",
            "var": "VarianceReduction", "stddev": "StdDevReduction", "entropy": "EntropyReduction", "cv": "CVReduction",
            "CV": "CVReduction", "median": "MedianReduction", "min": "MinReduction", "max": "MaxReduction"}
SET_REDUCERS = {"ttest": "TTestReduction", "wilcoxon": "MWUReduction", "mwu": "MWUReduction"}


def _bind():
    L = _lib.lib()
    if getattr(L, "_wt_dropin_bound", False):
        return L
    L.wtamd_host_alloc.restype = C.c_void_p
    L.wtamd_host_alloc.argtypes = [C.c_size_t]
    L.wtamd_host_free.argtypes = [C.c_void_p]
    L.wtamd_host_free.restype = None
    L.wtamd_ArrayReader.restype = C.c_void_p
    L.wtamd_ArrayReader.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
    L.wtamd_BigWiggleReader.restype = C.c_void_p
    L.wtamd_BigWiggleReader.argtypes = [C.c_char_p, C.c_int]
    L.newMultiplexer.restype = C.c_void_p
    L.newMultiplexer.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char]
    L.newMultiset.restype = C.c_void_p
    L.newMultiset.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    for name in set(REDUCERS.values()) | set(SET_REDUCERS.values()):
        f = getattr(L, name)
        f.restype = C.c_void_p
        f.argtypes = [C.c_void_p]
    L.wtamd_iterator_next_block.restype = C.c_int64
    L.wtamd_iterator_next_block.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_void_p)]
    L.wtamd_drain.restype = C.c_int64
    L.wtamd_drain.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    L.seek.restype = None
    L.seek.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    L._wt_dropin_bound = True
    return L


class PinnedArray:
    """numpy view of page-locked host memory (wtamd_host_alloc)."""

    def __init__(self, n, dtype):
        L = _bind()
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        self.ptr = L.wtamd_host_alloc(max(self.n, 1) * self.dtype.itemsize)
        if not self.ptr:
            raise MemoryError("wtamd_host_alloc(%d bytes)" % (self.n * self.dtype.itemsize))
        ct = np.ctypeslib.as_ctypes_type(self.dtype)
        self.array = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(ct)), shape=(max(self.n, 1),))[:self.n]

    def free(self):
        if self.ptr:
            _bind().wtamd_host_free(self.ptr)
            self.ptr = None


def array_reader(chrom_names, seg_off, start_ptr, finish_ptr, value_ptr, default_value=0.0):
    """wtamd_ArrayReader over raw pointers (int addresses); seg_off: n_chrom + 1 offsets."""
    L = _bind()
    names = (C.c_char_p * len(chrom_names))(*[n.encode() for n in chrom_names])
    so = np.ascontiguousarray(seg_off, np.int64)
    return L.wtamd_ArrayReader(len(chrom_names), names, so.ctypes.data, start_ptr, finish_ptr, value_ptr, float(default_value))


def buffered_array_reader(chrom_names, seg_off, start_ptr, finish_ptr, value_ptr, default_value=0.0):
    """wtamd_BufferedArrayReader: the same arrays behind the reference's buffered-reader protocol (a producer thread
    pushing one interval at a time, csrc/wt_bufreader.h)."""
    L = _bind()
    L.wtamd_BufferedArrayReader.restype = C.c_void_p
    L.wtamd_BufferedArrayReader.argtypes = L.wtamd_ArrayReader.argtypes
    names = (C.c_char_p * len(chrom_names))(*[n.encode() for n in chrom_names])
    so = np.ascontiguousarray(seg_off, np.int64)
    return L.wtamd_BufferedArrayReader(len(chrom_names), names, so.ctypes.data, start_ptr, finish_ptr, value_ptr, float(default_value))


def bigwig_reader(path, box=True):
    """wtamd_BigWiggleReader: the reference's BigWiggleReader role (bigWiggleReader.c:147-151), bulk-capable."""
    return _bind().wtamd_BigWiggleReader(path.encode(), int(box))


def bigwig_readers(paths, box=True):
    """wtamd_BigWiggleReaders: n files opened side by side."""
    L = _bind()
    L.wtamd_BigWiggleReaders.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_void_p)]
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    out = (C.c_void_p * len(paths))()
    if L.wtamd_BigWiggleReaders(len(paths), arr, int(box), out) != 0:
        raise _lib.WtamdError("wtamd_BigWiggleReaders failed")
    return list(out)


def multiplexer(iters, strict=False):
    L = _bind()
    arr = (C.c_void_p * len(iters))(*iters)
    return L.newMultiplexer(arr, len(iters), b"\x01" if strict else b"\x00")


def reducer(op, iters, n_set0=0, strict=False):
    """The reducer iterator the reference's parser would build for `op` over the child iterators."""
    L = _bind()
    if op in SET_REDUCERS:
        ms = (C.c_void_p * 2)(multiplexer(iters[:n_set0], strict), multiplexer(iters[n_set0:], strict))
        keep = ms                       # newMultiset keeps the caller's array (multiSet.c:118)
        r = getattr(L, SET_REDUCERS[op])(L.newMultiset(ms, 2))
        _KEEP.append(keep)
        return r
    return getattr(L, REDUCERS[op])(multiplexer(iters, strict))


_KEEP = []


def seek(wi, chrom, start, finish):
    """The reference's seek() (wiggleIterator.c:67-70): restricts an iterator to chrom:[start, finish)."""
    _bind().seek(wi, chrom.encode(), int(start), int(finish))


def drain_blocks(wi, on_block=None):
    """Consumes a reducer through the block door; returns (runs, covered bp)."""
    L = _bind()
    chrom, s, f, v = C.c_char_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    runs = bp = 0
    while True:
        n = L.wtamd_iterator_next_block(wi, C.byref(chrom), C.byref(s), C.byref(f), C.byref(v))
        if n < 0:
            raise _lib.WtamdError("wtamd_iterator_next_block: not a reducer of this library")
        if n == 0:
            return runs, bp
        runs += n
        if on_block is not None:
            i32 = C.POINTER(C.c_int32)
            sa = np.ctypeslib.as_array(C.cast(s, i32), shape=(n,))
            fa = np.ctypeslib.as_array(C.cast(f, i32), shape=(n,))
            va = np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_double)), shape=(n,))
            bp += on_block(chrom.value, sa, fa, va)


def pipe_stats(wi):
    from .pipe import PipeStats
    L = _bind()
    st = PipeStats()
    L.wtamd_iterator_pipe_stats.argtypes = [C.c_void_p, C.POINTER(PipeStats)]
    if L.wtamd_iterator_pipe_stats(wi, C.byref(st)) != 0:
        return {}
    return {k: getattr(st, k) for k, _ in PipeStats._fields_}


def drain_pops(wi):
    """Consumes an iterator one pop at a time (in C); returns (runs, covered bp, value sum)."""
    L = _bind()
    bp, acc = C.c_int64(), C.c_double()
    n = L.wtamd_drain(wi, C.byref(bp), C.byref(acc))
    return n, bp.value, acc.value
