"""BigWig files -> RunLists through the library's section decoder (csrc/wt_bigwig.cpp).

Mirrors what the reference's BigWiggleReader hands to a Multiplexer (reference
src/bigWiggleReader.c): 1-based runs, chromosomes in strcmp order, runs cut at the reader's
10 000-bp stretch edges (``box=True``, the reference behaviour) -- but whole chromosomes at a
time instead of one ``pop()`` per interval.
"""
import ctypes as C

import numpy as np

from . import _lib
from .runlists import RunLists


class BigWig:
    def __init__(self, path):
        self._h = C.c_void_p()
        if _lib.lib().wtamd_bw_open(path.encode(), C.byref(self._h)) != 0:
            raise ValueError("File %s is not in BigWig format" % path)
        L = _lib.lib()
        n = L.wtamd_bw_n_chrom(self._h)
        self.chroms = {L.wtamd_bw_chrom_name(self._h, i).decode(): int(L.wtamd_bw_chrom_length(self._h, i))
                       for i in range(n)}

    def read(self, chrom, box=True):
        """(start, finish, value) numpy arrays of one chromosome."""
        L = _lib.lib()
        cap = 1 << 16
        while True:
            s, f, v = np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float32)
            n = L.wtamd_bw_read_chrom(self._h, chrom.encode(), int(box), cap, s.ctypes.data, f.ctypes.data,
                                      v.ctypes.data)
            if n < 0:
                raise IOError("BigWig decode failed")
            if n <= cap:
                return s[:n].copy(), f[:n].copy(), v[:n].copy()
            cap = int(n)

    def close(self):
        if self._h:
            _lib.lib().wtamd_bw_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_runlists(paths, defaults=None, box=True):
    """One track per BigWig file; chromosomes = union of the files' chromosomes in strcmp order."""
    files = [BigWig(p) for p in paths]
    names = sorted({c for f in files for c in f.chroms}, key=lambda s: s.encode())
    seg_off = [0]
    S, F, V = [], [], []
    for c in names:
        for f in files:
            if c in f.chroms:
                s, e, v = f.read(c, box)
            else:
                s, e, v = np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32)
            S.append(s); F.append(e); V.append(v)
            seg_off.append(seg_off[-1] + len(s))
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    return RunLists(len(names), len(files), seg_off, cat(S, np.int32), cat(F, np.int32), cat(V, np.float32),
                    defaults, names)
