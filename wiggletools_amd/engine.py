"""Python mirror of the bulk C ABI (include/wiggletools_amd.h, wtamd_*).

Operator names follow the reference CLI grammar (reference commandParser.c:46-61,
763-826): sum, mult(product), mean, var, stddev, entropy, CV, min, max, median,
ttest, wilcoxon(mwu).  PyTorch is used only to own device memory / streams.
"""
import ctypes as C

import numpy as np

from . import _lib
from .runlists import RunLists

OPS = {"sum": 0, "product": 1, "mult": 1, "mean": 2, "var": 3, "stddev": 4, "entropy": 5, "cv": 6, "CV": 6,
       "min": 7, "max": 8, "median": 9, "ttest": 10, "mwu": 11, "wilcoxon": 11}
STRICT_SET0, STRICT_SET1 = 1, 2


def opcode(op):
    if isinstance(op, str):
        if op not in OPS:
            raise ValueError("unknown reducer %r" % op)
        return OPS[op]
    return int(op)


def reducer_default(op, defaults):
    d = np.ascontiguousarray(defaults, np.float64)
    return _lib.lib().wtamd_reducer_default(opcode(op), len(d), d.ctypes.data)


MAP_OPS = {"scale": 0, "offset": 1, "ln": 2, "log": 3, "exp": 4, "expb": 5, "pow": 6, "abs": 7,
           "gt": 8, "gte": 9, "lt": 10, "lte": 11}


def map_default(op, param, default_value):
    return _lib.lib().wtamd_map_default(MAP_OPS[op], float(param), float(default_value))


def map_runlists(rl: RunLists, op, param=0.0):
    """The reference's `map`-able unary operator `op` over every track of `rl`, on device
    (wtamd_runs_map): returns a new RunLists with f64 values and transformed defaults."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    n = int(rl.seg_off[-1])
    n_seg = rl.n_chrom * rl.n_tracks
    seg = np.ascontiguousarray(rl.seg_off, np.int64)
    s = torch.from_numpy(np.ascontiguousarray(rl.start, np.int32)).to(dev)
    f = torch.from_numpy(np.ascontiguousarray(rl.finish, np.int32)).to(dev)
    v = torch.from_numpy(np.ascontiguousarray(rl.value)).to(dev)
    os_ = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    of = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    ov = torch.empty(max(n, 1), dtype=torch.float64, device=dev)
    oseg = np.zeros(n_seg + 1, np.int64)
    _lib.check(_lib.lib().wtamd_runs_map(MAP_OPS[op], float(param), n_seg, seg.ctypes.data, s.data_ptr(), f.data_ptr(),
                                         v.data_ptr(), 1 if rl.value.dtype == np.float64 else 0, os_.data_ptr(),
                                         of.data_ptr(), ov.data_ptr(), oseg.ctypes.data, None))
    m = int(oseg[-1])
    d = np.array([map_default(op, param, x) for x in rl.defaults], np.float64)
    return RunLists(rl.n_chrom, rl.n_tracks, oseg, os_[:m].cpu().numpy(), of[:m].cpu().numpy(), ov[:m].cpu().numpy(), d)


class TrackSet:
    """N tracks resident in HBM (wtamd_trackset)."""

    def __init__(self, handle, n_chrom, n_tracks, keep=None):
        self._h = handle
        self.n_chrom = n_chrom
        self.n_tracks = n_tracks
        self._keep = keep

    @staticmethod
    def _ranges(ranges, n_chrom):
        """ranges: None or per-chromosome (lo, hi) run-start bounds -> two int32 arrays (or None, None)."""
        if ranges is None:
            return None, None
        assert len(ranges) == n_chrom
        lo = np.ascontiguousarray([r[0] for r in ranges], np.int32)
        hi = np.ascontiguousarray([r[1] for r in ranges], np.int32)
        return lo, hi

    @classmethod
    def from_runlists(cls, rl: RunLists, ranges=None):
        L = _lib.lib()
        value = np.ascontiguousarray(rl.value)
        rlo, rhi = cls._ranges(ranges, rl.n_chrom)
        t = _lib.Tracks(rl.n_chrom, rl.n_tracks, rl.seg_off.ctypes.data, rl.start.ctypes.data,
                        rl.finish.ctypes.data, value.ctypes.data, int(value.dtype == np.float64),
                        rl.defaults.ctypes.data, rlo.ctypes.data if rlo is not None else None,
                        rhi.ctypes.data if rhi is not None else None)
        h = C.c_void_p()
        _lib.check(L.wtamd_trackset_create_host(C.byref(t), C.byref(h)))
        return cls(h, rl.n_chrom, rl.n_tracks)

    @classmethod
    def from_device(cls, n_chrom, n_tracks, seg_off, start, finish, value, defaults, ranges=None):
        """start/finish/value: torch CUDA tensors (int32,int32,float32|float64), kept alive by this object.
        seg_off / defaults: host numpy arrays."""
        import torch
        L = _lib.lib()
        assert start.is_cuda and finish.is_cuda and value.is_cuda
        assert start.dtype == torch.int32 and finish.dtype == torch.int32
        assert value.dtype in (torch.float32, torch.float64)
        seg_off = np.ascontiguousarray(seg_off, np.int64)
        defaults = np.ascontiguousarray(defaults, np.float64)
        rlo, rhi = cls._ranges(ranges, n_chrom)
        t = _lib.Tracks(n_chrom, n_tracks, seg_off.ctypes.data, start.data_ptr(), finish.data_ptr(),
                        value.data_ptr(), int(value.dtype == torch.float64), defaults.ctypes.data,
                        rlo.ctypes.data if rlo is not None else None, rhi.ctypes.data if rhi is not None else None)
        h = C.c_void_p()
        _lib.check(L.wtamd_trackset_create_device(C.byref(t), C.byref(h)))
        return cls(h, n_chrom, n_tracks, keep=(start, finish, value))

    def close(self):
        if self._h is not None:
            _lib.lib().wtamd_trackset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def max_runs(self):
        return int(_lib.lib().wtamd_trackset_max_runs(self._h))

    def index(self, op="mean", stream=None):
        _lib.check(_lib.lib().wtamd_trackset_index(self._h, opcode(op), stream))

    def stats(self):
        s = _lib.Stats()
        _lib.check(_lib.lib().wtamd_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in _lib.Stats._fields_}

    def validate(self):
        """(number of runs violating the input contract, global index of the first or -1)."""
        n, first = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().wtamd_trackset_validate(self._h, C.byref(n), C.byref(first)))
        return n.value, first.value

    def pearson(self):
        """Pearson correlation of the set's two tracks (reference `pearson a b`), computed on device."""
        out = C.c_double()
        _lib.check(_lib.lib().wtamd_pearson(self._h, C.byref(out)))
        return out.value

    def pearson_moments(self):
        """{n, sum_X, sum_Y, T_XX, T_XY, T_YY} of this track set's part of the genome (6 doubles)."""
        m = np.zeros(6, np.float64)
        _lib.check(_lib.lib().wtamd_pearson_moments(self._h, m.ctypes.data))
        return m

    # ---- host-output convenience (tests, drop-in layer) ----
    def reduce_host(self, op, flags=0, n_set0=0):
        """Returns (chrom, start, finish, value) numpy arrays."""
        cap = max(self.max_runs(), 1)
        s, f, v = np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float64)
        cro = np.zeros(self.n_chrom + 1, np.int64)
        runs = _lib.Runs(cap, s.ctypes.data, f.ctypes.data, v.ctypes.data, cro.ctypes.data)
        desc = _lib.ReduceDesc(opcode(op), flags, n_set0, 0)
        n = C.c_int64()
        _lib.check(_lib.lib().wtamd_reduce_host(self._h, C.byref(desc), C.byref(runs), C.byref(n)))
        n = n.value
        chrom = np.repeat(np.arange(self.n_chrom, dtype=np.int32), np.diff(cro))
        assert len(chrom) == n
        return chrom, s[:n].copy(), f[:n].copy(), v[:n].copy()

    def multiplex_host(self, flags=0):
        """Returns (chrom, start, finish, values[R,N], inplay[R,N])."""
        cap = max(self.max_runs(), 1)
        N = self.n_tracks
        s, f = np.empty(cap, np.int32), np.empty(cap, np.int32)
        cnt = np.empty(cap, np.float64)      # per-run inplay_count (multiplexer.h:27)
        cro = np.zeros(self.n_chrom + 1, np.int64)
        tile = np.empty((cap, N), np.float64)
        ip = np.empty((cap, N), np.uint8)
        runs = _lib.Runs(cap, s.ctypes.data, f.ctypes.data, cnt.ctypes.data, cro.ctypes.data)
        n = C.c_int64()
        _lib.check(_lib.lib().wtamd_multiplex_host(self._h, flags, C.byref(runs), tile.ctypes.data, ip.ctypes.data,
                                                   C.byref(n)))
        n = n.value
        chrom = np.repeat(np.arange(self.n_chrom, dtype=np.int32), np.diff(cro))
        return chrom, s[:n].copy(), f[:n].copy(), tile[:n].copy(), ip[:n].copy()

    # ---- device-resident path (bench, pipelines) ----
    def alloc_runs(self, capacity=None):
        import torch
        cap = max(int(capacity if capacity is not None else self.max_runs()), 1)
        dev = torch.device("cuda", torch.cuda.current_device())
        return DeviceRuns(torch.empty(cap, dtype=torch.int32, device=dev),
                          torch.empty(cap, dtype=torch.int32, device=dev),
                          torch.empty(cap, dtype=torch.float64, device=dev),
                          torch.zeros(self.n_chrom + 1, dtype=torch.int64, device=dev))

    def reduce(self, op, out, flags=0, n_set0=0, stream=None, sync=True):
        """Multiplex+reduce on device into `out` (DeviceRuns). Returns run count when sync."""
        desc = _lib.ReduceDesc(opcode(op), flags, n_set0, 0)
        runs = out.as_struct()
        n = C.c_int64(-1)
        _lib.check(_lib.lib().wtamd_reduce(self._h, C.byref(desc), C.byref(runs),
                                           C.byref(n) if sync else None, stream))
        if sync:
            out.n = n.value
            return n.value
        return None


class DeviceRuns:
    def __init__(self, start, finish, value, chrom_run_off):
        self.start, self.finish, self.value, self.chrom_run_off = start, finish, value, chrom_run_off
        self.n = 0

    def as_struct(self):
        return _lib.Runs(self.start.numel(), self.start.data_ptr(), self.finish.data_ptr(),
                         self.value.data_ptr(), self.chrom_run_off.data_ptr())

    def auc(self, n=None, stream=None):
        r = self.as_struct()
        out = C.c_double()
        _lib.check(_lib.lib().wtamd_runs_auc(C.byref(r), int(self.n if n is None else n), C.byref(out), stream))
        return out.value

    def mean(self, n=None, stream=None):
        """meanI of the run list (reference MeanIntegrator): length-weighted mean of the non-NaN runs."""
        r = self.as_struct()
        out = C.c_double()
        _lib.check(_lib.lib().wtamd_runs_mean(C.byref(r), int(self.n if n is None else n), C.byref(out), stream))
        return out.value

    def compress(self, out=None, n=None, stream=None):
        """Merged run list (reference CompressionWiggleIterator) as a new DeviceRuns."""
        import torch
        n = int(self.n if n is None else n)
        if out is None:
            cap = max(n, 1)
            dev = self.start.device
            out = DeviceRuns(torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev),
                             torch.empty(cap, dtype=torch.float64, device=dev), torch.zeros_like(self.chrom_run_off))
        a, b = self.as_struct(), out.as_struct()
        m = C.c_int64()
        _lib.check(_lib.lib().wtamd_runs_compress(C.byref(a), n, self.chrom_run_off.numel() - 1, C.byref(b), C.byref(m),
                                                  stream))
        out.n = m.value
        return out

    def to_host(self):
        n = self.n
        cro = self.chrom_run_off.cpu().numpy()
        chrom = np.repeat(np.arange(len(cro) - 1, dtype=np.int32), np.diff(cro))
        return chrom, self.start[:n].cpu().numpy(), self.finish[:n].cpu().numpy(), self.value[:n].cpu().numpy()
