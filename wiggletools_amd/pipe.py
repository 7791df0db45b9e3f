"""Python mirror of the streaming pipeline of the C ABI (include/wiggletools_amd.h, wtamd_pipe_*).

A pipe owns a few batch slots with pinned host staging, device buffers and pinned output; batches
are shipped asynchronously (H2D, window index + multiplex/reduce kernels, D2H on three HIP
streams) -- see csrc/wt_pipe.h.  This module only moves pointers: numpy views of the pinned
arrays are filled by the caller.  It is what the drop-in layer's Feeder (csrc/wt_iter_abi.cpp)
does in C++ for lazy WiggleIterators.
"""
import ctypes as C

import numpy as np

from . import _lib
from .engine import opcode

OP_MULTIPLEX = 12


class PipeConfig(C.Structure):
    _fields_ = [("n_tracks", C.c_int32), ("n_slots", C.c_int32), ("defaults", C.c_void_p),
                ("desc", _lib.ReduceDesc), ("max_intervals", C.c_int64), ("max_runs", C.c_int64),
                ("flags", C.c_uint32), ("reserved", C.c_int32)]


class PipeBatch(C.Structure):
    _fields_ = [("capacity", C.c_int64), ("seg_off", C.c_void_p), ("start", C.c_void_p), ("finish", C.c_void_p),
                ("value32", C.c_void_p), ("value64", C.c_void_p)]


class PipeResult(C.Structure):
    _fields_ = [("n_runs", C.c_int64), ("start", C.c_void_p), ("finish", C.c_void_p), ("value", C.c_void_p),
                ("tile", C.c_void_p), ("inplay", C.c_void_p), ("covered_bp", C.c_int64), ("n_intervals", C.c_int64),
                ("integ_valid", C.c_int32), ("reserved", C.c_int32), ("integ", C.c_double * 6)]


class PipeStats(C.Structure):
    _fields_ = [("batches", C.c_int64), ("intervals", C.c_int64), ("runs", C.c_int64), ("covered_bp", C.c_int64),
                ("h2d_bytes", C.c_int64), ("d2h_bytes", C.c_int64), ("kernel_ms", C.c_double), ("h2d_ms", C.c_double),
                ("d2h_ms", C.c_double), ("delta_batches", C.c_int32), ("n_slots", C.c_int32),
                ("host_submit_ms", C.c_double), ("host_wait_ms", C.c_double),
                ("bw_sections", C.c_int64), ("bw_decode_ms", C.c_double)]


def _view(ptr, n, dtype):
    if not ptr or n <= 0:
        return np.zeros(0, dtype)
    ct = np.ctypeslib.as_ctypes_type(dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(int(n),))


def _bind(L):
    if getattr(L, "_wt_pipe_bound", False):
        return L
    L.wtamd_pipe_create.argtypes = [C.POINTER(PipeConfig), C.POINTER(C.c_void_p)]
    L.wtamd_pipe_destroy.argtypes = [C.c_void_p]
    L.wtamd_pipe_destroy.restype = None
    L.wtamd_pipe_acquire.argtypes = [C.c_void_p, C.POINTER(PipeBatch)]
    L.wtamd_pipe_grow.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.POINTER(PipeBatch)]
    L.wtamd_pipe_submit.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32]
    L.wtamd_pipe_cancel.argtypes = [C.c_void_p]
    L.wtamd_pipe_collect.argtypes = [C.c_void_p, C.POINTER(PipeResult)]
    L.wtamd_pipe_release.argtypes = [C.c_void_p]
    L.wtamd_pipe_in_flight.argtypes = [C.c_void_p]
    L.wtamd_pipe_get_stats.argtypes = [C.c_void_p, C.POINTER(PipeStats)]
    L.wtamd_last_error.restype = C.c_char_p
    L._wt_pipe_bound = True
    return L


class Pipe:
    def __init__(self, n_tracks, op, defaults=None, flags=0, n_set0=0, max_intervals=1 << 16, max_runs=1 << 21,
                 n_slots=3, lib=None, compress=False):
        self.L = _bind(lib if lib is not None else _lib.lib())
        self.n_tracks = int(n_tracks)
        self.op = OP_MULTIPLEX if op == "multiplex" else opcode(op)
        self._defaults = np.ascontiguousarray(np.zeros(n_tracks) if defaults is None else defaults, np.float64)
        cfg = PipeConfig(self.n_tracks, n_slots, self._defaults.ctypes.data,
                         _lib.ReduceDesc(self.op, flags, n_set0, 0), max_intervals, max_runs, 1 if compress else 0, 0)
        h = C.c_void_p()
        self._check(self.L.wtamd_pipe_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._b = None

    def _check(self, rc):
        if rc != 0:
            raise _lib.WtamdError("wtamd error %d: %s" % (rc, self.L.wtamd_last_error().decode()))

    def close(self):
        if getattr(self, "_h", None) is not None:
            self.L.wtamd_pipe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- producer side ----
    def acquire(self):
        b = PipeBatch()
        self._check(self.L.wtamd_pipe_acquire(self._h, C.byref(b)))
        self._b = b
        return b.capacity

    def grow(self, used, min_capacity, want_f64=False):
        self._check(self.L.wtamd_pipe_grow(self._h, used, min_capacity, int(want_f64), C.byref(self._b)))
        return self._b.capacity

    def staging(self):
        """numpy views (seg_off, start, finish, value32, value64 or None) of the acquired slot."""
        b = self._b
        return (_view(b.seg_off, self.n_tracks + 1, np.int64), _view(b.start, b.capacity, np.int32),
                _view(b.finish, b.capacity, np.int32), _view(b.value32, b.capacity, np.float32),
                _view(b.value64, b.capacity, np.float64) if b.value64 else None)

    def submit(self, lo, hi, f64=False):
        self._check(self.L.wtamd_pipe_submit(self._h, int(f64), int(lo), int(hi)))
        self._b = None

    def cancel(self):
        self._check(self.L.wtamd_pipe_cancel(self._h))
        self._b = None

    # ---- consumer side ----
    def in_flight(self):
        return self.L.wtamd_pipe_in_flight(self._h)

    def collect(self, copy=True):
        """(start, finish, value[, tile, inplay]) of the oldest batch; views into pinned memory unless copy."""
        r = PipeResult()
        self._check(self.L.wtamd_pipe_collect(self._h, C.byref(r)))
        n = r.n_runs
        out = [_view(r.start, n, np.int32), _view(r.finish, n, np.int32), _view(r.value, n, np.float64)]
        if self.op == OP_MULTIPLEX:
            out += [_view(r.tile, n * self.n_tracks, np.float64).reshape(n, self.n_tracks),
                    _view(r.inplay, n * self.n_tracks, np.uint8).reshape(n, self.n_tracks)]
        if copy:
            out = [a.copy() for a in out]
        self.last_covered_bp = r.covered_bp
        return tuple(out)

    def release(self):
        self._check(self.L.wtamd_pipe_release(self._h))

    def stats(self):
        s = PipeStats()
        self._check(self.L.wtamd_pipe_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in PipeStats._fields_}


def stream_runlists(rl, op, batch_bp, flags=0, n_set0=0, depth=2, lib=None, max_runs=None, max_intervals=64, compress=False):
    """Feeds a RunLists through a Pipe, chromosome by chromosome, `batch_bp` run starts per batch,
    `depth` batches in flight; returns (chrom, start, finish, value[, tile, inplay]) concatenated.
    The cuts follow the drop-in layer's rules: a batch holds every interval overlapping or
    touching [lo, hi] plus, per track, the first interval beyond hi."""
    N = rl.n_tracks
    p = Pipe(N, op, rl.defaults, flags, n_set0, max_intervals=max_intervals,
             max_runs=max_runs if max_runs is not None else max(batch_bp, 1), n_slots=depth + 1, lib=lib, compress=compress)
    f64 = rl.value.dtype == np.float64
    pieces, chroms, pending = [], [], []

    def drain_one():
        c = pending.pop(0)
        pieces.append(p.collect(copy=True))
        chroms.append(np.full(len(pieces[-1][0]), c, np.int32))
        p.release()

    for c in range(rl.n_chrom):
        segs = [(int(rl.seg_off[c * N + i]), int(rl.seg_off[c * N + i + 1])) for i in range(N)]
        nonempty = [s for s in segs if s[1] > s[0]]
        if not nonempty:
            continue
        lo = min(int(rl.start[a]) for a, b in nonempty)
        end = max(int(rl.finish[b - 1]) for a, b in nonempty)
        while lo < end:
            hi = lo + batch_bp
            sel = []
            for (a, b) in segs:
                s, f = rl.start[a:b], rl.finish[a:b]
                first = int(np.searchsorted(f, lo, side="left"))            # finish >= lo (touching the cut counts)
                last = int(np.searchsorted(s, hi, side="left"))             # start < hi ...
                if last < len(s) and not (last > first and f[last - 1] >= hi):
                    last += 1                                               # ... plus the sentinel beyond
                sel.append((a + first, a + max(last, first)))
            n = sum(y - x for x, y in sel)
            cap = p.acquire()
            if cap < n or f64:
                p.grow(0, max(n, cap), want_f64=f64)
            so, ss, sf, v32, v64 = p.staging()
            k = 0
            for i, (x, y) in enumerate(sel):
                so[i] = k
                ss[k:k + y - x] = rl.start[x:y]
                sf[k:k + y - x] = rl.finish[x:y]
                (v64 if f64 else v32)[k:k + y - x] = rl.value[x:y]
                k += y - x
            so[N] = k
            while p.in_flight() >= depth:
                drain_one()
            p.submit(lo, hi, f64)
            pending.append(c)
            lo = hi
    while pending:
        drain_one()
    st = p.stats()
    p.close()
    ncol = 5 if op == "multiplex" else 3
    if not pieces:
        return (np.zeros(0, np.int32),) * 3 + (np.zeros(0),), st
    cat = [np.concatenate(chroms)] + [np.concatenate([q[j] for q in pieces]) for j in range(ncol)]
    return tuple(cat), st
