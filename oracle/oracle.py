"""ctypes front-end of the checker libraries (TEST INFRASTRUCTURE ONLY).

* ``libwt_oracle.so``   -- our plain-C restatement (oracle/wt_oracle.c)
* ``libref_harness.so`` -- driver of the compiled reference (oracle/_ref/...)

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this module.  The product package (wiggletools_amd/) never does.

Tracks are passed as a plain dict (see wiggletools_amd.runlists.RunLists.as_dict):
    n_chrom, n_tracks, seg_off[int64], start[int32], finish[int32],
    value[float64], defaults[float64]
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

OPS = {"sum": 0, "product": 1, "mean": 2, "var": 3, "stddev": 4, "entropy": 5, "cv": 6,
       "min": 7, "max": 8, "median": 9, "ttest": 10, "mwu": 11}
STRICT_SET0, STRICT_SET1 = 1, 2


class _Tracks(C.Structure):
    _fields_ = [("n_chrom", C.c_int32), ("n_tracks", C.c_int32),
                ("seg_off", C.c_void_p), ("start", C.c_void_p), ("finish", C.c_void_p),
                ("value", C.c_void_p), ("defaults", C.c_void_p)]


def _locked_make(target):
    """One make at a time per checkout (pytest-xdist workers call this side by side); the Makefile builds every library
    to a temporary name and rename()s it, so a process that has the old file dlopen'ed keeps a whole file."""
    import fcntl
    with open(os.path.join(_HERE, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-s", "-C", _HERE, target])


def _newer(out, deps):
    return os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps if os.path.exists(d))


def build(force=False):
    """(Re)build the checker libraries with oracle/Makefile."""
    hdr = os.path.join(_HERE, "..", "include", "wiggletools_amd.h")
    # (the same dependencies as the Makefile's rules: a file make would not rebuild must not count as stale here)
    need = force or not (_newer(os.path.join(_HERE, "libwt_oracle.so"), [os.path.join(_HERE, "wt_oracle.c")])
                         and _newer(os.path.join(_HERE, "libref_harness.so"), [os.path.join(_HERE, "ref_harness.c"), hdr]))
    ref_missing = not os.path.exists(os.path.join(_HERE, "_ref", "libwiggletools_ref.so"))
    if need or (ref_missing and os.path.isdir("/root/reference/src")):
        _locked_make("all")


def build_mixed(force=False):
    """The mixed-link libraries (reference translation units + a drop-in library); see oracle/Makefile.  Skipped when
    both are newer than the drop-in libraries they link against."""
    if not os.path.isdir("/root/reference/src"):
        return
    mk = os.path.join(_HERE, "Makefile")
    pairs = (("amd", os.path.join(_HERE, "..", "wiggletools_amd", "csrc", "libwiggletools_amd.so")),
             ("emu", os.path.join(_HERE, "..", "tests", "emu", "libwt_dropin_emu.so")))
    if force or not all(_newer(os.path.join(_HERE, "_ref", "libwiggletools_mixed_%s.so" % v), [lib, mk])
                        for v, lib in pairs if os.path.exists(lib)):
        _locked_make("mixed")


def mixed_path(variant):
    p = os.path.join(_HERE, "_ref", "libwiggletools_mixed_%s.so" % variant)
    return p if os.path.exists(p) else None


_oracle = None
_ref = None


def _pack(t):
    keep = {
        "seg_off": np.ascontiguousarray(t["seg_off"], dtype=np.int64),
        "start": np.ascontiguousarray(t["start"], dtype=np.int32),
        "finish": np.ascontiguousarray(t["finish"], dtype=np.int32),
        "value": np.ascontiguousarray(t["value"], dtype=np.float64),
        "defaults": np.ascontiguousarray(t["defaults"], dtype=np.float64),
    }
    s = _Tracks(int(t["n_chrom"]), int(t["n_tracks"]),
                keep["seg_off"].ctypes.data, keep["start"].ctypes.data, keep["finish"].ctypes.data,
                keep["value"].ctypes.data, keep["defaults"].ctypes.data)
    return s, keep


def oracle_lib():
    global _oracle
    if _oracle is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "libwt_oracle.so"))
        L.wto_reduce.restype = C.c_int64
        L.wto_reduce.argtypes = [C.POINTER(_Tracks), C.c_int, C.c_uint, C.c_int, C.c_int64,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wto_multiplex.restype = C.c_int64
        L.wto_multiplex.argtypes = [C.POINTER(_Tracks), C.c_uint, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wto_reducer_default.restype = C.c_double
        L.wto_reducer_default.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.wto_auc.restype = C.c_double
        L.wto_auc.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wto_compress.restype = C.c_int64
        L.wto_compress.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wto_tdist_Q.restype = C.c_double
        L.wto_tdist_Q.argtypes = [C.c_double, C.c_double]
        L.wto_ttest_stat.restype = None
        L.wto_ttest_stat.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _oracle = L
    return _oracle


class Harness:
    """A private copy of libref_harness.so bound to ONE library that exports the reference's
    C API (the compiled reference, or wiggletools_amd's drop-in library)."""

    def __init__(self, lib_path, tag):
        import shutil
        build()
        src = os.path.join(_HERE, "libref_harness.so")
        dst = os.path.join(_HERE, "libref_harness_%s.so" % tag)
        if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
            shutil.copyfile(src, dst)
        L = C.CDLL(dst)
        self.lib_path = lib_path
        L.ref_open.argtypes = [C.c_char_p]
        if L.ref_open(lib_path.encode()) != 0:
            raise RuntimeError("ref_open(%s) failed" % lib_path)
        sig = [C.POINTER(_Tracks), C.c_int, C.c_uint, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_reduce.restype = C.c_int64
        L.ref_reduce.argtypes = sig
        L.ref_reduce2.restype = C.c_int64
        L.ref_reduce2.argtypes = [C.POINTER(_Tracks), C.c_int, C.c_int, C.c_uint, C.c_int64,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_reduce_seek.restype = C.c_int64
        L.ref_reduce_seek.argtypes = [C.POINTER(_Tracks), C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_multiplex.restype = C.c_int64
        L.ref_multiplex.argtypes = [C.POINTER(_Tracks), C.c_uint, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_multiset.restype = C.c_int64
        L.ref_multiset.argtypes = [C.POINTER(_Tracks), C.c_int, C.c_uint, C.c_int64,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_reducer_default.restype = C.c_double
        L.ref_reducer_default.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.ref_write_reduce.restype = C.c_int64
        L.ref_write_reduce.argtypes = [C.POINTER(_Tracks), C.c_int, C.c_uint, C.c_char_p, C.c_int]
        L.ref_mwrite.restype = C.c_int64
        L.ref_mwrite.argtypes = [C.POINTER(_Tracks), C.c_uint, C.c_char_p, C.c_int]
        L.ref_pearson.restype = C.c_double
        L.ref_pearson.argtypes = [C.POINTER(_Tracks)]
        L.ref_auc_of_reduce.restype = C.c_double
        L.ref_auc_of_reduce.argtypes = [C.POINTER(_Tracks), C.c_int, C.c_uint]
        L.ref_door_integrate.restype = C.c_double
        L.ref_door_integrate.argtypes = [C.POINTER(_Tracks), C.c_int, C.c_uint, C.c_int, C.c_void_p]
        L.ref_door_integrate_seek.restype = C.c_int
        L.ref_door_integrate_seek.argtypes = [C.POINTER(_Tracks), C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_set_compress_mode.argtypes = [C.c_int]
        L.ref_set_compress_mode.restype = None
        L.ref_set_modes.argtypes = [C.c_int, C.c_int]
        L.ref_set_modes.restype = None
        L.ref_set_map.argtypes = [C.c_int, C.c_double]
        L.ref_set_map.restype = None
        L.ref_reduce_seek_held.restype = C.c_int64
        L.ref_reduce_seek_held.argtypes = [C.POINTER(_Tracks), C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int,
                                           C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_multiset_seek_held.restype = C.c_int64
        L.ref_multiset_seek_held.argtypes = [C.POINTER(_Tracks), C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int64,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.L = L

    def write_reduce(self, t, op, path, bedgraph=False, flags=0):
        """The library's TeeWiggleIterator (`write` / `write_bg`) over its reducer `op`; returns the text."""
        s, keep = _pack(t)
        n = self.L.ref_write_reduce(C.byref(s), _opcode(op), flags, str(path).encode(), int(bedgraph))
        if n < 0:
            raise RuntimeError("ref_write_reduce returned %d" % n)
        return open(path).read()

    def mwrite(self, t, path, bedgraph=False, flags=0):
        s, keep = _pack(t)
        n = self.L.ref_mwrite(C.byref(s), flags, str(path).encode(), int(bedgraph))
        if n < 0:
            raise RuntimeError("ref_mwrite returned %d" % n)
        return open(path).read()

    def pearson(self, t):
        s, keep = _pack(t)
        return self.L.ref_pearson(C.byref(s))

    def auc_of_reduce(self, t, op, flags=0):
        s, keep = _pack(t)
        return self.L.ref_auc_of_reduce(C.byref(s), _opcode(op), flags)

    def door_integrate(self, t, kind, op="mean", flags=0):
        """The tested library's fused integrator doors (wtamd_AUCIntegrator / wtamd_MeanIntegrator over reducer `op`,
        wtamd_PearsonIntegrator over the 2-track Multiplexer); returns (result, pops, d2h bytes, runs)."""
        s, keep = _pack(t)
        info = np.zeros(3, np.int64)
        k = {"auc": 0, "mean": 1, "pearson": 2}[kind]
        r = self.L.ref_door_integrate(C.byref(s), _opcode(op) if k < 2 else 0, flags, k, info.ctypes.data)
        return r, int(info[0]), int(info[1]), int(info[2])

    def door_integrate_seek(self, t, kind, regions, op="mean", flags=0, pre_pops=0):
        """The doors driven like `apply`: pre_pops pops, then seek + drain per region (chrom index, start, finish);
        returns [value before the first seek, value after region 0, ...]."""
        s, keep = _pack(t)
        reg = np.ascontiguousarray(np.array(regions, np.int32).reshape(-1, 3))
        out = np.zeros(1 + len(reg), np.float64)
        k = {"auc": 0, "mean": 1, "pearson": 2}[kind]
        rc = self.L.ref_door_integrate_seek(C.byref(s), _opcode(op) if k < 2 else 0, flags, k, pre_pops, len(reg),
                                            reg.ctypes.data, out.ctypes.data)
        if rc != 0:
            raise RuntimeError("ref_door_integrate_seek returned %d" % rc)
        return out

    def set_compress_mode(self, on):
        """write_reduce asks the reducer to merge its runs on the device (wtamd_iterator_compress_output)."""
        self.L.ref_set_compress_mode(int(on))

    def set_map(self, op=None, param=0.0):
        """`map <op>`: every child is wrapped in one operator iterator -- the compiled reference's own
        (unaryOps.c) or the tested library's wtamd_MapIterator.  None switches it off."""
        self.L.ref_set_map(-1 if op is None else MAP_OPS[op], float(param))

    def set_modes(self, child_mode=0, block_mode=0):
        """child_mode 1: children are the tested library's own bulk-capable wtamd_ArrayReader (float32
        values); block_mode 1: one-sample reducer output is taken through wtamd_iterator_next_block."""
        self.L.ref_set_modes(child_mode, block_mode)

    def reduce(self, t, op, flags=0, n_set0=0):
        s, keep = _pack(t)
        cap = _bound(t)
        oc, os_, of, ov = _alloc(cap)
        code = _opcode(op)
        if code >= 10:
            n = self.L.ref_reduce2(C.byref(s), code, n_set0, flags, cap,
                                   oc.ctypes.data, os_.ctypes.data, of.ctypes.data, ov.ctypes.data)
        else:
            n = self.L.ref_reduce(C.byref(s), code, flags, cap,
                                  oc.ctypes.data, os_.ctypes.data, of.ctypes.data, ov.ctypes.data)
        return _trim(n, (oc, os_, of, ov))

    def reduce_seek(self, t, op, chrom, start, finish, flags=0):
        s, keep = _pack(t)
        cap = _bound(t)
        oc, os_, of, ov = _alloc(cap)
        n = self.L.ref_reduce_seek(C.byref(s), _opcode(op), flags, chrom, start, finish, cap,
                                   oc.ctypes.data, os_.ctypes.data, of.ctypes.data, ov.ctypes.data)
        return _trim(n, (oc, os_, of, ov))

    def reduce_seek_held(self, t, op, chrom, start, finish, flags=0, n_set0=0):
        """`seek chrom start finish <reducer>` the CLI's way: children hold their data until the seek."""
        s, keep = _pack(t)
        cap = _bound(t)
        oc, os_, of, ov = _alloc(cap)
        n = self.L.ref_reduce_seek_held(C.byref(s), _opcode(op), n_set0, flags, chrom, start, finish, cap,
                                        oc.ctypes.data, os_.ctypes.data, of.ctypes.data, ov.ctypes.data)
        return _trim(n, (oc, os_, of, ov))

    def multiset_seek_held(self, t, n_set0, chrom, start, finish, flags=0):
        s, keep = _pack(t)
        cap = _bound(t)
        N = int(t["n_tracks"])
        oc, os_, of, _ = _alloc(cap)
        tile = np.zeros((cap, N), np.float64)
        ip = np.zeros((cap, N), np.uint8)
        n = self.L.ref_multiset_seek_held(C.byref(s), n_set0, flags, chrom, start, finish, cap, oc.ctypes.data,
                                          os_.ctypes.data, of.ctypes.data, tile.ctypes.data, ip.ctypes.data)
        return _trim(n, (oc, os_, of, tile, ip))

    def multiplex(self, t, flags=0):
        s, keep = _pack(t)
        cap = _bound(t)
        N = int(t["n_tracks"])
        oc, os_, of, _ = _alloc(cap)
        tile = np.empty((cap, N), np.float64)
        ip = np.empty((cap, N), np.uint8)
        n = self.L.ref_multiplex(C.byref(s), flags, cap, oc.ctypes.data, os_.ctypes.data, of.ctypes.data,
                                 tile.ctypes.data, ip.ctypes.data)
        return _trim(n, (oc, os_, of, tile, ip))

    def multiset(self, t, n_set0, flags=0):
        s, keep = _pack(t)
        cap = _bound(t)
        N = int(t["n_tracks"])
        oc, os_, of, _ = _alloc(cap)
        tile = np.zeros((cap, N), np.float64)
        ip = np.zeros((cap, N), np.uint8)
        n = self.L.ref_multiset(C.byref(s), n_set0, flags, cap, oc.ctypes.data, os_.ctypes.data, of.ctypes.data,
                                tile.ctypes.data, ip.ctypes.data)
        return _trim(n, (oc, os_, of, tile, ip))

    def reducer_default(self, op, defaults):
        d = np.ascontiguousarray(defaults, np.float64)
        return self.L.ref_reducer_default(_opcode(op), len(d), d.ctypes.data)


def have_ref():
    build()
    return os.path.exists(os.path.join(_HERE, "_ref", "libwiggletools_ref.so"))


def ref_lib():
    global _ref
    if _ref is None:
        build()
        path = os.path.join(_HERE, "_ref", "libwiggletools_ref.so")
        if not os.path.exists(path):
            raise RuntimeError("compiled reference not available (oracle/_ref missing)")
        L = C.CDLL(os.path.join(_HERE, "libref_harness.so"))
        L.ref_open.argtypes = [C.c_char_p]
        if L.ref_open(path.encode()) != 0:
            raise RuntimeError("ref_open failed")
        sig = [C.POINTER(_Tracks), C.c_int, C.c_uint, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_reduce.restype = C.c_int64
        L.ref_reduce.argtypes = sig
        L.ref_reduce_compressed.restype = C.c_int64
        L.ref_reduce_compressed.argtypes = sig
        L.ref_reduce_seek.restype = C.c_int64
        L.ref_reduce_seek.argtypes = [C.POINTER(_Tracks), C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_multiplex.restype = C.c_int64
        L.ref_multiplex.argtypes = [C.POINTER(_Tracks), C.c_uint, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_multiset.restype = C.c_int64
        L.ref_multiset.argtypes = [C.POINTER(_Tracks), C.c_int, C.c_uint, C.c_int64,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_reducer_default.restype = C.c_double
        L.ref_reducer_default.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.ref_auc_of_reduce.restype = C.c_double
        L.ref_auc_of_reduce.argtypes = [C.POINTER(_Tracks), C.c_int, C.c_uint]
        L.ref_pearson.restype = C.c_double
        L.ref_pearson.argtypes = [C.POINTER(_Tracks)]
        L.ref_time_reduce.restype = C.c_double
        L.ref_time_reduce.argtypes = [C.POINTER(_Tracks), C.c_int, C.c_uint, C.c_void_p, C.c_void_p]
        L.ref_reduce_files.restype = C.c_int64
        L.ref_reduce_files.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_uint, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
        _ref = L
    return _ref


_ref_h = None


def ref_harness():
    """The compiled reference behind the same Harness class that drives the drop-in library."""
    global _ref_h
    if _ref_h is None:
        build()
        path = os.path.join(_HERE, "_ref", "libwiggletools_ref.so")
        if not os.path.exists(path):
            raise RuntimeError("compiled reference not available (oracle/_ref missing)")
        _ref_h = Harness(path, "ref")
    return _ref_h


def _bound(t):
    # every run starts at an interval start or finish
    return 2 * int(len(t["start"])) + 8


def _alloc(cap):
    return (np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float64))


def _trim(n, arrs):
    if n < 0:
        raise RuntimeError("checker returned %d" % n)
    return tuple(a[:n].copy() for a in arrs)


def _opcode(op):
    return OPS[op] if isinstance(op, str) else int(op)


def reduce(t, op, flags=0, n_set0=0):
    """Oracle restatement: returns (chrom, start, finish, value) numpy arrays."""
    L = oracle_lib()
    s, keep = _pack(t)
    cap = _bound(t)
    oc, os_, of, ov = _alloc(cap)
    n = L.wto_reduce(C.byref(s), _opcode(op), flags, n_set0, cap,
                     oc.ctypes.data, os_.ctypes.data, of.ctypes.data, ov.ctypes.data)
    return _trim(n, (oc, os_, of, ov))


def multiplex(t, flags=0):
    """Oracle restatement of the Multiplexer tile: (chrom,start,finish,values[R,N],inplay[R,N])."""
    L = oracle_lib()
    s, keep = _pack(t)
    cap = _bound(t)
    N = int(t["n_tracks"])
    oc, os_, of, _ = _alloc(cap)
    tile = np.empty((cap, N), np.float64)
    ip = np.empty((cap, N), np.uint8)
    n = L.wto_multiplex(C.byref(s), flags, cap, oc.ctypes.data, os_.ctypes.data, of.ctypes.data,
                        tile.ctypes.data, ip.ctypes.data)
    return _trim(n, (oc, os_, of, tile, ip))


def reducer_default(op, defaults):
    d = np.ascontiguousarray(defaults, np.float64)
    return oracle_lib().wto_reducer_default(_opcode(op), len(d), d.ctypes.data)


def auc(start, finish, value):
    s = np.ascontiguousarray(start, np.int32)
    f = np.ascontiguousarray(finish, np.int32)
    v = np.ascontiguousarray(value, np.float64)
    return oracle_lib().wto_auc(len(s), s.ctypes.data, f.ctypes.data, v.ctypes.data)


def pearson(t):
    """Pearson correlation of tracks 0 and 1 (reference PearsonIntegrator over a 2-track Multiplexer)."""
    c, s, f, vals, ip = multiplex(t)
    d = np.asarray(t["defaults"], np.float64)
    x = np.ascontiguousarray(np.where(ip[:, 0] != 0, vals[:, 0], d[0]), np.float64)
    y = np.ascontiguousarray(np.where(ip[:, 1] != 0, vals[:, 1], d[1]), np.float64)
    s = np.ascontiguousarray(s, np.int32)
    f = np.ascontiguousarray(f, np.int32)
    L = oracle_lib()
    L.wto_pearson.restype = C.c_double
    L.wto_pearson.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return L.wto_pearson(len(s), s.ctypes.data, f.ctypes.data, x.ctypes.data, y.ctypes.data)


MAP_OPS = {"scale": 0, "offset": 1, "ln": 2, "log": 3, "exp": 4, "expb": 5, "pow": 6, "abs": 7,
           "gt": 8, "gte": 9, "lt": 10, "lte": 11}


def map_values(op, param, values):
    """Oracle restatement of the value-wise unary operators: (out f64, keep u8)."""
    L = oracle_lib()
    L.wto_map.restype = None
    L.wto_map.argtypes = [C.c_int, C.c_double, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    v = np.ascontiguousarray(values, np.float64)
    out = np.empty(len(v), np.float64)
    keep = np.empty(len(v), np.uint8)
    L.wto_map(MAP_OPS[op], float(param), len(v), v.ctypes.data, out.ctypes.data, keep.ctypes.data)
    return out, keep


def map_default(op, param, d):
    L = oracle_lib()
    L.wto_map_default.restype = C.c_double
    L.wto_map_default.argtypes = [C.c_int, C.c_double, C.c_double]
    return L.wto_map_default(MAP_OPS[op], float(param), float(d))


def ref_map(t, track, op, param):
    """The compiled reference's operator iterator over one track: (chrom, start, finish, value, default)."""
    L = ref_lib()
    L.ref_map.restype = C.c_int64
    L.ref_map.argtypes = [C.POINTER(_Tracks), C.c_int, C.c_int, C.c_double, C.c_int64, C.c_void_p, C.c_void_p,
                          C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    s, keep = _pack(t)
    cap = int(len(np.asarray(t["start"]))) + 1
    oc, os_, of, ov = _alloc(cap)
    dflt = C.c_double()
    n = L.ref_map(C.byref(s), track, MAP_OPS[op], float(param), cap, oc.ctypes.data, os_.ctypes.data, of.ctypes.data,
                  ov.ctypes.data, C.byref(dflt))
    assert n >= 0, n
    return oc[:n], os_[:n], of[:n], ov[:n], dflt.value


def compress(chrom, start, finish, value):
    c = np.array(chrom, np.int32)
    s = np.array(start, np.int32)
    f = np.array(finish, np.int32)
    v = np.array(value, np.float64)
    n = oracle_lib().wto_compress(len(s), c.ctypes.data, s.ctypes.data, f.ctypes.data, v.ctypes.data)
    return c[:n], s[:n], f[:n], v[:n]


def tdist_Q(t, nu):
    return oracle_lib().wto_tdist_Q(float(t), float(nu))


def ttest_stat(n1, n2, values, inplay):
    v = np.ascontiguousarray(values, np.float64)
    ip = np.ascontiguousarray(inplay, np.uint8)
    t = C.c_double()
    nu = C.c_double()
    oracle_lib().wto_ttest_stat(n1, n2, v.ctypes.data, ip.ctypes.data, C.byref(t), C.byref(nu))
    return t.value, nu.value


# ---------------- compiled reference ----------------

def ref_reduce(t, op, flags=0, compressed=False):
    L = ref_lib()
    s, keep = _pack(t)
    cap = _bound(t)
    oc, os_, of, ov = _alloc(cap)
    fn = L.ref_reduce_compressed if compressed else L.ref_reduce
    n = fn(C.byref(s), _opcode(op), flags, cap, oc.ctypes.data, os_.ctypes.data, of.ctypes.data, ov.ctypes.data)
    return _trim(n, (oc, os_, of, ov))


def ref_reduce_seek(t, op, chrom, start, finish, flags=0):
    L = ref_lib()
    s, keep = _pack(t)
    cap = _bound(t)
    oc, os_, of, ov = _alloc(cap)
    n = L.ref_reduce_seek(C.byref(s), _opcode(op), flags, chrom, start, finish, cap,
                          oc.ctypes.data, os_.ctypes.data, of.ctypes.data, ov.ctypes.data)
    return _trim(n, (oc, os_, of, ov))


def ref_multiplex(t, flags=0):
    L = ref_lib()
    s, keep = _pack(t)
    cap = _bound(t)
    N = int(t["n_tracks"])
    oc, os_, of, _ = _alloc(cap)
    tile = np.empty((cap, N), np.float64)
    ip = np.empty((cap, N), np.uint8)
    n = L.ref_multiplex(C.byref(s), flags, cap, oc.ctypes.data, os_.ctypes.data, of.ctypes.data,
                        tile.ctypes.data, ip.ctypes.data)
    return _trim(n, (oc, os_, of, tile, ip))


def ref_multiset(t, n_set0, flags=0):
    L = ref_lib()
    s, keep = _pack(t)
    cap = _bound(t)
    N = int(t["n_tracks"])
    oc, os_, of, _ = _alloc(cap)
    tile = np.zeros((cap, N), np.float64)
    ip = np.zeros((cap, N), np.uint8)
    n = L.ref_multiset(C.byref(s), n_set0, flags, cap, oc.ctypes.data, os_.ctypes.data, of.ctypes.data,
                       tile.ctypes.data, ip.ctypes.data)
    return _trim(n, (oc, os_, of, tile, ip))


def ref_reducer_default(op, defaults):
    d = np.ascontiguousarray(defaults, np.float64)
    return ref_lib().ref_reducer_default(_opcode(op), len(d), d.ctypes.data)


def ref_auc_of_reduce(t, op, flags=0):
    s, keep = _pack(t)
    return ref_lib().ref_auc_of_reduce(C.byref(s), _opcode(op), flags)


def ref_pearson(t):
    s, keep = _pack(t)
    return ref_lib().ref_pearson(C.byref(s))


def ref_time_reduce(t, op, flags=0):
    """Times the compiled reference: returns (seconds, runs, covered_bp)."""
    s, keep = _pack(t)
    runs = C.c_int64()
    bp = C.c_int64()
    sec = ref_lib().ref_time_reduce(C.byref(s), _opcode(op), flags, C.byref(runs), C.byref(bp))
    return sec, runs.value, bp.value


def ref_reduce_files(paths, op, flags=0, cap=1 << 20):
    """Reference reducer over text files read by the reference's own readers."""
    L = ref_lib()
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    oc, os_, of, ov = _alloc(cap)
    names = C.create_string_buffer(1 << 16)
    n = L.ref_reduce_files(len(paths), arr, _opcode(op), flags, cap,
                           oc.ctypes.data, os_.ctypes.data, of.ctypes.data, ov.ctypes.data, names, 1 << 16)
    out = _trim(n, (oc, os_, of, ov))
    return out + (names.value.decode().split("\n") if n > 0 else [],)
