"""ctypes wrapper of the CPU emulator of the HIP kernels (test-only)."""
import ctypes as C
import os

import numpy as np

from . import build as _build

_lib = None
_variant = None         # (name, flags): the emulator compiled with extra flags (use_variant)


def use_variant(name=None, flags=()):
    """Switches the emulator library: None = the default build, else tests/emu/libwt_emu_<name>.so compiled with `flags`."""
    global _lib, _variant
    _variant = (name, tuple(flags)) if name else None
    _lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_build.build_variant(*_variant) if _variant else _build.build())
        L.wtemu_reduce.restype = C.c_longlong
        L.wtemu_reduce.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_int, C.c_uint, C.c_int, C.c_longlong,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def reduce(t, op, flags=0, n_set0=0, ppt=None, T=None, multiplex=False, ranges=None, chunk=None, global_scratch=None, delta_T=None, no_delta=None,
           no_walk=None, walk_T=None, walk_S=None, walk_capp=None, walk_ov=None, walk_pair=None, mwalk=None):
    """t: RunLists.  Returns (chrom, start, finish, value) [+ (tile, inplay) if multiplex], info."""
    from oracle.oracle import OPS
    opcode = 12 if multiplex else (OPS[op] if isinstance(op, str) else int(op))
    old = {k: os.environ.get(k) for k in ("WTAMD_PPT", "WTAMD_T", "WTAMD_CHUNK", "WTAMD_GLOBAL_SCRATCH", "WTAMD_DELTA_T", "WTAMD_NO_DELTA", "WTAMD_DELTA_MIN_TRACKS",
                                             "WTAMD_NO_WALK", "WTAMD_WALK_T", "WTAMD_WALK_S", "WTAMD_WALK_CAPP", "WTAMD_WALK_OV", "WTAMD_WALK_PAIR", "WTAMD_MWALK")}
    try:
        for k, v in (("WTAMD_PPT", ppt), ("WTAMD_T", T), ("WTAMD_CHUNK", chunk), ("WTAMD_GLOBAL_SCRATCH", global_scratch), ("WTAMD_DELTA_T", delta_T), ("WTAMD_NO_DELTA", no_delta), ("WTAMD_DELTA_MIN_TRACKS", 1),
                     ("WTAMD_NO_WALK", no_walk), ("WTAMD_WALK_T", walk_T), ("WTAMD_WALK_S", walk_S), ("WTAMD_WALK_CAPP", walk_capp), ("WTAMD_WALK_OV", walk_ov), ("WTAMD_WALK_PAIR", walk_pair), ("WTAMD_MWALK", mwalk)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        cap = 2 * t.n_intervals + 8
        N = t.n_tracks
        os_, of, ov = np.empty(cap, np.int32), np.empty(cap, np.int32), np.full(cap, -7.0, np.float64)
        cro = np.zeros(t.n_chrom + 1, np.int64)
        tile = np.zeros((cap, N), np.float64) if multiplex else None
        ip = np.zeros((cap, N), np.uint8) if multiplex else None
        info = np.zeros(16, np.int64)
        if ranges is not None:
            rlo = np.ascontiguousarray([r[0] for r in ranges], np.int32)
            rhi = np.ascontiguousarray([r[1] for r in ranges], np.int32)
        value = np.ascontiguousarray(t.value)
        n = lib().wtemu_reduce(t.n_chrom, N, t.seg_off.ctypes.data, t.start.ctypes.data, t.finish.ctypes.data,
                               value.ctypes.data, int(value.dtype == np.float64), t.defaults.ctypes.data,
                               opcode, flags, n_set0, cap, os_.ctypes.data, of.ctypes.data, ov.ctypes.data,
                               cro.ctypes.data, tile.ctypes.data if multiplex else None,
                               ip.ctypes.data if multiplex else None, info.ctypes.data,
                               rlo.ctypes.data if ranges is not None else None,
                               rhi.ctypes.data if ranges is not None else None)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if n < 0:
        raise RuntimeError("emulator returned %d" % n)
    assert cro[-1] == n
    chrom = np.repeat(np.arange(t.n_chrom, dtype=np.int32), np.diff(cro))
    out = (chrom, os_[:n].copy(), of[:n].copy(), ov[:n].copy())
    if multiplex:
        out = (chrom, os_[:n].copy(), of[:n].copy(), tile[:n].copy(), ip[:n].copy())
    return out, dict(W=int(info[0]), T=int(info[1]), lds=int(info[2]), n_windows=int(info[3]), n_chunks=int(info[6]), scratch_slab=int(info[7]), delta=int(info[8]), delta_bad=int(info[9]), delta_redo=int(info[10]), patched=int(info[11]), walk=int(info[12]), walk_rounds=int(info[13]), walk_fallback=int(info[14]),
                     covered_bp=int(info[4]), n_intervals=int(info[5]))
