"""Loader of the C-ABI shared library (wiggletools_amd/csrc/libwiggletools_amd.so).

There is NO fallback: if the HIP library is missing the import of the engine
fails loudly (the oracle under oracle/ is test infrastructure, never a
substitute).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WTAMD_LIB") or os.path.join(_HERE, "csrc", "libwiggletools_amd.so")


class Tracks(C.Structure):
    _fields_ = [("n_chrom", C.c_int32), ("n_tracks", C.c_int32), ("seg_off", C.c_void_p),
                ("start", C.c_void_p), ("finish", C.c_void_p), ("value", C.c_void_p),
                ("value_is_f64", C.c_int32), ("defaults", C.c_void_p),
                ("range_lo", C.c_void_p), ("range_hi", C.c_void_p)]


class ReduceDesc(C.Structure):
    _fields_ = [("op", C.c_int32), ("flags", C.c_uint32), ("n_set0", C.c_int32), ("reserved", C.c_int32)]


class Runs(C.Structure):
    _fields_ = [("capacity", C.c_int64), ("start", C.c_void_p), ("finish", C.c_void_p),
                ("value", C.c_void_p), ("chrom_run_off", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("n_runs", C.c_int64), ("covered_bp", C.c_int64), ("n_intervals", C.c_int64),
                ("n_windows", C.c_int64), ("window_bp", C.c_int32), ("lds_bytes", C.c_int32),
                ("index_ms", C.c_float), ("reduce_ms", C.c_float), ("kernel", C.c_int32), ("patched_windows", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "wiggletools_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    try:
        if os.environ.get("WTAMD_NO_TORCH"):     # (tools/cli_cold.py: a process that never touches torch, like the reference's CLI)
            raise ImportError
        # PyTorch's HIP runtime goes first: a process that loads this library (and with it /opt/rocm's
        # libamdhip64) before torch ends up with two runtimes and torch.cuda.is_available() == False
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)   # RTLD_LOCAL: the library exports the reference's symbol names (pop, seek, ...)
    L.wtamd_last_error.restype = C.c_char_p
    L.wtamd_version.restype = C.c_char_p
    L.wtamd_device_count.restype = C.c_int
    L.wtamd_set_device.argtypes = [C.c_int]
    L.wtamd_trackset_create_host.argtypes = [C.POINTER(Tracks), C.POINTER(C.c_void_p)]
    L.wtamd_trackset_create_device.argtypes = [C.POINTER(Tracks), C.POINTER(C.c_void_p)]
    L.wtamd_trackset_destroy.argtypes = [C.c_void_p]
    L.wtamd_trackset_destroy.restype = None
    L.wtamd_trackset_max_runs.argtypes = [C.c_void_p]
    L.wtamd_trackset_max_runs.restype = C.c_int64
    L.wtamd_trackset_index.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.wtamd_reduce.argtypes = [C.c_void_p, C.POINTER(ReduceDesc), C.POINTER(Runs), C.POINTER(C.c_int64), C.c_void_p]
    L.wtamd_reduce_host.argtypes = [C.c_void_p, C.POINTER(ReduceDesc), C.POINTER(Runs), C.POINTER(C.c_int64)]
    L.wtamd_multiplex_host.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Runs), C.c_void_p, C.c_void_p,
                                       C.POINTER(C.c_int64)]
    L.wtamd_runs_auc.argtypes = [C.POINTER(Runs), C.c_int64, C.POINTER(C.c_double), C.c_void_p]
    L.wtamd_runs_map.argtypes = [C.c_int, C.c_double, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.wtamd_map_default.argtypes = [C.c_int, C.c_double, C.c_double]
    L.wtamd_map_default.restype = C.c_double
    L.wtamd_runs_mean.argtypes = [C.POINTER(Runs), C.c_int64, C.POINTER(C.c_double), C.c_void_p]
    L.wtamd_runs_compress.argtypes = [C.POINTER(Runs), C.c_int64, C.c_int32, C.POINTER(Runs), C.POINTER(C.c_int64), C.c_void_p]
    L.wtamd_trackset_validate.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.wtamd_pearson.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.wtamd_pearson_moments.argtypes = [C.c_void_p, C.c_void_p]
    L.wtamd_pearson_merge.argtypes = [C.c_void_p, C.c_void_p]
    L.wtamd_pearson_merge.restype = None
    L.wtamd_pearson_finish.argtypes = [C.c_void_p]
    L.wtamd_pearson_finish.restype = C.c_double
    L.wtamd_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.wtamd_bw_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.wtamd_bw_close.argtypes = [C.c_void_p]
    L.wtamd_bw_close.restype = None
    L.wtamd_bw_n_chrom.argtypes = [C.c_void_p]
    L.wtamd_bw_chrom_name.argtypes = [C.c_void_p, C.c_int]
    L.wtamd_bw_chrom_name.restype = C.c_char_p
    L.wtamd_bw_chrom_length.argtypes = [C.c_void_p, C.c_int]
    L.wtamd_bw_chrom_length.restype = C.c_uint32
    L.wtamd_bw_read_chrom.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.wtamd_bw_read_chrom.restype = C.c_int64
    L.wtamd_reducer_default.argtypes = [C.c_int, C.c_int, C.c_void_p]
    L.wtamd_reducer_default.restype = C.c_double
    _lib = L
    return L


class WtamdError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise WtamdError("wtamd error %d: %s" % (rc, lib().wtamd_last_error().decode()))
