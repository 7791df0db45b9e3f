"""Where the time of opening the bench's BigWig files goes (WTAMD_BENCH_BWDIR; WTAMD_TRACE_OPEN=1 prints per-file timings)."""
import ctypes as C, glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wiggletools_amd import _lib
L = _lib.lib()
paths = sorted(glob.glob(os.path.join(os.environ["WTAMD_BENCH_BWDIR"], "t*.bw")))
L.wtamd_BigWiggleReader.restype = C.c_void_p
L.wtamd_BigWiggleReader.argtypes = [C.c_char_p, C.c_int]
L.wtamd_BigWiggleReaders.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_void_p)]
arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths]); out = (C.c_void_p * len(paths))()
t0 = time.perf_counter()
L.wtamd_BigWiggleReaders(len(paths), arr, 1, out)
print("wtamd_BigWiggleReaders(%d), first thing in the process: %.1f ms" % (len(paths), (time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter()
for p in paths[:20]: L.wtamd_BigWiggleReader(p.encode(), 1)
print("wtamd_BigWiggleReader x 20 (one thread): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter()
L.wtamd_BigWiggleReaders(len(paths), arr, 1, out)
print("wtamd_BigWiggleReaders(%d) again: %.1f ms" % (len(paths), (time.perf_counter() - t0) * 1e3))
