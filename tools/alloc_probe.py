"""How long do big HIP allocations take on this box?  (hipMalloc / hipHostMalloc / frees, by size)"""
import ctypes as C
import time

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipHostFree.argtypes = [C.c_void_p]
hip.hipSetDevice(0)
p = C.c_void_p()
hip.hipMalloc(C.byref(p), 1 << 20); hip.hipFree(p)
for name, fn, free, sizes in (("hipMalloc", lambda q, n: hip.hipMalloc(C.byref(q), n), hip.hipFree, (64 << 20, 256 << 20, 1 << 30, 3 << 30)),
                             ("hipHostMalloc", lambda q, n: hip.hipHostMalloc(C.byref(q), n, 0), hip.hipHostFree, (16 << 20, 64 << 20, 256 << 20, 1 << 30))):
    for n in sizes:
        for rep in range(2):
            q = C.c_void_p()
            t0 = time.perf_counter(); rc = fn(q, n); t1 = time.perf_counter()
            free(q); t2 = time.perf_counter()
            print("%-14s %6d MB  alloc %8.2f ms  free %8.2f ms  rc %d" % (name, n >> 20, (t1 - t0) * 1e3, (t2 - t1) * 1e3, rc))
