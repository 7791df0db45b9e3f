"""What a `wiggletools mean *.bw` PROCESS gets: a fresh interpreter that has not touched the GPU.  Like a C program linked
against it, the process loads libwiggletools_amd.so FIRST (library_load_s: the dynamic loader bringing in the HIP runtime's
shared objects -- libamdhip64, libhsa-runtime64, libamd_comgr -- which is the runtime's price, reported on its own), then
the harness's own imports (not the product's), then the clock starts:
    wtamd_BigWiggleReaders -> newMultiplexer -> <Op>Reduction -> the last block of runs on the host.
No torch, no earlier GPU call: the HIP runtime's start-up, the code object, hardware queues, file opens, page-locking and
device allocations are all inside `seconds`.  Prints one JSON line.
usage: python tools/cli_cold.py <dir with g000.bw ...> [n_files] [op] [genome_bp]"""
import ctypes
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    os.environ["WTAMD_NO_TORCH"] = "1"          # this process links the system's HIP runtime only
    t_l0 = time.perf_counter()
    ctypes.CDLL(os.environ.get("WTAMD_LIB") or os.path.join(ROOT, "wiggletools_amd", "csrc", "libwiggletools_amd.so"), mode=ctypes.RTLD_GLOBAL)
    t_l1 = time.perf_counter()
    sys.path.insert(0, ROOT)
    from wiggletools_amd import dropin          # numpy + the ctypes signatures
    dropin._bind()
    d = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    op = sys.argv[3] if len(sys.argv) > 3 else "mean"
    genome_bp = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    paths = sorted(glob.glob(os.path.join(d, "g*.bw")))[:n]
    t0 = time.perf_counter()
    readers = dropin.bigwig_readers(paths, box=True)
    t_open = time.perf_counter()
    r = dropin.reducer(op, readers, n_set0=len(paths) // 2)
    t_red = time.perf_counter()
    first = []
    runs, _ = dropin.drain_blocks(r, on_block=lambda c, a, b, v: (first.append(time.perf_counter()) if not first else None, 0)[1])
    t1 = time.perf_counter()
    st = dropin.pipe_stats(r)
    out = {"seconds": t1 - t0, "runs": runs, "library_load_s": t_l1 - t_l0, "open_readers_s": t_open - t0, "reducer_ctor_s": t_red - t_open,
           "first_block_at_s": (first[0] - t0) if first else None, "batches": st.get("batches"), "sum_device_decode_ms": st.get("bw_decode_ms"),
           "host_submit_ms": st.get("host_submit_ms"), "host_wait_ms": st.get("host_wait_ms"), "files": len(paths),
           "seconds_with_library_load": (t1 - t0) + (t_l1 - t_l0)}
    if genome_bp:
        out["bp_per_s"] = genome_bp / (t1 - t0)
        out["bp_per_s_with_library_load"] = genome_bp / out["seconds_with_library_load"]
    print(json.dumps(out))
