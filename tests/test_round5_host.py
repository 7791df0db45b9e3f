"""Host-side pieces of round 5 that need no GPU: the descriptor table grown when the library is loaded (csrc/wt_bigwig.cpp,
DESIGN 6.6), the bench's untimed read-through of freshly written files, MWUReduction's table (csrc/wt_plan.h) against the reference's
own expression (setComparisons.c:361-366, :386-387)."""
import math
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "wiggletools_amd", "csrc", "libwiggletools_amd.so")


def _fdsize_after_load(extra_env):
    code = ("import ctypes, re; ctypes.CDLL(%r); import time; time.sleep(0.5);"
            "print(int(re.search(r'FDSize:\\s+(\\d+)', open('/proc/self/status').read()).group(1)))" % LIB)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra_env), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-800:]
    return int(r.stdout.strip().splitlines()[-1])


def test_library_load_grows_the_descriptor_table():
    """A process that loads the library has room for thousands of descriptors before it opens its first file: no open() of a
    later, multi-threaded phase crosses the table's end (on the 256-CPU GPU hosts that is a 140-180 ms synchronize_rcu() for
    every thread that needs the new table: 16 of 100 fopen() calls of a fresh process, round 5)."""
    import resource
    soft = resource.getrlimit(resource.RLIMIT_NOFILE)[0]
    want = min(4096, soft if soft != resource.RLIM_INFINITY else 4096)
    assert _fdsize_after_load({}) >= want // 2          # (the kernel rounds the table to a power of two >= the descriptor asked for)
    assert _fdsize_after_load({"WTAMD_NO_FD_GROW": "1"}) < want // 2 or want <= 256


def test_bench_read_through(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    paths = []
    for k in range(5):
        p = tmp_path / ("f%d.bin" % k)
        p.write_bytes(os.urandom(100000 + 777 * k))
        paths.append(str(p))
    total, seconds = bench.read_through(paths, threads=3)
    assert total == sum(os.path.getsize(p) for p in paths) and seconds >= 0


def test_mwu_table_is_the_references_expression():
    """wtemu / engine fill the table with wt_mwu_make_table; here the same arithmetic in Python (double, the platform's erf):
    entry k = 2 erf(-(k / 2) / sigma), mu and sigma from C integer divisions; the emulator's MWU values must be entries of it."""
    from emu import emu
    from wiggletools_amd.runlists import synth
    n1, n2 = 7, 9
    mu = float((n1 * n2) // 2)
    sigma = math.sqrt(float((n1 * n2 * (n1 + n2 + 1)) // 12))
    table = set()
    k = 0
    while True:
        v = 2 * math.erf((mu - (mu + 0.5 * k)) / sigma) if k else 2 * math.erf(0.0 / sigma)
        table.add(v)
        if v == -2.0:
            break
        k += 1
    t = synth(n1 + n2, [3000], mean_run=5, seed=3, dtype=np.float32, value_levels=9)
    got, info = emu.reduce(t, "mwu", n_set0=n1)
    vals = got[3][~np.isnan(got[3])]
    assert len(vals) > 500 and all(float(v) in table for v in vals)
