"""round 6: the host's share of a resident step: wall time of index / reduce / stats / synchronize for one chromosome of C2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from wiggletools_amd import engine, synthgen
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
for c in (20, 10):
    L = bench.GRCH38[c]
    acc = {}
    for rep in range(6):
        seg, s, f, v = synthgen.device_tracks(bench.SEED, [L], 100, 16.0, 0.02, 800, dev, chrom_ids=[c])
        t = {}
        t0 = time.perf_counter(); ts = engine.TrackSet.from_device(1, 100, seg, s, f, v, np.zeros(100)); out = ts.alloc_runs(); torch.cuda.synchronize(); t["create+alloc"] = time.perf_counter() - t0
        t0 = time.perf_counter(); ts.index("mean", stream); t["index call"] = time.perf_counter() - t0
        t0 = time.perf_counter(); ts.reduce("mean", out, stream=stream, sync=False); t["reduce call"] = time.perf_counter() - t0
        t0 = time.perf_counter(); st = ts.stats(); t["stats (waits)"] = time.perf_counter() - t0
        t0 = time.perf_counter(); torch.cuda.synchronize(); t["synchronize"] = time.perf_counter() - t0
        t["gpu index_ms"] = st["index_ms"] / 1e3; t["gpu reduce_ms"] = st["reduce_ms"] / 1e3
        ts.close(); del ts, out, s, f, v
        if rep:
            for k, x in t.items(): acc[k] = acc.get(k, 0) + x / 5
    print("chrom", c + 1, {k: round(x * 1e3, 3) for k, x in acc.items()}, "ms")
