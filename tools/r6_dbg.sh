#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6_dbg; mkdir -p $OUT; cd $R
WTAMD_TRACE=1 python - > $OUT/dbg.log 2>&1 <<'PY'
import numpy as np, sys, torch
sys.path.insert(0, "tests")
from wiggletools_amd import engine, synthgen
device = torch.device("cuda", 0)
for L in (3000000, 46709983):
    seg, s, f, v = synthgen.device_tracks(20260927, [L], 100, 16.0, 0.02, 800, device, chrom_ids=[20])
    ts = engine.TrackSet.from_device(1, 100, seg, s, f, v, np.zeros(100))
    out = ts.alloc_runs()
    ts.index("ttest", None)
    n = ts.reduce("ttest", out, n_set0=50, sync=True)
    print(L, "stats", n, ts.stats())
    n = ts.reduce("ttest", out, n_set0=50, sync=True)
    print(L, "stats2", n, ts.stats())
    ts.close()
PY
tail -20 $OUT/dbg.log
timeout 900 python -m pytest tests -q -m gpu -x -k "ttest" > $OUT/tests.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/tests.log
