#!/bin/bash
cd $GRAFT_REPO_ROOT
python -u - 2>&1 <<'PY' | tail -30
import numpy as np, sys
sys.path.insert(0, "tests")
from wiggletools_amd import engine
from wiggletools_amd.runlists import synth
import oracle.oracle as O
seed = 7
rng = np.random.default_rng(7100 + seed)
n = int(rng.choice([8, 16, 33, 100, 200, 900]))
n1 = int(rng.integers(3, n - 2)) if seed % 2 else n // 2
t = synth(n, [int(rng.integers(3000, 90000)), 900], mean_run=float(rng.choice([1, 3, 16, 60, 3000])), seed=seed,
          gap_prob=float(rng.choice([0, 0.05, 0.5])), dtype=np.float32, value_levels=800)
ts = engine.TrackSet.from_runlists(t)
got = ts.reduce_host("ttest", flags=0, n_set0=n1)
exp = O.reduce(t.as_dict(), "ttest", flags=0, n_set0=n1)
g, e = got[3], exp[3]
m = ~np.isnan(e)
rel = np.abs(g[m] - e[m]) / np.maximum(np.abs(e[m]), 1e-300)
order = np.argsort(-rel)[:12]
for i in order:
    print("p oracle %.17g device %.17g rel %.3g" % (e[m][i], g[m][i], rel[i]))
print("n", n, "n1", n1, "median rel", np.median(rel), "count > 1e-10", int((rel > 1e-10).sum()), "of", len(rel))
import os
os.environ["WTAMD_NO_DELTA_TTEST"] = "1"
ts2 = engine.TrackSet.from_runlists(t)
g2 = ts2.reduce_host("ttest", flags=0, n_set0=n1)[3]
print("general vs delta on device: equal bits", np.array_equal(g2, g, equal_nan=True), "max rel", np.nanmax(np.abs(g2 - g) / np.maximum(np.abs(g), 1e-300)))
PY
