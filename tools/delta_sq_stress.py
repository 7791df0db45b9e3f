"""One-off stress (round 5): the difference-array kernels with squares in the emulator -- parked edge runs, scans on the first 512 of
768 lanes, totals-only scan state -- against the oracle on random track sets: many tracks with few runs, long runs, gaps, tiny and
huge chromosomes, both coverage predicates.  Usage: python tools/delta_sq_stress.py [n_cases]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from emu import emu  # noqa: E402
from helpers import assert_runs_equal  # noqa: E402
from wiggletools_amd.runlists import synth  # noqa: E402
import oracle.oracle as O  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
O.build()
orc = O
done = 0
for seed in range(n_cases):
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([8, 9, 33, 100, 257, 500, 800]))
    clens = [int(rng.choice([300, 4096, 4097, 9000, 40000]))] + ([int(rng.integers(1, 3000))] if seed % 3 == 0 else [])
    mean_run = float(rng.choice([1, 2, 7, 16, 300, 6000]))
    t = synth(n, clens, mean_run=mean_run, gap_prob=float(rng.choice([0.0, 0.02, 0.4])), seed=seed)
    d = t.as_dict()
    for op in ("var", "stddev", "cv"):
        for flags in (0, 1):
            exp = orc.reduce(d, op, flags=flags)
            for sq_t in (None, "512", "704"):
                if sq_t is None:
                    os.environ.pop("WTAMD_DELTA_SQ_T", None)
                else:
                    os.environ["WTAMD_DELTA_SQ_T"] = sq_t
                got, info = emu.reduce(t, op, flags=flags)
                assert info["delta"] == 1, info
                assert_runs_equal(got, exp, 1e-12, "seed %d n %d run %g op %s strict %d T %s" % (seed, n, mean_run, op, flags, sq_t))
                done += 1
os.environ.pop("WTAMD_DELTA_SQ_T", None)
print("ok: %d comparisons" % done)
