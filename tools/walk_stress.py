"""One-off stress of the walking median against the bitmap kernel on the GPU: random track sets of chromosome-like sizes,
both lane layouts, plan overrides; every result must be bit-identical.  python tools/walk_stress.py [cases]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wiggletools_amd import engine
from wiggletools_amd.runlists import synth

def run(t, env, flags):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ts = engine.TrackSet.from_runlists(t)
        got = ts.reduce_host("median", flags=flags)
        k = ts.stats()["kernel"]
        ts.close()
        return got, k
    finally:
        for key, v in old.items():
            if v is None: os.environ.pop(key, None)
            else: os.environ[key] = v

def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(77)
    t0 = time.time()
    for c in range(cases):
        n = int(rng.choice([1, 2, 5, 16, 31, 64, 77, 100, 128]))
        lens = [int(rng.integers(20000, 700000)), int(rng.integers(1, 40000))]
        mr = float(rng.choice([1, 2, 4, 16, 50, 300]))
        t = synth(n, lens, mean_run=mr, seed=1000 + c, gap_prob=float(rng.choice([0, 0.02, 0.3, 0.8])), dtype=np.float32,
                  value_levels=int(rng.choice([2, 7, 800, 1 << 20])), nan_prob=float(rng.choice([0, 0, 1e-4])),
                  defaults=(rng.integers(-2, 3, n).astype(np.float64) / 2.0 if rng.random() < 0.3 else None))
        flags = int(rng.choice([0, 0, 1]))
        ref, k0 = run(t, {"WTAMD_NO_WALK": "1"}, flags)
        assert k0 == 0
        envs = [{}, {"WTAMD_WALK_PAIR": "0"}, {"WTAMD_WALK_T": str(int(rng.choice([64, 128, 256]))), "WTAMD_WALK_S": str(int(rng.choice([4, 8, 16, 32])))},
                {"WTAMD_WALK_CAPP": str(int(rng.choice([2, 4, 8]))), "WTAMD_WALK_OV": str(int(rng.choice([0, 3, 2048])))}]
        for env in envs:
            got, k = run(t, env, flags)
            assert k == 2, (env, k)
            assert len(got[0]) == len(ref[0]), (c, env, len(got[0]), len(ref[0]))
            for j in range(3):
                assert np.array_equal(got[j], ref[j]), (c, env, j)
            assert np.array_equal(got[3], ref[3], equal_nan=True), (c, env, "values")
        print("case %d ok: N %d, %d runs out, mean run %g" % (c, n, len(ref[0]), mr), flush=True)
    print("walk_stress: %d cases x 4 plans identical to the bitmap kernel in %.1f s" % (cases, time.time() - t0))

if __name__ == "__main__":
    main()
