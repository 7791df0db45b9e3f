"""GPU-box debugging aid: one small difference-array launch per process, stderr visible."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wiggletools_amd.runlists import synth
from wiggletools_amd import engine as E

n, clen, mr = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
t = synth(n, [clen], mean_run=mr, seed=1)
ts = E.TrackSet.from_runlists(t)
got = ts.reduce_host("mean")
print("ok", n, clen, mr, len(got[0]), ts.stats()["window_bp"], flush=True)
