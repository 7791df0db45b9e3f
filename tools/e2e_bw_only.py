"""Runs only the BigWig-files-to-result leg of bench.py (device-side inflate + decode) -- for rocprofv3 traces."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 47.0
    tracks = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    torch.cuda.set_device(0)
    reps = int(os.environ.get("WTAMD_E2E_REPS", "1"))
    for k in range(reps):       # (the second run of a process finds the pinned pool and the hardware queues warm)
        print(json.dumps(bench.e2e_bigwig("mean", tracks, 16.0, mbp, torch.device("cuda", 0))))
