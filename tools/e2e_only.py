"""Runs only the end-to-end (drop-in layer) leg of bench.py -- for rocprofv3 traces of the pipeline."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
    torch.cuda.set_device(0)
    print(json.dumps(bench.e2e_dropin("mean", 100, 16.0, mbp, torch.device("cuda", 0))))
