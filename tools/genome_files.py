"""Runs only the whole-genome BigWig-files-to-result leg of bench.py (e2e_bigwig_genome): N files x 24 chromosomes ->
device inflate + decode -> reducer -> runs on the host.  WTAMD_BENCH_BWDIR=<dir> keeps the files for later runs (A/B of
libraries and settings: WTAMD_LIB, WTAMD_INFLATE_RING, WTAMD_BW_BATCH_SECTIONS, ...)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
    tracks = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    op = sys.argv[3] if len(sys.argv) > 3 else "mean"
    torch.cuda.set_device(0)
    only = [int(x) for x in os.environ["WTAMD_GENOME_ONLY"].split(",")] if os.environ.get("WTAMD_GENOME_ONLY") else None
    r = bench.e2e_bigwig_genome(op, tracks, 16.0, scale, torch.device("cuda", 0), only=only, fresh_process=bool(os.environ.get("WTAMD_FRESH")))
    r["env"] = {k: v for k, v in os.environ.items() if k.startswith("WTAMD_")}
    print(json.dumps(r))
