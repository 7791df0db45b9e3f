"""Generates tests/golden/reference_fixtures.json by running the COMPILED REFERENCE
(oracle/_ref, built from /root/reference/src by oracle/Makefile) over the
reference's own test fixtures (copied verbatim as data files next to this
script: fixedStep.wig, variableStep.wig, overlapping.bed) through the
reference's own text readers.

Run in the build container only (needs /root/reference to build oracle/_ref):
    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

OPS = ["sum", "product", "mean", "var", "stddev", "entropy", "cv", "min", "max", "median"]
SETS = {
    "fixed_variable": ["fixedStep.wig", "variableStep.wig"],
    "fixed_variable_bed": ["fixedStep.wig", "variableStep.wig", "overlapping.bed"],
    "variable_fixed": ["variableStep.wig", "fixedStep.wig"],
    "single": ["variableStep.wig"],
}


def enc(v):
    return [None if np.isnan(x) else float(x) for x in v]


def main():
    out = {"generator": "tests/golden/make_golden.py", "source": "compiled reference v1.2.11 (oracle/_ref)",
           "cases": []}
    for sname, files in SETS.items():
        paths = [os.path.join(HERE, f) for f in files]
        for strict in (0, 1):
            for op in OPS:
                c, s, f, v, names = O.ref_reduce_files(paths, op, flags=strict)
                out["cases"].append({"set": sname, "files": files, "op": op, "strict": strict,
                                     "chrom_names": names, "chrom": c.tolist(), "start": s.tolist(),
                                     "finish": f.tolist(), "value": enc(v)})
    with open(os.path.join(HERE, "reference_fixtures.json"), "w") as fh:
        json.dump(out, fh, indent=0, separators=(",", ":"))
    print("wrote %d cases" % len(out["cases"]))


if __name__ == "__main__":
    main()
