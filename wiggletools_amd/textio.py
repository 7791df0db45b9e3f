"""Minimal host-side text readers (wig / bedGraph / bed) producing RunLists.

Mirrors what the reference's text readers hand to a Multiplexer so that parity
tests can start from the reference's own fixture files:
  * wig fixedStep / variableStep / bedGraph lines .. reference wigReader.c:110-186,
    output wrapped in run compression (wigReader.c:248, unaryOps.c:235-253)
  * bed .. reference bedReader.c:27-84: value 1, 0-based -> 1-based, overlapping
    regions merged by the union operator the Multiplexer applies
    (multiplexer.c:163, unaryOps.c:60-96)
Not a performance path (the bulk BigWig decoder is the "next" row of SURVEY 8f).
"""
import math

import numpy as np

from .runlists import RunLists


def _compress(recs):
    out = []
    for (c, s, f, v) in recs:
        if out:
            pc, ps, pf, pv = out[-1]
            if pc == c and s == pf and ((math.isnan(v) and math.isnan(pv)) or abs(v - pv) < 0.000001):
                out[-1] = (pc, ps, f, pv)
                continue
        out.append((c, s, f, v))
    return out


def _union(recs):
    out = []
    for (c, s, f, v) in recs:
        if out and out[-1][0] == c and out[-1][2] > s:
            pc, ps, pf, pv = out[-1]
            out[-1] = (pc, ps, max(pf, f), pv)
        else:
            out.append((c, s, f, v))
    return out


def read_wig(path):
    recs = []
    mode, chrom, pos, step, span = "bg", None, 0, 1, 1
    with open(path) as fh:
        for line in fh:
            line = line.rstrip("\n")
            if not line or line[0] == "#" or line.startswith("track"):
                continue
            if line.startswith("variableStep") or line.startswith("fixedStep"):
                mode = "var" if line.startswith("variableStep") else "fix"
                span = 1
                toks = line.replace("=", " ").split()
                kv = dict(zip(toks[1::2], toks[2::2]))
                chrom = kv["chrom"]
                span = int(kv.get("span", 1))
                if mode == "fix":
                    step = int(kv["step"])
                    pos = int(kv["start"]) - step
                continue
            w = line.split()
            if len(w) == 4:
                recs.append((w[0], int(w[1]) + 1, int(w[2]) + 1, float(w[3])))
            elif len(w) == 2:
                s = int(w[0])
                recs.append((chrom, s, s + span, float(w[1])))
            elif len(w) == 1:
                pos += step
                recs.append((chrom, pos, pos + span, float(w[0])))
            else:
                raise ValueError("Badly formatted wiggle or bed graph line: %r" % line)
    return _compress(recs)


def read_bed(path):
    recs = []
    with open(path) as fh:
        for line in fh:
            if not line.strip() or line[0] == "#":
                continue
            w = line.split()
            recs.append((w[0], int(w[1]) + 1, int(w[2]) + 1, 1.0))
    return _union(recs)


def read_any(path):
    if path.endswith(".bed"):
        return read_bed(path)
    if path.endswith(".wig") or path.endswith(".bg"):
        return read_wig(path)
    raise ValueError("unsupported text format: %s" % path)


def load_runlists(paths, defaults=None, dtype=np.float64):
    """Reads every file, returns RunLists with chromosomes in strcmp order."""
    per_file = [read_any(p) for p in paths]
    names = sorted({r[0] for recs in per_file for r in recs}, key=lambda s: s.encode())
    idx = {n: k for k, n in enumerate(names)}
    tracks = []
    for recs in per_file:
        by_c = [[] for _ in names]
        for (c, s, f, v) in recs:
            by_c[idx[c]].append((s, f, v))
        tracks.append(by_c)
    return RunLists.from_lists(tracks, defaults, names, dtype)
