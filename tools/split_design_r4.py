"""Cuts round 4's DESIGN.md (git: f894be8) into one file per heading under /tmp/design_parts -- the input of tools/assemble_design.py.
Usage: git show f894be8:DESIGN.md | python tools/split_design_r4.py /tmp/design_parts"""
import os
import re
import sys

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
L = sys.stdin.read().split("\n")
heads = [(i, l) for i, l in enumerate(L) if l.startswith("## ") or l.startswith("### ")]
for n, (i, l) in enumerate(heads):
    j = heads[n + 1][0] if n + 1 < len(heads) else len(L)
    name = re.sub(r"[^A-Za-z0-9]+", "_", l)[:60]
    open(os.path.join(out, "%03d_%s.md" % (n, name)), "w").write("\n".join(L[i:j]))
