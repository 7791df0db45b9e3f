#!/usr/bin/env python
"""Generates tests/golden/setcmp_fixtures.json -- INDEPENDENT pins for the two functions of the
reference's src/setComparisons.c, which cannot be compiled in this image (it includes
<gsl/gsl_cdf.h>; GSL is absent and un-vendored, and writing a stand-in is not allowed).

Nothing here calls oracle/ or the product.  Expected values come from:

* Mann-Whitney U (setComparisons.c:269-390), tie-free columns: with no ties the reference's rank
  scan (:328-359) adds, for every set-0 element, the number of set-1 elements sorted before it --
  which is scipy.stats.mannwhitneyu(x, y).statistic (U of the first sample).  On top of that the
  constructor's integer divisions (:386-387)  mu = n1*n2 / 2,  sigma = sqrt(n1*n2*(N+1) / 12)  and
  the result  2*erf(-|U1 - mu| / sigma)  (:361-366).
* Mann-Whitney U, ties: hand traces of the scan's state machine (`ties`, `previousTies`, :335-355),
  written out in tests/test_setcmp_golden.py; only the resulting U1 is stored here.
* Welch t-test (:60-117), every track in play: t and nu from the closed form of :90-113, summed
  sequentially in track order as the source does; the tail 2*Q_t(t, nu) from scipy.stats.t.sf
  (GSL's gsl_cdf_tdist_Q computes the same function; its version is unpinned).

Run:  python tests/golden/make_setcmp_golden.py   (needs numpy + scipy; this container only)
"""
import json
import math
import os

import numpy as np
from scipy import stats

HERE = os.path.dirname(os.path.abspath(__file__))


def mwu_value(U1, n1, n2):
    mu = float((n1 * n2) // 2)                                  # setComparisons.c:386, C int division
    sigma = math.sqrt(float((n1 * n2 * (n1 + n2 + 1)) // 12))   # :387
    num = (mu - U1) if U1 > mu else (U1 - mu)                   # :361-366
    if sigma == 0.0:
        z = float("nan") if num == 0.0 else math.copysign(float("inf"), num)
    else:
        z = num / sigma
    return float("nan") if math.isnan(z) else 2.0 * math.erf(z)


def ttest_stats(x, y):
    n1, n2 = len(x), len(y)
    s1 = s2 = q1 = q2 = 0.0
    for v in x:
        s1 += v; q1 += v * v                                    # :69-81 (sequential, track order)
    for v in y:
        s2 += v; q2 += v * v
    m1, m2 = s1 / n1, s2 / n2
    v1, v2 = q1 / n1 - m1 * m1, q2 / n2 - m2 * m2               # :90-95
    if v1 + v2 == 0:
        return None
    den = v1 / n1 + v2 / n2
    t = abs((m1 - m2) / math.sqrt(den))                         # :106-109
    nu = den * den / ((v1 * v1) / (n1 * n1 * (n1 - 1)) + (v2 * v2) / (n2 * n2 * (n2 - 1)))   # :113
    return t, nu


def main():
    rng = np.random.default_rng(20260927)
    out = {"mwu_tie_free": [], "mwu_ties": [], "ttest": []}
    for (n1, n2) in ((1, 1), (1, 4), (2, 3), (5, 4), (7, 12), (50, 50), (3, 60)):
        P = 24
        cols = []
        for _ in range(P):
            # N distinct values per column, exact in float32: k/8 with distinct k
            k = rng.choice(4000, size=n1 + n2, replace=False).astype(np.float64) / 8.0 - 100.0
            cols.append(k)
        M = np.array(cols)                                      # [P, N]
        exp = []
        for p in range(P):
            x, y = M[p, :n1], M[p, n1:]
            U1 = float(stats.mannwhitneyu(x, y, alternative="two-sided", method="asymptotic").statistic)
            assert U1 == float(sum((y < xv).sum() for xv in x))
            exp.append(mwu_value(U1, n1, n2))
        out["mwu_tie_free"].append({"n1": n1, "n2": n2, "values": M.tolist(),
                                    "expected": [None if math.isnan(v) else v for v in exp]})
    # hand-traced tie cases: (set 0 values, set 1 values, U1 of the trace in tests/test_setcmp_golden.py)
    for x, y, U1 in (([1, 2], [2, 3], 0.5),
                     ([1, 1, 5], [1, 4, 6], 2.0),
                     ([2, 2, 7], [2, 2, 7], 2.5),
                     ([1, 3], [1, 5], 2.0),
                     ([4, 4, 4], [4, 4], 1.0)):
        v = mwu_value(U1, len(x), len(y))
        out["mwu_ties"].append({"set0": x, "set1": y, "U1": U1, "expected": None if math.isnan(v) else v})
    for (n1, n2) in ((3, 3), (3, 7), (10, 4), (50, 50)):
        P = 24
        M = (rng.integers(0, 800, size=(P, n1 + n2)).astype(np.float64)) / 8.0
        M[0, :] = 3.0                                           # var1 + var2 == 0 -> NaN (:98)
        rows = []
        for p in range(P):
            st = ttest_stats(M[p, :n1].tolist(), M[p, n1:].tolist())
            if st is None:
                rows.append({"t": None, "nu": None, "expected": None})
            else:
                t, nu = st
                rows.append({"t": t, "nu": nu, "expected": 2.0 * float(stats.t.sf(t, nu))})
        out["ttest"].append({"n1": n1, "n2": n2, "values": M.tolist(), "rows": rows})
    with open(os.path.join(HERE, "setcmp_fixtures.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote setcmp_fixtures.json:", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
