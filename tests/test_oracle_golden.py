"""Oracle restatement vs the committed golden vectors.

tests/golden/reference_fixtures.json was produced by the COMPILED REFERENCE over
the reference's own fixtures (tests/golden/make_golden.py).  The literal vectors
below are the ones SURVEY.md 8c lists (captured from the reference binary's
`write_bg` output, 0-based there, 1-based here) -- an independent pin, and the
only pin available for `wilcoxon` (setComparisons.c cannot be compiled here).
"""
import json
import os

import numpy as np
import pytest

from helpers import assert_runs_equal
from wiggletools_amd.textio import load_runlists

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    with open(os.path.join(G, "reference_fixtures.json")) as fh:
        return json.load(fh)["cases"]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "%s-%s-%d" % (c["set"], c["op"], c["strict"]))
def test_oracle_reproduces_reference_fixture_outputs(oracle, case):
    t = load_runlists([os.path.join(G, f) for f in case["files"]])
    assert t.chrom_names[:len(case["chrom_names"])] == case["chrom_names"] or \
        [t.chrom_names[c] for c in sorted(set(case["chrom"]))] == case["chrom_names"]
    got = oracle.reduce(t.as_dict(), case["op"], flags=case["strict"])
    names_got = [t.chrom_names[c] for c in got[0]]
    names_exp = [case["chrom_names"][c] for c in case["chrom"]]
    assert names_got == names_exp
    exp_v = np.array([np.nan if v is None else v for v in case["value"]], np.float64)
    assert_runs_equal((np.zeros(len(got[0])), got[1], got[2], got[3]),
                      (np.zeros(len(exp_v)), np.array(case["start"]), np.array(case["finish"]), exp_v), 0.0,
                      "%s %s" % (case["set"], case["op"]))


SURVEY_8C = {  # op -> values over chr1 runs [1,2) .. [10,11), files fixedStep.wig variableStep.wig
    "sum": [1, 3, 2, 6, 4, 9, 6, 12, 8, 9],
    "mean": [0.5, 1.5, 1, 3, 2, 4.5, 3, 6, 4, 4.5],
    "var": [0.25, 0.25, 0.5, 0, 2, 0.25, 4.5, 1, 8, 10.125],
    "stddev": [0.5, 0.5, 1, 0, 2, 0.5, 3, 1, 4, 4.5],
    "entropy": [0.5, 0.5, 1, 0, 2, 0.5, 3, 1, 4, 4.5],
    "cv": [1, 0.333333, 1, 0, 1, 0.111111, 1, 0.166667, 1, 1],
    "median": [1, 2, 2, 3, 4, 5, 6, 7, 8, 9],
    "max": [1, 2, 2, 3, 4, 5, 6, 7, 8, 9],
    "min": [0, 1, 0, 3, 0, 4, 0, 5, 0, 0],
    "product": [0, 2, 0, 9, 0, 20, 0, 35, 0, 0],
}


@pytest.mark.parametrize("op", sorted(SURVEY_8C))
def test_oracle_matches_survey_vectors(oracle, op):
    t = load_runlists([os.path.join(G, "fixedStep.wig"), os.path.join(G, "variableStep.wig")])
    c, s, f, v = oracle.reduce(t.as_dict(), op)
    assert s.tolist() == list(range(1, 11)) and f.tolist() == list(range(2, 12))
    assert np.allclose(v, SURVEY_8C[op], rtol=0, atol=5e-7)   # survey values are %lf-printed (6 decimals)


def test_oracle_mean_strict_survey_vector(oracle):
    t = load_runlists([os.path.join(G, "fixedStep.wig"), os.path.join(G, "variableStep.wig")])
    c, s, f, v = oracle.reduce(t.as_dict(), "mean", flags=1)
    assert s.tolist() == [1, 2, 4, 6, 8] and f.tolist() == [2, 3, 5, 7, 9]
    assert v.tolist() == [0.5, 1.5, 3, 4.5, 6]


def test_oracle_wilcoxon_survey_vector(oracle):
    """`wilcoxon fixedStep.wig variableStep.wig : overlapping.bed fixedStep.wig`."""
    files = ["fixedStep.wig", "variableStep.wig", "overlapping.bed", "fixedStep.wig"]
    t = load_runlists([os.path.join(G, f) for f in files])
    c, s, f, v = oracle.reduce(t.as_dict(), "mwu", n_set0=2)
    exp = [-1.990645, -1.990645, -1.041, 0, -1.041, -1.041, -1.041, -1.041, -1.685402, -1.685402]
    assert c.tolist() == [0] * 10
    assert s.tolist() == list(range(1, 11)) and f.tolist() == list(range(2, 12))
    assert np.allclose(v, exp, rtol=0, atol=5e-7)


def test_oracle_auc_mean_survey_vector(oracle):
    t = load_runlists([os.path.join(G, "fixedStep.wig"), os.path.join(G, "variableStep.wig")])
    c, s, f, v = oracle.reduce(t.as_dict(), "mean")
    assert oracle.auc(s, f, v) == 30.0


def test_oracle_pearson_reference_expected(oracle):
    """reference test/expected/pearson.txt == -0.028968 (test/test.py:104): the oracle's restatement
    of PearsonPop, and the compiled reference when available."""
    t = load_runlists([os.path.join(G, "fixedStep.wig"), os.path.join(G, "variableStep.wig")])
    assert abs(oracle.pearson(t.as_dict()) - (-0.028968)) < 5e-7
    if oracle.have_ref():
        assert oracle.ref_pearson(t.as_dict()) == oracle.pearson(t.as_dict())


def test_tdist_tail_against_scipy(oracle):
    """gsl_cdf_tdist_Q stand-in vs scipy.stats.t.sf (GSL itself is absent: parity unpinned)."""
    st = pytest.importorskip("scipy.stats")
    rng = np.random.default_rng(3)
    for _ in range(300):
        t = float(rng.random() * 12)
        nu = float(rng.random() * 200 + 0.5)
        a, b = oracle.tdist_Q(t, nu), st.t.sf(t, nu)
        assert abs(a - b) <= 1e-9 * b + 1e-300, (t, nu, a, b)
