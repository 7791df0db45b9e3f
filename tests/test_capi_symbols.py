"""The C-ABI library loads on a GPU-less host and exports every symbol include/wiggletools_amd.h
declares (no compute call is made here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "wiggletools_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(wtamd_[a-z_0-9]+)\s*\(", txt))
    names |= set(re.findall(r"^[A-Za-z_][\w \*]*?\b(\w+)\s*\([^;{]*\);", txt, flags=re.M))
    names -= {"void", "pop_fn"}
    return {n for n in names if not n.startswith("WTAMD_")}


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from wiggletools_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared()
    must = {"newMultiplexer", "newMultiset", "popMultiplexer", "seekMultiplexer", "runMultiplexer",
            "newCoreMultiplexer", "popMultiset", "seekMultiset", "SumReduction", "ProductReduction",
            "MeanReduction", "VarianceReduction", "StdDevReduction", "EntropyReduction", "CVReduction",
            "MedianReduction", "MinReduction", "MaxReduction", "SelectReduction", "FillInReduction",
            "TTestReduction", "MWUReduction", "newWiggleIterator", "pop", "seek", "runWiggleIterator",
            "destroyWiggleIterator", "wtamd_reduce", "wtamd_reduce_host", "wtamd_trackset_create_host",
            "wtamd_trackset_create_device", "wtamd_multiplex_host", "wtamd_runs_auc", "wtamd_reducer_default", "wtamd_pearson", "wtamd_runs_compress", "wtamd_trackset_validate", "wtamd_runs_mean", "wtamd_runs_map", "wtamd_map_default",
            "FTestReduction", "wtamd_pipe_create", "wtamd_pipe_acquire", "wtamd_pipe_grow", "wtamd_pipe_submit",
            "wtamd_pipe_collect", "wtamd_pipe_release", "wtamd_pipe_cancel", "wtamd_pipe_destroy", "wtamd_pipe_get_stats"}
    assert must <= declared, must - declared
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing


def test_struct_layouts_match_the_reference_abi():
    """Sizes / offsets of the three ABI structs on LP64 (reference wiggleIterator.h:21-35,
    multiplexer.h:21-36, multiSet.h:20-30)."""
    import subprocess
    import tempfile
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "wiggletools_amd.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(struct wiggleIterator_st), offsetof(struct wiggleIterator_st, value),
         offsetof(struct wiggleIterator_st, done), offsetof(struct wiggleIterator_st, data),
         offsetof(struct wiggleIterator_st, pop), offsetof(struct wiggleIterator_st, default_value),
         offsetof(struct wiggleIterator_st, append));
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(struct multiplexer_st), offsetof(struct multiplexer_st, values),
         offsetof(struct multiplexer_st, count), offsetof(struct multiplexer_st, inplay),
         offsetof(struct multiplexer_st, done), offsetof(struct multiplexer_st, data));
  printf("%zu %zu %zu %zu\n", sizeof(struct multiset_st), offsetof(struct multiset_st, values),
         offsetof(struct multiset_st, multis), offsetof(struct multiset_st, done));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "a.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "a.c"), "-o",
                               os.path.join(d, "a")])
        out = subprocess.check_output([os.path.join(d, "a")]).decode().split("\n")
    assert out[0].split() == ["88", "16", "32", "40", "48", "72", "80"]
    assert out[1].split() == ["104", "16", "32", "40", "56", "96"]
    assert out[2].split() == ["72", "16", "40", "48"]


def test_reducer_defaults_match_oracle(oracle):
    import numpy as np
    from wiggletools_amd import engine
    rng = np.random.default_rng(7)
    cases = [np.zeros(3), np.array([1.0, 2.0, 3.5]), np.array([0.1, 0.2, 0.7, 1e-3]),
             np.array([1.0, np.nan, 2.0]), np.array([5.0]), rng.random(17) * 100 - 50, np.array([0.0, 0.0, 1.0, 2.0])]
    for d in cases:
        for op in ("sum", "product", "mean", "var", "stddev", "entropy", "cv", "min", "max", "median", "ttest", "mwu"):
            a, b = engine.reducer_default(op, d), oracle.reducer_default(op, d)
            assert (np.isnan(a) and np.isnan(b)) or a == b, (op, d, a, b)
