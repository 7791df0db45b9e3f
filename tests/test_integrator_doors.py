"""Fused integrator doors of the drop-in layer: wtamd_AUCIntegrator / wtamd_MeanIntegrator over a reducer,
wtamd_PearsonIntegrator over a 2-track Multiplexer (reference AUCIntegrator / MeanIntegrator / PearsonIntegrator,
src/statistics.c:62-127,414-465, built by commandParser.c:653-704).  Handed this library's own reducer / Multiplexer
they integrate ON THE DEVICE batch by batch -- no per-position run crosses PCIe (SURVEY 8f-4's stated purpose) --
and must equal the COMPILED REFERENCE's integrators over the reference's reducers to rounding (the reference sums
sequentially, the device in slices merged in order); with WTAMD_NO_FUSED_INTEGRATORS=1 they are the reference's
per-run pass-through on the host and equal it bit for bit.

"emu": host logic over the emulated pipeline (CPU); "amd" (-m gpu): the product (HIP kernels)."""
import numpy as np
import pytest

from helpers import random_case
from test_dropin import _get


@pytest.fixture(params=["emu", pytest.param("amd", marks=pytest.mark.gpu)])
def H(request, oracle):
    return _get(oracle, request.param)


def _close(a, b, rel=1e-9):
    if np.isnan(a) or np.isnan(b):
        return np.isnan(a) and np.isnan(b)
    return abs(a - b) <= rel * max(1.0, abs(a), abs(b))


def _mean_of(oracle, d, op, flags):
    c, s, f, v = oracle.reduce(d, op, flags=flags)
    ok = ~np.isnan(v)
    ln = (f.astype(np.int64) - s)[ok].astype(np.float64)
    return float((ln * v[ok]).sum() / ln.sum()) if ln.sum() > 0 else float("nan")


@pytest.mark.parametrize("seed", range(6))
def test_fused_auc_and_mean_vs_reference(oracle, H, seed, monkeypatch):
    if not oracle.have_ref():
        pytest.skip("compiled reference not available")
    if seed >= 3:       # many small batches: the integrals are merged across ~100 batches
        monkeypatch.setenv("WTAMD_MIN_SPAN", "64")
        monkeypatch.setenv("WTAMD_BATCH_INTERVALS", "200")
    t = random_case(8800 + seed, max_len=6000)
    d = t.as_dict()
    for op in ("mean", "sum", "max", "median", "var"):
        for flags in (0, 1):
            want = oracle.ref_auc_of_reduce(d, op, flags)
            got, pops, d2h, runs = H.door_integrate(d, "auc", op, flags)
            assert _close(got, want), (op, flags, got, want)
            # fused: one pop per batch, not per run -- and (almost) nothing but counters came back over PCIe
            # (the first two batches -- 65 536 bp each by default -- are on their way before the door exists)
            if seed >= 3 and runs > 2000:
                assert pops < runs // 4, (pops, runs)
                assert d2h < 16 * runs // 2, (d2h, runs)
            got_m, pops, d2h, runs = H.door_integrate(d, "mean", op, flags)
            assert _close(got_m, _mean_of(oracle, d, op, flags)), (op, flags)


@pytest.mark.parametrize("seed", range(4))
def test_fused_pearson_vs_reference(oracle, H, seed, monkeypatch):
    if not oracle.have_ref():
        pytest.skip("compiled reference not available")
    if seed >= 2:
        monkeypatch.setenv("WTAMD_MIN_SPAN", "64")
        monkeypatch.setenv("WTAMD_TILE_BYTES", "4096")
    t = random_case(8900 + seed, n_tracks=2, max_len=8000)
    d = t.as_dict()
    want = oracle.ref_pearson(d)
    got, pops, _, _ = H.door_integrate(d, "pearson")
    assert _close(got, want, 1e-8), (got, want)
    n_runs = len(oracle.multiplex(d)[0])
    if n_runs > 2000:
        assert pops < n_runs // 4


def test_host_pass_through_is_the_reference(oracle, H, monkeypatch):
    """WTAMD_NO_FUSED_INTEGRATORS=1 (and any foreign source): one pop per run, sums in the reference's order."""
    if not oracle.have_ref():
        pytest.skip("compiled reference not available")
    monkeypatch.setenv("WTAMD_NO_FUSED_INTEGRATORS", "1")
    t = random_case(8950, max_len=3000)
    d = t.as_dict()
    for op in ("mean", "max"):
        want = oracle.ref_auc_of_reduce(d, op, 0)
        got, pops, _, runs = H.door_integrate(d, "auc", op, 0)
        assert got == want or (np.isnan(got) and np.isnan(want))
        assert pops == len(oracle.reduce(d, op)[0])         # (the constructor primes with the first run, wiggleIterator.c:32)
    t2 = random_case(8951, n_tracks=2, max_len=3000)
    got, pops, _, _ = H.door_integrate(t2.as_dict(), "pearson")
    want = oracle.ref_pearson(t2.as_dict())
    assert got == want or (np.isnan(got) and np.isnan(want))


def test_empty_and_golden(oracle, H):
    """AUC mean fixedStep.wig variableStep.wig = 30.0, pearson = -0.028968 (test/expected/pearson.txt)."""
    import os
    from wiggletools_amd.textio import load_runlists
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    d = load_runlists([os.path.join(G, "fixedStep.wig"), os.path.join(G, "variableStep.wig")]).as_dict()
    got, _, _, _ = H.door_integrate(d, "auc", "mean", 0)
    assert got == 30.0
    got, _, _, _ = H.door_integrate(d, "pearson")
    assert abs(got - (-0.028968)) < 5e-7
    e = dict(d, seg_off=np.zeros_like(d["seg_off"]), start=d["start"][:0], finish=d["finish"][:0], value=d["value"][:0])
    got, pops, _, _ = H.door_integrate(e, "auc", "mean", 0)
    assert got == 0.0
    got, _, _, _ = H.door_integrate(e, "mean", "mean", 0)
    assert np.isnan(got)


def _auc(runs):
    c, s, f, v = runs
    ok = ~np.isnan(v)
    return float(((f.astype(np.int64) - s)[ok].astype(np.float64) * v[ok]).sum())


@pytest.mark.parametrize("pre_pops", [0, 1, 3])
def test_fused_auc_seek_per_region(oracle, H, pre_pops, monkeypatch):
    """A fused integrator that is SEEKED -- `apply`'s use of it (one seek per region), and a seek in the middle of a
    pass, when the source's pipes are still integrating on the device (the round-3 advisor's finding: the priming pop of
    the seek read a batch that had shipped no runs and crashed).  The sums go on across seeks (statistics.c:38-43):
    value after region k = value before the first seek + the AUC of the reducer over the regions so far (oracle over
    the clipped tracks; the reference itself is not consulted for a seek on a primed Multiplexer)."""
    from test_dropin import clip
    monkeypatch.setenv("WTAMD_MIN_SPAN", "64")         # many batches: the pass is under way when the seek comes
    monkeypatch.setenv("WTAMD_BATCH_INTERVALS", "200")
    t = random_case(8900, n_tracks=6, n_chrom=2, max_len=9000)
    d = t.as_dict()
    regions = [(0, 100, 3000), (0, 2500, 2600), (1, 1, 50), (1, 10, 4000)]
    for op in ("mean", "max"):
        got = H.door_integrate_seek(d, "auc", regions, op=op, pre_pops=pre_pops)
        want = got[0]
        # (got[0]: what the constructor's own pop and the pre_pops absorbed -- whole batches, not runs)
        for k, (c, s, f) in enumerate(regions):
            want += _auc(oracle.reduce(clip(t, c, s, f).as_dict(), op))
            assert _close(got[1 + k], want), (op, pre_pops, k, got[1 + k], want)


def test_fused_pearson_seek(oracle, H, monkeypatch):
    """wtamd_PearsonIntegrator seeked mid-stream and per region: the moments go on across seeks (statistics.c:406-410),
    so the value is the correlation over everything absorbed so far -- whole batches here, single runs in the reference --
    and has no closed expectation; what is checked is that every seek re-primes cleanly (no batch without runs is read)
    and that the result stays a correlation.  Seeking a region TWICE in a row doubles every weight and leaves the
    correlation of that region alone."""
    monkeypatch.setenv("WTAMD_MIN_SPAN", "64")
    monkeypatch.setenv("WTAMD_BATCH_INTERVALS", "200")
    t = random_case(8950, n_tracks=2, n_chrom=2, max_len=9000)
    d = t.as_dict()
    for pre in (0, 2):
        got = H.door_integrate_seek(d, "pearson", [(0, 100, 3000), (1, 10, 4000), (1, 10, 4000)], pre_pops=pre)
        assert np.all(np.isfinite(got[1:])) and np.all(np.abs(got[1:]) <= 1.0 + 1e-12), got
