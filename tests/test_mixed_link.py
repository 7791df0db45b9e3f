"""Mixed link at the drop-in boundary: the reference's OWN translation units that
read `struct multiplexer_st` fields or consume reducers -- statistics.c (PearsonIntegrator :414-465,
AUCIntegrator :103-127), mWigWriter.c (TeeMultiplexer, which reads chrom / start / finish /
values[] / inplay[] / default_values / count / done directly, :182-197), wigWriter.c
(TeeWiggleIterator with CompressionWiggleIterator in front, :261-276) -- compiled unmodified
WITHOUT multiplexer.c / multiSet.c / reducers.c and linked against the drop-in library
(oracle/Makefile target `mixed`), running on top of the drop-in Multiplexer / reducers in one
process.  Everything is compared with the all-reference build: text output byte for byte.

"emu": drop-in layer over the emulated pipeline (CPU); "amd" (-m gpu): the product library.
"""
import os

import numpy as np
import pytest

from helpers import random_case


_cache = {}


@pytest.fixture(params=["emu", pytest.param("amd", marks=pytest.mark.gpu)])
def M(request, oracle):
    if not oracle.have_ref():
        pytest.skip("compiled reference not available")
    if request.param not in _cache:
        if request.param == "emu":
            from emu import build as emu_build
            emu_build.build_dropin()
        if "built" not in _cache:
            oracle.build_mixed()
            _cache["built"] = True
        p = oracle.mixed_path(request.param)
        if p is not None and request.param == "amd":
            import torch
            assert torch.cuda.is_available()
        _cache[request.param] = oracle.Harness(p, "mixed_" + request.param) if p is not None else None
    if _cache[request.param] is None:
        pytest.skip("mixed-link library not built")
    return _cache[request.param]


@pytest.mark.parametrize("seed", range(4))
def test_reference_writers_over_dropin_reducers(oracle, M, tmp_path, seed):
    """`write` (compressed, fixedStep / bedGraph mix, wigWriter.c:55-119) and `write_bg` of the
    drop-in reducers through the reference's TeeWiggleIterator == the all-reference build."""
    R = oracle.ref_harness()
    t = random_case(9600 + seed, max_len=4000, dtype=np.float32)
    d = t.as_dict()
    for op in ("mean", "sum", "max", "median"):
        for bg in (False, True):
            a = M.write_reduce(d, op, tmp_path / "a.txt", bedgraph=bg)
            b = R.write_reduce(d, op, tmp_path / "b.txt", bedgraph=bg)
            assert a == b, (op, bg)
            assert len(b) > 0 or t.n_intervals == 0


@pytest.mark.parametrize("seed", range(3))
def test_reference_mwrite_over_dropin_multiplexer(oracle, M, tmp_path, seed):
    """TeeMultiplexer reads the drop-in Multiplexer's fields per pop (mWigWriter.c:182-197)."""
    R = oracle.ref_harness()
    t = random_case(9700 + seed, n_tracks=4, max_len=3000)
    d = t.as_dict()
    for strict in (0, 1):
        for bg in (False, True):
            a = M.mwrite(d, tmp_path / "a.txt", bedgraph=bg, flags=strict)
            b = R.mwrite(d, tmp_path / "b.txt", bedgraph=bg, flags=strict)
            assert a == b, (strict, bg)


@pytest.mark.parametrize("seed", range(4))
def test_reference_integrators_over_dropin(oracle, M, seed):
    """The reference's PearsonIntegrator walks the drop-in Multiplexer (inplay[], values[],
    iters[i]->default_value, start, finish: statistics.c:432-442,464); its AUCIntegrator consumes the
    drop-in reducer.  Same arithmetic on the same per-run values: bit-identical results."""
    R = oracle.ref_harness()
    from wiggletools_amd.runlists import synth
    t = synth(2, [5000, 800], mean_run=6, gap_prob=0.1, seed=40 + seed)
    if seed % 2:
        t.defaults[:] = [0.5, -1.0]
    d = t.as_dict()
    assert M.pearson(d) == R.pearson(d)
    t = random_case(9800 + seed, max_len=3000)
    d = t.as_dict()
    for op in ("mean", "max"):
        assert M.auc_of_reduce(d, op) == R.auc_of_reduce(d, op), op


@pytest.mark.parametrize("seed", range(4))
def test_device_side_compression_under_the_reference_writer(oracle, M, tmp_path, seed, monkeypatch):
    """Writer hand-off: the reducer merges its runs on the device before they
    cross PCIe (wtamd_iterator_compress_output), the reference's TeeWiggleIterator still wraps it
    in its own CompressionWiggleIterator (wigWriter.c:263-267) -- the rule is idempotent, also
    across the batch seams where a group stays split -- and the text is byte-identical."""
    R = oracle.ref_harness()
    from wiggletools_amd.runlists import synth
    monkeypatch.setenv("WTAMD_MIN_SPAN", "300")
    monkeypatch.setenv("WTAMD_BATCH_INTERVALS", "500")
    # few levels + tiny differences: long mergeable stretches, uncertain neighbours (|dv| < 2e-6)
    t = synth(3, [6000, 900], mean_run=5, gap_prob=0.05, seed=50 + seed, value_levels=2, dtype=np.float64)
    rng = np.random.default_rng(seed)
    t.value[:] = t.value + rng.integers(0, 3, len(t.value)) * 4e-7
    d = t.as_dict()
    M.set_compress_mode(1)
    try:
        for op in ("mean", "max", "sum"):
            a = M.write_reduce(d, op, tmp_path / "a.txt")
            b = R.write_reduce(d, op, tmp_path / "b.txt")
            assert a == b, op
    finally:
        M.set_compress_mode(0)


@pytest.mark.parametrize("mop,param", [("scale", 0.5), ("offset", -1.25), ("abs", 0.0), ("gt", 1.5), ("lte", 2.0)])
def test_reference_writers_over_mapped_dropin_reducers(oracle, M, tmp_path, mop, param):
    """`write mean map <op> ...` and `mwrite map <op> ...`: the reference's writers over the drop-in reducer /
    Multiplexer whose children are wtamd_MapIterator handles (chains evaluated inside the pipeline), against
    the all-reference build with the reference's own operator iterators.  The operators whose values are
    exact on every libm: the text is byte-identical (the transcendental ones are compared as numbers in
    tests/test_dropin.py)."""
    R = oracle.ref_harness()
    t = random_case(9900, n_tracks=5, max_len=3000, dtype=np.float32)
    rng = np.random.default_rng(9)
    t.value[:] = (t.value * rng.choice([1.0, -1.0, 0.0], size=len(t.value), p=[0.6, 0.3, 0.1])).astype(np.float32)
    d = t.as_dict()
    try:
        M.set_map(mop, param); R.set_map(mop, param)
        for op in ("mean", "max"):
            for bg in (False, True):
                a = M.write_reduce(d, op, tmp_path / "a.txt", bedgraph=bg)
                b = R.write_reduce(d, op, tmp_path / "b.txt", bedgraph=bg)
                assert a == b and len(b) > 0, (mop, op, bg)
        a = M.mwrite(d, tmp_path / "a.txt", bedgraph=True, flags=0)
        b = R.mwrite(d, tmp_path / "b.txt", bedgraph=True, flags=0)
        assert a == b, mop
    finally:
        M.set_map(None); R.set_map(None)


@pytest.mark.parametrize("seed", range(3))
def test_reference_integrator_vs_fused_door(oracle, M, seed, monkeypatch):
    """`AUC mean ...`: the reference's own AUCIntegrator (statistics.c:103-127) over the drop-in MeanReduction -- it
    pops every run after a full D2H -- against wtamd_AUCIntegrator over the same reducer, which integrates on the
    device and ships no run; PearsonIntegrator likewise.  Equal to rounding."""
    monkeypatch.setenv("WTAMD_MIN_SPAN", "128")
    monkeypatch.setenv("WTAMD_BATCH_INTERVALS", "500")
    t = random_case(9900 + seed, max_len=8000, dtype=np.float32)
    d = t.as_dict()
    for op in ("mean", "sum", "median"):
        ref = M.auc_of_reduce(d, op)
        got, pops, d2h, runs = M.door_integrate(d, "auc", op)
        assert abs(got - ref) <= 1e-9 * max(1.0, abs(ref)) or (np.isnan(got) and np.isnan(ref)), (op, got, ref)
        if runs > 4000:
            assert d2h < 8 * runs          # the reference's route ships 16 bytes per run
    t2 = random_case(9950 + seed, n_tracks=2, max_len=8000, dtype=np.float32)
    ref = M.pearson(t2.as_dict())
    got, pops, _, _ = M.door_integrate(t2.as_dict(), "pearson")
    assert abs(got - ref) <= 1e-8 * max(1.0, abs(ref)) or (np.isnan(got) and np.isnan(ref))
