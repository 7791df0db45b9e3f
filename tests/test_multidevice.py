"""WTAMD_DEVICES: one pipeline per GPU inside the drop-in layer -- the Feeder deals batches (chromosome x run-start
range: the reference's sharding unit, python/wiggletools/parallelWiggleTools.py:63-68,103-113) round robin to the
pipes and collects them in submission order, which is (strcmp(chrom), start) order (multiplexer.c:56).  Results must
be bit-identical to one device.  CPU: the host logic over 2 / 3 emulated pipes; `-m gpu`: two pipes on the one GPU
of the test box (a count above the number of devices wraps around -- every pipe still has its own streams, slots
and staging, which is what the dealing logic has to get right)."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import random_case
from test_bwreader import _bind, _blocks, _reduce, _write_set
from test_dropin import ALL_MULTIPLEX_OPS, _get, _tol
from helpers import assert_runs_equal


@pytest.fixture(params=["emu", pytest.param("amd", marks=pytest.mark.gpu)])
def backend(request):
    return request.param


@pytest.mark.parametrize("devices", ["2", "3"])
def test_reducers_on_several_pipes(oracle, backend, devices, monkeypatch):
    """test_dropin_tiny_batches with the batches dealt to 2 / 3 pipes: hundreds of seams, every reducer."""
    H = _get(oracle, backend)
    monkeypatch.setenv("WTEMU_DEVICES", devices)
    monkeypatch.setenv("WTAMD_DEVICES", devices)
    monkeypatch.setenv("WTAMD_MIN_SPAN", "7")
    monkeypatch.setenv("WTAMD_BATCH_INTERVALS", "20")
    monkeypatch.setenv("WTEMU_PIPE_CAP", "3")
    for seed in range(3):
        t = random_case(7700 + seed, max_len=1500)
        d = t.as_dict()
        for op in ALL_MULTIPLEX_OPS:
            for strict in (0, 1):
                exp = oracle.reduce(d, op, flags=strict)
                got = H.reduce(d, op, flags=strict)
                assert_runs_equal(got, exp, _tol(op), "devices %s seed %d op %s strict %d" % (devices, seed, op, strict))
        if t.n_tracks >= 2:
            n0 = t.n_tracks // 2
            exp = oracle.reduce(d, "mwu", n_set0=n0)
            got = H.reduce(d, "mwu", n_set0=n0)
            assert_runs_equal(got, exp, 1e-12, "mwu")


def test_integrator_doors_on_two_pipes(oracle, backend, monkeypatch):
    if not oracle.have_ref():
        pytest.skip("compiled reference not available")
    H = _get(oracle, backend)
    monkeypatch.setenv("WTEMU_DEVICES", "2")
    monkeypatch.setenv("WTAMD_DEVICES", "2")
    monkeypatch.setenv("WTAMD_MIN_SPAN", "64")
    monkeypatch.setenv("WTAMD_BATCH_INTERVALS", "200")
    t = random_case(7801, max_len=5000)
    d = t.as_dict()
    want = oracle.ref_auc_of_reduce(d, "mean", 0)
    got, pops, d2h, runs = H.door_integrate(d, "auc", "mean", 0)
    assert abs(got - want) <= 1e-9 * max(1.0, abs(want)) or (np.isnan(got) and np.isnan(want))


def test_bigwig_files_on_two_pipes(oracle, backend, tmp_path, monkeypatch):
    """File-byte batches (device-side inflate) dealt to two pipes, with read-ahead: == one pipe, run for run."""
    if backend == "emu":
        from emu.build import build_dropin
        L = _bind(C.CDLL(build_dropin()))
    else:
        from wiggletools_amd import _lib
        L = _bind(_lib.lib())
    paths = _write_set(tmp_path, 5, seed=41, block=37)
    monkeypatch.setenv("WTAMD_BW_BATCH_BYTES", "3000")
    monkeypatch.setenv("WTAMD_BATCH_RUNS", "1500")
    res = {}
    for devices in ("1", "2"):
        monkeypatch.setenv("WTEMU_DEVICES", devices)
        monkeypatch.setenv("WTAMD_DEVICES", devices)
        wi, keep = _reduce(L, paths, "MeanReduction")
        res[devices] = _blocks(L, wi)
    assert len(res["1"]) > 1000 and res["1"] == res["2"]
