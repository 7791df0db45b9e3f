"""The streaming pipeline of the C ABI (wtamd_pipe_*) driven directly: batches cut by the caller,
several in flight, results in submission order == the oracle over the whole tracks.

"emu": the API contract against the emulated pipe (CPU); "amd" (-m gpu): the product's pipe --
pinned staging, three HIP streams, events, grow-only buffers, patch after the counters."""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_runs_equal, random_case

EXACT = {"sum", "product", "mean", "min", "max", "median"}


def _lib_for(backend):
    if backend == "amd":
        import torch
        assert torch.cuda.is_available()
        return None
    from emu import build as emu_build
    return C.CDLL(emu_build.build_dropin())


@pytest.fixture(params=["emu", pytest.param("amd", marks=pytest.mark.gpu)])
def lib(request):
    return _lib_for(request.param)


@pytest.mark.parametrize("seed", range(6))
def test_pipe_matches_oracle(oracle, lib, seed):
    from wiggletools_amd.pipe import stream_runlists
    rng = np.random.default_rng(seed)
    t = random_case(8100 + seed, max_len=5000, dtype=np.float32 if seed % 2 else np.float64)
    d = t.as_dict()
    for op in ("mean", "sum", "var", "median", "max"):
        for (bp, depth) in ((int(rng.integers(5, 60)), 1), (257, 2), (100000, 3)):
            got, st = stream_runlists(t, op, bp, depth=depth, lib=lib)
            assert_runs_equal(got, oracle.reduce(d, op), 0.0 if op in EXACT else 1e-12, "%s bp %d depth %d" % (op, bp, depth))
            assert st["runs"] == len(got[0]) and st["n_slots"] == depth + 1


def test_pipe_two_sample_and_tile(oracle, lib):
    from wiggletools_amd.pipe import stream_runlists
    t = random_case(8200, n_tracks=8, max_len=4000)
    d = t.as_dict()
    for op in ("ttest", "mwu"):
        got, _ = stream_runlists(t, op, 300, n_set0=4, lib=lib)
        assert_runs_equal(got, oracle.reduce(d, op, n_set0=4), 1e-9, op)
    got, _ = stream_runlists(t, "multiplex", 300, lib=lib)
    exp = oracle.multiplex(d)
    assert len(got[0]) == len(exp[0])
    for a, b in zip((got[0], got[1], got[2], got[4], got[5]), exp):
        assert np.array_equal(a, b, equal_nan=True)
    assert np.array_equal(got[3], exp[4].sum(axis=1))       # the run's inplay_count (multiplexer.h:27)


def test_pipe_difference_array_with_patched_windows(oracle, lib, monkeypatch):
    """Float tracks, zero defaults: the exact difference-array kernel runs; NaN / Inf / a wide
    dynamic range in a few windows make it hand those windows to the patch kernel, which the pipe
    launches once the batch's counters are back -- before the runs are shipped."""
    from wiggletools_amd.pipe import stream_runlists
    from wiggletools_amd.runlists import synth
    monkeypatch.setenv("WTAMD_DELTA_MIN_TRACKS", "1")
    t = synth(6, [240000], mean_run=9, seed=3, dtype=np.float32)        # (8192-bp windows: a batch needs > 4 per inexact one)
    v = t.value
    v[100] = np.nan
    v[len(v) // 2] = np.inf
    v[len(v) // 3] = 2.0 ** -120
    v[len(v) // 3 + 1] = 2.0 ** 100
    d = t.as_dict()
    for op in ("sum", "mean"):
        got, st = stream_runlists(t, op, 80000, depth=2, lib=lib)
        assert_runs_equal(got, oracle.reduce(d, op), 0.0, op)
        assert st["delta_batches"] == st["batches"] > 1


def test_pipe_device_side_compression(oracle, lib):
    """WTAMD_PIPE_COMPRESS: every batch is merged on the device by CompressionWiggleIterator's rule, except
    for its head (the runs before the first run that leads a group whatever preceded it: they may
    belong to the previous batch's last group).  Contract: a consumer that applies the reference's
    wrapper to this output gets what it gets on the uncompressed runs -- checked with the oracle's
    restatement of the wrapper (itself pinned on the compiled reference, tests/test_oracle_vs_ref.py),
    for batches of any size."""
    from wiggletools_amd.pipe import stream_runlists
    from wiggletools_amd.runlists import synth
    t = synth(3, [9000, 700], mean_run=4, gap_prob=0.05, seed=8, value_levels=2, dtype=np.float64)
    rng = np.random.default_rng(1)
    t.value[:] = t.value + rng.integers(0, 3, len(t.value)) * 4e-7
    t.value[50:60] = np.nan
    d = t.as_dict()
    for op in ("mean", "max"):
        plain = oracle.reduce(d, op)
        exp = oracle.compress(*plain)
        for bp in (1 << 20, 700, 97):
            got, st = stream_runlists(t, op, bp, depth=2, lib=lib, compress=True)
            assert_runs_equal(oracle.compress(*got), exp, 0.0, "%s batches of %d bp" % (op, bp))
            assert len(exp[0]) <= len(got[0]) < len(plain[0])
            assert int((got[2] - got[1]).sum()) == int((plain[2] - plain[1]).sum())      # same coverage


def test_pipe_contract_errors(lib):
    from wiggletools_amd import _lib
    from wiggletools_amd.pipe import Pipe
    p = Pipe(3, "mean", n_slots=2, lib=lib, max_runs=1000)
    with pytest.raises(_lib.WtamdError):
        p.collect()                         # nothing in flight
    p.acquire()
    with pytest.raises(_lib.WtamdError):
        p.acquire()                         # one slot at a time
    so, ss, sf, v32, v64 = p.staging()
    assert v64 is None
    so[:] = 0
    p.submit(1, 100)                        # an empty batch is legal
    assert p.in_flight() == 1
    s, f, v = p.collect()
    assert len(s) == 0
    with pytest.raises(_lib.WtamdError):
        p.collect()                         # previous result not released
    p.release()
    with pytest.raises(_lib.WtamdError):
        p.release()
    p.acquire(); p.cancel()
    p.close()
    with pytest.raises(_lib.WtamdError):
        Pipe(4, "ttest", n_set0=2, lib=lib) if lib is None else (_ for _ in ()).throw(_lib.WtamdError("emu skips the descriptor check"))
