"""Size-independent properties at the bench's own size (BASELINE config C2: mean / sum over 100
float tracks, mean run 16 bp, GRCh38 x 1/8 = 386 Mbp, 2.4e9 input runs, 3.9e8 output runs) -- far
beyond what the oracle can finish, so the checks are properties, not comparisons with it:

 * two independent algorithms agree bit for bit: the exact difference-array kernel and the general
   bitmap kernel (same run count, same coordinates, same f64 bit patterns);
 * run-list invariants: start < finish, sorted and non-overlapping inside every chromosome;
 * a checksum of checksums: AUC(sum over tracks) == sum over tracks of AUC(track).  All values are
   k/8, so both sides are exact in f64 whatever the summation order, and must be EQUAL;
 * the covered-bp counter equals the sum of the run lengths.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize("scale", [0.125])
def test_gpu_full_size_properties(scale, monkeypatch):
    import torch
    import bench
    from wiggletools_amd import synthgen
    from wiggletools_amd import engine

    dev = torch.device("cuda", 0)
    N = 100
    chrom_lens = [max(int(x * scale), 1) for x in bench.GRCH38]
    seg_off, start, finish, value = synthgen.device_tracks(7, chrom_lens, N, 16.0, 0.02, 800, dev)
    stream = torch.cuda.current_stream().cuda_stream
    ts0 = engine.TrackSet.from_device(len(chrom_lens), N, seg_off, start, finish, value, np.zeros(N))
    assert ts0.validate() == (0, -1), "generator produced runs that violate the input contract"
    ts0.close()

    def run(op, no_delta):
        if no_delta:
            monkeypatch.setenv("WTAMD_NO_DELTA", "1")
        else:
            monkeypatch.delenv("WTAMD_NO_DELTA", raising=False)
        ts = engine.TrackSet.from_device(len(chrom_lens), N, seg_off, start, finish, value, np.zeros(N))
        out = ts.alloc_runs()
        n = ts.reduce(op, out, stream=stream, sync=True)
        st = ts.stats()
        ts.close()
        return out, n, st

    a, na, sta = run("sum", no_delta=False)
    assert sta["kernel"] == 1, "difference-array kernel expected for float tracks with zero defaults"
    b, nb, stb = run("sum", no_delta=True)
    assert stb["kernel"] == 0
    assert na == nb and na > 3e8
    assert torch.equal(a.start[:na], b.start[:nb]) and torch.equal(a.finish[:na], b.finish[:nb])
    assert torch.equal(a.value[:na].view(torch.int64), b.value[:nb].view(torch.int64)), "kernels disagree bitwise"
    assert torch.equal(a.chrom_run_off, b.chrom_run_off)
    del b

    # run-list invariants
    s, f = a.start[:na], a.finish[:na]
    assert bool((s < f).all())
    cro = a.chrom_run_off.cpu().numpy()
    gap_ok = s[1:] >= f[:-1]
    first = torch.zeros(na - 1, dtype=torch.bool, device=dev)
    idx = torch.as_tensor(cro[1:-1], device=dev) - 1          # pairs straddling a chromosome boundary
    idx = idx[(idx >= 0) & (idx < na - 1)]
    first[idx] = True
    assert bool((gap_ok | first).all()), "runs overlap or are out of order inside a chromosome"
    covered = int((f.to(torch.int64) - s.to(torch.int64)).sum().item())
    assert covered == sta["covered_bp"]

    # checksum of checksums, exact: AUC(sum over tracks) == sum over tracks of AUC(track)
    lens = finish.to(torch.int64) - start.to(torch.int64)
    k8 = (value.to(torch.float64) * 8).to(torch.int64)
    expect = int((lens * k8).sum().item())                      # in units of 1/8
    a.n = na
    auc = a.auc(stream=stream)
    assert auc * 8 == float(expect) and expect < 2 ** 53, (auc, expect / 8)

    # a handful of values the difference-array kernel cannot take (NaN, Inf, a 2^-120 speck): those
    # windows are patched by the general kernel -- and the whole output still equals the general
    # kernel's, bit for bit, NaN runs included
    g = torch.Generator(device=dev); g.manual_seed(1)
    where = torch.randint(0, value.numel(), (12,), device=dev, generator=g)
    saved = value[where].clone()
    value[where] = torch.tensor([float("nan"), float("inf"), 2.0 ** -120] * 4, device=dev, dtype=value.dtype)
    p, np_, stp = run("sum", no_delta=False)
    assert stp["kernel"] == 1 and 1 <= stp["patched_windows"] <= 12, stp
    q, nq, stq = run("sum", no_delta=True)
    assert np_ == nq == na
    assert torch.equal(p.start[:np_], q.start[:nq]) and torch.equal(p.finish[:np_], q.finish[:nq])
    assert torch.equal(p.value[:np_].view(torch.int64), q.value[:nq].view(torch.int64)), "patched windows differ"
    assert int(torch.isnan(p.value[:np_]).sum().item()) > 0
    value[where] = saved
    del p, q

    # mean = sum / N, the reference's own expression (reducers.c:399-400)
    m, nm, stm = run("mean", no_delta=False)
    assert nm == na and stm["kernel"] == 1
    # (a tensor divisor: torch turns division by a Python scalar into a multiplication by 1/N)
    assert torch.equal(m.value[:nm], a.value[:na] / torch.full((na,), float(N), dtype=torch.float64, device=dev))


@pytest.mark.parametrize("op,kw", [("max", {}), ("var", {}), ("cv", {}), ("ttest", dict(n_set0=50)),
                                   ("median", {}), ("mwu", dict(n_set0=50))])
def test_gpu_plans_agree_bitwise_at_scale(op, kw, monkeypatch):
    """The same reduction through different execution plans of the general kernel -- 4 positions
    per lane vs 1, all tracks resident vs chunks of 37 (bitmaps rebuilt per chunk and pass), narrow
    vs wide workgroups -- must give identical bits: 100 tracks, 62 Mbp (31 Mbp for median / MWU)."""
    import torch
    import bench
    from wiggletools_amd import synthgen
    from wiggletools_amd import engine

    dev = torch.device("cuda", 0)
    N = 100
    scale = 0.01 if op in ("median", "mwu") else 0.02
    chrom_lens = [max(int(x * scale), 1) for x in bench.GRCH38]
    seg_off, start, finish, value = synthgen.device_tracks(3, chrom_lens, N, 16.0, 0.02, 800, dev)
    stream = torch.cuda.current_stream().cuda_stream
    plans = [{}, {"WTAMD_CHUNK": "37"}]
    plans.append({"WTAMD_T": "64"} if op in ("median", "mwu") else {"WTAMD_PPT": "1", "WTAMD_T": "256"})
    if op == "max":
        # round 6: Max / Min default to the difference-array kernel's segment tree; the general kernel's plans with it switched off
        plans = [dict(p, WTAMD_NO_DELTA_MINMAX="1") for p in plans] + [{}]
    if op == "ttest":
        # round 6: TTestReduction's default is the difference-array kernel (exact integer sums per set); the plans of the general
        # kernel with it switched off -- and the default beside them: the generator's k/8 values make every route's sums exact,
        # so t, the degrees of freedom and with them the tail agree bit for bit
        plans = [dict(p, WTAMD_NO_DELTA_TTEST="1") for p in plans] + [{}]
    ref = None
    for env in plans:
        for k in ("WTAMD_CHUNK", "WTAMD_PPT", "WTAMD_T", "WTAMD_NO_DELTA_TTEST", "WTAMD_NO_DELTA_MINMAX"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ts = engine.TrackSet.from_device(len(chrom_lens), N, seg_off, start, finish, value, np.zeros(N))
        out = ts.alloc_runs()
        n = ts.reduce(op, out, stream=stream, sync=True, **kw)
        if op in ("ttest", "max"):
            assert ts.stats()["kernel"] == (0 if ("WTAMD_NO_DELTA_TTEST" in env or "WTAMD_NO_DELTA_MINMAX" in env) else 1), (env, ts.stats())
        ts.close()
        got = (n, out.start[:n].clone(), out.finish[:n].clone(), out.value[:n].view(torch.int64).clone())
        del out
        if ref is None:
            ref = got
            assert n > 1e7
        else:
            assert got[0] == ref[0], (env, got[0], ref[0])
            assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), env
            assert torch.equal(got[3], ref[3]), "plan %s changes the value bits of %s" % (env, op)


@pytest.mark.parametrize("op,N,scale", [("var", 100, 0.02), ("stddev", 100, 0.02), ("cv", 100, 0.02),
                                        ("var", 500, 0.01), ("stddev", 500, 0.01)])
def test_gpu_var_family_two_algorithms_at_scale(op, N, scale, monkeypatch):
    """Variance family at 62 Mbp x 100 tracks -- and at BASELINE config 3's width, 500 tracks x 31 Mbp (the 768-lane layout of
    the launches with squares; the general kernel in chunks of tracks) -- through two independent algorithms: exact integer sums of
    the scaled mantissas and of their squares with a 128-bit finish (difference arrays)
    vs the reference's two sequential f64 passes per position (general kernel).  Coordinates and run
    count identical; values within 1e-12 relative (the reference-order result carries ~N roundings,
    the integer route one)."""
    import torch
    import bench
    from wiggletools_amd import engine, synthgen

    dev = torch.device("cuda", 0)
    chrom_lens = [max(int(x * scale), 1) for x in bench.GRCH38]
    seg_off, start, finish, value = synthgen.device_tracks(5, chrom_lens, N, 16.0, 0.02, 800, dev)
    stream = torch.cuda.current_stream().cuda_stream
    res = []
    for no_delta in (False, True):
        if no_delta:
            monkeypatch.setenv("WTAMD_NO_DELTA_VAR", "1")
        else:
            monkeypatch.delenv("WTAMD_NO_DELTA_VAR", raising=False)
        ts = engine.TrackSet.from_device(len(chrom_lens), N, seg_off, start, finish, value, np.zeros(N))
        out = ts.alloc_runs()
        n = ts.reduce(op, out, stream=stream, sync=True)
        assert ts.stats()["kernel"] == (0 if no_delta else 1)
        ts.close()
        res.append((n, out.start[:n].clone(), out.finish[:n].clone(), out.value[:n].clone()))
        del out
    (na, sa, fa, va), (nb, sb, fb, vb) = res
    assert na == nb and na > 1e7
    assert torch.equal(sa, sb) and torch.equal(fa, fb)
    assert torch.equal(torch.isnan(va), torch.isnan(vb))
    ok = ~torch.isnan(va)
    err = (va[ok] - vb[ok]).abs()
    tol = 1e-12 * torch.maximum(vb[ok].abs(), torch.tensor(1e-300, dtype=torch.float64, device=dev))
    assert bool((err <= tol).all()), float((err / tol).max())
