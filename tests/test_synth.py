"""The synthetic workload generator (bench plumbing): the HIP kernels and their numpy
mirror produce the same tracks, the tracks honour the run-list contract, and a region regenerated
on its own equals the same region of the whole."""
import numpy as np
import pytest


def test_mirror_region_consistency():
    from wiggletools_amd import synthgen as synth
    whole = synth.region_runs(77, 0, 3, 50000, 0, 50000, 16.0, 0.02, 800)
    part = synth.region_runs(77, 0, 3, 50000, 12000, 30000, 16.0, 0.02, 800)
    m = (whole[0] - 1 >= 12000) & (whole[0] - 1 < 30000)
    for a, b in zip(part, whole):
        assert np.array_equal(a, b[m])
    s, f, v = whole
    assert (f > s).all() and (s[1:] >= f[:-1]).all() and f[-1] <= 50001
    assert abs(np.mean(np.diff(np.concatenate([[1], f])) ) - 16.0) < 1.0        # mean run ~ l (gaps merge into the next diff)
    assert 0.01 < 1 - (f - s).sum() / 50000 < 0.03                               # ~2 % of the bp dropped
    assert set(np.unique(v * 8) % 1) == {0.0} and v.max() < 100.0
    dense = synth.region_runs(5, 0, 0, 1000, 0, 1000, 1.0, 0.0, 800)
    assert len(dense[0]) == 1000 and (dense[1] - dense[0] == 1).all()


@pytest.mark.gpu
def test_device_generator_equals_mirror():
    import torch
    from wiggletools_amd import engine, synthgen as synth
    lens = [70001, 4096, 33]
    ids = [4, 9, 1]
    for mean_run in (1.0, 16.0, 200.0):
        seg, s, f, v = synth.device_tracks(20260927, lens, 7, mean_run, 0.02, 800, chrom_ids=ids)
        h = synth.host_runlists(20260927, lens, 7, mean_run, 0.02, 800, chrom_ids=ids)
        assert np.array_equal(seg, h.seg_off)
        assert np.array_equal(s.cpu().numpy(), h.start) and np.array_equal(f.cpu().numpy(), h.finish)
        assert np.array_equal(v.cpu().numpy(), h.value)
        ts = engine.TrackSet.from_device(len(lens), 7, seg, s, f, v, np.zeros(7))
        assert ts.validate() == (0, -1)
