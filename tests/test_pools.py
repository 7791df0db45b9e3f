"""The process-wide pools behind the pipes (csrc/wt_pipe.h: page-locked staging and device buffers): a second reducer of the
same shape in one process must not page-lock or map anything again, and must give the same runs."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tracks(n_tracks, length, seed):
    rng = np.random.default_rng(seed)
    out = []
    for t in range(n_tracks):
        lens = rng.geometric(1 / 16.0, int(length / 16 * 1.1))
        ends = np.cumsum(lens)
        ends = ends[ends < length]
        s = np.concatenate([[1], ends[:-1] + 1]).astype(np.int32)
        f = (ends + 1).astype(np.int32)
        v = (rng.integers(0, 800, len(s)) / 8.0).astype(np.float32)
        out.append((s, f, v))
    return out


def test_second_reducer_allocates_nothing():
    import torch
    assert torch.cuda.is_available()
    from wiggletools_amd import _lib, dropin
    L = _lib.lib()
    tracks = _tracks(8, 24_000_000, 5)          # 1.5 M intervals per track: staging well above the pools' 1 MB threshold
    keep, readers_args = [], []
    for s, f, v in tracks:
        ps, pf, pv = dropin.PinnedArray(len(s), np.int32), dropin.PinnedArray(len(s), np.int32), dropin.PinnedArray(len(s), np.float32)
        ps.array[:] = s; pf.array[:] = f; pv.array[:] = v
        keep.append((ps, pf, pv))
        readers_args.append((["chr1"], [0, len(s)], ps.ptr, pf.ptr, pv.ptr))

    def stats():
        a = (C.c_int64 * 6)()
        L.wtamd_pool_stats(a)
        return list(a)

    def run():
        r = dropin.reducer("mean", [dropin.array_reader(*a) for a in readers_args])
        sums = []
        runs, bp = dropin.drain_blocks(r, on_block=lambda c, a, b, v: (sums.append(float(np.nansum(v))), 0)[1])
        return runs, bp, float(np.sum(sums))

    first = run()
    s1 = stats()
    second = run()
    s2 = stats()
    assert second == first and first[0] > 1_000_000
    assert s1[0] > 0 and s1[3] > 0, s1                      # the first reducer did go to the runtime ...
    assert s2[0] == s1[0] and s2[1] == s1[1], (s1, s2)      # ... the second page-locked nothing
    assert s2[3] == s1[3] and s2[4] == s1[4], (s1, s2)      # ... and mapped nothing
    assert s2[2] > 0 and s2[5] > 0                          # its buffers rest in the pools again
    L.wtamd_pool_trim()
    s3 = stats()
    assert s3[2] == 0 and s3[5] == 0, s3                    # nothing rests in the pools after a trim
    third = run()
    assert third == first and stats()[3] > s3[3]            # and the next reducer goes to the runtime again
    for ps, pf, pv in keep:
        ps.free(); pf.free(); pv.free()
