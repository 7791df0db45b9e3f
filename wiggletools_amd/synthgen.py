"""Synthetic workload of SURVEY 8d (bench / test plumbing): counter-based run lists.

`device_tracks` drives the HIP generator (csrc/wt_synth.hip) and returns torch CUDA tensors in the
wtamd_tracks layout; `region_runs` is its numpy mirror for one (chromosome, track, region) -- the
same hash, so the CPU baseline (bench.py) regenerates exactly the tracks the GPU processed without
moving them, and the tests pin the generator itself.
"""
import ctypes as C

import numpy as np

from . import _lib

GOLD = np.uint64(0x9E3779B97F4A7C15)
M1 = np.uint64(0xBF58476D1CE4E5B9)
M2 = np.uint64(0x94D049BB133111EB)
RUNK = np.uint64(0xD6E8FEB86659FD93)
MASK32 = np.uint64(0xFFFFFFFF)


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * M1
    z = (z ^ (z >> np.uint64(27))) * M2
    return z ^ (z >> np.uint64(31))


def _base(seed, c, t):
    with np.errstate(over="ignore"):
        return _mix(np.uint64(seed) ^ (np.uint64(c) << np.uint64(40)) ^ (np.uint64(t) << np.uint64(8)) ^ np.uint64(0x5bd1e995))


def _thresholds(mean_run, gap_prob):
    bp = 0 if mean_run <= 1.0 else int(4294967296.0 / mean_run)
    return np.uint64(bp), np.uint64(int(4294967296.0 * gap_prob))


def region_runs(seed, c, t, chrom_len, lo, hi, mean_run=16.0, gap_prob=0.02, levels=800):
    """Runs of (chromosome c, track t) whose 0-based start lies in [lo, hi): (start, finish, value)
    with 1-based inclusive start / exclusive finish; the last run keeps its true finish."""
    bp_t, gap_t = _thresholds(mean_run, gap_prob)
    base = _base(seed, c, t)
    hi = min(hi, chrom_len)
    with np.errstate(over="ignore"):
        def hashes(a, b):
            x = np.arange(a, b, dtype=np.uint64)
            return _mix(base + GOLD * (x + np.uint64(1)))

        def is_bp(a, b):
            h = hashes(a, b)
            m = ((h & MASK32) < bp_t) if bp_t else np.ones(b - a, bool)
            if a == 0 and b > 0:
                m[0] = True
            return m, h

        m, h = is_bp(lo, hi)
        pos = np.nonzero(m)[0] + lo
        if len(pos) == 0:
            return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32)
        # finish of the last run: the first breakpoint at or after hi
        end, a = chrom_len, hi
        while a < chrom_len:
            b = min(a + 4096, chrom_len)
            mm, _ = is_bp(a, b)
            nz = np.nonzero(mm)[0]
            if len(nz):
                end = a + int(nz[0])
                break
            a = b
        fin = np.concatenate([pos[1:], [end]])
        g = _mix(h[pos - lo] ^ RUNK)
        keep = (g & MASK32) >= gap_t
        val = ((g >> np.uint64(40)) & MASK32) % np.uint64(levels)
    return ((pos[keep] + 1).astype(np.int32), (fin[keep] + 1).astype(np.int32),
            (val[keep].astype(np.float32) * np.float32(0.125)))


def host_runlists(seed, chrom_lens, n_tracks, mean_run=16.0, gap_prob=0.02, levels=800, region=None, chrom_ids=None):
    """numpy mirror of device_tracks as a RunLists (small cases / CPU baseline samples).
    region: optional (lo, hi) of 0-based run starts applied to every chromosome."""
    from .runlists import RunLists
    S, F, V, so = [], [], [], [0]
    ids = list(range(len(chrom_lens))) if chrom_ids is None else list(chrom_ids)
    for c, clen in enumerate(chrom_lens):
        lo, hi = (0, clen) if region is None else region
        cseed = chrom_seed(seed, ids[c])
        for t in range(n_tracks):
            s, f, v = region_runs(cseed, 0, t, clen, lo, hi, mean_run, gap_prob, levels)
            S.append(s); F.append(f); V.append(v)
            so.append(so[-1] + len(s))
    return RunLists(len(chrom_lens), n_tracks, so, np.concatenate(S), np.concatenate(F), np.concatenate(V))


def _bind():
    L = _lib.lib()
    if not getattr(L, "_wt_synth_bound", False):
        L.wtamd_synth_plan.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_void_p]
        L.wtamd_synth_count.argtypes = [C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int,
                                        C.c_void_p, C.c_void_p]
        L.wtamd_synth_fill.argtypes = [C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L._wt_synth_bound = True
    return L


def device_tracks(seed, chrom_lens, n_tracks, mean_run=16.0, gap_prob=0.02, levels=800, device=None, chrom_ids=None):
    """(seg_off[np.int64], start, finish, value) with the three arrays as torch CUDA tensors.
    chrom_ids: the generator's chromosome index of every entry of chrom_lens (default 0..n-1), so a
    work item holding only chromosome 7 gets chromosome 7's runs."""
    import torch
    L = _bind()
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    n_chrom = len(chrom_lens)
    ids = list(range(n_chrom)) if chrom_ids is None else list(chrom_ids)
    S, F, V, seg = [], [], [], [0]
    stream = torch.cuda.current_stream().cuda_stream
    # the kernels key their hash on the position of a chromosome in the list they are given: one
    # launch per chromosome with `seed` folded with the chromosome's own id keeps items independent
    for k, clen in enumerate(chrom_lens):
        lens = np.ascontiguousarray([clen], np.int32)
        cseed = chrom_seed(seed, ids[k])
        nb = C.c_int64()
        first = np.zeros(n_tracks + 1, np.int64)
        _lib.check(L.wtamd_synth_plan(1, lens.ctypes.data, n_tracks, C.byref(nb), first.ctypes.data))
        counts = torch.empty(nb.value, dtype=torch.int64, device=dev)
        _lib.check(L.wtamd_synth_count(cseed, 1, lens.ctypes.data, n_tracks, mean_run, gap_prob, levels, counts.data_ptr(), stream))
        incl = torch.cumsum(counts, 0)
        off = incl - counts
        total = int(incl[-1].item())
        s = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        f = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        v = torch.empty(max(total, 1), dtype=torch.float32, device=dev)
        _lib.check(L.wtamd_synth_fill(cseed, 1, lens.ctypes.data, n_tracks, mean_run, gap_prob, levels, off.data_ptr(),
                                      s.data_ptr(), f.data_ptr(), v.data_ptr(), stream))
        firsts = torch.from_numpy(first[:-1]).to(dev)
        so = np.append(off[firsts].cpu().numpy(), total)     # first run of every track inside this chromosome
        base = seg[-1]
        for t in range(n_tracks):
            seg.append(base + int(so[t + 1]))
        S.append(s[:total]); F.append(f[:total]); V.append(v[:total])
        del counts, incl, off
    if n_chrom == 1:
        return np.array(seg, np.int64), S[0], F[0], V[0]
    return np.array(seg, np.int64), torch.cat(S), torch.cat(F), torch.cat(V)


def chrom_seed(seed, chrom_id):
    """Seed of one chromosome's tracks: work items are generated independently of each other."""
    with np.errstate(over="ignore"):
        return int(_mix(np.uint64(seed) + GOLD * np.uint64(chrom_id + 1)))
