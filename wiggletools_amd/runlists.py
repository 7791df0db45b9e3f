"""Host-side container for N tracks of sorted, non-overlapping runs (SoA).

This is the layout include/wiggletools_amd.h documents for ``wtamd_tracks``:
for chromosome ``c`` (index in strcmp order, cf. reference multiplexer.c:56) and
track ``i`` the runs are the slice ``seg_off[c*N+i] : seg_off[c*N+i+1]`` of the
parallel arrays ``start`` (1-based, inclusive), ``finish`` (exclusive), ``value``.
"""
import numpy as np


class RunLists:
    def __init__(self, n_chrom, n_tracks, seg_off, start, finish, value, defaults=None, chrom_names=None):
        self.n_chrom = int(n_chrom)
        self.n_tracks = int(n_tracks)
        self.seg_off = np.ascontiguousarray(seg_off, np.int64)
        self.start = np.ascontiguousarray(start, np.int32)
        self.finish = np.ascontiguousarray(finish, np.int32)
        self.value = np.ascontiguousarray(value)
        if self.value.dtype not in (np.float32, np.float64):
            self.value = self.value.astype(np.float64)
        self.defaults = (np.zeros(self.n_tracks, np.float64) if defaults is None
                         else np.ascontiguousarray(defaults, np.float64))
        self.chrom_names = chrom_names or ["c%05d" % c for c in range(self.n_chrom)]
        assert len(self.seg_off) == self.n_chrom * self.n_tracks + 1
        assert len(self.defaults) == self.n_tracks

    @classmethod
    def from_lists(cls, tracks, defaults=None, chrom_names=None, dtype=np.float64):
        """tracks[i][c] = iterable of (start, finish, value); every track lists every chromosome."""
        n_tracks = len(tracks)
        n_chrom = len(tracks[0]) if n_tracks else 0
        seg_off = [0]
        s, f, v = [], [], []
        for c in range(n_chrom):
            for i in range(n_tracks):
                for (a, b, x) in tracks[i][c]:
                    s.append(a); f.append(b); v.append(x)
                seg_off.append(len(s))
        return cls(n_chrom, n_tracks, seg_off, np.array(s, np.int32), np.array(f, np.int32),
                   np.array(v, dtype), defaults, chrom_names)

    @property
    def n_intervals(self):
        return int(self.seg_off[-1])

    def as_dict(self):
        """Form consumed by oracle/oracle.py (values widened to float64)."""
        return dict(n_chrom=self.n_chrom, n_tracks=self.n_tracks, seg_off=self.seg_off,
                    start=self.start, finish=self.finish, value=self.value.astype(np.float64),
                    defaults=self.defaults)

    def subset(self, track_ids):
        """New RunLists with the given tracks (in that order)."""
        N = self.n_tracks
        seg_off = [0]
        s, f, v = [], [], []
        for c in range(self.n_chrom):
            for i in track_ids:
                lo, hi = self.seg_off[c * N + i], self.seg_off[c * N + i + 1]
                s.append(self.start[lo:hi]); f.append(self.finish[lo:hi]); v.append(self.value[lo:hi])
                seg_off.append(seg_off[-1] + int(hi - lo))
        cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
        return RunLists(self.n_chrom, len(track_ids), seg_off, cat(s, np.int32), cat(f, np.int32),
                        cat(v, self.value.dtype), self.defaults[list(track_ids)], self.chrom_names)


def synth(n_tracks, chrom_lens, mean_run=16, gap_prob=0.02, seed=20260927, dtype=np.float32,
          nan_prob=0.0, defaults=None, value_levels=800, first_start=1):
    """Synthetic tracks after SURVEY.md 8d: run length ~ Geometric(1/mean_run) (>=1),
    value = k/8 with k ~ U{0..value_levels-1} (exact in f32/f64, ties on purpose),
    a run is dropped (gap) with probability gap_prob."""
    rng = np.random.default_rng(seed)
    seg_off = [0]
    S, F, V = [], [], []
    for c, clen in enumerate(chrom_lens):
        for i in range(n_tracks):
            if clen <= 0:
                seg_off.append(seg_off[-1]); continue
            est = int(clen / max(mean_run, 1) * 1.2) + 16
            lens = np.empty(0, np.int64)
            while lens.sum() < clen:
                lens = np.concatenate([lens, rng.geometric(1.0 / mean_run, est).astype(np.int64)])
            ends = np.cumsum(lens)
            k = int(np.searchsorted(ends, clen, side="left")) + 1
            ends = ends[:k].copy(); ends[-1] = clen
            starts = np.concatenate([[0], ends[:-1]])
            keep = rng.random(k) >= gap_prob
            vals = rng.integers(0, value_levels, k).astype(np.float64) / 8.0
            if nan_prob > 0:
                vals[rng.random(k) < nan_prob] = np.nan
            S.append((starts[keep] + first_start).astype(np.int32))
            F.append((ends[keep] + first_start).astype(np.int32))
            V.append(vals[keep].astype(dtype))
            seg_off.append(seg_off[-1] + int(keep.sum()))
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    return RunLists(len(chrom_lens), n_tracks, seg_off, cat(S, np.int32), cat(F, np.int32), cat(V, dtype),
                    defaults)
