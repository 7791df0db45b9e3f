// wt_delta.h -- exact difference-array path for Sum / Mean over float tracks (device + -DWT_EMU).
//
// The general kernel (wt_core.h) visits every track at every run start: O(tracks x runs) f64
// adds, VALU-bound on MI355X (measured: VALUBusy 64 %, 12 % of the HBM roofline).  For
// SumReduction / MeanReduction (reference reducers.c:294-307, 375-401) the work can drop to
// O(intervals): add +v at an interval's start and -v at its finish, prefix-sum over the window.
// Floating-point prefix sums would not reproduce the reference's track-order f64 summation --
// unless NO addition rounds.  That is decidable per window:
//   * every value is a float, v = m * 2^(e-150) with |m| < 2^24 (e = biased exponent, 1..254;
//     denormals: e = 1 without the hidden bit);
//   * with emin / emax the smallest / largest exponent of the non-zero values the window touches,
//     every value is an integer multiple of q = 2^(emin-150) and every partial sum of at most N
//     of them is below N * 2^(emax-emin+24) * q;
//   * if N * 2^(emax-emin+24) <= 2^53 every partial sum -- in ANY order -- is exactly
//     representable in f64, so the reference's sequential sum is exact and equals the integer sum
//     of the scaled mantissas times q.  Bit-identical, not approximately equal.
// Windows that fail the test (wide dynamic range, NaN / Inf) still emit their runs and are recorded
// (WT_CTR_DELTA_BAD, bad_list, bad_goff); the general kernel then rewrites the values of just those
// windows (wt_patch_kernel), or of everything if they are many.  Preconditions checked by the host:
// float tracks, all default values == 0 (absent tracks then add +0.0, which never changes a
// sum), op in {SUM, MEAN}.
//
// Per window (W = 8 * workgroup size positions):
//   ranges   per chunk of T tracks: interval range of every track, scanned into one flat space
//   pass 1   exponent range of the window's values (4 B / run) -- only for a workgroup's FIRST
//            window: afterwards the unit exponent is guessed and checked on the side
//            (wt_delta_window_verdict), so value[] is read once
//   pass 2   acc[p] += +-scaled mantissa, ev[p] += start | finish<<16    (12 B / run)
//            runs spanning w0 go to the window base instead (not a breakpoint)
//   scan     one lane per 8 positions: running sum, running coverage, breakpoint byte,
//            emitted byte (breakpoint & coverage predicate & range), value of the run
//   then run-count scan, look-back, staging of the runs in LDS at their rank, coalesced copy-out.
#ifndef WT_DELTA_H_
#define WT_DELTA_H_

#define WT_DELTA_K 8            // positions per lane: one byte of the U / E bitmaps
#define WT_DELTA_GROUP 16       // lanes per group in the hierarchical scan
#ifndef WT_DELTA_PARK
#define WT_DELTA_PARK 1         // runs that cross a window edge are parked and applied once per wavefront -- by the launches with squares (see wt_delta_apply_tile)
#endif
#ifndef WT_DELTA_U
#define WT_DELTA_U 4            // flat interval indices per lane and tile (round 2, with the prefetch really in flight: 4 beats 8 by 4 % at 100 tracks, loses 1 % at 500; round 1 measured the opposite with the prefetch serialised)
#endif
#define WT_DELTA_TILE (64 * WT_DELTA_U)
#ifndef WT_DELTA_COPY2
#define WT_DELTA_COPY2 1
#endif
#ifndef WT_DELTA_ONE_TILE
#define WT_DELTA_ONE_TILE 1     // a wavefront whose only tile this is applies it without the next tile's loads in front (round 6: mean run 200 -5.5 %; 0: as before)
#endif
#define WT_DELTA_TF 2048        // tiles whose first track is tabulated (beyond: binary search)
#define WT_MAX_DELTA_T 1024     // largest workgroup of the difference-array kernels (an 8192-bp window)
#define WT_DELTA_SQ_T0 768      // workgroup (and launch bound) of the launches with squares: 12 wavefronts share the passes, the first 8 run the scans

struct WtDeltaShared {
    long long base_v;           // scaled sum of the intervals spanning w0
    int32_t base_c;             // their number
    int32_t emin, emax, bad;    // exponent range of the window's non-zero values, NaN/Inf seen
    unsigned long long base_qa, base_qb;    // squares of the intervals spanning w0 (split, see WT_DELTA_QSHIFT)
    // two-sample launches (TTestReduction, wt_delta_scan3_tt): the same four for the SECOND set, and "a position of this
    // window cancels too much for the exact sums to stand in for the reference's rounded ones" (the window is then patched)
    long long base_v1;
    unsigned long long base_qa1, base_qb1;
    int32_t base_c1, risk;
};

// Var / StdDev / CV by difference arrays (round 2).  Beside S = sum of the scaled mantissas m_i of
// the tracks in play, the window accumulates Q = sum of m_i^2 -- exactly: m_i < 2^53 / N (the span
// test), so m_i^2 < 2^106 / N^2 does not fit 64 bits; it is split at bit WT_DELTA_QSHIFT into
// a = m^2 >> 40 and b = m^2 & (2^40 - 1), accumulated in two u64 arrays (sums of at most N terms:
// < 2^66 / N and < N * 2^40; N >= 8 is required) with wrap-around subtraction at the run's finish.
// Per position, in 128-bit integers and with n = tracks in play:
//   var    (reducers.c:428-479: mean over all N, squares over the tracks IN PLAY only)
//          sum (mean - v)^2 = q^2 / N^2 * [ N^2 Q - (2N - n) S^2 ]
//   stddev (reducers.c:511-563; entropy runs the same pop, :665: absent tracks enter as 0)
//          sum (mean - x)^2 = q^2 / N   * [ N Q - S^2 ]
// so the only roundings are the final conversions -- the reference's two sequential f64 passes
// carry ~N ulp of their own; agreement is ~1e-14 relative (tests: 1e-12), coordinates bit-exact.
#define WT_DELTA_QSHIFT 40

struct WtDeltaCtx {
    long long *acc;             // [W] scaled value deltas
    uint32_t *ev;               // [W] starts (low half) | finishes (high half)
    long long *ltv;             // [T] lane totals (value)
    int32_t *ltc;               // [T] lane totals (coverage)
    long long *gtv;             // [T / 16] group totals
    int32_t *gtc;
    long long *tbase;           // [T] global index of the first interval of the chunk's track t in this window minus tpfx[t], in BYTES of a 4-byte column: byte offset = tbase[t] + 4 * flat
    uint32_t *tpfx;             // [T + 1] exclusive prefix of the tracks' interval counts (flat index space)
    uint16_t *tfirst;           // [WT_DELTA_TF] first track of every tile of the flat space
    uint32_t *tdef;             // [T] non-zero defaults (P.delta_df): the float bits of the chunk's track t's default value
    unsigned long long *qa, *qb;        // [W] each: deltas of the squares' high / low parts (delta_q launches)
    unsigned long long *ltqa, *ltqb;    // [T] lane totals
    unsigned long long *gtqa, *gtqb;    // [T / 16] group (emulator) / wave (device) totals
    WtDeltaShared *dsh;
    int32_t np;                 // accumulator entries per array: W, or 2 W with the second set's positions behind the first's (two-sample launches)
};

// per-lane registers across the scan's barriers.  Round 5: only the lane's TOTALS and what the wavefront's scan makes of them -- the
// inclusive prefixes of its 8 positions (8 x value, coverage and, with squares, two more 64-bit sums: up to 56 registers) used to be
// carried from scan 1 to scan 3; scan 3 now reads the lane's 8 positions from LDS a second time (224 bytes) and forms the prefixes as
// it goes.  The launches with squares, at 1024 lanes and 128 registers each, had 115 registers in scratch memory because of them.
// Sum / Mean keep their prefixes in registers as before (24 of them: they fit, and at mean run 200, where the scans are a fifth of a
// window, the second read cost 7 %).
struct WtDeltaLane {
    long long pv[WT_DELTA_K];   // Sum / Mean: inclusive prefix of the lane's value deltas
    int32_t pc[WT_DELTA_K];     // Sum / Mean: inclusive prefix of the lane's coverage deltas
    uint32_t evmask;            // Sum / Mean: bit k: position k is a true breakpoint
    long long tv;               // sum of the lane's value deltas
    int32_t tc;                 // ... of its coverage deltas
    long long wv;               // device: sum of the value deltas of the wave's lanes before this one
    int32_t wc;                 // device: same for the coverage deltas
    unsigned long long tqa, tqb;    // delta_q launches: the lane's sums of the squares' parts
    unsigned long long wqa, wqb;
};

WT_DEV void wt_delta_ctx_init(WtDeltaCtx &d, const WtParams &P, char *lds) {
    d.acc = (long long *) (lds + P.off_acc);
    d.ev = (uint32_t *) (lds + P.off_ev);
    d.ltv = (long long *) (lds + P.off_ltv);
    d.ltc = (int32_t *) (lds + P.off_ltc);
    d.gtv = (long long *) (lds + P.off_gtv);
    d.gtc = (int32_t *) (lds + P.off_gtc);
    d.tbase = (long long *) (lds + P.off_tbase);
    d.tpfx = (uint32_t *) (lds + P.off_tpfx);
    d.tfirst = (uint16_t *) (lds + P.off_tfirst);
    d.tdef = (uint32_t *) (lds + P.off_tdef);
    const int ns = P.delta_ns > 1 ? P.delta_ns : 1;     // sets (two-sample launches: 2)
    d.np = ns * P.W;
    d.qa = (unsigned long long *) (lds + P.off_qa);
    d.qb = d.qa + d.np;
    d.ltqa = (unsigned long long *) (lds + P.off_ltq);
    d.ltqb = d.ltqa + ns * (P.W / WT_DELTA_K);          // (W / K = lanes of the scans)
    d.gtqa = (unsigned long long *) (lds + P.off_gtq);
    d.gtqb = d.gtqa + ns * (P.W / WT_DELTA_K / WT_DELTA_GROUP);
    d.dsh = (WtDeltaShared *) (lds + P.off_dsh);
}

#ifdef WT_EMU
WT_DEV void wt_lds_add32(uint32_t *p, uint32_t v) { *p += v; }
WT_DEV void wt_lds_addi32(int32_t *p, int32_t v) { *p += v; }
WT_DEV void wt_lds_max32(int32_t *p, int32_t v) { if (v > *p) *p = v; }
#else
WT_DEV void wt_lds_add32(uint32_t *p, uint32_t v) { atomicAdd((unsigned int *) p, (unsigned int) v); }
WT_DEV void wt_lds_addi32(int32_t *p, int32_t v) { atomicAdd(p, v); }
WT_DEV void wt_lds_max32(int32_t *p, int32_t v) { atomicMax(p, v); }
#endif

// largest exponent span a window of N tracks may have: N * 2^(span+24) <= 2^53
WT_DEV int wt_delta_max_span(int n_tracks) {
    int lg = 0;
    while ((1ll << lg) < (long long) n_tracks) lg++;
    return 29 - lg;
}

template <bool QQ = false, bool TT = false>
WT_DEV void wt_delta_zero(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int tid, int nt) {
    const int np = TT ? 2 * P.W : P.W;
    for (int x = tid; x < np; x += nt) { d.acc[x] = 0; d.ev[x] = 0; }
    if constexpr (QQ)
        for (int x = tid; x < np; x += nt) { d.qa[x] = 0; d.qb[x] = 0; }
    if (tid == 0) {
        d.dsh->base_v = 0; d.dsh->base_c = 0;
        if constexpr (QQ) { d.dsh->base_qa = 0; d.dsh->base_qb = 0; }
        if constexpr (TT) { d.dsh->base_v1 = 0; d.dsh->base_c1 = 0; d.dsh->base_qa1 = 0; d.dsh->base_qb1 = 0; d.dsh->risk = 0; }
        // (non-zero defaults are terms of every position's sum: their exponents belong to every window's range)
        d.dsh->emin = P.def_emin; d.dsh->emax = P.def_emax; d.dsh->bad = 0;
    }
}

// m^2 split at bit WT_DELTA_QSHIFT (m = |scaled mantissa| < 2^53)
WT_DEV void wt_delta_square(unsigned long long m, unsigned long long &a, unsigned long long &b) {
#ifdef WT_EMU
    const unsigned __int128 sq = (unsigned __int128) m * m;
    const unsigned long long hi = (unsigned long long) (sq >> 64), lo = (unsigned long long) sq;
#else
    const unsigned long long hi = __umul64hi(m, m), lo = m * m;
#endif
    a = (hi << (64 - WT_DELTA_QSHIFT)) | (lo >> WT_DELTA_QSHIFT);
    b = lo & ((1ull << WT_DELTA_QSHIFT) - 1ull);
}

// ---- the window's intervals as ONE flat index space ----
// A chunk of up to T tracks (T = workgroup size): lane t looks up track c0 + t's interval range
// (first interval with finish >= w0 .. last one that can start before w1, from the window index),
// an exclusive scan of the counts gives every track its slice [tpfx[t], tpfx[t+1]) of the flat
// space.  The passes then stride over flat indices: 64-lane tiles of 4 x 64 consecutive indices,
// so loads are coalesced inside a track and every lane is busy whatever the tracks' densities.
// (`row`, `chrom`: of the window -- the current one's header, or the NEXT one's, see wt_delta_kernel)
WT_DEV void wt_delta_ranges1(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int c0, int tid, int nt, long long row, int chrom) {
    const int N = P.n_tracks;
    const int g = c0 + tid;
    long long n = 0, first = 0;
    if (g < N) {
        const uint32_t *row0 = P.widx + (size_t) row * N;
        const long long seg = (long long) chrom * N + g;
        const long long off = P.seg_off[seg];
        const long long cnt = P.seg_off[seg + 1] - off;
        const long long lo = row0[g];
        long long hi = row0[N + g];
        if (hi >= cnt) hi = cnt - 1;
        n = hi >= lo ? hi - lo + 1 : 0;
        first = off + lo;
    }
    d.tbase[tid] = first * 4;
    d.ltc[tid] = (int32_t) n;
    if (P.delta_df) d.tdef[tid] = g < N ? __builtin_bit_cast(uint32_t, (float) P.defaults[g]) : 0u;
}
WT_DEV void wt_delta_ranges1(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int c0, int tid, int nt) {
    wt_delta_ranges1(P, c, d, c0, tid, nt, c.sh->row, c.sh->chrom);
}

WT_DEV void wt_delta_ranges2(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int tid, int nt) {
    const int ngroups = nt / WT_DELTA_GROUP;
    if (tid >= ngroups) return;
    int32_t sc = 0;
    for (int x = 0; x < WT_DELTA_GROUP; x++) sc += d.ltc[tid * WT_DELTA_GROUP + x];
    d.gtc[tid] = sc;
}

WT_DEV void wt_delta_ranges3(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int tid, int nt, uint32_t tile_runs = WT_DELTA_TILE) {
    const int grp = tid / WT_DELTA_GROUP;
    uint32_t pfx = 0;
    for (int x = 0; x < grp; x++) pfx += (uint32_t) d.gtc[x];
    for (int x = grp * WT_DELTA_GROUP; x < tid; x++) pfx += (uint32_t) d.ltc[x];
    const uint32_t n = (uint32_t) d.ltc[tid];
    d.tpfx[tid] = pfx;
    d.tbase[tid] -= 4ll * (long long) pfx;
    if (tid == nt - 1) d.tpfx[nt] = pfx + n;
    // first track of every tile of WT_DELTA_TILE flat indices (the slices partition the flat
    // space, so every tile start lies in exactly one non-empty slice)
    for (uint32_t b = (pfx + tile_runs - 1) / tile_runs; b * tile_runs < pfx + n && b < WT_DELTA_TF; b++)
        d.tfirst[b] = (uint16_t) tid;
}

// smallest track slot i' >= i whose slice holds flat index jj (jj < tpfx[nt])
WT_DEV int wt_delta_find(const uint32_t *tpfx, int nt, uint32_t jj, int i) {
    int lo = i, hi = nt - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (jj >= tpfx[mid + 1]) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// byte offsets (into a 4-byte column) of the lane's WT_DELTA_U flat indices of tile `tb` (-1: past the end)
template <int U = WT_DELTA_U>
WT_DEV void wt_delta_tile(const WtDeltaCtx &d, int nt, uint32_t M, uint32_t tb, int lane, long long (&g)[U]) {
    const uint32_t tile = tb / (64u * U);
    int i = tile < WT_DELTA_TF ? (int) d.tfirst[tile] : wt_delta_find(d.tpfx, nt, tb, 0);
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t jj = tb + (uint32_t) lane + 64u * (uint32_t) u;
        g[u] = -1;
        if (jj < M) {
            while (jj >= d.tpfx[i + 1]) i++;
            g[u] = d.tbase[i] + 4ll * (long long) jj;
        }
    }
}

// pass 1: exponent range of every value the window may use.  The loads of the wave's next tile
// are issued before the current one is consumed (the passes are latency-, not bandwidth-bound).
template <int U = WT_DELTA_U>
WT_DEV void wt_delta_pass1(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int tid, int nt) {
    const int wave = tid >> 6, lane = tid & 63, nwaves = nt >> 6;
    const uint32_t *val = (const uint32_t *) P.value;
    const uint32_t M = d.tpfx[nt];
    const uint32_t step = (uint32_t) nwaves * (64u * U);
    int emin = 255, emax = 0, bad = 0;
    uint32_t cur[U], nxt[U];
    uint32_t tb = (uint32_t) wave * (64u * U);
    if (tb < M) {
        long long g[U];
        wt_delta_tile<U>(d, nt, M, tb, lane, g);
#pragma unroll
        for (int u = 0; u < U; u++) cur[u] = g[u] >= 0 ? *(const uint32_t *) ((const char *) val + g[u]) : 0u;
    }
    for (; tb < M; tb += step) {
        if (tb + step < M) {
            long long g[U];
            wt_delta_tile<U>(d, nt, M, tb + step, lane, g);
#pragma unroll
            for (int u = 0; u < U; u++) nxt[u] = g[u] >= 0 ? *(const uint32_t *) ((const char *) val + g[u]) : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = (int) ((cur[u] >> 23) & 0xffu);
            if (e == 0xff) bad = 1;
            if ((cur[u] & 0x7fffffffu) != 0u) {
                const int ee = e ? e : 1;
                emin = ee < emin ? ee : emin;
                emax = ee > emax ? ee : emax;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) cur[u] = nxt[u];
    }
    if (emin <= emax) { wt_lds_min32(&d.dsh->emin, emin); wt_lds_max32(&d.dsh->emax, emax); }
    if (bad) wt_lds_max32(&d.dsh->bad, 1);
}

// After pass 1 (and a barrier): is the window exact, and what is the unit exponent?  Pure function
// of the LDS verdict fields, evaluated by every lane that needs it.
WT_DEV bool wt_delta_verdict(const WtParams &P, const WtDeltaCtx &d, int &emin) {
    int lo = d.dsh->emin, hi = d.dsh->emax;
    if (lo > hi) { lo = 1; hi = 1; }            // no non-zero value at all
    emin = lo;
    return !d.dsh->bad && (hi - lo) <= wt_delta_max_span(P.n_tracks);
}

// ---- pass 2 ----
// Per interval the wave spends instruction ISSUE, not bandwidth (DESIGN A.1; round 5's phase profile, DESIGN 4.1, reads differently: the pass runs at 0.81 of the HBM peak), so this is written for
// the instruction count (round 3; read off the ISA):
//   * w0 and the window width are arguments (SGPRs): read from LDS inside the loop they cost an
//     `s_waitcnt lgkmcnt(0)` per interval, which also waited for the previous interval's four atomics;
//   * the signed scaled mantissa is built in 32 bits and shifted once; the finish takes ds_sub_u64;
//   * a value below the unit (`e < scale`) shifts by a wrapped count: the sum is garbage, and the
//     window is redone anyway (wt_delta_window_verdict sees emin < guess) -- no select per value;
//   * the exponent range is the unsigned min / max of the values' magnitude BITS (zero excluded from
//     the minimum by the wrap of `key - 1`): four instructions instead of nine.
struct WtDeltaRange {
    uint32_t kmax, kmin;        // max of (bits & 0x7fffffff); min of (bits & 0x7fffffff) - 1 (zero wraps to the top)
};

// the float with bits `vb` in units of 2^(scale - 150): signed mantissa, shifted (a wrapped shift count -- a value
// below the unit -- gives garbage, see above)
WT_DEV long long wt_delta_scaled(uint32_t vb, int scale) {
    const uint32_t e = (vb >> 23) & 0xffu;
    const uint32_t m = (vb & 0x7fffffu) | ((e < 1u ? e : 1u) << 23);    // hidden bit unless denormal / zero
    const int32_t sgn = (int32_t) vb >> 31;
    const int32_t sm = (int32_t) ((m ^ (uint32_t) sgn) - (uint32_t) sgn);
    return (long long) ((unsigned long long) (long long) sm << (((e > 1u ? e : 1u) - (uint32_t) scale) & 63u));
}

// DF (non-zero defaults, Sum / Mean): a track that is absent contributes its default, so the window's base holds
// the sum of all defaults and an interval adds (value - default) while it lasts -- `db`: the default's bits.
// TT (two-sample launches): `db` is not a default's bits but the run's POSITION OFFSET -- 0 for a track of the first set, W for one of
// the second, whose accumulator entries lie behind the first's -- and the window's base is kept per set.
template <bool QQ = false, bool DF = false, bool TT = false>
WT_DEV void wt_delta_apply(WtDeltaCtx &d, WtCtx &c, int32_t w0, uint32_t width, int32_t s, int32_t f, uint32_t vb, uint32_t db,
                           int scale, bool ok, int32_t &my_next, WtDeltaRange &R) {
    const uint32_t key = vb & 0x7fffffffu;
    R.kmax = key > R.kmax ? key : R.kmax;
    R.kmin = key - 1u < R.kmin ? key - 1u : R.kmin;
    long long vi = wt_delta_scaled(vb, scale);
    if (DF) vi -= wt_delta_scaled(db, scale);
    const uint32_t po = TT ? db : 0u;
#ifdef WT_EMU
    if (!ok) vi = 0;            // (a window known not to be exact: the device adds garbage, the patch kernel rewrites the values)
#endif
    const uint32_t cs = (uint32_t) (s - w0), cf = (uint32_t) (f - w0);
    if (cs < width && cf < width) {             // the common case: the run lies inside the window -- no branches
        wt_lds_add64((unsigned long long *) &d.acc[po + cs], (unsigned long long) vi);
        wt_lds_add32(&d.ev[po + cs], 1u);
        wt_lds_sub64((unsigned long long *) &d.acc[po + cf], (unsigned long long) vi);
        wt_lds_add32(&d.ev[po + cf], 0x10000u);
        if (QQ && vi) {
            unsigned long long a, b;
            wt_delta_square((unsigned long long) (vi < 0 ? -vi : vi), a, b);
            wt_lds_add64(&d.qa[po + cs], a); wt_lds_add64(&d.qb[po + cs], b);
            wt_lds_sub64(&d.qa[po + cf], a); wt_lds_sub64(&d.qb[po + cf], b);
        }
        return;
    }
    const int32_t w1 = w0 + (int32_t) width;
    if (f == w0) { wt_lds_add32(&d.ev[po], 0x00010001u); return; }    // true breakpoint at w0, covers nothing here
    if (s >= w1) { my_next = s < my_next ? s : my_next; return; }
    unsigned long long qa = 0, qb = 0;
    if (QQ && vi) wt_delta_square((unsigned long long) (vi < 0 ? -vi : vi), qa, qb);
    if (s < w0) {                               // spans w0: part of the window's base, not a breakpoint
        const bool second = TT && po != 0u;
        wt_lds_add64((unsigned long long *) (second ? &d.dsh->base_v1 : &d.dsh->base_v), (unsigned long long) vi);
        wt_lds_addi32(second ? &d.dsh->base_c1 : &d.dsh->base_c, 1);
        if (QQ && vi) { wt_lds_add64(second ? &d.dsh->base_qa1 : &d.dsh->base_qa, qa); wt_lds_add64(second ? &d.dsh->base_qb1 : &d.dsh->base_qb, qb); }
    } else {
        if (vi) wt_lds_add64((unsigned long long *) &d.acc[po + cs], (unsigned long long) vi);
        wt_lds_add32(&d.ev[po + cs], 1u);
        if (QQ && vi) { wt_lds_add64(&d.qa[po + cs], qa); wt_lds_add64(&d.qb[po + cs], qb); }
    }
    if (f < w1) {
        if (vi) wt_lds_sub64((unsigned long long *) &d.acc[po + cf], (unsigned long long) vi);
        wt_lds_add32(&d.ev[po + cf], 0x10000u);
        if (QQ && vi) { wt_lds_sub64(&d.qa[po + cf], qa); wt_lds_sub64(&d.qb[po + cf], qb); }
    } else {
        my_next = f < my_next ? f : my_next;
    }
}

// ---- runs that cross a window edge wait for the end of the pass (round 5) ----
// A run that is not wholly inside the window -- it spans w0 (part of the window's base), ends beyond w1, starts at or
// after w1 (the index is generous), finishes exactly at w0 -- takes wt_delta_apply's other branch: ~90 instructions and
// ten branches, executed by the whole wavefront for the one or two lanes that need it.  Every track has one run across
// each edge, so at 100 tracks and mean run 16 one wave row in four went that way (at mean run 200: every row), and
// forcing every run onto the branch-free path (results wrong: profiles/r05_delta_pass2_experiments.txt, B) made the pass
// 20 % shorter.  So the loop only PARKS such a run in three registers of its lane; a lane that already holds one makes
// the wavefront work the parked ones off first (rare: two of a lane's ~50 runs per window), and what is parked when the
// pass ends is worked off once -- one trip through the long branch per wavefront and window instead of a dozen.
// Integer adds commute: bit-identical.  (`mask`: wave-uniform, which lanes hold a run.)
template <bool DF>
struct WtDeltaPend {
    unsigned long long mask;
    int32_t s, f;
    uint32_t b;
    uint32_t d[DF ? 1 : 1];
};
#ifdef WT_EMU
WT_DEV unsigned long long wt_delta_ballot(bool c) { return c ? 1ull : 0ull; }           // (the emulator's lanes run alone)
WT_DEV bool wt_delta_in_mask(unsigned long long m, int lane) { return m != 0ull; }
#else
WT_DEV unsigned long long wt_delta_ballot(bool c) { return __builtin_amdgcn_ballot_w64(c); }
WT_DEV bool wt_delta_in_mask(unsigned long long m, int lane) { return (m >> lane) & 1ull; }
#endif
template <bool QQ, bool DF, bool TT = false>
WT_DEV void wt_delta_flush(WtDeltaCtx &d, WtCtx &c, WtDeltaPend<DF || TT> &pn, int lane, int32_t w0, uint32_t width, int scale, bool ok,
                           int32_t &my_next) {
    if (wt_delta_in_mask(pn.mask, lane)) {
        WtDeltaRange scrap;     // (the exponent range took the run when it was parked)
        scrap.kmax = 0u; scrap.kmin = 0xffffffffu;
        wt_delta_apply<QQ, DF, TT>(d, c, w0, width, pn.s, pn.f, pn.b, (DF || TT) ? pn.d[0] : 0u, scale, ok, my_next, scrap);
    }
    pn.mask = 0ull;
}
// one run of the pass: inside the window -> the four (QQ: eight) atomics; not -> parked
template <bool QQ, bool DF, bool TT = false>
WT_DEV void wt_delta_apply_or_park(WtDeltaCtx &d, WtCtx &c, WtDeltaPend<DF || TT> &pn, int lane, bool valid, int32_t w0, uint32_t width,
                                   int32_t s, int32_t f, uint32_t vb, uint32_t db, int scale, bool ok, int32_t &my_next, WtDeltaRange &R) {
    const uint32_t rs = (uint32_t) (s - w0), rf = (uint32_t) (f - w0);
    const bool inside = rs < width && rf < width;
    const uint32_t cs = TT ? rs + db : rs, cf = TT ? rf + db : rf;          // (TT: `db` is the set's position offset)
    if (valid) {
        const uint32_t key = vb & 0x7fffffffu;
        R.kmax = key > R.kmax ? key : R.kmax;
        R.kmin = key - 1u < R.kmin ? key - 1u : R.kmin;
    }
    if (valid && inside) {
        long long vi = wt_delta_scaled(vb, scale);
        if (DF) vi -= wt_delta_scaled(db, scale);
#ifdef WT_EMU
        if (!ok) vi = 0;
#endif
        wt_lds_add64((unsigned long long *) &d.acc[cs], (unsigned long long) vi);
        wt_lds_add32(&d.ev[cs], 1u);
        wt_lds_sub64((unsigned long long *) &d.acc[cf], (unsigned long long) vi);
        wt_lds_add32(&d.ev[cf], 0x10000u);
        if (QQ && vi) {
            unsigned long long a, b;
            wt_delta_square((unsigned long long) (vi < 0 ? -vi : vi), a, b);
            wt_lds_add64(&d.qa[cs], a); wt_lds_add64(&d.qb[cs], b);
            wt_lds_sub64(&d.qa[cf], a); wt_lds_sub64(&d.qb[cf], b);
        }
    }
    const bool park = valid && !inside;
    const unsigned long long pm = wt_delta_ballot(park);
    if (pm & pn.mask) wt_delta_flush<QQ, DF, TT>(d, c, pn, lane, w0, width, scale, ok, my_next);     // (wave-uniform, rare)
    pn.s = park ? s : pn.s;
    pn.f = park ? f : pn.f;
    pn.b = park ? vb : pn.b;
    if (DF || TT) pn.d[0] = park ? db : pn.d[0];
    pn.mask |= pm;
}

// the lane's WT_DELTA_U intervals of one tile (same software pipeline as pass 1)
template <bool DF, int U = WT_DELTA_U>
struct WtDeltaBatch {
    int32_t s[U], f[U];
    uint32_t b[U];
    uint32_t d[DF ? U : 1];    // DF: bits of the interval's track's default
};

// The loads are UNCONDITIONAL and always in range: a flat index past the end is clamped to the last
// one (the applying side masks it), a whole tile past the end re-reads the last tile.  With a branch (or
// an exec-masked region the compiler may skip) around them, the number of loads in flight is unknown to
// the compiler, and the consumer of the PREVIOUS tile gets `s_waitcnt vmcnt(0)` -- the prefetch is
// waited for before the tile it was meant to overlap is applied (round 2, read off the ISA).
// The track of a flat index: tfirst[] gives the tile's first one; the lane keeps the end of its
// current slice and the slice's byte offset in registers and only walks tpfx[] when an index crosses it.
// (TT: B.d[] carries the run's position offset -- 0 / W by the set of its track, slot i of the chunk starting at track c0)
template <bool DF, bool TT = false, int U = WT_DELTA_U>
WT_DEV void wt_delta_fetch(const WtParams &P, const WtDeltaCtx &d, int nt, uint32_t M, uint32_t tb, int lane, WtDeltaBatch<DF || TT, U> &B, int c0 = 0) {
    const uint32_t last = (M - 1u) / (64u * U) * (64u * U);
    const uint32_t tbe = tb < last ? tb : last;
    const uint32_t tile = tbe / (64u * U);
    int i = tile < WT_DELTA_TF ? (int) d.tfirst[tile] : wt_delta_find(d.tpfx, nt, tbe, 0);
    uint32_t hi = d.tpfx[i + 1];
    long long dl = d.tbase[i];
    uint32_t db = DF ? d.tdef[i] : 0u;
    if (TT) db = c0 + i >= P.n_set0 ? (uint32_t) P.W : 0u;
#pragma unroll
    for (int u = 0; u < U; u++) {
        uint32_t jj = tbe + (uint32_t) lane + 64u * (uint32_t) u;
        jj = jj < M - 1u ? jj : M - 1u;
        if (jj >= hi) {
            do { i++; hi = d.tpfx[i + 1]; } while (jj >= hi);
            dl = d.tbase[i];
            if (DF) db = d.tdef[i];
            if (TT) db = c0 + i >= P.n_set0 ? (uint32_t) P.W : 0u;
        }
        const long long ob = dl + ((long long) jj << 2);
        B.s[u] = *(const int32_t *) ((const char *) P.start + ob);
        B.f[u] = *(const int32_t *) ((const char *) P.finish + ob);
        B.b[u] = *(const uint32_t *) ((const char *) P.value + ob);
        if (DF || TT) B.d[u] = db;
    }
}

// (Round 5 tried "contiguous runs share an atomic" here: where a track's runs are contiguous the lane of the previous run subtracts at
// the very position where this lane adds, so one lane can do both -- two LDS atomics per run instead of four, the facts passed one lane
// along by DPP wave shifts, bit-identical.  No effect on MI355X: C2 66.50 against 66.83 ms, C3 70.58 against 70.49; the code is
// tools/experiments/r5_delta_merged_atomics.patch, the record DESIGN 4.1.)

// every interval of the tile at flat index `tb`; only the window's last tile can be partial
template <bool QQ = false, bool DF = false, bool TT = false, int U = WT_DELTA_U>
WT_DEV void wt_delta_apply_tile(WtDeltaCtx &d, WtCtx &c, const WtDeltaBatch<DF || TT, U> &B, uint32_t tb, uint32_t M, int lane, int32_t w0,
                                uint32_t width, int scale, bool ok, int32_t &my_next, WtDeltaRange &R, WtDeltaPend<DF || TT> &pn) {
    // WT_DELTA_PARK: 1 = the launches with squares park (their long branch is twice as long and, at 500 tracks and 4096-bp
    // windows, every second wave row took it: -10 %); Sum / Mean do not (C2 the same to 1 % on a fast box, 4 % slower under
    // the profiler on a slow one: two more spilled registers and 5 % more HBM traffic); 2 = everybody parks; 0 = nobody.
    if constexpr (WT_DELTA_PARK == 2 || (WT_DELTA_PARK == 1 && QQ)) {
        if (tb + (64u * U) <= M) {
#pragma unroll
            for (int u = 0; u < U; u++)
                wt_delta_apply_or_park<QQ, DF, TT>(d, c, pn, lane, true, w0, width, B.s[u], B.f[u], B.b[u], (DF || TT) ? B.d[u] : 0u, scale, ok, my_next, R);
        } else {
#pragma unroll
            for (int u = 0; u < U; u++)
                wt_delta_apply_or_park<QQ, DF, TT>(d, c, pn, lane, tb + (uint32_t) lane + 64u * (uint32_t) u < M, w0, width, B.s[u], B.f[u], B.b[u],
                                               (DF || TT) ? B.d[u] : 0u, scale, ok, my_next, R);
        }
    } else {
        if (tb + (64u * U) <= M) {
#pragma unroll
            for (int u = 0; u < U; u++)
                wt_delta_apply<QQ, DF, TT>(d, c, w0, width, B.s[u], B.f[u], B.b[u], (DF || TT) ? B.d[u] : 0u, scale, ok, my_next, R);
        } else {
#pragma unroll
            for (int u = 0; u < U; u++)
                if (tb + (uint32_t) lane + 64u * (uint32_t) u < M)
                    wt_delta_apply<QQ, DF, TT>(d, c, w0, width, B.s[u], B.f[u], B.b[u], (DF || TT) ? B.d[u] : 0u, scale, ok, my_next, R);
        }
    }
}

// `scale`: exponent of one unit of the scaled mantissas; `ok`: false -> the window is known not to be
// exact (only coordinates matter); `collect`: also publish the exponent range of the values (the
// speculative single-pass flavour, see wt_delta_window_verdict); `stats`: count the intervals;
// `ntr`: tracks of this chunk (DF: their defaults go to the window's base).
// `c0`: first track of the chunk (TT: the set of a track is a matter of its number).
template <bool QQ = false, bool DF = false, bool TT = false, int U = WT_DELTA_U>
WT_DEV void wt_delta_pass2(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int scale, bool ok, bool collect, bool stats,
                           int tid, int nt, int ntr = 0, int c0 = 0) {
    const int wave = wt_uniform32(tid >> 6), lane = tid & 63, nwaves = nt >> 6;     // (uniform: the tile tests stay scalar)
    const uint32_t M = (uint32_t) wt_uniform32((int32_t) d.tpfx[nt]);
    const int32_t w0 = wt_uniform32(c.sh->w0);
    const uint32_t width = (uint32_t) (wt_uniform32(c.sh->w1) - w0);
    const uint32_t step = (uint32_t) nwaves * (64u * U);
    int32_t my_next = 0x7fffffff;
    WtDeltaRange R;
    R.kmax = 0u; R.kmin = 0xffffffffu;
    if (DF) {
        // every track's default, present or not, in the units of this pass
        unsigned long long dsum = (tid < ntr && ok) ? (unsigned long long) wt_delta_scaled(d.tdef[tid], scale) : 0ull;
        dsum = wt_wave_sum_u64(dsum);
        if (dsum && wt_wave_leader(lane)) wt_lds_add64((unsigned long long *) &d.dsh->base_v, dsum);
    }
    uint32_t tb = (uint32_t) wave * (64u * U);
    if (tb < M) {
        // two register sets take turns (a `cur = nxt` copy is 12 moves per tile, and it put the wait for the
        // prefetched tile at the end of the iteration that issued it)
        WtDeltaBatch<DF || TT, U> A, B;
        WtDeltaPend<DF || TT> pn;
        pn.mask = 0ull; pn.s = 0; pn.f = 0; pn.b = 0u; pn.d[0] = 0u;
        wt_delta_fetch<DF, TT, U>(P, d, nt, M, tb, lane, A, c0);
#if WT_DELTA_ONE_TILE
        // a wavefront with ONE tile (sparse windows: a tile per wavefront or fewer) applies it without a second set of loads in front
        if (tb + step >= M) wt_delta_apply_tile<QQ, DF, TT, U>(d, c, A, tb, M, lane, w0, width, scale, ok, my_next, R, pn);
        else
#endif
        for (;;) {
            wt_delta_fetch<DF, TT, U>(P, d, nt, M, tb + step, lane, B, c0);    // (past the end: harmless re-reads of the last tile)
            wt_delta_apply_tile<QQ, DF, TT, U>(d, c, A, tb, M, lane, w0, width, scale, ok, my_next, R, pn);
            tb += step;
            if (tb >= M) break;
            wt_delta_fetch<DF, TT, U>(P, d, nt, M, tb + step, lane, A, c0);
            wt_delta_apply_tile<QQ, DF, TT, U>(d, c, B, tb, M, lane, w0, width, scale, ok, my_next, R, pn);
            tb += step;
            if (tb >= M) break;
        }
        if (pn.mask) wt_delta_flush<QQ, DF, TT>(d, c, pn, lane, w0, width, scale, ok, my_next);      // the runs across the window's edges
    }
    my_next = wt_wave_min_i32(my_next);
    if (my_next != 0x7fffffff && wt_wave_leader(lane)) wt_lds_min32(&c.sh->next_bp, my_next);
    if (collect) {
        // back to exponents: largest / smallest non-zero magnitude of the wave (denormals count as exponent 1)
        const uint32_t kmax = wt_wave_max_u32(R.kmax), kmin = wt_wave_min_u32(R.kmin) + 1u;
        if (kmax != 0u && wt_wave_leader(lane)) {
            const int emax = (int) (kmax >> 23), emin = (int) (kmin >> 23);
            wt_lds_min32(&d.dsh->emin, emin ? emin : 1);
            wt_lds_max32(&d.dsh->emax, emax ? emax : 1);
            if (emax == 0xff) wt_lds_max32(&d.dsh->bad, 1);
        }
    }
    if (stats && tid == 0 && M) wt_lds_add64(&c.sh->n_intervals, (unsigned long long) M);
}

// Single-pass speculation.  Reading value[] twice costs a third more HBM traffic, and the passes
// are bandwidth-bound, so a workgroup that already knows a unit exponent (`guess`: the smallest
// exponent it has met so far) scales with it right away and gathers the window's true exponent
// range on the side.  Any unit <= the window's smallest exponent is as exact as the smallest
// itself provided the span test holds for it, so the result does not depend on the guess (nor on
// which workgroup got which window).  After the pass:
//   returns 1  the guess was fine: go on
//   returns 0  a value lay below the guess, or the span test fails for the guess: the window is
//              redone with its own smallest exponent `lo` (verdict `ok`)
WT_DEV int wt_delta_window_verdict(const WtParams &P, const WtDeltaCtx &d, int guess, int &lo, bool &ok) {
    int a = d.dsh->emin, b = d.dsh->emax;
    const bool bad = d.dsh->bad != 0;
    const int R = wt_delta_max_span(P.n_tracks);
    if (a > b) { lo = guess; ok = !bad; return bad ? 0 : 1; }      // no non-zero value: any unit will do
    lo = a;
    ok = !bad && (b - a) <= R;
    return (!bad && a >= guess && (b - guess) <= R) ? 1 : 0;
}

// One lane: the window is not provably exact.  It still emits its runs (coordinates, run count,
// look-back) so that the output stays ordered; it is recorded so that the general kernel can
// rewrite the values of just these windows afterwards (wt_patch_kernel).
WT_DEV void wt_delta_mark_bad(const WtParams &P, WtCtx &c, long long k) {
    const unsigned long long slot = wt_glb_add64(&P.counters[WT_CTR_DELTA_BAD], 1ull);
    c.sh->bad_slot = (int32_t) slot;
    if (P.bad_list) P.bad_list[slot] = (int32_t) k;
}
// after the look-back (lanes 0 .. 15 of the wave that ran it): where the window's runs start, and where the runs of each
// of its 16 sub-ranges start (the emitted bitmap's prefix counts are there: wt_delta_escan_wave) -- the patch kernel's
// narrower windows then need no order among themselves (round 4: a patched 8192-bp window was four 2048-bp windows done
// one after the other by one workgroup, because each one's output offset was the previous one's plus its run count).
#define WT_BAD_SUB 16
WT_DEV void wt_delta_note_offset(const WtParams &P, WtCtx &c, int lane) {
    if (c.sh->bad_slot >= 0 && P.bad_goff && lane < WT_BAD_SUB)
        P.bad_goff[(long long) c.sh->bad_slot * WT_BAD_SUB + lane] = c.sh->goffset + (long long) c.epfx[(lane * P.n_words) / WT_BAD_SUB];
}

// redo of a window: clear the accumulators only (the exponent range is kept)
template <bool QQ = false, bool TT = false>
WT_DEV void wt_delta_rezero(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int tid, int nt) {
    const int np = TT ? 2 * P.W : P.W;
    for (int x = tid; x < np; x += nt) { d.acc[x] = 0; d.ev[x] = 0; }
    if constexpr (QQ)
        for (int x = tid; x < np; x += nt) { d.qa[x] = 0; d.qb[x] = 0; }
    if (tid == 0) {
        d.dsh->base_v = 0; d.dsh->base_c = 0;
        if constexpr (QQ) { d.dsh->base_qa = 0; d.dsh->base_qb = 0; }
        if constexpr (TT) { d.dsh->base_v1 = 0; d.dsh->base_c1 = 0; d.dsh->base_qa1 = 0; d.dsh->base_qb1 = 0; }
    }
}

// A lane's 8 consecutive accumulator entries in registers.  On the device as 16-byte LDS reads (ds_read_b128): entry by entry the lanes of
// a wavefront are 32 or 64 bytes apart and eight of them meet in every bank -- 16 LDS cycles per 8-byte read instead of 2; the wide reads
// see a quarter of that (round 6: the scans' strided reads were 4 000 of a window's LDS cycles, a tenth of a sparse window's time).
// `p`: 16-byte aligned (the arrays' offsets are multiples of 16, a lane's first entry is a multiple of 8).
#ifdef WT_EMU
WT_DEV void wt_lds_load8(const uint32_t *p, uint32_t (&v)[WT_DELTA_K]) { for (int k = 0; k < WT_DELTA_K; k++) v[k] = p[k]; }
WT_DEV void wt_lds_load8(const unsigned long long *p, unsigned long long (&v)[WT_DELTA_K]) { for (int k = 0; k < WT_DELTA_K; k++) v[k] = p[k]; }
#else
WT_DEV void wt_lds_load8(const uint32_t *p, uint32_t (&v)[WT_DELTA_K]) {
    typedef uint32_t wt_u32x4 __attribute__((ext_vector_type(4)));
    const wt_u32x4 a = ((const wt_u32x4 *) p)[0], b = ((const wt_u32x4 *) p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
WT_DEV void wt_lds_load8(const unsigned long long *p, unsigned long long (&v)[WT_DELTA_K]) {
    typedef unsigned long long wt_u64x2 __attribute__((ext_vector_type(2)));
    const wt_u64x2 a = ((const wt_u64x2 *) p)[0], b = ((const wt_u64x2 *) p)[1], c = ((const wt_u64x2 *) p)[2], e = ((const wt_u64x2 *) p)[3];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y; v[6] = e.x; v[7] = e.y;
}
#endif
#ifdef WT_EMU
WT_DEV void wt_lds_load4(const uint32_t *p, uint32_t (&v)[4]) { for (int k = 0; k < 4; k++) v[k] = p[k]; }
WT_DEV void wt_lds_load2(const uint32_t *p, uint32_t (&v)[2]) { v[0] = p[0]; v[1] = p[1]; }
#else
WT_DEV void wt_lds_load4(const uint32_t *p, uint32_t (&v)[4]) {     // (16-byte aligned)
    typedef uint32_t wt_u32x4 __attribute__((ext_vector_type(4)));
    const wt_u32x4 a = *(const wt_u32x4 *) p;
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
WT_DEV void wt_lds_load2(const uint32_t *p, uint32_t (&v)[2]) {     // (8-byte aligned)
    typedef uint32_t wt_u32x2 __attribute__((ext_vector_type(2)));
    const wt_u32x2 a = *(const wt_u32x2 *) p;
    v[0] = a.x; v[1] = a.y;
}
#endif
WT_DEV void wt_lds_load8(const long long *p, long long (&v)[WT_DELTA_K]) { wt_lds_load8((const unsigned long long *) p, (unsigned long long (&)[WT_DELTA_K]) v); }

// scan step 1: the lane's 8 positions
template <bool QQ = false>
WT_DEV void wt_delta_scan1(const WtParams &P, WtCtx &c, WtDeltaCtx &d, WtDeltaLane &L, int tid, int nt) {
    const int p0 = tid * WT_DELTA_K;
    long long rv = 0;
    int32_t rc = 0;
    uint32_t evm = 0;
    uint32_t e8[WT_DELTA_K];
    long long a8[WT_DELTA_K];
    wt_lds_load8(d.ev + p0, e8);
    wt_lds_load8(d.acc + p0, a8);
#pragma unroll
    for (int k = 0; k < WT_DELTA_K; k++) {
        const uint32_t e = e8[k];
        rv += a8[k];
        rc += (int32_t) (e & 0xffffu) - (int32_t) (e >> 16);
        if constexpr (!QQ) { L.pv[k] = rv; L.pc[k] = rc; evm |= (e != 0u ? 1u : 0u) << k; }
    }
    if constexpr (!QQ) L.evmask = evm;
    L.tv = rv;
    L.tc = rc;
    d.ltv[tid] = rv;
    d.ltc[tid] = rc;
    if constexpr (QQ) {
        unsigned long long ra = 0, rb = 0;
        unsigned long long qa8[WT_DELTA_K], qb8[WT_DELTA_K];
        wt_lds_load8(d.qa + p0, qa8);
        wt_lds_load8(d.qb + p0, qb8);
#pragma unroll
        for (int k = 0; k < WT_DELTA_K; k++) { ra += qa8[k]; rb += qb8[k]; }
        L.tqa = ra; L.tqb = rb;
        d.ltqa[tid] = ra; d.ltqb[tid] = rb;
    }
}

template <bool QQ = false>
WT_DEV void wt_delta_scan2(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int tid, int nt) {
    const int ngroups = nt / WT_DELTA_GROUP;
    if (tid >= ngroups) return;
    long long sv = 0;
    int32_t sc = 0;
    for (int x = 0; x < WT_DELTA_GROUP; x++) {
        sv += d.ltv[tid * WT_DELTA_GROUP + x];
        sc += d.ltc[tid * WT_DELTA_GROUP + x];
    }
    d.gtv[tid] = sv;
    d.gtc[tid] = sc;
    if constexpr (QQ) {
        unsigned long long sa = 0, sb = 0;
        for (int x = 0; x < WT_DELTA_GROUP; x++) { sa += d.ltqa[tid * WT_DELTA_GROUP + x]; sb += d.ltqb[tid * WT_DELTA_GROUP + x]; }
        d.gtqa[tid] = sa; d.gtqb[tid] = sb;
    }
}

// s / n for an integer 1 <= n < 2^15 (the track count; `y` = 1.0 / n, correctly rounded), correctly rounded -- bit for bit the quotient the
// reference computes (reducers.c:375-401: sum / count) in three instructions instead of the compiler's division sequence (two
// v_div_scale, a quarter-rate v_rcp_f64, six fma, v_div_fmas, v_div_fixup: it is a fifth of the scan's arithmetic, round 6).
//   q0 = RN(s y) is within two ulp of s / n;  r = s - n q0 is exact (a multiple of ulp(q0), at most 17 bits: one fma);
//   q0 + r / n = s / n exactly, and q0 + r y differs from that by 2^-105 relative;  the quotient of a double by an integer below 2^15 is
//   never closer than 2^-68 relative to the midpoint of two doubles, nor is it one (n m would need 54 bits or more), so
//   RN(q0 + r y) = RN(s / n).  No underflow: |s| >= 2^-149 or 0, so |s / n| >= 2^-164.  tests/test_div_by_count.py, wtemu_div_n_mismatches.
WT_DEV double wt_div_n(double s, double n, double y) {
    const double q0 = s * y;
    const double r = __builtin_fma(-n, q0, s);
    return __builtin_fma(r, y, q0);
}

// scan step 3: running sum / coverage of every position, breakpoint and emitted bytes, run values
template <int OP>
WT_DEV void wt_delta_scan3(const WtParams &P, WtCtx &c, WtDeltaCtx &d, const WtDeltaLane &L,
                           WtLane<WT_DELTA_K> &out, int emin, int tid, int nt) {
    long long bv = d.dsh->base_v;
    int32_t bc = d.dsh->base_c;
    constexpr bool QQ = OP == WT_OP_VAR || OP == WT_OP_STDDEV || OP == WT_OP_ENTROPY || OP == WT_OP_CV;
    unsigned long long bqa = QQ ? d.dsh->base_qa : 0ull, bqb = QQ ? d.dsh->base_qb : 0ull;
#ifdef WT_EMU
    const int grp = tid / WT_DELTA_GROUP;
    for (int x = 0; x < grp; x++) { bv += d.gtv[x]; bc += d.gtc[x]; }
    for (int x = grp * WT_DELTA_GROUP; x < tid; x++) { bv += d.ltv[x]; bc += d.ltc[x]; }
    if (QQ) {
        for (int x = 0; x < grp; x++) { bqa += d.gtqa[x]; bqb += d.gtqb[x]; }
        for (int x = grp * WT_DELTA_GROUP; x < tid; x++) { bqa += d.ltqa[x]; bqb += d.ltqb[x]; }
    }
#else
    bv += (long long) wt_waves_before64((const unsigned long long *) d.gtv, 0, tid >> 6, tid & 63);     // waves before this one (wt_delta_scan_w1)
    bc += (int32_t) wt_waves_before32((const uint32_t *) d.gtc, 0, tid >> 6, tid & 63);
    bv += L.wv;
    bc += L.wc;
    if (QQ) {
        bqa += wt_waves_before64(d.gtqa, 0, tid >> 6, tid & 63);
        bqb += wt_waves_before64(d.gtqb, 0, tid >> 6, tid & 63);
        bqa += L.wqa;
        bqb += L.wqb;
    }
#endif
    const int N = P.n_tracks;
    const bool strict = (P.flags & WT_STRICT_SET0) != 0;
    const int p0 = tid * WT_DELTA_K;
    // 2^(emin - 150): the weight of one unit of the scaled mantissas
    const double q = __builtin_bit_cast(double, (uint64_t) (emin - 150 + 1023) << 52);
    const long long room = (long long) c.sh->emit_hi - ((long long) c.sh->w0 + p0);
    uint32_t em = 0, evmask = 0;
    if constexpr (!QQ) evmask = L.evmask;
    const double dn = (double) N, yn = 1.0 / dn;            // (Mean: wt_div_n)
    uint32_t e8[QQ ? WT_DELTA_K : 1];
    long long a8[QQ ? WT_DELTA_K : 1];
    unsigned long long qa8[QQ ? WT_DELTA_K : 1], qb8[QQ ? WT_DELTA_K : 1];
    if constexpr (QQ) {
        wt_lds_load8(d.ev + p0, (uint32_t (&)[WT_DELTA_K]) e8);
        wt_lds_load8(d.acc + p0, (long long (&)[WT_DELTA_K]) a8);
        wt_lds_load8(d.qa + p0, (unsigned long long (&)[WT_DELTA_K]) qa8);
        wt_lds_load8(d.qb + p0, (unsigned long long (&)[WT_DELTA_K]) qb8);
    }
#pragma unroll
    for (int k = 0; k < WT_DELTA_K; k++) {
        if constexpr (!QQ) {
            const int32_t cov = bc + L.pc[k];
            const bool pred = strict ? (cov == N) : (cov > 0);       // multiplexer.c:120,125
            if (((evmask >> k) & 1u) && pred && k < room) em |= 1u << k;
            const double s = (double) (bv + L.pv[k]) * q;
            out.res[k] = (OP == WT_OP_MEAN) ? wt_div_n(s, dn, yn) : s;
        } else {
            // the lane's 8 positions once more (scan 1 kept only their totals): running sums from the lane's base on
            const uint32_t e = e8[k];
            bv += a8[k];
            bc += (int32_t) (e & 0xffffu) - (int32_t) (e >> 16);
            bqa += qa8[k]; bqb += qb8[k];
            evmask |= (e != 0u ? 1u : 0u) << k;
            const int32_t cov = bc;
            const bool pred = strict ? (cov == N) : (cov > 0);       // multiplexer.c:120,125
            if (e != 0u && pred && k < room) em |= 1u << k;
            // exact integers: S, n = cov, Q = (A << 40) + B (see WT_DELTA_QSHIFT)
            const long long S = bv;
            const unsigned long long sa = (unsigned long long) (S < 0 ? -S : S);
            const unsigned __int128 Q = ((unsigned __int128) bqa << WT_DELTA_QSHIFT) + (unsigned __int128) bqb;
            const unsigned __int128 s2 = (unsigned __int128) sa * sa;
            unsigned __int128 I;
            if (OP == WT_OP_VAR) I = (unsigned __int128) ((unsigned long long) N * (unsigned long long) N) * Q - (unsigned __int128) (unsigned long long) (2 * N - cov) * s2;
            else I = (unsigned __int128) (unsigned long long) N * Q - s2;
            const double Id = (double) (unsigned long long) (I >> 64) * 18446744073709551616.0 + (double) (unsigned long long) I;
            double r;
            if (OP == WT_OP_VAR) {
                r = Id * q * q / ((double) N * (double) N) / N;                 // reducers.c:476  sum / count
                if (N < 2) r = wt_nan();                                      // :464
            } else {
                r = sqrt(Id * q * q / N / N);                                 // :558  sqrt(sum / count)
                if (OP == WT_OP_CV) {
                    const double mean = (double) S * q / N;
                    r = mean == 0 ? wt_nan() : r / mean;                      // :707-711
                }
            }
            out.res[k] = r;
        }
    }
    ((uint8_t *) c.U)[tid] = (uint8_t) evmask;
    ((uint8_t *) c.E)[tid] = (uint8_t) em;
}

// ---- MaxReduction / MinReduction over float tracks with zero defaults (round 6): a segment tree of range updates ----
// Reference reducers.c:125-168 / 192-235: the largest (smallest) of every track's value -- or, for a track not in play, its default (0 for
// track 0 whatever its default: the seed) -- NaN if any of them is NaN.  The general kernel gathers all N tracks at every run start (VALU
// bound, 0.15 of the HBM roofline).  With all defaults 0 the answer at position p is
//     max(v of the runs covering p)            if every track is in play there (coverage == N),
//     max(that, 0)                             otherwise,
// and "the runs covering p" is a RANGE UPDATE per run: the window's positions are the leaves of a segment tree (2 W u32 nodes where
// Sum / Mean keep their W 64-bit accumulators: the same 64 KB), a run [l, r) takes atomic max on its <= 2 log2(r - l) canonical nodes
// (order-preserving keys of the float bits, wt_key32), and a position's value is the max over its log2(W) + 1 ancestors -- read by the scan,
// 25 LDS reads for a lane's 8 positions (the levels above the lane's 8 leaves are common to them).  O(runs x log(run length)) instead of
// O(tracks x positions); max is idempotent and order-free, so the result is the reference's bit for bit: a float widened to double is
// exact, and the one place where the reference's order shows -- which of -0.0 and +0.0 it keeps -- is kept out: a window holding a NaN
// (the reference answers NaN there) or a -0.0 is recorded (wt_delta_mark_bad) and redone by the general kernel, like a window
// Sum / Mean cannot prove exact.  Breakpoints, coverage, emitted runs: the ev[] counters of Sum / Mean, unchanged.
template <bool ISMAX>
WT_DEV void wt_delta_tree_update(uint32_t *tree, uint32_t W, uint32_t l, uint32_t r, uint32_t key) {
    for (l += W, r += W; l < r; l >>= 1, r >>= 1) {
        if (l & 1u) { if (ISMAX) wt_lds_umax32(&tree[l], key); else wt_lds_umin32(&tree[l], key); l++; }
        if (r & 1u) { --r; if (ISMAX) wt_lds_umax32(&tree[r], key); else wt_lds_umin32(&tree[r], key); }
    }
}

template <bool ISMAX>
WT_DEV void wt_delta_zero_mm(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int tid, int nt) {
    uint32_t *tree = (uint32_t *) d.acc;
    const uint32_t none = ISMAX ? 0u : 0xffffffffu;         // below / above every key
    for (int x = tid; x < 2 * P.W; x += nt) tree[x] = none;
    for (int x = tid; x < P.W; x += nt) d.ev[x] = 0;
    if (tid == 0) { d.dsh->base_v = 0; d.dsh->base_c = 0; d.dsh->emin = 255; d.dsh->emax = 0; d.dsh->bad = 0; }
}

// one run: the coverage / breakpoint counters exactly as wt_delta_apply keeps them, and the range update
template <bool ISMAX>
WT_DEV void wt_delta_apply_mm(WtDeltaCtx &d, WtCtx &c, int32_t w0, uint32_t width, int32_t s, int32_t f, uint32_t vb, int32_t &my_next, bool &bad) {
    bad |= (vb & 0x7fffffffu) > 0x7f800000u || vb == 0x80000000u;       // NaN, -0.0
    const uint32_t key = wt_key32(__builtin_bit_cast(float, vb));
    const uint32_t cs = (uint32_t) (s - w0), cf = (uint32_t) (f - w0);
    uint32_t l, r;
    if (cs < width && cf < width) {             // the common case: the run lies inside the window
        wt_lds_add32(&d.ev[cs], 1u);
        wt_lds_add32(&d.ev[cf], 0x10000u);
        l = cs; r = cf;
    } else {
        const int32_t w1 = w0 + (int32_t) width;
        if (f == w0) { wt_lds_add32(&d.ev[0], 0x00010001u); return; }    // true breakpoint at w0, covers nothing here
        if (s >= w1) { my_next = s < my_next ? s : my_next; return; }
        if (s < w0) { wt_lds_addi32(&d.dsh->base_c, 1); l = 0u; }        // spans w0: part of the window's base coverage, not a breakpoint
        else { wt_lds_add32(&d.ev[cs], 1u); l = cs; }
        if (f < w1) { wt_lds_add32(&d.ev[cf], 0x10000u); r = cf; }
        else { my_next = f < my_next ? f : my_next; r = width; }
    }
    wt_delta_tree_update<ISMAX>((uint32_t *) d.acc, width, l, r, key);
}

// the pass over the window's runs (the flat index space and the tile pipeline of wt_delta_pass2)
template <bool ISMAX>
WT_DEV void wt_delta_pass_mm(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int tid, int nt) {
    const int wave = wt_uniform32(tid >> 6), lane = tid & 63, nwaves = nt >> 6;
    const uint32_t M = (uint32_t) wt_uniform32((int32_t) d.tpfx[nt]);
    const int32_t w0 = wt_uniform32(c.sh->w0);
    const uint32_t width = (uint32_t) (wt_uniform32(c.sh->w1) - w0);
    const uint32_t step = (uint32_t) nwaves * WT_DELTA_TILE;
    int32_t my_next = 0x7fffffff;
    bool bad = false;
    uint32_t tb = (uint32_t) wave * WT_DELTA_TILE;
    if (tb < M) {
        WtDeltaBatch<false> A, B;
        auto apply = [&](const WtDeltaBatch<false> &T, uint32_t at) {
            if (at + WT_DELTA_TILE <= M) {
#pragma unroll
                for (int u = 0; u < WT_DELTA_U; u++) wt_delta_apply_mm<ISMAX>(d, c, w0, width, T.s[u], T.f[u], T.b[u], my_next, bad);
            } else {
#pragma unroll
                for (int u = 0; u < WT_DELTA_U; u++)
                    if (at + (uint32_t) lane + 64u * (uint32_t) u < M) wt_delta_apply_mm<ISMAX>(d, c, w0, width, T.s[u], T.f[u], T.b[u], my_next, bad);
            }
        };
        wt_delta_fetch<false>(P, d, nt, M, tb, lane, A);
#if WT_DELTA_ONE_TILE
        if (tb + step >= M) apply(A, tb);
        else
#endif
        for (;;) {
            wt_delta_fetch<false>(P, d, nt, M, tb + step, lane, B);    // (past the end: harmless re-reads of the last tile)
            apply(A, tb);
            tb += step;
            if (tb >= M) break;
            wt_delta_fetch<false>(P, d, nt, M, tb + step, lane, A);
            apply(B, tb);
            tb += step;
            if (tb >= M) break;
        }
    }
    my_next = wt_wave_min_i32(my_next);
    if (my_next != 0x7fffffff && wt_wave_leader(lane)) wt_lds_min32(&c.sh->next_bp, my_next);
    if (wt_delta_ballot(bad) != 0ull && wt_wave_leader(lane)) wt_lds_max32(&d.dsh->bad, 1);
    if (tid == 0 && M) wt_lds_add64(&c.sh->n_intervals, (unsigned long long) M);
}

// scan step 1: the lane's coverage total (the values need no prefix)
WT_DEV void wt_delta_scan1_mm(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int32_t &tc, int tid, int nt) {
    const int p0 = tid * WT_DELTA_K;
    int32_t rc = 0;
    uint32_t e8[WT_DELTA_K];
    wt_lds_load8(d.ev + p0, e8);
#pragma unroll
    for (int k = 0; k < WT_DELTA_K; k++) {
        const uint32_t e = e8[k];
        rc += (int32_t) (e & 0xffffu) - (int32_t) (e >> 16);
    }
    tc = rc;
    d.ltc[tid] = rc;
}
WT_DEV void wt_delta_scan2_mm(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int tid, int nt) {     // (emulator: group totals)
    const int ngroups = nt / WT_DELTA_GROUP;
    if (tid >= ngroups) return;
    int32_t sc = 0;
    for (int x = 0; x < WT_DELTA_GROUP; x++) sc += d.ltc[tid * WT_DELTA_GROUP + x];
    d.gtc[tid] = sc;
}

// scan step 3: coverage at every position, breakpoint and emitted bytes, and the run values from the tree
// (`wc`: device -- the coverage deltas of the wave's lanes before this one)
template <bool ISMAX>
WT_DEV void wt_delta_scan3_mm(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int32_t wc, WtLane<WT_DELTA_K> &out, int tid, int nt) {
    int32_t bc = d.dsh->base_c;
#ifdef WT_EMU
    const int grp = tid / WT_DELTA_GROUP;
    for (int x = 0; x < grp; x++) bc += d.gtc[x];
    for (int x = grp * WT_DELTA_GROUP; x < tid; x++) bc += d.ltc[x];
#else
    bc += (int32_t) wt_waves_before32((const uint32_t *) d.gtc, 0, tid >> 6, tid & 63);
    bc += wc;
#endif
    const int N = P.n_tracks;
    const bool strict = (P.flags & WT_STRICT_SET0) != 0;
    const int p0 = tid * WT_DELTA_K;
    const long long room = (long long) c.sh->emit_hi - ((long long) c.sh->w0 + p0);
    const uint32_t *tree = (const uint32_t *) d.acc;
    const uint32_t W = (uint32_t) P.W;
    auto best = [](uint32_t a, uint32_t b) { return ISMAX ? (a > b ? a : b) : (a < b ? a : b); };
    // the ancestors above the lane's 8 leaves are the same for all of them: node (p0 + W) >> 3 and up
    uint32_t top = ISMAX ? 0u : 0xffffffffu;
    for (uint32_t n = ((uint32_t) p0 + W) >> 3; n >= 1u; n >>= 1) top = best(top, tree[n]);
    uint32_t em = 0, evmask = 0;
    uint32_t e8[WT_DELTA_K];
    wt_lds_load8(d.ev + p0, e8);
    // the lane's 8 leaves, their 4 parents and 2 grandparents: consecutive nodes, read wide (wt_lds_load8: the lanes are 8 nodes apart)
    uint32_t lf[WT_DELTA_K], par[4], gp[2];
    wt_lds_load8(tree + W + (uint32_t) p0, lf);
    wt_lds_load4(tree + ((W + (uint32_t) p0) >> 1), par);
    wt_lds_load2(tree + ((W + (uint32_t) p0) >> 2), gp);
#pragma unroll
    for (int k = 0; k < WT_DELTA_K; k++) {
        const uint32_t e = e8[k];
        bc += (int32_t) (e & 0xffffu) - (int32_t) (e >> 16);
        evmask |= (e != 0u ? 1u : 0u) << k;
        const bool pred = strict ? (bc == N) : (bc > 0);            // multiplexer.c:120,125
        if (e != 0u && pred && k < room) em |= 1u << k;
        const uint32_t key = best(best(lf[k], par[k >> 1]), best(gp[k >> 2], top));
        double v = (double) wt_unkey32(key);
        // a track that is not in play enters with its default, 0 (reducers.c:141-152; track 0: the seed, :143-146)
        if (bc < N) v = ISMAX ? (v < 0.0 ? 0.0 : v) : (v > 0.0 ? 0.0 : v);
        out.res[k] = v;
    }
    ((uint8_t *) c.U)[tid] = (uint8_t) evmask;
    ((uint8_t *) c.E)[tid] = (uint8_t) em;
}

// ---- two-sample launches: TTestReduction by difference arrays (round 6) ----
// Reference setComparisons.c:35-121: at every run where BOTH sets have a track in play (:48-54), per set the sum and the sum of
// squares of the values IN PLAY (:69-81; defaults play no part) and the set's size as the count (:66-67), then Welch's t, its
// degrees of freedom and the Student tail.  The two sums per set are what the launches with squares accumulate -- S = sum of
// the scaled mantissas, Q = sum of their squares, exactly -- so the window keeps them twice: the second set's accumulator
// entries lie W behind the first's (wt_delta_fetch: a run's position offset), 56 bytes per position, a 2048-bp window.
// From (double) S * q and (double) Q * q^2 on, the arithmetic is the reference's, operation for operation: where the
// reference's own sums do not round (values on a coarse grid, counts) the result is the reference's bit for bit; where they
// do, its sum of squares carries up to n ulp and the difference of (:91-92) `meanSq - mean * mean` magnifies that by
// meanSq / var.  A window with an emitted position where var * 2^10 < meanSq in a set is therefore recorded like a window
// whose exponent range is too wide (`risk` -> wt_delta_mark_bad): the general kernel, which adds in the reference's order,
// rewrites its values.
// The scans of a two-sample launch are split BY SET: lanes [0, nts) own set 0's 8 positions each, lanes [nts, 2 nts) set 1's (whole
// wavefronts either way) -- half the registers of a lane that carried both sets (which spilled 80), and twice the wavefronts.
//   A  wt_delta_scan1_tt     the lane's totals of its set + the wave-level prefix (device) / group totals (emulator, scan2)
//   B  wt_delta_scan3_tt     running sums at the lane's 8 positions, left IN PLACE as what the test needs: acc[] <- (double) S q,
//                            qa[] <- (double) Q q^2, ev[] <- coverage << 1 | breakpoint
//   C  wt_delta_combine_tt   one lane per POSITION, all the workgroup's lanes: both sets' sums -> emitted?, t, degrees of freedom;
//                            the breakpoint / emitted words are the wavefront's ballots
//   D  wt_delta_tail_tt      the Student tail of the emitted positions (beside the look-back of wave 0)
struct WtDeltaLane2 {
    long long tv;               // the lane's totals of ITS set: value deltas,
    int32_t tc;                 // ... coverage deltas,
    unsigned long long tqa, tqb;            // ... the squares' two parts
    long long wv;               // device: the same summed over the wave's lanes before this one
    int32_t wc;
    unsigned long long wqa, wqb;
};

// (tid: 0 .. 2 nts - 1; the lane's set is tid / nts, its positions 8 (tid % nts) ..)
WT_DEV void wt_delta_scan1_tt(const WtParams &P, WtCtx &c, WtDeltaCtx &d, WtDeltaLane2 &L, int tid, int nts) {
    const int s = tid >= nts ? 1 : 0;
    const int o = s * P.W + (tid - s * nts) * WT_DELTA_K;
    long long rv = 0;
    int32_t rc = 0;
    unsigned long long ra = 0, rb = 0;
    uint32_t e8[WT_DELTA_K];
    long long a8[WT_DELTA_K];
    unsigned long long qa8[WT_DELTA_K], qb8[WT_DELTA_K];
    wt_lds_load8(d.ev + o, e8);
    wt_lds_load8(d.acc + o, a8);
    wt_lds_load8(d.qa + o, qa8);
    wt_lds_load8(d.qb + o, qb8);
#pragma unroll
    for (int k = 0; k < WT_DELTA_K; k++) {
        const uint32_t e = e8[k];
        rv += a8[k];
        rc += (int32_t) (e & 0xffffu) - (int32_t) (e >> 16);
        ra += qa8[k];
        rb += qb8[k];
    }
    L.tv = rv; L.tc = rc; L.tqa = ra; L.tqb = rb;
    d.ltv[tid] = rv;
    d.ltc[tid] = rc;
    d.ltqa[tid] = ra;
    d.ltqb[tid] = rb;
}

// (the emulator's middle step: group totals; groups never straddle the sets: nts is a multiple of WT_DELTA_GROUP)
WT_DEV void wt_delta_scan2_tt(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int tid, int nts) {
    const int ngroups = 2 * nts / WT_DELTA_GROUP;
    if (tid >= ngroups) return;
    long long sv = 0;
    int32_t sc = 0;
    unsigned long long sa = 0, sb = 0;
    for (int x = 0; x < WT_DELTA_GROUP; x++) {
        const int j = tid * WT_DELTA_GROUP + x;
        sv += d.ltv[j]; sc += d.ltc[j]; sa += d.ltqa[j]; sb += d.ltqb[j];
    }
    d.gtv[tid] = sv; d.gtc[tid] = sc; d.gtqa[tid] = sa; d.gtqb[tid] = sb;
}

// Welch's t and its degrees of freedom from the two sets' sums, setComparisons.c:88-113 operation for operation
// (the same lines as the general kernel's wt_eval_finish<TTEST>); t = NaN where the reference answers NaN (:98).
WT_DEV void wt_ttest_stat(double sum1, double sumsq1, double sum2, double sumsq2, int na, int nb, double &t_out, double &nu_out, bool &risk) {
    const double m1 = sum1 / na, m2 = sum2 / nb;
    const double msq1 = sumsq1 / na, msq2 = sumsq2 / nb;
    const double var1 = msq1 - m1 * m1, var2 = msq2 - m2 * m2;
    double t = (m1 - m2) / sqrt(var1 / na + var2 / nb);
    if (t < 0) t = -t;
    const double den = var1 / na + var2 / nb;
    const double c1 = (double) ((long long) na * na * (na - 1));
    const double c2 = (double) ((long long) nb * nb * (nb - 1));
    nu_out = den * den / ((var1 * var1) / c1 + (var2 * var2) / c2);
    t_out = (var1 + var2 == 0) ? wt_nan() : t;
    risk = (var1 * 1024.0 < msq1) || (var2 * 1024.0 < msq2);
}

// phase B: the running sums of the lane's set at its 8 positions, left in place for wt_delta_combine_tt
WT_DEV void wt_delta_scan3_tt(const WtParams &P, WtCtx &c, WtDeltaCtx &d, const WtDeltaLane2 &L, int emin, int tid, int nts) {
    const int s = tid >= nts ? 1 : 0;
    long long bv = s ? d.dsh->base_v1 : d.dsh->base_v;
    int32_t bc = s ? d.dsh->base_c1 : d.dsh->base_c;
    unsigned long long bqa = s ? d.dsh->base_qa1 : d.dsh->base_qa, bqb = s ? d.dsh->base_qb1 : d.dsh->base_qb;
#ifdef WT_EMU
    const int grp = tid / WT_DELTA_GROUP, grp0 = s * nts / WT_DELTA_GROUP;
    for (int x = grp0; x < grp; x++) { bv += d.gtv[x]; bc += d.gtc[x]; bqa += d.gtqa[x]; bqb += d.gtqb[x]; }
    for (int x = grp * WT_DELTA_GROUP; x < tid; x++) { bv += d.ltv[x]; bc += d.ltc[x]; bqa += d.ltqa[x]; bqb += d.ltqb[x]; }
#else
    {   // the set's waves before this one
        const int w0_ = s * (nts >> 6), w1_ = tid >> 6, ln_ = tid & 63;
        bv += (long long) wt_waves_before64((const unsigned long long *) d.gtv, w0_, w1_, ln_);
        bc += (int32_t) wt_waves_before32((const uint32_t *) d.gtc, w0_, w1_, ln_);
        bqa += wt_waves_before64(d.gtqa, w0_, w1_, ln_);
        bqb += wt_waves_before64(d.gtqb, w0_, w1_, ln_);
    }
    bv += L.wv; bc += L.wc; bqa += L.wqa; bqb += L.wqb;
#endif
    const int o = s * P.W + (tid - s * nts) * WT_DELTA_K;
    const double q = __builtin_bit_cast(double, (uint64_t) (emin - 150 + 1023) << 52);      // the weight of one unit of the scaled mantissas
    double *sum = (double *) d.acc, *sumsq = (double *) d.qa;
    uint32_t e8[WT_DELTA_K];
    long long a8[WT_DELTA_K];
    unsigned long long qa8[WT_DELTA_K], qb8[WT_DELTA_K];
    wt_lds_load8(d.ev + o, e8);
    wt_lds_load8(d.acc + o, a8);
    wt_lds_load8(d.qa + o, qa8);
    wt_lds_load8(d.qb + o, qb8);
#pragma unroll
    for (int k = 0; k < WT_DELTA_K; k++) {
        const uint32_t e = e8[k];
        bv += a8[k];
        bc += (int32_t) (e & 0xffffu) - (int32_t) (e >> 16);
        bqa += qa8[k]; bqb += qb8[k];
        // exact integers: S and Q = (A << 40) + B (see WT_DELTA_QSHIFT); S * q is the reference's sum whenever that did not round
        const unsigned __int128 Q = ((unsigned __int128) bqa << WT_DELTA_QSHIFT) + (unsigned __int128) bqb;
        sum[o + k] = (double) bv * q;
        sumsq[o + k] = ((double) (unsigned long long) (Q >> 64) * 18446744073709551616.0 + (double) (unsigned long long) Q) * q * q;
        d.ev[o + k] = ((uint32_t) bc << 1) | (e != 0u ? 1u : 0u);
    }
}

// phase C: one lane per position (wave `tid >> 6` takes the 64-position words wave, wave + nwaves, ...).  t (NaN: no emitted run
// starts here, or the reference's own NaN) and the degrees of freedom go to the second set's slots of the position -- this
// lane has just read them -- for wt_delta_tail_tt.
WT_DEV void wt_delta_combine_tt(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int tid, int nt) {
    const int na = P.n_set0, nb = P.n_tracks - P.n_set0;
    const bool strict0 = (P.flags & WT_STRICT_SET0) != 0, strict1 = (P.flags & WT_STRICT_SET1) != 0;
    const long long room = (long long) c.sh->emit_hi - (long long) c.sh->w0;
    const int lane = tid & 63, nwaves = nt >> 6;
    double *sum = (double *) d.acc, *sumsq = (double *) d.qa;
    bool any_risk = false;
    for (int w = tid >> 6; w < P.n_words; w += nwaves) {
        const int p = w * 64 + lane;
        const uint32_t cv0 = d.ev[p], cv1 = d.ev[P.W + p];
        const bool eany = ((cv0 | cv1) & 1u) != 0u;
        const int32_t c0 = (int32_t) (cv0 >> 1), c1 = (int32_t) (cv1 >> 1);
        const bool pred = (strict0 ? c0 == na : c0 > 0) && (strict1 ? c1 == nb : c1 > 0);     // multiplexer.c:120,125; setComparisons.c:48-54
        const bool emit = eany && pred && p < room;
        double t, nu;
        bool risk;
        wt_ttest_stat(sum[p], sumsq[p], sum[P.W + p], sumsq[P.W + p], na, nb, t, nu, risk);
        any_risk |= emit && risk;
        sum[P.W + p] = emit ? t : wt_nan();
        sumsq[P.W + p] = nu;
#ifdef WT_EMU
        if (lane == 0) { c.U[w] = 0; c.E[w] = 0; }         // (the emulator's lanes run one after the other, lane 0 of a word first)
        c.U[w] |= (uint64_t) (eany ? 1 : 0) << lane;
        c.E[w] |= (uint64_t) (emit ? 1 : 0) << lane;
#else
        const uint64_t ub = __builtin_amdgcn_ballot_w64(eany), eb = __builtin_amdgcn_ballot_w64(emit);
        if (lane == 0) { c.U[w] = ub; c.E[w] = eb; }
#endif
    }
    if (any_risk) d.dsh->risk = 1;
}

// the tail of every emitted position of the window: all lanes, consecutive lanes consecutive positions
WT_DEV void wt_delta_tail_tt(const WtParams &P, WtDeltaCtx &d, int tid, int nt) {
    double *res = (double *) (d.acc + P.W);
    const double *dof = (const double *) (d.qa + P.W);
    for (int p = tid; p < P.W; p += nt) {
        const double t = res[p];
        if (!wt_isnan(t)) res[p] = wt_ttest_tail(t, dof[p]);        // setComparisons.c:117
    }
}

// before the staging of a two-sample launch: the lane's 8 results, from where wt_delta_tail_tt left them
WT_DEV void wt_delta_load_res_tt(const WtParams &P, const WtDeltaCtx &d, WtLane<WT_DELTA_K> &out, int tid) {
    const double *res = (const double *) (d.acc + P.W + tid * WT_DELTA_K);
#pragma unroll
    for (int k = 0; k < WT_DELTA_K; k++) out.res[k] = res[k];
}

// Output staging.  A lane owns 8 consecutive positions, so writing its runs straight to HBM makes
// every store instruction touch 64 scattered 4/8-byte pieces.  Instead the runs are first placed
// in LDS at their rank inside the window (acc[] / ev[] are dead after the scan and are exactly big
// enough: one f64 and one u32 per position), then copied out with consecutive lanes writing
// consecutive runs.  The u32 holds the run's start and finish, both relative to w0 (16 bits each;
// WT_DELTA_FAR: the next breakpoint lies beyond the window, i.e. sh->next_bp): the staging lane has its
// byte of the breakpoint bitmap in a register, so only the run that ends beyond the lane's own 8
// positions needs the bitmap walk (round 3: the copying lanes used to walk it for every run -- three
// dependent LDS round trips per run, 8 runs per lane).
#define WT_DELTA_FAR 0xffffu
// Where run number r of the window is staged: one spare entry after every 32 (round 6).  A lane's runs are consecutive, so at r itself the
// lanes of a wavefront store 8 entries apart where breakpoints are dense: eight lanes to a bank, 32 LDS cycles per 8-byte store instead
// of 6, and the staging was a twentieth of a window.  With the spare entries lane i's k-th store goes to bank (8 i + i / 4 + k) mod 32:
// no two lanes of a group meet, and the copy-out still reads consecutive entries.  (acc[] / ev[] are W + W / 32 entries: wt_make_delta_plan)
#define WT_STAGE_AT(r) ((r) + ((r) >> 5))
template <int OP>
WT_DEV void wt_delta_stage(const WtParams &P, WtCtx &c, WtDeltaCtx &d, const WtLane<WT_DELTA_K> &L, int tid, int nt) {
    const unsigned em = ((const uint8_t *) c.E)[tid];
    if (!em) return;
    const unsigned um = ((const uint8_t *) c.U)[tid];
    const int p0 = tid * WT_DELTA_K;
    const int w = p0 >> 6, b0 = p0 & 63;
    const uint64_t below0 = b0 ? wt_mask_incl(b0 - 1) : 0ull;
    unsigned idx = c.epfx[w] + (unsigned) wt_popc64(c.E[w] & below0);
    // first breakpoint after the lane's last position
    const int32_t after = wt_next_breakpoint(P, c, p0 + WT_DELTA_K - 1);
    const int32_t w0 = c.sh->w0;
    const uint32_t after_rel = after < c.sh->w1 ? (uint32_t) (after - w0) : WT_DELTA_FAR;
    double *sv = (double *) d.acc;
    uint32_t *sp = d.ev;
#pragma unroll
    for (int k = 0; k < WT_DELTA_K; k++) {
        if (!((em >> k) & 1u)) continue;
        const unsigned higher = k + 1 < WT_DELTA_K ? um >> (k + 1) : 0u;
        const uint32_t fin_rel = higher ? (uint32_t) (p0 + k + 1 + wt_ctz64((uint64_t) higher)) : after_rel;
        sv[WT_STAGE_AT(idx)] = L.res[k];
        sp[WT_STAGE_AT(idx)] = (uint32_t) (p0 + k) | (fin_rel << 16);
        idx++;
    }
}

#ifndef WT_EMU
// Sum / Mean on the device (round 6): the lane's emitted byte, breakpoint byte and rank come in registers (wt_delta_scan3_cov); the first
// lane of each of the window's WT_BAD_SUB sub-ranges leaves its rank in epfx[32 + j] for wt_delta_note_offset_ep.
template <int OP>
WT_DEV void wt_delta_stage_ep(const WtParams &P, WtCtx &c, WtDeltaCtx &d, const WtLane<WT_DELTA_K> &L, unsigned em, unsigned um, unsigned idx, int tid, int nt) {
    const int per = nt / WT_BAD_SUB;
    if (c.sh->bad_slot >= 0 && tid % per == 0) c.epfx[32 + tid / per] = idx;
    if (!em) return;
    const int p0 = tid * WT_DELTA_K;
    const int32_t after = wt_next_breakpoint(P, c, p0 + WT_DELTA_K - 1);
    const int32_t w0 = c.sh->w0;
    const uint32_t after_rel = after < c.sh->w1 ? (uint32_t) (after - w0) : WT_DELTA_FAR;
    double *sv = (double *) d.acc;
    uint32_t *sp = d.ev;
#pragma unroll
    for (int k = 0; k < WT_DELTA_K; k++) {
        if (!((em >> k) & 1u)) continue;
        const unsigned higher = k + 1 < WT_DELTA_K ? um >> (k + 1) : 0u;
        const uint32_t fin_rel = higher ? (uint32_t) (p0 + k + 1 + wt_ctz64((uint64_t) higher)) : after_rel;
        sv[WT_STAGE_AT(idx)] = L.res[k];
        sp[WT_STAGE_AT(idx)] = (uint32_t) (p0 + k) | (fin_rel << 16);
        idx++;
    }
}
WT_DEV void wt_delta_note_offset_ep(const WtParams &P, WtCtx &c, int lane) {
    if (c.sh->bad_slot >= 0 && P.bad_goff && lane < WT_BAD_SUB)
        P.bad_goff[(long long) c.sh->bad_slot * WT_BAD_SUB + lane] = c.sh->goffset + (long long) c.epfx[32 + lane];
}
#endif

WT_DEV void wt_delta_copy_out(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int tid, int nt) {
    const int n = c.sh->n_emit;
    const long long goff = c.sh->goffset;
    const int32_t w0 = c.sh->w0, far = c.sh->next_bp;
    const double *sv = (const double *) d.acc;
    const uint32_t *sp = d.ev;
    unsigned long long bp = 0;
#if !defined(WT_EMU) && WT_DELTA_COPY2
    // two consecutive runs per lane and trip: 8- and 16-byte stores (4- / 8-byte aligned: the unaligned access mode of global memory) --
    // half the store instructions of a phase that is bound by issuing them (round 6)
    struct __attribute__((packed, aligned(4))) I2 { int32_t x[2]; };
    struct __attribute__((packed, aligned(8))) D2 { double x[2]; };
    for (int i = 2 * tid; i < n; i += 2 * nt) {
        const unsigned a0 = WT_STAGE_AT((unsigned) i);          // (run i + 1 is staged right behind: i is even)
        const bool two = i + 1 < n;
        const uint32_t pf0 = sp[a0], pf1 = two ? sp[a0 + 1] : 0u;
        const double v0 = sv[a0], v1 = two ? sv[a0 + 1] : 0.0;
        const int32_t st0 = w0 + (int32_t) (pf0 & 0xffffu), st1 = w0 + (int32_t) (pf1 & 0xffffu);
        const int32_t fin0 = (pf0 >> 16) == WT_DELTA_FAR ? far : w0 + (int32_t) (pf0 >> 16);
        const int32_t fin1 = (pf1 >> 16) == WT_DELTA_FAR ? far : w0 + (int32_t) (pf1 >> 16);
        bp += (unsigned long long) (fin0 - st0) + (two ? (unsigned long long) (fin1 - st1) : 0ull);
        const long long o = goff + i;
        if (two && o + 1 < P.capacity) {
            I2 s2, f2; D2 d2;
            s2.x[0] = st0; s2.x[1] = st1; f2.x[0] = fin0; f2.x[1] = fin1; d2.x[0] = v0; d2.x[1] = v1;
            *(I2 *) (P.o_start + o) = s2;
            *(I2 *) (P.o_finish + o) = f2;
            *(D2 *) (P.o_value + o) = d2;
        } else if (o < P.capacity) {
            P.o_start[o] = st0; P.o_finish[o] = fin0; P.o_value[o] = v0;
        }
    }
#else
    for (int i = tid; i < n; i += nt) {
        const uint32_t pf = sp[WT_STAGE_AT((unsigned) i)];
        const int32_t st = w0 + (int32_t) (pf & 0xffffu);
        const int32_t fin = (pf >> 16) == WT_DELTA_FAR ? far : w0 + (int32_t) (pf >> 16);
        bp += (unsigned long long) (fin - st);
        const long long o = goff + i;
        if (o >= P.capacity) continue;
        P.o_start[o] = st;
        P.o_finish[o] = fin;
        P.o_value[o] = sv[WT_STAGE_AT((unsigned) i)];
    }
#endif
    bp = wt_wave_sum_u64(bp);
    if (bp && wt_wave_leader(tid & 63)) wt_lds_add64(&c.sh->bp_sum, bp);
}

// next non-empty word of U after every word (wt_next_breakpoint's jump table); one lane per word,
// each looks ahead until it finds one -- one step where breakpoints are dense
WT_DEV void wt_delta_nextw(const WtParams &P, WtCtx &c, int tid, int nt) {
    for (int w = nt - 1 - tid; w < P.n_words; w += nt) {      // the LAST lanes: wave 0 is busy with the look-back
        int x = w + 1;
        while (x < P.n_words && c.U[x] == 0) x++;
        c.nextw[w] = (int16_t) (x < P.n_words ? x : -1);
    }
}


#ifndef WT_EMU
// ---------------------------------------------------------------------------
// Device flavours of the per-window scans: the wave-level part runs on lane shuffles, so every
// scan is "local + wave scan | barrier | add the totals of the waves before" -- one barrier and
// no serial loop over lanes.  Measured on MI355X: barriers and single-lane chains, not bandwidth,
// bound this kernel (dropping ONE barrier took the 100-track mean from 1.69 to 1.48 ms).  The
// LDS-only versions above are what the CPU emulator executes (it has no lane shuffles).
// ---------------------------------------------------------------------------
// (wave-wide inclusive scans: DPP, wt_core.h)
WT_DEV unsigned wt_wave_scan_u32(unsigned v, int lane) {
    return wt_wave_scan32(v, 0u, [](uint32_t a, uint32_t b) { return a + b; });
}
WT_DEV long long wt_wave_scan_i64(long long v, int lane) { return (long long) wt_wave_scan_add64((unsigned long long) v); }

// ranges: lookup + wave-local exclusive prefix; the wave totals go to gtc[wave]
WT_DEV void wt_delta_ranges_w1(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int c0, int tid, int nt, long long row, int chrom) {
    wt_delta_ranges1(P, c, d, c0, tid, nt, row, chrom);
    const int lane = tid & 63;
    const unsigned n = (unsigned) d.ltc[tid];
    const unsigned incl = wt_wave_scan_u32(n, lane);
    d.tpfx[tid] = incl - n;
    if (lane == 63) d.gtc[tid >> 6] = (int32_t) incl;
}
WT_DEV void wt_delta_ranges_w1(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int c0, int tid, int nt) {
    wt_delta_ranges_w1(P, c, d, c0, tid, nt, c.sh->row, c.sh->chrom);
}

WT_DEV void wt_delta_ranges_w2(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int tid, int nt, uint32_t tile_runs = WT_DELTA_TILE) {
    const int wave = tid >> 6;
    uint32_t pfx = d.tpfx[tid];
    pfx += wt_waves_before32((const uint32_t *) d.gtc, 0, wave, tid & 63);
    const uint32_t n = (uint32_t) d.ltc[tid];
    d.tpfx[tid] = pfx;
    d.tbase[tid] -= 4ll * (long long) pfx;
    if (tid == nt - 1) d.tpfx[nt] = pfx + n;
    for (uint32_t b = (pfx + tile_runs - 1) / tile_runs; b * tile_runs < pfx + n && b < WT_DELTA_TF; b++)
        d.tfirst[b] = (uint16_t) tid;
}

// value / coverage scan, step 1: the lane's 8 positions + the wave-level prefix
template <bool QQ = false>
WT_DEV void wt_delta_scan_w1(const WtParams &P, WtCtx &c, WtDeltaCtx &d, WtDeltaLane &L, int tid, int nt) {
    wt_delta_scan1<QQ>(P, c, d, L, tid, nt);
    const int lane = tid & 63;
    const long long rv = L.tv;
    const int32_t rc = L.tc;
    const long long iv = wt_wave_scan_i64(rv, lane);
    const int32_t ic = (int32_t) wt_wave_scan_u32((unsigned) rc, lane);
    L.wv = iv - rv;
    L.wc = ic - rc;
    if (lane == 63) { d.gtv[tid >> 6] = iv; d.gtc[tid >> 6] = ic; }
    if constexpr (QQ) {
        const unsigned long long ra = L.tqa, rb = L.tqb;
        const unsigned long long ia = (unsigned long long) wt_wave_scan_i64((long long) ra, lane);
        const unsigned long long ib = (unsigned long long) wt_wave_scan_i64((long long) rb, lane);
        L.wqa = ia - ra;
        L.wqb = ib - rb;
        if (lane == 63) { d.gtqa[tid >> 6] = ia; d.gtqb[tid >> 6] = ib; }
    }
}

// ... of a min / max launch: coverage only
WT_DEV void wt_delta_scan_w1_mm(const WtParams &P, WtCtx &c, WtDeltaCtx &d, int32_t &wc, int tid, int nt) {
    int32_t tc;
    wt_delta_scan1_mm(P, c, d, tc, tid, nt);
    const int lane = tid & 63;
    const int32_t ic = (int32_t) wt_wave_scan_u32((unsigned) tc, lane);
    wc = ic - tc;
    if (lane == 63) d.gtc[tid >> 6] = ic;
}

// ... of a two-sample launch: the lane's set's four totals (gt*[tid >> 6]: the wave totals; waves are whole in one set)
WT_DEV void wt_delta_scan_w1_tt(const WtParams &P, WtCtx &c, WtDeltaCtx &d, WtDeltaLane2 &L, int tid, int nts) {
    wt_delta_scan1_tt(P, c, d, L, tid, nts);
    const int lane = tid & 63, wave = tid >> 6;
    const long long iv = wt_wave_scan_i64(L.tv, lane);
    const int32_t ic = (int32_t) wt_wave_scan_u32((unsigned) L.tc, lane);
    const unsigned long long ia = (unsigned long long) wt_wave_scan_i64((long long) L.tqa, lane);
    const unsigned long long ib = (unsigned long long) wt_wave_scan_i64((long long) L.tqb, lane);
    L.wv = iv - L.tv;
    L.wc = ic - L.tc;
    L.wqa = ia - L.tqa;
    L.wqb = ib - L.tqb;
    if (lane == 63) { d.gtv[wave] = iv; d.gtc[wave] = ic; d.gtqa[wave] = ia; d.gtqb[wave] = ib; }
}

// escan on wave 0: run-count prefix of the emitted bitmap; returns the window's run count
WT_DEV unsigned wt_delta_escan_wave(const WtParams &P, WtCtx &c, int lane) {
    // one lane per 64-position word, two words per lane for the 8192-bp window (128 words)
    const unsigned v0 = lane < P.n_words ? (unsigned) wt_popc64(c.E[lane]) : 0u;
    const unsigned incl0 = wt_wave_scan_u32(v0, lane);
    if (lane < P.n_words) c.epfx[lane + 1] = incl0;
    if (lane == 0) c.epfx[0] = 0;
    unsigned total = wt_wave_last32(incl0);
    if (P.n_words > 64) {
        const unsigned v1 = lane + 64 < P.n_words ? (unsigned) wt_popc64(c.E[lane + 64]) : 0u;
        const unsigned incl1 = wt_wave_scan_u32(v1, lane) + total;
        if (lane + 64 < P.n_words) c.epfx[lane + 65] = incl1;
        total = wt_wave_last32(incl1);
    }
    return total;
}

// ---------------------------------------------------------------------------
// Round 6: Sum / Mean publish their run count BEFORE the values are computed.  The look-back of a window waits for its
// predecessors to PUBLISH (8-11 % of a window), and a count needs only the coverage scan: scan 3 is split into the
// breakpoint / emitted bytes (wt_delta_scan3_cov) -- then wave 0 scans the counts and publishes -- and the values
// (wt_delta_scan3_val: the conversions and Mean's division), which the other wavefronts compute meanwhile and wave 0 right after.
// ---------------------------------------------------------------------------
// (`em`: the lane's emitted byte; returns the emitted runs of the wavefront's lanes before this one; the wavefront's total goes to
//  epfx[wave] -- the run-count scan over the 64-bit words of E, wt_delta_escan_wave, is not needed: a lane's rank is the sum of the
//  wavefronts before it, wt_waves_before32, and this)
WT_DEV uint32_t wt_delta_scan3_cov(const WtParams &P, WtCtx &c, WtDeltaCtx &d, const WtDeltaLane &L, uint32_t &em_out, int tid, int nt) {
    int32_t bc = d.dsh->base_c;
    bc += (int32_t) wt_waves_before32((const uint32_t *) d.gtc, 0, tid >> 6, tid & 63);
    bc += L.wc;
    const int N = P.n_tracks;
    const bool strict = (P.flags & WT_STRICT_SET0) != 0;
    const int p0 = tid * WT_DELTA_K;
    const long long room = (long long) c.sh->emit_hi - ((long long) c.sh->w0 + p0);
    const uint32_t evmask = L.evmask;
    uint32_t em = 0;
#pragma unroll
    for (int k = 0; k < WT_DELTA_K; k++) {
        const int32_t cov = bc + L.pc[k];
        const bool pred = strict ? (cov == N) : (cov > 0);       // multiplexer.c:120,125
        if (((evmask >> k) & 1u) && pred && k < room) em |= 1u << k;
    }
    ((uint8_t *) c.U)[tid] = (uint8_t) evmask;
    ((uint8_t *) c.E)[tid] = (uint8_t) em;
    em_out = em;
    const uint32_t cnt = (uint32_t) wt_popc32(em);
    const uint32_t incl = wt_wave_scan_u32(cnt, tid & 63);
    if ((tid & 63) == 63) c.epfx[tid >> 6] = incl;
    return incl - cnt;
}
template <int OP>
WT_DEV void wt_delta_scan3_val(const WtParams &P, WtCtx &c, WtDeltaCtx &d, const WtDeltaLane &L, WtLane<WT_DELTA_K> &out, int emin, int tid, int nt) {
    long long bv = d.dsh->base_v;
    bv += (long long) wt_waves_before64((const unsigned long long *) d.gtv, 0, tid >> 6, tid & 63);
    bv += L.wv;
    const int N = P.n_tracks;
    const double q = __builtin_bit_cast(double, (uint64_t) (emin - 150 + 1023) << 52);
    const double dn = (double) N, yn = 1.0 / dn;            // (Mean: wt_div_n)
#pragma unroll
    for (int k = 0; k < WT_DELTA_K; k++) {
        const double s = (double) (bv + L.pv[k]) * q;
        out.res[k] = (OP == WT_OP_MEAN) ? wt_div_n(s, dn, yn) : s;
    }
}

#endif

#endif  // WT_DELTA_H_
