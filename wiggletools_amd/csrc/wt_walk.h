// wt_walk.h -- MedianReduction by walking (device + -DWT_EMU), included by wt_core.h after wt_delta.h.
//
// The bitmap kernel (wt_reduce_kernel) evaluates every run of the output from scratch: N gathers and a sorting
// network per position, although between neighbouring runs only the tracks with a breakpoint there change -- ~6 of
// 100 at the benchmark's density (reducers.c:780-813 itself re-sorts all N values per pop).  Here a STRETCH of S
// consecutive positions of the window is walked by the same lane -- or, by default, by a PAIR of lanes that hold half the
// tracks each -- which carries the column of the current values (order-preserving keys, in LDS, [row][lane]) from one
// position to the next.  Per window (a barrier between the phases):
//
//   ranges   the window's runs of every track as one flat index space (wt_delta_ranges1, wt_walk_ranges*)
//   pass     one pass over the runs: an EVENT at a run's start (the track takes the run's value) and one at its finish (the
//            track falls back to its default) unless the next run of the track starts right there.  An event is counted
//            in its position's counter word (with the coverage change it brings) and placed in one of the position's
//            fixed slots of the workgroup's slab (global memory), beyond them in an overflow list.  A run that covers the
//            first position of a stretch from before it writes its value into that stretch's column: the columns start
//            out right without any look-up.
//   emits    from the counter words alone: which positions start an emitted run; the window's run count is scanned and
//            published to the look-back chain before anything else happens
//   walk     position by position: apply the events (old key out, new key in, the counts of keys below / not above the
//            current median m kept up to date), and when the wanted rank k = N/2 has left [lt, le) move m: ONE sweep over
//            the column collects the 2 / 4 / 8 nearest keys on the side m has to move to (ties are handled as a multiset,
//            so the result is the exact order statistic, bit for bit what a sort gives).  A stretch's first median
//            starts from the last one any lane of the workgroup found (same distribution: a few ranks off).
//            Pair mode: each lane sweeps its half, counts and candidates are exchanged inside the pair (DPP).
//   write    the look-back completes, the lanes write their runs
//   (fallback, a window with too many events beyond their slots: offsets from the counts, a second pass sorting the
//    events by position into the same slab, in rounds of as many stretches as fit; same walk)
//
// Eligibility (host): float tracks, float-exact defaults, N <= 128.
// The output is the bitmap kernel's, bit for bit (same keys, same order statistic, NaN if any value is NaN).
// DESIGN 4.5 has the measurements.
#ifndef WT_WALK_H_
#define WT_WALK_H_
#ifdef WT_EMU
#include <atomic>
#endif

#define WT_WALK_NANKEY 0xfffffffeu     // every NaN (no other value has this key: wt_walk_key)
#define WT_WALK_INC 0x10000u           // event: the track becomes covered
#define WT_WALK_DEC 0x20000u           // event: the track stops being covered
#define WT_WALK_MAX_S 32               // (the lanes' position masks are 32 bits)
#define WT_WALK_PAD 8                  // col[] has a multiple of this many rows (the sweeps' block)
// a position's counter word: events (12 bits) | of which the track becomes covered (10 bits) | stops being covered (10 bits)
#define WT_WALK_NMASK 0xfffu
#define WT_WALK_CINC (1u << 12)
#define WT_WALK_CDEC (1u << 22)
// pair mode, while the events sit in their slots: events of the even tracks (7 bits) | of the odd tracks (7) | covered (8) |
// uncovered (8) -- every lane of a pair has its own slots (half of the position's) and its own count
#define WT_WALK_PNMASK 0x7fu
#define WT_WALK_PCINC (1u << 14)
#define WT_WALK_PCDEC (1u << 22)
#ifndef WT_WALK_PAIR_DEFAULT
#define WT_WALK_PAIR_DEFAULT 1          // two lanes per stretch (WTAMD_WALK_PAIR=0: one)
#endif
#define WT_WALK_TF 256                 // tiles of the flat run space whose first track is tabulated (wt_delta.h tabulates 2048: 4 KB)
#define WT_WALK_OV_SCAN 32             // events beyond their positions' slots a window may have and still be walked from the slots

struct alignas(8) WtWalkEvent { uint32_t key, meta; };     // meta: track | WT_WALK_INC | WT_WALK_DEC
struct WtWalkOvf { uint32_t pos, key, meta; };             // an event that did not fit its position's fixed slots

struct WtWalkCtx {
    uint32_t *col;      // [N][T] current keys, one column per lane
    uint32_t *cnt;      // [W] events per position (count pass); slot countdown (scatter); the lanes' results (walk)
    uint32_t *off;      // [W + 1] fallback: first event of every position (index into the window's event sequence); in the slab
    uint32_t *tot;      // [T] scan scratch
    uint32_t *base;     // [T + 1] scan result
    uint32_t *gt;       // [T / 64 + 1] wave totals
    int32_t *ncov;      // [T] tracks covering the position before the lane's first one
    int32_t *fe;        // [T] first position (window-relative) of the lane's stretch that has an event, or -1
    uint32_t *dkey;     // [N] keys of the defaults
    uint32_t *guess;    // [1] a recent median of this workgroup (where a stretch's first selection starts)
    WtWalkEvent *slab;  // this workgroup's events (global): capp slots per position, or (fallback) the sorted sequence
    WtWalkOvf *ovf;     // ... behind the fixed slots: the events beyond a position's slots
    uint32_t *novf;     // [1] how many of those (LDS)
    uint32_t cap;       // events the slab holds (sorted use)
    uint32_t ov_cap;    // entries of ovf[]
    int capp;           // fixed slots per position (a power of two)
    int npad;           // rows of col[]: N rounded up to the sweeps' block (the extra rows hold 0xffffffff)
    int S;              // positions per stretch (a power of two)
    int logS;
    int pair;           // 1: TWO lanes per stretch -- lane 2q + h holds the tracks i = h (mod 2) of stretch q (row i >> 1): half the
                        //    column per lane, twice the lanes per CU; counts and candidates are exchanged inside the lane pair
    int nstr;           // stretches of the window (lanes, or half of them)
    int mwu;            // 1: MWUReduction's walk (wt_mwalk.h; pair mode): the lane's "half" is its SET, row r of lane h holds track
                        //    r of set h -- tracks are renumbered 2 r + h wherever this file says "track" (wt_mw_vtrack)
};

// MWUReduction by walking: track -> its number in the pair mode's terms (parity = set = the lane of the pair, >> 1 = row), and back
WT_DEV int wt_mw_vtrack(const WtParams &P, int trk) { return trk < P.n_set0 ? 2 * trk : 2 * (trk - P.n_set0) + 1; }
WT_DEV int wt_mw_row_track(const WtParams &P, int r, int h) {
    const int n1 = P.n_set0, n2 = P.n_tracks - P.n_set0;
    if (h == 0) return r < n1 ? r : -1;
    return r < n2 ? n1 + r : -1;
}

WT_DEV void wt_walk_ctx_init(WtWalkCtx &w, const WtParams &P, char *lds, char *slab) {
    w.col = (uint32_t *) (lds + P.off_wcol);
    w.cnt = (uint32_t *) (lds + P.off_wcnt);
    w.tot = (uint32_t *) (lds + P.off_wtot);
    w.base = (uint32_t *) (lds + P.off_wbase);
    w.gt = (uint32_t *) (lds + P.off_wgt);
    w.ncov = (int32_t *) (lds + P.off_wncov);
    w.fe = (int32_t *) (lds + P.off_wfe);
    w.dkey = (uint32_t *) (lds + P.off_wdk);
    w.guess = (uint32_t *) (lds + P.off_wguess);
    w.slab = (WtWalkEvent *) slab;
    w.cap = (uint32_t) ((long long) P.walk_off_at / (long long) sizeof(WtWalkEvent));
    w.off = (uint32_t *) (slab + P.walk_off_at);
    w.capp = P.walk_capp;
    w.ovf = (WtWalkOvf *) (slab + (size_t) P.W * (size_t) P.walk_capp * sizeof(WtWalkEvent));
    w.ov_cap = (uint32_t) (P.walk_ov < WT_WALK_OV_SCAN ? P.walk_ov : WT_WALK_OV_SCAN);
    w.novf = w.guess + 1;
    w.pair = P.walk_pair;
    w.mwu = P.walk_mwu;
    w.nstr = P.W / P.walk_S;
    const int n_big = P.n_set0 > P.n_tracks - P.n_set0 ? P.n_set0 : P.n_tracks - P.n_set0;
    w.npad = ((w.mwu ? n_big : (w.pair ? (P.n_tracks + 1) / 2 : P.n_tracks)) + WT_WALK_PAD - 1) & ~(WT_WALK_PAD - 1);
    w.S = P.walk_S;
    w.logS = 0;
    while ((1 << w.logS) < w.S) w.logS++;
}

// ---- the two lanes of a stretch (pair mode): the partner lane's value ----
#ifdef WT_EMU
// (the emulator runs the two lanes of a pair on two threads, one pair at a time: a rendezvous through a mailbox)
struct WtEmuPairBox {
    std::atomic<uint32_t> box[2];
    std::atomic<int> count{0}, sense{0};
    int local_sense[2] = {0, 0};
    void barrier(int h) {
        const int sns = (local_sense[h] ^= 1);
        if (count.fetch_add(1) == 1) { count.store(0); sense.store(sns); }
        else while (sense.load() != sns) { }
    }
    uint32_t xchg(int h, uint32_t x) {
        box[h].store(x);
        barrier(h);
        const uint32_t r = box[1 - h].load();
        barrier(h);
        return r;
    }
};
inline WtEmuPairBox &wt_emu_pairbox() { static WtEmuPairBox b; return b; }
WT_DEV uint32_t wt_pair_xchg(uint32_t x, int tid) { return wt_emu_pairbox().xchg(tid & 1, x); }
#else
// quad_perm [1, 0, 3, 2]: lanes 2q and 2q + 1 swap (both are active wherever this is called: a pair never diverges)
WT_DEV uint32_t wt_pair_xchg(uint32_t x, int tid) { return (uint32_t) __builtin_amdgcn_mov_dpp((int) x, 0xB1, 0xf, 0xf, false); }
#endif

// order-preserving key of the float with bits `vb`; every NaN gets the same one, above +Inf
WT_DEV uint32_t wt_walk_key(uint32_t vb) {
    return (vb & 0x7fffffffu) > 0x7f800000u ? WT_WALK_NANKEY : wt_key32(__builtin_bit_cast(float, vb));
}

// once per workgroup
WT_DEV void wt_walk_defaults(const WtParams &P, WtWalkCtx &w, int tid, int nt) {
    for (int i = tid; i < P.n_tracks; i += nt) w.dkey[i] = wt_walk_key(__builtin_bit_cast(uint32_t, (float) P.defaults[i]));
    if (tid == 0) w.guess[0] = 0x80000000u;        // (the key of 0.0; any key will do)
}

// per window: no events, nothing covered, every column holds the defaults
WT_DEV void wt_walk_zero(const WtParams &P, const WtCtx &c, WtWalkCtx &w, int tid, int nt) {
    for (int x = tid; x < P.W; x += nt) w.cnt[x] = 0;
    if (tid < w.nstr || w.mwu) w.ncov[tid] = 0;         // (MWU: per lane -- the tracks of the lane's SET in play)
    const int N = P.n_tracks, h = w.pair ? tid & 1 : 0;
    for (int r = 0; r < w.npad; r++) {
        const int i = w.mwu ? wt_mw_row_track(P, r, h) : (w.pair ? 2 * r + h : r);      // the track of row r of this lane's column
        w.col[r * nt + tid] = (i >= 0 && i < N) ? w.dkey[i] : 0xffffffffu;      // (rows past the tracks: above every key, never NaN's)
    }
    if (tid == 0) w.novf[0] = 0;
}

// ---- the window's runs as the flat index space of wt_delta.h (wt_delta_ranges1 + the scans below: the same as
// wt_delta_ranges2 / 3 and wt_delta_ranges_w2 with a shorter table of tile starts) ----
#ifdef WT_EMU
WT_DEV void wt_walk_ranges3(WtDeltaCtx &d, int tid, int nt) {
    uint32_t pfx = 0;
    for (int x = 0; x < tid; x++) pfx += (uint32_t) d.ltc[x];
    const uint32_t n = (uint32_t) d.ltc[tid];
    d.tpfx[tid] = pfx;
    d.tbase[tid] -= 4ll * (long long) pfx;
    if (tid == nt - 1) d.tpfx[nt] = pfx + n;
    for (uint32_t b = (pfx + WT_DELTA_TILE - 1) / WT_DELTA_TILE; b * WT_DELTA_TILE < pfx + n && b < WT_WALK_TF; b++)
        d.tfirst[b] = (uint16_t) tid;
}
#else
WT_DEV void wt_walk_ranges_w2(WtDeltaCtx &d, int tid, int nt) {
    const int wave = tid >> 6;
    uint32_t pfx = d.tpfx[tid];
    pfx += wt_waves_before32((const uint32_t *) d.gtc, 0, wave, tid & 63);
    const uint32_t n = (uint32_t) d.ltc[tid];
    d.tpfx[tid] = pfx;
    d.tbase[tid] -= 4ll * (long long) pfx;
    if (tid == nt - 1) d.tpfx[nt] = pfx + n;
    for (uint32_t b = (pfx + WT_DELTA_TILE - 1) / WT_DELTA_TILE; b * WT_DELTA_TILE < pfx + n && b < WT_WALK_TF; b++)
        d.tfirst[b] = (uint16_t) tid;
}
#endif

struct WtWalkBatch {
    int32_t s[WT_DELTA_U], f[WT_DELTA_U];
    int32_t ps[WT_DELTA_U], ns[WT_DELTA_U];     // finish of the track's previous run, start of its next one (of the FILE, not the window)
    uint32_t b[WT_DELTA_U];
    int trk[WT_DELTA_U];
    bool first[WT_DELTA_U], last[WT_DELTA_U];   // the run is the track's first / last of the window
};

// (unconditional, always in range: see wt_delta_fetch)
WT_DEV void wt_walk_fetch(const WtParams &P, const WtDeltaCtx &d, const WtWalkCtx &w, int nt, uint32_t M, uint32_t tb, int lane, WtWalkBatch &B) {
    const uint32_t lastt = (M - 1u) / WT_DELTA_TILE * WT_DELTA_TILE;
    const uint32_t tbe = tb < lastt ? tb : lastt;
    const uint32_t tile = tbe / WT_DELTA_TILE;
    int i = tile < WT_WALK_TF ? (int) d.tfirst[tile] : wt_delta_find(d.tpfx, nt, tbe, 0);
    uint32_t lo = d.tpfx[i], hi = d.tpfx[i + 1];
    long long dl = d.tbase[i];
#pragma unroll
    for (int u = 0; u < WT_DELTA_U; u++) {
        uint32_t jj = tbe + (uint32_t) lane + 64u * (uint32_t) u;
        jj = jj < M - 1u ? jj : M - 1u;
        if (jj >= hi) {
            do { i++; lo = hi; hi = d.tpfx[i + 1]; } while (jj >= hi);
            dl = d.tbase[i];
        }
        const long long ob = dl + ((long long) jj << 2);
        // first / last run of the track IN THIS WINDOW: what the events need to know.  A first run that starts inside the
        // window follows a run that ended before it (the range starts at the first run with finish >= w0): it covers anew,
        // like the first run of the file.  A last run that ends inside the window is the file's last (the range ends at
        // the first run with finish >= w1): nothing starts where it ends.
        const bool fst = jj == lo, lst = jj + 1u == hi;
        B.s[u] = *(const int32_t *) ((const char *) P.start + ob);
        B.f[u] = *(const int32_t *) ((const char *) P.finish + ob);
        B.b[u] = *(const uint32_t *) ((const char *) P.value + ob);
        B.ps[u] = *(const int32_t *) ((const char *) P.finish + (fst ? ob : ob - 4));
        B.ns[u] = *(const int32_t *) ((const char *) P.start + (lst ? ob : ob + 4));
        B.trk[u] = i;
        B.first[u] = fst; B.last[u] = lst;
    }
}

#ifdef WT_EMU
WT_DEV uint32_t wt_lds_add_rtn(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
#else
WT_DEV uint32_t wt_lds_add_rtn(uint32_t *p, uint32_t v) { return atomicAdd((unsigned int *) p, v); }
#endif

// the fields of a counter word (as the first pass leaves it)
WT_DEV uint32_t wt_walk_cnt_n(const WtWalkCtx &w, uint32_t v) { return w.pair ? (v & WT_WALK_PNMASK) + ((v >> 7) & WT_WALK_PNMASK) : (v & WT_WALK_NMASK); }
WT_DEV int wt_walk_cnt_dcov(const WtWalkCtx &w, uint32_t v) {
    return w.pair ? (int) ((v >> 14) & 0xffu) - (int) ((v >> 22) & 0xffu) : (int) ((v >> 12) & 0x3ffu) - (int) (v >> 22);
}

// an event of position cs: counted, and placed in one of the position's fixed slots -- or, beyond them, in the overflow list
WT_DEV void wt_walk_place(WtWalkCtx &w, int32_t cs, uint32_t key, uint32_t meta) {
    // (the coverage changes ride in the same word: the window's run count is known before any event is applied)
    uint32_t slot, cap, at;
    if (w.pair) {
        const uint32_t half = meta & 1u;        // (the track's parity: the lane of the pair that holds it)
        slot = (wt_lds_add_rtn(&w.cnt[cs], (1u << (7u * half)) + ((meta & WT_WALK_INC) ? WT_WALK_PCINC : 0u) + ((meta & WT_WALK_DEC) ? WT_WALK_PCDEC : 0u)) >> (7u * half)) & WT_WALK_PNMASK;
        cap = (uint32_t) w.capp >> 1;
        at = (((uint32_t) cs << 1) | half) * cap;
    } else {
        slot = wt_lds_add_rtn(&w.cnt[cs], 1u + ((meta & WT_WALK_INC) ? WT_WALK_CINC : 0u) + ((meta & WT_WALK_DEC) ? WT_WALK_CDEC : 0u)) & WT_WALK_NMASK;
        cap = (uint32_t) w.capp;
        at = (uint32_t) cs * cap;
    }
    if (slot < cap) {
        WtWalkEvent e;
        e.key = key; e.meta = meta;
        w.slab[at + slot] = e;
    } else {
        const uint32_t j = wt_lds_add_rtn(w.novf, 1u);
        if (j < w.ov_cap) {
            WtWalkOvf o;
            o.pos = (uint32_t) cs; o.key = key; o.meta = meta;
            w.ovf[j] = o;
        }
    }
}

// The first pass over ONE TILE of runs (round 5).  wt_walk_count1 below does a run at a time: up to two LDS atomics WITH
// return per run (the event's slot in its position), each followed by the store that needs the returned slot -- eight
// dependent LDS round trips per tile and lane, behind whatever the other wavefronts have queued, at two wavefronts per
// SIMD.  Here the tile's (up to) eight atomics and the four reads of the tracks' default keys are issued back to back and
// waited for once; the events are stored and the columns initialised afterwards.  Same events, same counter words; the
// order in which a position's slots are handed out was never defined (the lanes race for them).
#ifndef WT_WALK_COUNT_TILE
#define WT_WALK_COUNT_TILE 1
#endif
WT_DEV uint32_t wt_walk_place_inc(const WtWalkCtx &w, uint32_t meta) {
    if (w.pair) return (1u << (7u * (meta & 1u))) + ((meta & WT_WALK_INC) ? WT_WALK_PCINC : 0u) + ((meta & WT_WALK_DEC) ? WT_WALK_PCDEC : 0u);
    return 1u + ((meta & WT_WALK_INC) ? WT_WALK_CINC : 0u) + ((meta & WT_WALK_DEC) ? WT_WALK_CDEC : 0u);
}
// `old`: what the atomic on cnt[cs] returned
WT_DEV void wt_walk_place_store(WtWalkCtx &w, int32_t cs, uint32_t key, uint32_t meta, uint32_t old) {
    uint32_t slot, cap, at;
    if (w.pair) {
        const uint32_t half = meta & 1u;
        slot = (old >> (7u * half)) & WT_WALK_PNMASK;
        cap = (uint32_t) w.capp >> 1;
        at = (((uint32_t) cs << 1) | half) * cap;
    } else {
        slot = old & WT_WALK_NMASK;
        cap = (uint32_t) w.capp;
        at = (uint32_t) cs * cap;
    }
    if (slot < cap) {
        WtWalkEvent e;
        e.key = key; e.meta = meta;
        w.slab[at + slot] = e;
    } else {
        const uint32_t j = wt_lds_add_rtn(w.novf, 1u);
        if (j < w.ov_cap) {
            WtWalkOvf o;
            o.pos = (uint32_t) cs; o.key = key; o.meta = meta;
            w.ovf[j] = o;
        }
    }
}
WT_DEV void wt_walk_count_tile(const WtParams &P, WtWalkCtx &w, int nt, int32_t w0, int32_t width, const WtWalkBatch &B, uint32_t tb, uint32_t M,
                               int lane, int32_t &my_next) {
    bool inw[WT_DELTA_U], ds[WT_DELTA_U], df[WT_DELTA_U];
    int32_t cs[WT_DELTA_U], cf[WT_DELTA_U];
    uint32_t key[WT_DELTA_U], kf[WT_DELTA_U], ms[WT_DELTA_U], mf[WT_DELTA_U], rs[WT_DELTA_U], rf[WT_DELTA_U];
    int vt[WT_DELTA_U];
#pragma unroll
    for (int u = 0; u < WT_DELTA_U; u++) {
        const bool valid = tb + (uint32_t) lane + 64u * (uint32_t) u < M;
        const int32_t s = B.s[u], f = B.f[u];
        cs[u] = s - w0; cf[u] = f - w0;         // cf >= 0: the window's runs finish at or beyond w0
        inw[u] = valid && cs[u] < width;
        if (valid && cs[u] >= width) my_next = s < my_next ? s : my_next;
        if (inw[u] && cf[u] >= width) my_next = f < my_next ? f : my_next;
        key[u] = wt_walk_key(B.b[u]);
        vt[u] = w.mwu ? wt_mw_vtrack(P, B.trk[u]) : B.trk[u];
        ds[u] = inw[u] && cs[u] >= 0;
        df[u] = inw[u] && cf[u] < width && (B.last[u] || B.ns[u] != f);    // (else the next run's start event says it all)
        ms[u] = (uint32_t) vt[u] | ((B.first[u] || B.ps[u] != s) ? WT_WALK_INC : 0u);
        mf[u] = (uint32_t) vt[u] | WT_WALK_DEC;
    }
    // everything that has to come back from the LDS, in one go
#pragma unroll
    for (int u = 0; u < WT_DELTA_U; u++) kf[u] = w.dkey[B.trk[u]];
#pragma unroll
    for (int u = 0; u < WT_DELTA_U; u++) rs[u] = ds[u] ? wt_lds_add_rtn(&w.cnt[cs[u]], wt_walk_place_inc(w, ms[u])) : 0u;
#pragma unroll
    for (int u = 0; u < WT_DELTA_U; u++) rf[u] = df[u] ? wt_lds_add_rtn(&w.cnt[cf[u]], wt_walk_place_inc(w, mf[u])) : 0u;
#pragma unroll
    for (int u = 0; u < WT_DELTA_U; u++) {
        if (ds[u]) wt_walk_place_store(w, cs[u], key[u], ms[u], rs[u]);
        if (df[u]) wt_walk_place_store(w, cf[u], kf[u], mf[u], rf[u]);
    }
    // the lanes whose first position a lies in (s, f]: the run covers the position before a
#pragma unroll
    for (int u = 0; u < WT_DELTA_U; u++) {
        if (!inw[u]) continue;
        int l = cs[u] < 0 ? 0 : (cs[u] >> w.logS) + 1;     // (no division: 30 instructions each on this machine)
        int lh = cf[u] >> w.logS;
        if (lh > w.nstr - 1) lh = w.nstr - 1;
        const int row = (vt[u] >> w.pair) * nt, half = vt[u] & w.pair;
        for (; l <= lh; l++) {
            w.col[row + ((l << w.pair) | half)] = key[u];
            wt_lds_addi32(&w.ncov[w.mwu ? ((l << 1) | half) : l], 1);
        }
    }
}

// one run, first pass: its events counted and placed, the columns it covers from before initialised
WT_DEV void wt_walk_count1(const WtParams &P, WtWalkCtx &w, int nt, int32_t w0, int32_t width, int trk, int32_t s, int32_t f, int32_t ps,
                           int32_t ns, bool first, bool last, uint32_t vb, int32_t &my_next) {
    const int32_t cs = s - w0, cf = f - w0;     // cf >= 0: the window's runs finish at or beyond w0
    if (cs >= width) { my_next = s < my_next ? s : my_next; return; }
    const uint32_t key = wt_walk_key(vb);
    const int vt = w.mwu ? wt_mw_vtrack(P, trk) : trk;      // (the track's number in the events and the columns)
    if (cs >= 0) wt_walk_place(w, cs, key, (uint32_t) vt | ((first || ps != s) ? WT_WALK_INC : 0u));
    if (cf < width) {
        if (last || ns != f) wt_walk_place(w, cf, w.dkey[trk], (uint32_t) vt | WT_WALK_DEC);      // (else the next run's start event says it all)
    } else {
        my_next = f < my_next ? f : my_next;
    }
    // the lanes whose first position a lies in (s, f]: the run covers the position before a
    int l = cs < 0 ? 0 : (cs >> w.logS) + 1;       // (no division: 30 instructions each on this machine)
    int lh = cf >> w.logS;
    if (lh > w.nstr - 1) lh = w.nstr - 1;
    const int row = (vt >> w.pair) * nt, half = vt & w.pair;
    for (; l <= lh; l++) {
        w.col[row + ((l << w.pair) | half)] = key;
        wt_lds_addi32(&w.ncov[w.mwu ? ((l << 1) | half) : l], 1);
    }
}

// one run, scatter pass of the fallback (cnt[] counts down: the slots of a position are handed out from the last to the first)
WT_DEV void wt_walk_scatter1(const WtParams &P, WtWalkCtx &w, int32_t w0, int32_t width, uint32_t ev0, uint32_t ev1, int trk, int32_t s,
                             int32_t f, int32_t ps, int32_t ns, bool first, bool last, uint32_t vb) {
    const int32_t cs = s - w0, cf = f - w0;
    if (cs >= width) return;
    if (cs >= 0) {
        const uint32_t at = w.off[cs];
        if (at >= ev0 && at < ev1) {                            // (positions of the lanes of this round)
#ifdef WT_EMU
            const uint32_t old = (w.cnt[cs]--) & WT_WALK_NMASK;
#else
            const uint32_t old = atomicSub((unsigned int *) &w.cnt[cs], 1u) & WT_WALK_NMASK;
#endif
            WtWalkEvent e;
            e.key = wt_walk_key(vb);
            e.meta = (uint32_t) (w.mwu ? wt_mw_vtrack(P, trk) : trk) | ((first || ps != s) ? WT_WALK_INC : 0u);
            w.slab[at - ev0 + old - 1u] = e;
        }
    }
    if (cf < width && (last || ns != f)) {
        const uint32_t at = w.off[cf];
        if (at >= ev0 && at < ev1) {
#ifdef WT_EMU
            const uint32_t old = (w.cnt[cf]--) & WT_WALK_NMASK;
#else
            const uint32_t old = atomicSub((unsigned int *) &w.cnt[cf], 1u) & WT_WALK_NMASK;
#endif
            WtWalkEvent e;
            e.key = w.dkey[trk];
            e.meta = (uint32_t) (w.mwu ? wt_mw_vtrack(P, trk) : trk) | WT_WALK_DEC;
            w.slab[at - ev0 + old - 1u] = e;
        }
    }
}

// The passes over the window's runs.  SCATTER == false: the first pass -- events counted per position and placed in the
// position's fixed slots (all a window of ordinary data needs).  SCATTER == true: the fallback's second pass, the events
// whose position's first slot lies in [ev0, ev1) to their place in the sorted sequence.
template <bool SCATTER>
WT_DEV void wt_walk_pass(const WtParams &P, WtCtx &c, WtWalkCtx &w, WtDeltaCtx &d, uint32_t ev0, uint32_t ev1, int tid, int nt) {
    const int wave = wt_uniform32(tid >> 6), lane = tid & 63, nwaves = nt >> 6;
    const uint32_t M = (uint32_t) wt_uniform32((int32_t) d.tpfx[nt]);
    const int32_t w0 = wt_uniform32(c.sh->w0);
    const int32_t width = wt_uniform32(c.sh->w1) - w0;
    const uint32_t step = (uint32_t) nwaves * WT_DELTA_TILE;
    int32_t my_next = 0x7fffffff;
    auto apply = [&](const WtWalkBatch &B, uint32_t tb) {
#if WT_WALK_COUNT_TILE
        if (!SCATTER) { wt_walk_count_tile(P, w, nt, w0, width, B, tb, M, lane, my_next); return; }
#endif
#pragma unroll
        for (int u = 0; u < WT_DELTA_U; u++)
            if (tb + (uint32_t) lane + 64u * (uint32_t) u < M) {
                if (SCATTER) wt_walk_scatter1(P, w, w0, width, ev0, ev1, B.trk[u], B.s[u], B.f[u], B.ps[u], B.ns[u], B.first[u], B.last[u], B.b[u]);
                else wt_walk_count1(P, w, nt, w0, width, B.trk[u], B.s[u], B.f[u], B.ps[u], B.ns[u], B.first[u], B.last[u], B.b[u], my_next);
            }
    };
    uint32_t tb = (uint32_t) wave * WT_DELTA_TILE;
    if (tb < M) {
        // two register sets take turns: the next tile's loads are in flight while this one is applied (a third set, two
        // tiles ahead, measured no faster and costs registers and code)
        WtWalkBatch A, B;
        wt_walk_fetch(P, d, w, nt, M, tb, lane, A);
        for (;;) {
            wt_walk_fetch(P, d, w, nt, M, tb + step, lane, B);      // (past the end: harmless re-reads of the last tile)
            apply(A, tb);
            tb += step;
            if (tb >= M) break;
            wt_walk_fetch(P, d, w, nt, M, tb + step, lane, A);
            apply(B, tb);
            tb += step;
            if (tb >= M) break;
        }
    }
    if (!SCATTER) {
        my_next = wt_wave_min_i32(my_next);
        if (my_next != 0x7fffffff && wt_wave_leader(lane)) wt_lds_min32(&c.sh->next_bp, my_next);
        if (tid == 0 && M) wt_lds_add64(&c.sh->n_intervals, (unsigned long long) M);
    }
}

// ---- workgroup-wide exclusive prefix of one value per lane: a() / barrier / b() -> base[0 .. nt] ----
#ifdef WT_EMU
WT_DEV void wt_walk_scan_a(WtWalkCtx &w, uint32_t v, int tid, int nt) { w.tot[tid] = v; }
WT_DEV void wt_walk_scan_b(WtWalkCtx &w, int tid, int nt) {
    uint32_t pfx = 0;
    for (int x = 0; x < tid; x++) pfx += w.tot[x];
    w.base[tid] = pfx;
    if (tid == nt - 1) w.base[nt] = pfx + w.tot[tid];
}
#else
WT_DEV void wt_walk_scan_a(WtWalkCtx &w, uint32_t v, int tid, int nt) {
    const int lane = tid & 63;
    const uint32_t incl = wt_wave_scan_u32(v, lane);
    w.tot[tid] = incl - v;
    if (lane == 63) w.gt[tid >> 6] = incl;
}
WT_DEV void wt_walk_scan_b(WtWalkCtx &w, int tid, int nt) {
    const int wave = tid >> 6, nwaves = nt >> 6;
    uint32_t pfx = w.tot[tid], all = 0;
    for (int x = 0; x < nwaves; x++) { const uint32_t g = w.gt[x]; if (x < wave) pfx += g; all += g; }
    w.base[tid] = pfx;
    if (tid == nt - 1) w.base[nt] = all;
}
#endif

// offsets, step 1: the lane's S positions
// (pair mode: the even lane of a pair speaks for the stretch; base[q << pair] is stretch q's prefix)
WT_DEV void wt_walk_offsets1(const WtParams &P, WtWalkCtx &w, int tid, int nt) {
    const int S = w.S, a = (tid >> w.pair) * S;
    uint32_t sum = 0;
    if (!(w.pair && (tid & 1)))
        for (int s = 0; s < S; s++) {
            const uint32_t v = w.cnt[a + s], n = wt_walk_cnt_n(w, v);
            // (pair mode: from here on the word holds the position's events as ONE count, like without pairs)
            if (w.pair) w.cnt[a + s] = n | ((v >> 14) & 0xffu) << 12 | ((v >> 22) & 0xffu) << 22;
            sum += n;
        }
    wt_walk_scan_a(w, sum, tid, nt);
}
// step 2 (after wt_walk_scan_b and a barrier): off[]
WT_DEV void wt_walk_offsets2(const WtParams &P, WtWalkCtx &w, int tid, int nt) {
    if (w.pair && (tid & 1)) return;
    const int S = w.S, q = tid >> w.pair, a = q * S;
    uint32_t o = w.base[tid];
    for (int s = 0; s < S; s++) { w.off[a + s] = o; o += w.cnt[a + s] & WT_WALK_NMASK; }
    if (q == w.nstr - 1) w.off[P.W] = o;
}

// The lanes [l0, l1) whose events one slab holds: l1 = the first lane at which they would not fit any more (every lane
// computes the same; a lane's own events always fit: cap >= 2 N S, wt_make_walk_plan).
// (l0, l1: stretches)
WT_DEV int wt_walk_round_end(const WtWalkCtx &w, int l0, int nt) {
    const uint32_t b0 = w.base[l0 << w.pair];
    if (w.base[nt] - b0 <= w.cap) return w.nstr;
    int l1 = l0 + 1;
    while (l1 < w.nstr && w.base[(l1 + 1) << w.pair] - b0 <= w.cap) l1++;
    return l1;
}

// ---- selection ----
#ifdef WT_EMU
WT_DEV bool wt_walk_any(bool x) { return x; }
#else
WT_DEV bool wt_walk_any(bool x) { return __ballot(x) != 0ull; }     // over the lanes that are active here
#endif
#define WT_WALK_RB WT_WALK_PAD       // keys of the column read ahead per block

// f(key) for every row of the lane's column (w.npad of them, a multiple of WT_WALK_RB: the rows past the tracks hold
// 0xffffffff); the LDS reads of the next block are issued before this one is consumed
template <class F>
WT_DEV void wt_walk_for_keys(const WtWalkCtx &w, int nt, int tid, F &&f) {
    const uint32_t *p = w.col + tid;
    const int R = w.npad;
    uint32_t cur[WT_WALK_RB], nxt[WT_WALK_RB];
#pragma unroll
    for (int u = 0; u < WT_WALK_RB; u++) cur[u] = p[u * nt];
    for (int i = 0; i < R; i += WT_WALK_RB) {
        if (i + WT_WALK_RB < R) {
#pragma unroll
            for (int u = 0; u < WT_WALK_RB; u++) nxt[u] = p[(i + WT_WALK_RB + u) * nt];
        }
#pragma unroll
        for (int u = 0; u < WT_WALK_RB; u++) f(cur[u]);
#pragma unroll
        for (int u = 0; u < WT_WALK_RB; u++) cur[u] = nxt[u];
    }
}

// lt = #{keys < m}, le = #{keys <= m}
WT_DEV void wt_walk_recount(const WtWalkCtx &w, int N, int nt, int tid, uint32_t m, int &lt, int &le) {
    int a = 0, b = 0;
    wt_walk_for_keys(w, nt, tid, [&](uint32_t x) { a += x < m ? 1 : 0; b += x <= m ? 1 : 0; });     // (the pad rows are above every m)
    lt = a; le = b;
}

// compare-exchange
#define WT_WALK_CE(x, y) do { const uint32_t lo_ = (x) < (y) ? (x) : (y); (y) = (x) < (y) ? (y) : (x); (x) = lo_; } while (0)
// b[0 .. WD) ascending (WD = 2, 4, 8: optimal networks of 1, 5, 19 exchanges)
template <int WD>
WT_DEV void wt_walk_sort_block(uint32_t (&b)[WD]) {
    if constexpr (WD == 2) {
        WT_WALK_CE(b[0], b[1]);
    } else if constexpr (WD == 4) {
        WT_WALK_CE(b[0], b[1]); WT_WALK_CE(b[2], b[3]); WT_WALK_CE(b[0], b[2]); WT_WALK_CE(b[1], b[3]); WT_WALK_CE(b[1], b[2]);
    } else {
        WT_WALK_CE(b[0], b[1]); WT_WALK_CE(b[2], b[3]); WT_WALK_CE(b[4], b[5]); WT_WALK_CE(b[6], b[7]);
        WT_WALK_CE(b[0], b[2]); WT_WALK_CE(b[1], b[3]); WT_WALK_CE(b[4], b[6]); WT_WALK_CE(b[5], b[7]);
        WT_WALK_CE(b[1], b[2]); WT_WALK_CE(b[5], b[6]); WT_WALK_CE(b[0], b[4]); WT_WALK_CE(b[3], b[7]);
        WT_WALK_CE(b[1], b[5]); WT_WALK_CE(b[2], b[6]);
        WT_WALK_CE(b[1], b[4]); WT_WALK_CE(b[3], b[6]);
        WT_WALK_CE(b[2], b[4]); WT_WALK_CE(b[3], b[5]);
        WT_WALK_CE(b[3], b[4]);
    }
}

// total over the lanes of a stretch (pair mode: this lane's and its partner's)
template <bool PAIR>
WT_DEV int wt_walk_tot(int x, int tid) { return PAIR ? x + (int) wt_pair_xchg((uint32_t) x, tid) : x; }

// One sweep: the WD smallest of the (flipped) keys above mf, ascending, ties kept (0xffffffff: none -- no key is 0 or ~0:
// wt_walk_key), then the move.  True: m and the counts are final.
// The column is taken WD keys at a time: the block is sorted by a network and merged with the running WD smallest
// (element-wise minimum against the reversed block leaves the WD smallest of both as a bitonic sequence, log2 WD
// exchange stages sort it) -- 12 instructions per key at WD = 8 where inserting key by key takes 20.
// lt / le: THIS LANE's counts (#{own keys < m}, #{own keys <= m}); LT / LE: the stretch's (pair mode: both lanes').
// Pair mode: each lane sweeps its half of the column, the two lists are exchanged and merged (both lanes hold the same
// WD candidates then), and every decision below is taken on the merged list and the totals -- the two lanes never diverge
// around an exchange.
template <int WD, bool PAIR>
WT_DEV bool wt_walk_move(const WtWalkCtx &w, int nt, int tid, int k, uint32_t &m, int &lt, int &le, int LT, int LE) {
    const bool up = k >= LE;
    const uint32_t flip = up ? 0u : 0xffffffffu;    // moving down is moving up among the complemented keys
    const uint32_t mf = m ^ flip;
    const int j = up ? k - LE : LT - 1 - k;         // wanted: the j-th smallest of the (flipped) keys above mf
    uint32_t a[WD];
#pragma unroll
    for (int q = 0; q < WD; q++) a[q] = 0xffffffffu;
    const uint32_t *p = w.col + tid;
    const int R = w.npad;                           // (a multiple of 8, hence of WD; the pad rows never pass the test below)
    uint32_t cur[WD], nxt[WD];
#pragma unroll
    for (int u = 0; u < WD; u++) cur[u] = p[u * nt];
    for (int i = 0; i < R; i += WD) {
        if (i + WD < R) {
#pragma unroll
            for (int u = 0; u < WD; u++) nxt[u] = p[(i + WD + u) * nt];
        }
        uint32_t b[WD];
#pragma unroll
        for (int u = 0; u < WD; u++) {
            const uint32_t x = cur[u] ^ flip;
            b[u] = x > mf ? x : 0xffffffffu;
        }
        wt_walk_sort_block<WD>(b);
#pragma unroll
        for (int u = 0; u < WD; u++) a[u] = a[u] < b[WD - 1 - u] ? a[u] : b[WD - 1 - u];
#pragma unroll
        for (int d = WD / 2; d >= 1; d >>= 1) {
#pragma unroll
            for (int u = 0; u < WD; u++)
                if ((u & d) == 0) WT_WALK_CE(a[u], a[u + d]);
        }
#pragma unroll
        for (int u = 0; u < WD; u++) cur[u] = nxt[u];
    }
    uint32_t g[WD];                                 // the stretch's WD candidates
    if (PAIR) {
        uint32_t o[WD];
#pragma unroll
        for (int u = 0; u < WD; u++) o[u] = wt_pair_xchg(a[u], tid);
#pragma unroll
        for (int u = 0; u < WD; u++) g[u] = a[u] < o[WD - 1 - u] ? a[u] : o[WD - 1 - u];
#pragma unroll
        for (int d = WD / 2; d >= 1; d >>= 1) {
#pragma unroll
            for (int u = 0; u < WD; u++)
                if ((u & d) == 0) WT_WALK_CE(g[u], g[u + d]);
        }
    } else {
#pragma unroll
        for (int u = 0; u < WD; u++) g[u] = a[u];
    }
    if (j >= WD) {                      // further away than the sweep reaches: go on from its far end
        m = g[WD - 1] ^ flip;
        wt_walk_recount(w, 0, nt, tid, m, lt, le);
        return false;
    }
    uint32_t gj = g[0];
#pragma unroll
    for (int q = 1; q < WD; q++) gj = q == j ? g[q] : gj;
    int last = 0;
#pragma unroll
    for (int q = 0; q < WD; q++) last = g[q] == gj ? q : last;
    m = gj ^ flip;
    if (last == WD - 1) {               // more keys equal to it may lie beyond the ones collected
        wt_walk_recount(w, 0, nt, tid, m, lt, le);
        return false;
    }
    // this lane's counts for the new m, from its own list: c1 keys between the old and the new m, c2 equal to the new one --
    // unless the list is full and ends at or before the new m (more of the lane's keys may lie there: recount)
    if (a[WD - 1] != 0xffffffffu && a[WD - 1] <= gj) {
        wt_walk_recount(w, 0, nt, tid, m, lt, le);
        return true;
    }
    int c1 = 0, c2 = 0;
#pragma unroll
    for (int q = 0; q < WD; q++) { c1 += a[q] < gj ? 1 : 0; c2 += a[q] == gj ? 1 : 0; }
    if (up) { const int l0 = le; lt = l0 + c1; le = l0 + c1 + c2; }
    else { const int l0 = lt; le = l0 - c1; lt = l0 - c1 - c2; }
    return true;
}

// Moves m to the key of rank k (0-based, ties as a multiset) of the stretch's column, given this lane's counts lt / le for the
// current m.  The sweep's width follows the furthest move any lane of the wave has to make (2, 4 or 8 keys).
template <bool PAIR>
WT_DEV void wt_walk_select(const WtWalkCtx &w, int nt, int tid, int k, uint32_t &m, int &lt, int &le) {
    for (;;) {
        const int LT = wt_walk_tot<PAIR>(lt, tid), LE = wt_walk_tot<PAIR>(le, tid);
        if (LT <= k && k < LE) return;
        const int j = k >= LE ? k - LE : LT - 1 - k;
        bool done;
        if (wt_walk_any(j > 3)) done = wt_walk_move<8, PAIR>(w, nt, tid, k, m, lt, le, LT, LE);
        else if (wt_walk_any(j > 1)) done = wt_walk_move<4, PAIR>(w, nt, tid, k, m, lt, le, LT, LE);
        else done = wt_walk_move<2, PAIR>(w, nt, tid, k, m, lt, le, LT, LE);
        if (done) return;
    }
}

// per-lane state across the rounds of a window
struct WtWalkLane {
    uint32_t evmask, emitmask;      // bit s: position s of the stretch has events / starts an emitted run
};

// The lane's stretch.  ev0: index (in the window's event sequence) of the first event the slab holds.
// Which of the lane's positions have events, and which start an emitted run -- from the counter words alone (events and
// coverage changes per position), before any event is applied: the window's run count goes to the look-back chain ahead of
// the walk, so that no successor ever waits for it.
WT_DEV void wt_walk_emits(const WtParams &P, const WtCtx &c, WtWalkCtx &w, WtWalkLane &L, int tid, int nt) {
    const int N = P.n_tracks, S = w.S, q = tid >> w.pair, a = q * S;         // (pair mode: both lanes of the stretch compute the same)
    const bool strict = (P.flags & WT_STRICT_SET0) != 0;
    const int32_t room = c.sh->emit_hi - (c.sh->w0 + a);       // positions of the stretch below the range end
    uint32_t evmask = 0, emitmask = 0;
    int ncov = w.ncov[q], fe = -1;
    for (int s = 0; s < S; s++) {
        const uint32_t v = w.cnt[a + s];
        if (!wt_walk_cnt_n(w, v)) continue;
        if (fe < 0) fe = a + s;
        evmask |= 1u << s;
        ncov += wt_walk_cnt_dcov(w, v);
        if ((strict ? ncov == N : ncov > 0) && s < room) emitmask |= 1u << s;      // multiplexer.c:120,125
    }
    w.fe[q] = fe;
    L.evmask = evmask; L.emitmask = emitmask;
}
// what the lane adds to the scan of the run counts (pair mode: the even lane speaks for the stretch)
WT_DEV uint32_t wt_walk_emit_count(const WtWalkCtx &w, const WtWalkLane &L, int tid) {
    return (w.pair && (tid & 1)) ? 0u : (uint32_t) wt_popc32(L.emitmask);
}

#if defined(WT_PROFILE) && !defined(WT_EMU)
#define WT_WALK_T0 unsigned long long wt_wt = __builtin_readcyclecounter()
#define WT_WALK_TICK(slot) do { const unsigned long long t_ = __builtin_readcyclecounter(); if (prof) prof[slot] += t_ - wt_wt; wt_wt = t_; } while (0)
#else
#define WT_WALK_T0 do { } while (0)
#define WT_WALK_TICK(slot) do { } while (0)
#endif
#define WT_WALK_EB 8        // events fetched per batch
// The lane's stretch.  (The slab was written by the other waves of this workgroup before the last barrier: same CU, same
// vector cache -- plain loads.)
//   FIXED   the events of position p are cnt[p] of the slots slab[p * capp ..] (+ the overflow list beyond capp)
//   !FIXED  (fallback) the sorted sequence: the events of p are slab[off[p] - ev0 .. off[p + 1] - ev0)
template <bool FIXED, bool PAIR>
WT_DEV void wt_walk_lane(const WtParams &P, const WtCtx &c, WtWalkCtx &w, WtWalkLane &L, uint32_t ev0, int tid, int nt,
                         unsigned long long *prof = nullptr) {
    const int N = P.n_tracks, S = w.S, a = (PAIR ? tid >> 1 : tid) * S, k = N / 2;
    const uint32_t half = PAIR ? (uint32_t) (tid & 1) : 0u;       // this lane's tracks: i = half (mod 2), row i >> 1
    const uint32_t emitmask = L.emitmask;       // (wt_walk_emits)
    uint32_t m = 0;
    int lt = 0, le = 0, nn = 0;
    bool have = false;              // the median of the column is known (lazily: only where a run is emitted)
    bool started = false;           // nn is initialised (at the stretch's first event)
    uint32_t *col = w.col + tid;
    const uint32_t novf = FIXED ? (w.novf[0] < w.ov_cap ? w.novf[0] : w.ov_cap) : 0u;

    auto apply = [&](const WtWalkEvent (&ev)[WT_WALK_EB], uint32_t nvalid) {
        // their tracks' old keys first (no track has two events at one position), then the updates
        uint32_t ok[WT_WALK_EB];
        uint32_t trk[WT_WALK_EB];
        bool mine[WT_WALK_EB];
#pragma unroll
        for (int u = 0; u < WT_WALK_EB; u++) {
            uint32_t t = ev[u].meta & 0xffffu;
            t = t < (uint32_t) N ? t : 0u;              // (a slot beyond the position's count holds anything)
            mine[u] = (uint32_t) u < nvalid && (!PAIR || FIXED || (t & 1u) == half);      // (FIXED: the lane's own slots hold its own tracks' events)
            trk[u] = PAIR ? t >> 1 : t;                 // its row in this lane's column (the partner's events: read, not applied)
            ok[u] = col[trk[u] * (uint32_t) nt];
        }
#pragma unroll
        for (int u = 0; u < WT_WALK_EB; u++) {
            if (!mine[u]) continue;
            const uint32_t nk = ev[u].key;
            col[trk[u] * (uint32_t) nt] = nk;
            if (have) {
                lt += (nk < m ? 1 : 0) - (ok[u] < m ? 1 : 0);
                le += (nk <= m ? 1 : 0) - (ok[u] <= m ? 1 : 0);
            }
            nn += (nk == WT_WALK_NANKEY ? 1 : 0) - (ok[u] == WT_WALK_NANKEY ? 1 : 0);
        }
    };
    // WT_WALK_EB events from slab index `from` on.  Sorted: the first `n` exist (the others: the last one again).  FIXED: the
    // position's slots as they are (the slab ends in the overflow list, at least WT_WALK_EB events long: no read past it).
    auto fetch = [&](uint32_t from, uint32_t n, WtWalkEvent (&ev)[WT_WALK_EB]) {
#pragma unroll
        for (int u = 0; u < WT_WALK_EB; u++) ev[u] = w.slab[from + (FIXED || (uint32_t) u < n ? (uint32_t) u : n - 1u)];
    };

    // The next position's first events are fetched right after this position's have been applied, i.e. before its
    // sweep -- which hides the loads.  FIXED: the slots of position a + s + 1 (used if that position has events);
    // sorted: the next events of the sequence, whatever position they belong to.
    WtWalkEvent pre[WT_WALK_EB];
    uint32_t pre_at = 0xffffffffu;              // slab index pre[] was fetched from
    uint32_t o = FIXED ? 0u : w.off[a];
    const uint32_t o_end = FIXED ? 0u : w.off[a + S];
    // FIXED: this lane's slots of a position (pair mode: its half of them -- the events of its own tracks)
    const uint32_t lcap = FIXED ? (PAIR ? (uint32_t) w.capp >> 1 : (uint32_t) w.capp) : 0u;
    auto slots_of = [&](int pos) { return PAIR ? ((((uint32_t) pos) << 1) | half) * lcap : (uint32_t) pos * lcap; };
    if (FIXED) { pre_at = slots_of(a); fetch(pre_at, 1u, pre); }
    else if (o_end != o) { fetch(o - ev0, o_end - o, pre); pre_at = o - ev0; }
    for (int s = 0; s < S; s++) {
        uint32_t n, from, ntot;
        if (FIXED) {
            const uint32_t v = w.cnt[a + s];
            ntot = wt_walk_cnt_n(w, v);
            n = PAIR ? (v >> (7u * half)) & WT_WALK_PNMASK : ntot;          // (the lane's own events)
            from = slots_of(a + s);
        } else { const uint32_t o1 = w.off[a + s + 1]; n = o1 - o; ntot = n; from = o - ev0; o = o1; }
        if (!ntot) continue;
        WT_WALK_T0;
        if (!started) {
            started = true;
            wt_walk_for_keys(w, nt, tid, [&](uint32_t x) { nn += x == WT_WALK_NANKEY ? 1 : 0; });
        }
        const uint32_t nslot = FIXED ? (n < lcap ? n : lcap) : n;
        for (uint32_t e = 0; e < nslot; e += WT_WALK_EB) {
            WtWalkEvent ev[WT_WALK_EB];
            const uint32_t left = nslot - e;
            if (e == 0 && pre_at == from) {
#pragma unroll
                for (int u = 0; u < WT_WALK_EB; u++) ev[u] = pre[u];
            } else {
                fetch(from + e, left, ev);
            }
            apply(ev, left);
        }
        if (FIXED && n > nslot) {               // the position's events beyond its slots: somewhere in the overflow list
            for (uint32_t j = 0; j < novf; j++) {
                const WtWalkOvf q = w.ovf[j];
                if (q.pos != (uint32_t) (a + s) || (PAIR && (q.meta & 1u) != half)) continue;
                WtWalkEvent ev[WT_WALK_EB];
#pragma unroll
                for (int u = 0; u < WT_WALK_EB; u++) { ev[u].key = q.key; ev[u].meta = q.meta; }
                apply(ev, 1u);
            }
        }
        if (FIXED) {
            if (s + 1 < S) { pre_at = slots_of(a + s + 1); fetch(pre_at, 1u, pre); }
        } else if (o < o_end) {
            pre_at = o - ev0;
            fetch(pre_at, o_end - o, pre);
        }
        WT_WALK_TICK(4);
        if (!((emitmask >> s) & 1u)) continue;
        if (!have) {
            const uint32_t g0 = w.guess[0], g1 = PAIR ? wt_pair_xchg(g0, tid) : g0;
            m = half ? g1 : g0;                         // (both lanes of a pair start from the even lane's reading)
            wt_walk_recount(w, N, nt, tid, m, lt, le);
            have = true;
            wt_walk_select<PAIR>(w, nt, tid, k, m, lt, le);
            WT_WALK_TICK(5);
        } else {
            wt_walk_select<PAIR>(w, nt, tid, k, m, lt, le);
            WT_WALK_TICK(6);
        }
        // (the position's count is not needed any more: the result lives there; pair mode: both lanes store the same)
        w.cnt[a + s] = wt_walk_tot<PAIR>(nn, tid) ? WT_WALK_NANKEY : m;
    }
    if (have) w.guess[0] = m;       // (any lane's: the next stretch's first selection starts there)
}

// first breakpoint after stretch q
WT_DEV int32_t wt_walk_next_after(const WtCtx &c, const WtWalkCtx &w, int q) {
    for (int l = q + 1; l < w.nstr; l++)
        if (w.fe[l] >= 0) return c.sh->w0 + w.fe[l];
    return c.sh->next_bp;
}

// the stretch's runs, at their global positions (base[]: exclusive prefix of the run counts; pair mode: the two lanes take
// every other run)
WT_DEV void wt_walk_write(const WtParams &P, WtCtx &c, const WtWalkCtx &w, const WtWalkLane &L, int tid, int nt) {
    if (!L.emitmask) return;
    const int S = w.S, q = tid >> w.pair, a = q * S, half = w.pair ? tid & 1 : 0;
    const int32_t w0 = c.sh->w0;
    long long idx = c.sh->goffset + (long long) w.base[q << w.pair];
    int32_t after = 0;
    bool have_after = false;
    unsigned long long bp = 0;
    int rank = 0;
    for (int s = 0; s < S; s++) {
        if (!((L.emitmask >> s) & 1u)) continue;
        const long long o = idx++;
        if (w.pair && (rank++ & 1) != half) continue;
        const uint32_t later = s + 1 < 32 ? (L.evmask >> (s + 1)) : 0u;
        int32_t fin;
        if (later) {
            fin = w0 + a + s + 1 + (int32_t) wt_ctz64((uint64_t) later);
        } else {
            if (!have_after) { after = wt_walk_next_after(c, w, q); have_after = true; }
            fin = after;
        }
        bp += (unsigned long long) (fin - (w0 + a + s));
        if (o >= P.capacity) continue;
        const uint32_t key = w.cnt[a + s];
        P.o_start[o] = w0 + a + s;
        P.o_finish[o] = fin;
        P.o_value[o] = key == WT_WALK_NANKEY ? wt_nan() : (double) wt_unkey32(key);
    }
    if (bp) wt_lds_add64(&c.sh->bp_sum, bp);
}

#endif  // WT_WALK_H_
