// wt_walk.h -- MedianReduction by walking (device + -DWT_EMU), included by wt_core.h after wt_delta.h.
//
// The bitmap kernel (wt_reduce_kernel) evaluates every run of the output from scratch: N gathers and a sorting
// network per position, although between neighbouring runs only the tracks with a breakpoint there change -- ~6 of
// 100 at the benchmark's density (reducers.c:780-813 itself re-sorts all N values per pop).  Here a lane owns S
// CONSECUTIVE positions of the window and carries the column of the N current values (order-preserving keys, in
// LDS, [track][lane]) from one to the next:
//
//   ranges   the window's runs of every track as one flat index space (wt_delta_ranges*)
//   count    per run: an EVENT at its start (the track takes the run's value) and one at its finish (the track falls
//            back to its default) unless the next run of the track starts right there; events are counted per
//            position (LDS atomics).  A run that covers the first position of a lane's stretch from before it
//            writes its value into that lane's column: the columns start out right without any look-up.
//   offsets  prefix sums of the counts: the events of a position are a contiguous slice of the workgroup's slab
//   scatter  second pass over the runs: every event to its slot (track, new key, coverage change)
//   walk     per lane: the median of its first position by the sorting network (as the bitmap kernel does for every
//            position), then position by position: apply the events (old key out, new key in, the counts of keys
//            below / not above the current median m kept up to date), and when the wanted rank k = N/2 has left
//            [lt, le) move m: ONE sweep over the column collects the four nearest keys on the side m has to move to
//            (ties are handled as a multiset, so the result is the exact order statistic, bit for bit what the
//            sort gives)
//   emit     run count scan, look-back, the lanes write their runs
//
// Eligibility (host): float tracks, float-exact defaults, N <= 128 (the first median uses the register network).
// The output is the bitmap kernel's, bit for bit (same keys, same order statistic, NaN if any value is NaN).
#ifndef WT_WALK_H_
#define WT_WALK_H_

#define WT_WALK_NANKEY 0xfffffffeu     // every NaN (no other value has this key: wt_walk_key)
#define WT_WALK_INC 0x10000u           // event: the track becomes covered
#define WT_WALK_DEC 0x20000u           // event: the track stops being covered
#define WT_WALK_MAX_S 16

struct alignas(8) WtWalkEvent { uint32_t key, meta; };     // meta: track | WT_WALK_INC | WT_WALK_DEC

struct WtWalkCtx {
    uint32_t *col;      // [N][T] current keys, one column per lane
    uint32_t *cnt;      // [W] events per position (count pass); slot countdown (scatter); the lanes' results (walk)
    uint32_t *off;      // [W + 1] first event of every position (index into the window's event sequence)
    uint32_t *tot;      // [T] scan scratch
    uint32_t *base;     // [T + 1] scan result
    uint32_t *gt;       // [T / 64 + 1] wave totals
    int32_t *ncov;      // [T] tracks covering the position before the lane's first one
    int32_t *fe;        // [T] first position (window-relative) of the lane's stretch that has an event, or -1
    uint32_t *dkey;     // [N] keys of the defaults
    WtWalkEvent *slab;  // this workgroup's events (global)
    uint32_t cap;       // events the slab holds
    int S;              // positions per lane
};

WT_DEV void wt_walk_ctx_init(WtWalkCtx &w, const WtParams &P, char *lds, char *slab) {
    w.col = (uint32_t *) (lds + P.off_wcol);
    w.cnt = (uint32_t *) (lds + P.off_wcnt);
    w.off = (uint32_t *) (lds + P.off_woff);
    w.tot = (uint32_t *) (lds + P.off_wtot);
    w.base = (uint32_t *) (lds + P.off_wbase);
    w.gt = (uint32_t *) (lds + P.off_wgt);
    w.ncov = (int32_t *) (lds + P.off_wncov);
    w.fe = (int32_t *) (lds + P.off_wfe);
    w.dkey = (uint32_t *) (lds + P.off_wdk);
    w.slab = (WtWalkEvent *) slab;
    w.cap = (uint32_t) (P.g_scratch_slab / (long long) sizeof(WtWalkEvent));
    w.S = P.walk_S;
}

// order-preserving key of the float with bits `vb`; every NaN gets the same one, above +Inf
WT_DEV uint32_t wt_walk_key(uint32_t vb) {
    return (vb & 0x7fffffffu) > 0x7f800000u ? WT_WALK_NANKEY : wt_key32(__builtin_bit_cast(float, vb));
}

// once per workgroup
WT_DEV void wt_walk_defaults(const WtParams &P, WtWalkCtx &w, int tid, int nt) {
    for (int i = tid; i < P.n_tracks; i += nt) w.dkey[i] = wt_walk_key(__builtin_bit_cast(uint32_t, (float) P.defaults[i]));
}

// per window: no events, nothing covered, every column holds the defaults
WT_DEV void wt_walk_zero(const WtParams &P, WtWalkCtx &w, int tid, int nt) {
    for (int x = tid; x < P.W; x += nt) w.cnt[x] = 0;
    w.ncov[tid] = 0;
    const int N = P.n_tracks;
    for (int i = 0; i < N; i++) w.col[i * nt + tid] = w.dkey[i];
}

// ---- the window's runs, twice (count, scatter): the flat index space of wt_delta.h ----
struct WtWalkBatch {
    int32_t s[WT_DELTA_U], f[WT_DELTA_U];
    int32_t ps[WT_DELTA_U], ns[WT_DELTA_U];     // finish of the track's previous run, start of its next one (of the FILE, not the window)
    uint32_t b[WT_DELTA_U];
    int trk[WT_DELTA_U];
    bool first[WT_DELTA_U], last[WT_DELTA_U];   // the run is the first / last of its (chromosome, track) segment
};

// (unconditional, always in range: see wt_delta_fetch)
WT_DEV void wt_walk_fetch(const WtParams &P, const WtDeltaCtx &d, int nt, uint32_t M, uint32_t tb, int lane, int chrom, WtWalkBatch &B) {
    const uint32_t lastt = (M - 1u) / WT_DELTA_TILE * WT_DELTA_TILE;
    const uint32_t tbe = tb < lastt ? tb : lastt;
    const uint32_t tile = tbe / WT_DELTA_TILE;
    int i = tile < WT_DELTA_TF ? (int) d.tfirst[tile] : wt_delta_find(d.tpfx, nt, tbe, 0);
    uint32_t hi = d.tpfx[i + 1];
    long long dl = d.tbase[i];
    const int N = P.n_tracks;
    long long s0 = P.seg_off[(long long) chrom * N + i] * 4, s1 = P.seg_off[(long long) chrom * N + i + 1] * 4;
#pragma unroll
    for (int u = 0; u < WT_DELTA_U; u++) {
        uint32_t jj = tbe + (uint32_t) lane + 64u * (uint32_t) u;
        jj = jj < M - 1u ? jj : M - 1u;
        if (jj >= hi) {
            do { i++; hi = d.tpfx[i + 1]; } while (jj >= hi);
            dl = d.tbase[i];
            s0 = P.seg_off[(long long) chrom * N + i] * 4; s1 = P.seg_off[(long long) chrom * N + i + 1] * 4;
        }
        const long long ob = dl + ((long long) jj << 2);
        const bool fst = ob <= s0, lst = ob + 4 >= s1;
        B.s[u] = *(const int32_t *) ((const char *) P.start + ob);
        B.f[u] = *(const int32_t *) ((const char *) P.finish + ob);
        B.b[u] = *(const uint32_t *) ((const char *) P.value + ob);
        B.ps[u] = *(const int32_t *) ((const char *) P.finish + (fst ? ob : ob - 4));
        B.ns[u] = *(const int32_t *) ((const char *) P.start + (lst ? ob : ob + 4));
        B.trk[u] = i;
        B.first[u] = fst; B.last[u] = lst;
    }
}

// one run, count pass
WT_DEV void wt_walk_count1(const WtParams &P, WtWalkCtx &w, int nt, int32_t w0, int32_t width, int trk, int32_t s, int32_t f, int32_t ns,
                           bool last, uint32_t vb, int32_t &my_next) {
    const int32_t cs = s - w0, cf = f - w0;     // cf >= 0: the window's runs finish at or beyond w0
    if (cs >= width) { my_next = s < my_next ? s : my_next; return; }
    if (cs >= 0) wt_lds_add32(&w.cnt[cs], 1u);
    if (cf < width) {
        if (last || ns != f) wt_lds_add32(&w.cnt[cf], 1u);       // (else the next run's start event says it all)
    } else {
        my_next = f < my_next ? f : my_next;
    }
    // the lanes whose first position a lies in (s, f]: the run covers the position before a
    const int S = w.S;
    int l = cs < 0 ? 0 : cs / S + 1;
    int lh = cf / S;
    if (lh > nt - 1) lh = nt - 1;
    if (l <= lh) {
        const uint32_t key = wt_walk_key(vb);
        for (; l <= lh; l++) {
            w.col[trk * nt + l] = key;
            wt_lds_addi32(&w.ncov[l], 1);
        }
    }
}

// one run, scatter pass (cnt[] counts down: the slots of a position are handed out from the last to the first)
WT_DEV void wt_walk_scatter1(const WtParams &P, WtWalkCtx &w, int32_t w0, int32_t width, uint32_t ev0, uint32_t ev1, int trk, int32_t s,
                             int32_t f, int32_t ps, int32_t ns, bool first, bool last, uint32_t vb) {
    const int32_t cs = s - w0, cf = f - w0;
    if (cs >= width) return;
    if (cs >= 0) {
        const uint32_t at = w.off[cs];
        if (at >= ev0 && at < ev1) {                            // (positions of the lanes of this round)
#ifdef WT_EMU
            const uint32_t old = w.cnt[cs]--;
#else
            const uint32_t old = atomicSub((unsigned int *) &w.cnt[cs], 1u);
#endif
            WtWalkEvent e;
            e.key = wt_walk_key(vb);
            e.meta = (uint32_t) trk | ((first || ps != s) ? WT_WALK_INC : 0u);
            w.slab[at - ev0 + old - 1u] = e;
        }
    }
    if (cf < width && (last || ns != f)) {
        const uint32_t at = w.off[cf];
        if (at >= ev0 && at < ev1) {
#ifdef WT_EMU
            const uint32_t old = w.cnt[cf]--;
#else
            const uint32_t old = atomicSub((unsigned int *) &w.cnt[cf], 1u);
#endif
            WtWalkEvent e;
            e.key = w.dkey[trk];
            e.meta = (uint32_t) trk | WT_WALK_DEC;
            w.slab[at - ev0 + old - 1u] = e;
        }
    }
}

// Both passes: SCATTER == false counts; SCATTER == true places the events whose position's first slot lies in [ev0, ev1).
template <bool SCATTER>
WT_DEV void wt_walk_pass(const WtParams &P, WtCtx &c, WtWalkCtx &w, WtDeltaCtx &d, uint32_t ev0, uint32_t ev1, int tid, int nt) {
    const int wave = wt_uniform32(tid >> 6), lane = tid & 63, nwaves = nt >> 6;
    const uint32_t M = (uint32_t) wt_uniform32((int32_t) d.tpfx[nt]);
    const int32_t w0 = wt_uniform32(c.sh->w0);
    const int32_t width = wt_uniform32(c.sh->w1) - w0;
    const int chrom = wt_uniform32(c.sh->chrom);
    const uint32_t step = (uint32_t) nwaves * WT_DELTA_TILE;
    int32_t my_next = 0x7fffffff;
    auto apply = [&](const WtWalkBatch &B, uint32_t tb) {
#pragma unroll
        for (int u = 0; u < WT_DELTA_U; u++)
            if (tb + (uint32_t) lane + 64u * (uint32_t) u < M) {
                if (SCATTER) wt_walk_scatter1(P, w, w0, width, ev0, ev1, B.trk[u], B.s[u], B.f[u], B.ps[u], B.ns[u], B.first[u], B.last[u], B.b[u]);
                else wt_walk_count1(P, w, nt, w0, width, B.trk[u], B.s[u], B.f[u], B.ns[u], B.last[u], B.b[u], my_next);
            }
    };
    uint32_t tb = (uint32_t) wave * WT_DELTA_TILE;
    if (tb < M) {
        WtWalkBatch A, B;
        wt_walk_fetch(P, d, nt, M, tb, lane, chrom, A);
        for (;;) {
            wt_walk_fetch(P, d, nt, M, tb + step, lane, chrom, B);
            apply(A, tb);
            tb += step;
            if (tb >= M) break;
            wt_walk_fetch(P, d, nt, M, tb + step, lane, chrom, A);
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
WT_DEV void wt_walk_offsets1(const WtParams &P, WtWalkCtx &w, int tid, int nt) {
    const int S = w.S, a = tid * S;
    uint32_t sum = 0;
    int fe = -1;
    for (int s = 0; s < S; s++) {
        const uint32_t n = w.cnt[a + s];
        if (n && fe < 0) fe = a + s;
        sum += n;
    }
    w.fe[tid] = fe;
    wt_walk_scan_a(w, sum, tid, nt);
}
// step 2 (after wt_walk_scan_b and a barrier): off[]
WT_DEV void wt_walk_offsets2(const WtParams &P, WtWalkCtx &w, int tid, int nt) {
    const int S = w.S, a = tid * S;
    uint32_t o = w.base[tid];
    for (int s = 0; s < S; s++) { w.off[a + s] = o; o += w.cnt[a + s]; }
    if (tid == nt - 1) w.off[P.W] = o;
}

// The lanes [l0, l1) whose events one slab holds: l1 = the first lane at which they would not fit any more (every lane
// computes the same; a lane's own events always fit: cap >= 2 N S, wt_make_walk_plan).
WT_DEV int wt_walk_round_end(const WtWalkCtx &w, int l0, int nt) {
    const uint32_t b0 = w.base[l0];
    if (w.base[nt] - b0 <= w.cap) return nt;
    int l1 = l0 + 1;
    while (l1 < nt && w.base[l1 + 1] - b0 <= w.cap) l1++;
    return l1;
}

// ---- selection ----
// lt = #{keys < m}, le = #{keys <= m}, nn = #{NaN}
WT_DEV void wt_walk_recount(const WtWalkCtx &w, int N, int nt, int tid, uint32_t m, int &lt, int &le) {
    int a = 0, b = 0;
    for (int i = 0; i < N; i++) {
        const uint32_t x = w.col[i * nt + tid];
        a += x < m ? 1 : 0;
        b += x <= m ? 1 : 0;
    }
    lt = a; le = b;
}

// Moves m to the key of rank k (0-based, ties as a multiset) given lt / le for the current m.
WT_DEV void wt_walk_select(const WtWalkCtx &w, int N, int nt, int tid, int k, uint32_t &m, int &lt, int &le) {
    for (;;) {
        if (lt <= k && k < le) return;
        const bool up = k >= le;
        // moving down is moving up among the complemented keys
        const uint32_t flip = up ? 0u : 0xffffffffu;
        const uint32_t mf = m ^ flip;
        const int j = up ? k - le : lt - 1 - k;         // wanted: the j-th smallest of the (flipped) keys above mf
        uint32_t a0 = 0xffffffffu, a1 = 0xffffffffu, a2 = 0xffffffffu, a3 = 0xffffffffu;       // (no key is 0 or ~0: wt_walk_key)
        for (int i = 0; i < N; i++) {
            const uint32_t x = w.col[i * nt + tid] ^ flip;
            uint32_t t = x > mf ? x : 0xffffffffu;
            uint32_t lo;
            lo = a0 < t ? a0 : t; t = a0 < t ? t : a0; a0 = lo;
            lo = a1 < t ? a1 : t; t = a1 < t ? t : a1; a1 = lo;
            lo = a2 < t ? a2 : t; t = a2 < t ? t : a2; a2 = lo;
            a3 = a3 < t ? a3 : t;
        }
        if (j > 3) {                        // further away than the sweep reaches: start again from its far end
            m = a3 ^ flip;
            wt_walk_recount(w, N, nt, tid, m, lt, le);
            continue;
        }
        const uint32_t aj = j == 0 ? a0 : (j == 1 ? a1 : (j == 2 ? a2 : a3));
        const int first = a0 == aj ? 0 : (a1 == aj ? 1 : (a2 == aj ? 2 : 3));
        const int last = a3 == aj ? 3 : (a2 == aj ? 2 : (a1 == aj ? 1 : 0));
        m = aj ^ flip;
        if (last == 3) {                    // more keys equal to it may lie beyond the four
            wt_walk_recount(w, N, nt, tid, m, lt, le);
            continue;
        }
        if (up) { lt = le + first; le = le + last + 1; }
        else { const int l0 = lt; le = l0 - first; lt = l0 - last - 1; }
        return;
    }
}

// The median of the lane's column from nothing: the bitmap kernel's register network (wt_eval_chunk), fed from LDS.
template <int NR>
WT_DEV uint32_t wt_walk_first_median(const WtWalkCtx &w, int N, int nt, int tid) {
    constexpr int H = NR / 2;
    uint32_t lo[H], hi[H];
    const int pad_lo = H - N / 2;
    wt_static_for<0, NR>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const int i = s - pad_lo;
        uint32_t v = s < pad_lo ? 0u : 0xffffffffu;
        if (i >= 0 && i < N) v = w.col[i * nt + tid];
        if constexpr (s < H) lo[s] = v; else hi[s - H] = v;
    });
    wt_sort_regs<H, false>(lo);
    wt_sort_regs<H, true>(hi);
    uint32_t m = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < H; i++) {
        const uint32_t x = lo[i] > hi[i] ? lo[i] : hi[i];
        m = x < m ? x : m;
    }
    return m;
}

// An event of the slab, written by another wave of this workgroup before the last barrier: read at agent scope, so
// that a line of the slab this CU's vector cache still holds from an earlier window is not what comes back.
WT_DEV WtWalkEvent wt_walk_event(const WtWalkEvent *p) {
#ifdef WT_EMU
    return *p;
#else
    const unsigned long long x = __hip_atomic_load((const unsigned long long *) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    WtWalkEvent e;
    e.key = (uint32_t) x; e.meta = (uint32_t) (x >> 32);
    return e;
#endif
}

// per-lane state across the rounds of a window
struct WtWalkLane {
    uint32_t evmask, emitmask;      // bit s: position s of the stretch has events / starts an emitted run
};

// The lane's stretch.  ev0: index (in the window's event sequence) of the first event the slab holds.
template <int NR>
WT_DEV void wt_walk_lane(const WtParams &P, const WtCtx &c, WtWalkCtx &w, WtWalkLane &L, uint32_t ev0, int tid, int nt) {
    const int N = P.n_tracks, S = w.S, a = tid * S, k = N / 2;
    const bool strict = (P.flags & WT_STRICT_SET0) != 0;
    const int32_t room = c.sh->emit_hi - (c.sh->w0 + a);       // positions of the stretch below the range end
    uint32_t evmask = 0, emitmask = 0;
    uint32_t o = w.off[a];
    if (w.off[a + S] != o) {
        uint32_t m = 0;
        int lt = 0, le = 0, nn = 0;
        bool have = false;          // the median of the column is known (lazily: only where a run is emitted)
        for (int i = 0; i < N; i++) nn += w.col[i * nt + tid] == WT_WALK_NANKEY ? 1 : 0;
        int ncov = w.ncov[tid];
        for (int s = 0; s < S; s++) {
            const uint32_t o1 = w.off[a + s + 1];
            if (o1 == o) continue;
            evmask |= 1u << s;
            for (uint32_t e = o; e < o1; e++) {
                const WtWalkEvent ev = wt_walk_event(&w.slab[e - ev0]);
                const int trk = (int) (ev.meta & 0xffffu);
                const uint32_t nk = ev.key;
                const uint32_t ok = w.col[trk * nt + tid];
                w.col[trk * nt + tid] = nk;
                if (have) {
                    lt += (nk < m ? 1 : 0) - (ok < m ? 1 : 0);
                    le += (nk <= m ? 1 : 0) - (ok <= m ? 1 : 0);
                }
                nn += (nk == WT_WALK_NANKEY ? 1 : 0) - (ok == WT_WALK_NANKEY ? 1 : 0);
                ncov += (int) ((ev.meta >> 16) & 1u) - (int) ((ev.meta >> 17) & 1u);
            }
            o = o1;
            const bool emit = (strict ? ncov == N : ncov > 0) && s < room;     // multiplexer.c:120,125
            if (!emit) continue;
            emitmask |= 1u << s;
            if (!have) {
                m = wt_walk_first_median<NR>(w, N, nt, tid);
                wt_walk_recount(w, N, nt, tid, m, lt, le);
                have = true;
            } else {
                wt_walk_select(w, N, nt, tid, k, m, lt, le);
            }
            w.cnt[a + s] = nn ? WT_WALK_NANKEY : m;         // (cnt[] is all zeros after the scatter pass: the results live there)
        }
    }
    L.evmask = evmask; L.emitmask = emitmask;
}

// first breakpoint after the lane's stretch
WT_DEV int32_t wt_walk_next_after(const WtCtx &c, const WtWalkCtx &w, int tid, int nt) {
    for (int l = tid + 1; l < nt; l++)
        if (w.fe[l] >= 0) return c.sh->w0 + w.fe[l];
    return c.sh->next_bp;
}

// the lane's runs, at their global positions (base[]: exclusive prefix of the lanes' run counts)
WT_DEV void wt_walk_write(const WtParams &P, WtCtx &c, const WtWalkCtx &w, const WtWalkLane &L, int tid, int nt) {
    if (!L.emitmask) return;
    const int S = w.S, a = tid * S;
    const int32_t w0 = c.sh->w0;
    long long idx = c.sh->goffset + (long long) w.base[tid];
    int32_t after = 0;
    bool have_after = false;
    unsigned long long bp = 0;
    for (int s = 0; s < S; s++) {
        if (!((L.emitmask >> s) & 1u)) continue;
        const uint32_t later = s + 1 < 32 ? (L.evmask >> (s + 1)) : 0u;
        int32_t fin;
        if (later) {
            fin = w0 + a + s + 1 + (int32_t) wt_ctz64((uint64_t) later);
        } else {
            if (!have_after) { after = wt_walk_next_after(c, w, tid, nt); have_after = true; }
            fin = after;
        }
        const long long o = idx++;
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
