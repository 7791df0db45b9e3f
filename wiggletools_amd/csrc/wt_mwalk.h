// wt_mwalk.h -- MWUReduction (Mann-Whitney U, reference src/setComparisons.c:269-370) by WALKING (device + -DWT_EMU),
// included by wt_core.h after wt_walk.h, whose window machinery it shares: the flat index space of the window's runs, one
// pass turning runs into events in per-position slots, stretches of S consecutive positions, the look-back chain.
//
// The bitmap kernel (wt_reduce_kernel<MWU>, wt_mwu_regs) evaluates every output run from scratch: a sorting network over
// set 0 and n1 x n2 compare / add pairs, ~16 000 lane instructions per position (253 wave-wide VALU instructions per output
// run, 0.014 of the HBM roofline for three rounds), although between neighbouring runs only ~6 of 100 tracks change.
// Round 4 dropped the incremental form because the reference's tie state machine (:328-359: `ties` / `previousTies` leak
// from one tie group into the next) "is a function of the sorted order, not of counts".  It is a function of LITTLE of the
// sorted order, though.  With the table sorted stably (set-0 entries first inside a group of equal values):
//
//   U1 = S + C,   S = #{(x, y) : x in set 0, y in set 1, y < x}      (:336, `index - prev` summed)
//
// and the correction C only moves at set-0 elements whose value also occurs in set 1 -- TIE GROUPS (value v, c0 = #set-0
// entries, c1 = #set-1 entries, both > 0) -- and, while the leaked state (T, P) = (ties, previousTies) is non-zero, by the
// CONSTANT (T - 2P) / 2 per set-0 element in between.  So C follows from the tie groups in value order, each with
// (c0, c1, r0 = #set-0 elements below v): wt_mw_eval.  All of it is integer arithmetic in halves: 2 U1 = 2 S + 2 C exactly.
//
// What a PAIR of lanes carries along its stretch (lane h holds set h's column of current keys in LDS, wt_walk.h's pair mode
// with "track parity" = set):
//   * S, as two partial sums: a set-0 element moving a -> b changes S by #{y < b} - #{y < a}, counted by the lane that
//     holds set 1 in ONE scan of its column; a set-1 element moving by #{x > b} - #{x > a}, counted by the other lane.
//     The k-th event of set 0 and the k-th of set 1 of a position are applied in the same step: each lane scans its column
//     for the partner's event (set 0's first: lane 0 writes before it scans, lane 1 scans before it writes);
//   * the tie groups (at most WT_MW_K, in registers, identical in both lanes): the same scans count the other set's
//     entries EQUAL to a and b, which is all it takes to update c0 / c1 or to open / close a group; r0 moves with every
//     set-0 event.  More groups than slots (data of a few distinct values): the position is evaluated by enumerating the
//     distinct set-0 values in order (wt_mw_eval_slow: ~2 n1 scans, about what the bitmap kernel spends) until the groups
//     fit again;
//   * NaN keys and tracks in play per set (the run is emitted where both sets have a track in play, :283-289 / the strict
//     flags); any NaN: NaN (:300-304, :314-318).
// State starts at the stretch's first position that has events: S and the groups from n1 scans per lane (wt_mw_init).
// The pairs of a wavefront do NOT walk in step by position (the first version did: every position cost the wavefront its
// busiest pair's events -- 8.2 steps where a pair has 4.1 on average, and 284 wave-wide VALU instructions per output run,
// more than the bitmap kernel's 253): every pair has its own cursor (position, step) and takes its next step each turn of
// the loop; the groups are kept in value order so that the state machine is one pass over the slots, cheap enough to run
// with every step -- a pair that has just applied the last event of a position stores that position's result.
// The value itself: |2 U1 - 2 mu| indexes the host's table (WtParams::mwu_table) -- bit-identical to the reference's
// 2 erf(-|U1 - mu| / sigma), no device libm.
//
// Eligibility (host, wt_pick_plan): float tracks, float-exact defaults, 1 <= n1, n2 <= 64.
#ifndef WT_MWALK_H_
#define WT_MWALK_H_

#ifndef WT_MW_K
#define WT_MW_K 12              // tie groups a lane pair keeps
#endif
#define WT_MW_NANRES 0xffffffffu
#define WT_MW_FREE 0xffffffffu  // key of a free slot (no value has it: wt_walk_key); the groups sit in front of the free slots, in value order

struct WtMwState {
    uint32_t key[WT_MW_K];      // the group's value (key), ascending; free slots (WT_MW_FREE) behind the groups
    uint32_t cnt[WT_MW_K];      // c0 | c1 << 8 | r0 << 16
    int Sp;                     // this lane's share of S
    int nn, cov;                // own set: NaN keys in the column, tracks in play
    bool have, ovf;             // S / groups initialised; more groups than slots (groups invalid until rebuilt)
};

// ---- scans of the lane's own column (w.npad rows, the pad rows hold 0xffffffff: below / equal to no key) ----
// a partner's element moved pa -> pb: what that changes in #{own keys < it} (+ for up, - for down: the keys in [lo, hi)), and
// the own keys equal to the old / the new value
WT_DEV void wt_mw_scan2(const WtWalkCtx &w, int nt, int tid, uint32_t pa, uint32_t pb, int &d_lt, int &eq_a, int &eq_b) {
    const uint32_t lo = pa < pb ? pa : pb, span = (pa < pb ? pb : pa) - lo;
    int between = 0, ea = 0, eb = 0;
    wt_walk_for_keys(w, nt, tid, [&](uint32_t x) {
        between += (x - lo) < span ? 1 : 0;         // lo <= x < hi (unsigned wrap: x < lo gives a huge difference)
        ea += x == pa ? 1 : 0;
        eb += x == pb ? 1 : 0;
    });
    d_lt = pa < pb ? between : -between;            // #{x < pb} - #{x < pa}
    eq_a = ea; eq_b = eb;
}
WT_DEV void wt_mw_scan1(const WtWalkCtx &w, int nt, int tid, uint32_t ka, int &lt_a, int &eq_a) {
    int la = 0, ea = 0;
    wt_walk_for_keys(w, nt, tid, [&](uint32_t x) { la += x < ka ? 1 : 0; ea += x == ka ? 1 : 0; });
    lt_a = la; eq_a = ea;
}
// the smallest key above v (0xffffffff: none -- the pad rows)
WT_DEV uint32_t wt_mw_next_above(const WtWalkCtx &w, int nt, int tid, uint32_t v) {
    uint32_t m = 0xffffffffu;
    wt_walk_for_keys(w, nt, tid, [&](uint32_t x) { m = (x > v && x < m) ? x : m; });
    return m;
}

// ---- the tie groups (both lanes of a pair hold the same; in value order, the free slots last) ----
WT_DEV void wt_mw_clear(WtMwState &st) {
#pragma unroll
    for (int k = 0; k < WT_MW_K; k++) { st.key[k] = WT_MW_FREE; st.cnt[k] = 0u; }
    st.ovf = false;
}
// group `key` loses one entry of set `set`; a group that loses its last entry of a set is no tie group any more: its slot
// closes, the groups behind it move up
WT_DEV void wt_mw_leave(WtMwState &st, uint32_t key, int set) {
    const uint32_t one = set ? 0x100u : 1u, mask = set ? 0xff00u : 0xffu;
    bool gone = false;
#pragma unroll
    for (int k = 0; k < WT_MW_K; k++) {
        const bool hit = st.key[k] == key;
        const uint32_t c = st.cnt[k] - (hit ? one : 0u);
        gone = gone || (hit && (c & mask) == 0u);
        // (from the closed slot on: the next slot's contents)
        st.key[k] = gone ? (k + 1 < WT_MW_K ? st.key[k + 1] : WT_MW_FREE) : st.key[k];
        st.cnt[k] = gone ? (k + 1 < WT_MW_K ? st.cnt[k + 1] : 0u) : c;
    }
}
// group `key` gains one entry of set `set`; false: there is no such group
WT_DEV bool wt_mw_add(WtMwState &st, uint32_t key, int set) {
    const uint32_t one = set ? 0x100u : 1u;
    bool found = false;
#pragma unroll
    for (int k = 0; k < WT_MW_K; k++) {
        const bool hit = st.key[k] == key;
        st.cnt[k] += hit ? one : 0u;
        found = found || hit;
    }
    return found;
}
// a new group (c0 | c1 << 8 | r0 << 16) at its place in value order; no free slot: the groups are invalid from here on (ovf)
WT_DEV void wt_mw_open(WtMwState &st, uint32_t key, uint32_t cnt) {
    if (st.key[WT_MW_K - 1] != WT_MW_FREE) { st.ovf = true; return; }
#pragma unroll
    for (int k = WT_MW_K - 1; k >= 1; k--) {
        const bool up = st.key[k - 1] > key;                    // slot k - 1 (a larger value, or free) moves to k
        const bool here = !up && st.key[k] > key;               // ... and the first slot that stays is followed by the new group
        st.key[k] = up ? st.key[k - 1] : (here ? key : st.key[k]);
        st.cnt[k] = up ? st.cnt[k - 1] : (here ? cnt : st.cnt[k]);
    }
    const bool first = st.key[0] > key;
    st.key[0] = first ? key : st.key[0];
    st.cnt[0] = first ? cnt : st.cnt[0];
}
WT_DEV bool wt_mw_has(const WtMwState &st, uint32_t key) {
    bool found = false;
#pragma unroll
    for (int k = 0; k < WT_MW_K; k++) found = found || st.key[k] == key;
    return found;
}
// a set-0 element moved a -> b: r0 = #{set-0 elements below the group's value} of every group
WT_DEV void wt_mw_shift(WtMwState &st, uint32_t a, uint32_t b) {
#pragma unroll
    for (int k = 0; k < WT_MW_K; k++) {
        const uint32_t v = st.key[k];
        const int d = (b < v ? 1 : 0) - (a < v ? 1 : 0);        // (free slots: both below WT_MW_FREE, d = 0)
        st.cnt[k] += (uint32_t) (d * 0x10000);
    }
}

// ---- the reference's scan (setComparisons.c:328-359) over one tie group, and over the stretch of tie-free set-0 elements
// before it.  C2 = 2 C; (T, P) = (ties, previousTies); m = c0, t = c1, r = r0; prev_end = r0 + c0 of the previous group.
//   tie-free elements while T != 0 (:337-346 with no set-1 entry behind them): previousTies unchanged, U1 += (T - 2P) / 2 each
//   group met with T == 0 (:347-354): the first element sets ties = t and adds t / 2; the others run through :337-346, the LAST
//     one with previousTies += t -- which equals ties: the state is reset (a single element: the state leaks)
//   group met with T != 0: every element through :337-346, the last one with previousTies += t, reset if that equals ties
WT_DEV void wt_mw_group(int m, int t, int r, int &prev_end, int &T, int &Pp, int &C2) {
    if (T) C2 += (r - prev_end) * (T - 2 * Pp);
    if (T == 0) {
        T = t; C2 += t;
        if (m >= 2) { C2 += (m - 2) * t; C2 -= t; T = 0; Pp = 0; }      // (previousTies = t = ties after the last one)
    } else {
        C2 += (m - 1) * (T - 2 * Pp);
        Pp += t;
        C2 += T - 2 * Pp;
        if (Pp == T) { T = 0; Pp = 0; }
    }
    prev_end = r + m;
}

// 2 C from the groups in the slots: one pass, they are in value order
WT_DEV int wt_mw_eval(const WtMwState &st, int n1) {
    int T = 0, Pp = 0, C2 = 0, prev_end = 0;
#pragma unroll
    for (int k = 0; k < WT_MW_K; k++) {
        const uint32_t cn = st.cnt[k];
        int t2 = T, p2 = Pp, c2 = C2, e2 = prev_end;
        wt_mw_group((int) (cn & 0xffu), (int) ((cn >> 8) & 0xffu), (int) (cn >> 16), e2, t2, p2, c2);
        const bool live = st.key[k] != WT_MW_FREE;
        T = live ? t2 : T; Pp = live ? p2 : Pp; C2 = live ? c2 : C2; prev_end = live ? e2 : prev_end;
    }
    if (T) C2 += (n1 - prev_end) * (T - 2 * Pp);
    return C2;
}

// More groups than slots: 2 C by enumerating the distinct set-0 values in order (lane 0 finds the next one, its
// multiplicity and rank; lane 1 counts the set-1 entries equal to it), and the slots refilled with the first WT_MW_K groups
// (ovf stays set if there are more).  Both lanes of the pair run this together.
WT_DEV int wt_mw_eval_slow(const WtWalkCtx &w, WtMwState &st, int n1, int nt, int tid) {
    const uint32_t half = (uint32_t) (tid & 1);
    wt_mw_clear(st);
    int T = 0, Pp = 0, C2 = 0, prev_end = 0, groups = 0;
    uint32_t v = 0u;
    for (;;) {
        const uint32_t mine = half ? 0u : wt_mw_next_above(w, nt, tid, v);
        const uint32_t other = wt_pair_xchg(mine, tid);
        const uint32_t nx = half ? other : mine;
        if (nx == 0xffffffffu) break;
        int lt = 0, eq = 0;
        wt_mw_scan1(w, nt, tid, nx, lt, eq);        // lane 0: r = lt, m = eq; lane 1: t = eq
        const uint32_t pk = wt_pair_xchg((uint32_t) lt | ((uint32_t) eq << 8), tid);
        const int m = half ? (int) ((pk >> 8) & 0xffu) : eq, r = half ? (int) (pk & 0xffu) : lt, t = half ? eq : (int) ((pk >> 8) & 0xffu);
        if (t > 0) {
            wt_mw_group(m, t, r, prev_end, T, Pp, C2);
            groups++;
            if (groups <= WT_MW_K) wt_mw_open(st, nx, (uint32_t) m | ((uint32_t) t << 8) | ((uint32_t) r << 16));
        }
        v = nx;
    }
    if (T) C2 += (n1 - prev_end) * (T - 2 * Pp);
    st.ovf = groups > WT_MW_K;
    return C2;
}

// S, the NaN count and the tie groups from scratch: row by row of set 0, lane 1 counts the set-1 entries below / equal to
// it, lane 0 the set-0 entries (the group's r0 and c0).
WT_DEV void wt_mw_init(const WtWalkCtx &w, WtMwState &st, int n1, int nt, int tid) {
    const uint32_t half = (uint32_t) (tid & 1);
    const uint32_t *col = w.col + tid;
    wt_mw_clear(st);
    int nn = 0;
    wt_walk_for_keys(w, nt, tid, [&](uint32_t x) { nn += x == WT_WALK_NANKEY ? 1 : 0; });
    st.nn = nn;
    int S = 0;
    for (int i = 0; i < n1; i++) {
        const uint32_t mine = half ? 0u : col[i * nt];
        const uint32_t other = wt_pair_xchg(mine, tid);
        const uint32_t x = half ? other : mine;
        int lt = 0, eq = 0;
        wt_mw_scan1(w, nt, tid, x, lt, eq);
        const uint32_t pk = wt_pair_xchg((uint32_t) lt | ((uint32_t) eq << 8), tid);
        const int lt1 = half ? lt : (int) (pk & 0xffu), eq1 = half ? eq : (int) ((pk >> 8) & 0xffu);       // in set 1
        const int lt0 = half ? (int) (pk & 0xffu) : lt, eq0 = half ? (int) ((pk >> 8) & 0xffu) : eq;       // in set 0
        S += half ? lt1 : 0;                          // (lane 1 keeps all of S to begin with)
        if (eq1 > 0 && !st.ovf && !wt_mw_has(st, x)) wt_mw_open(st, x, (uint32_t) eq0 | ((uint32_t) eq1 << 8) | ((uint32_t) lt0 << 16));
    }
    st.Sp = S;
    st.have = true;
}

// One step: this lane's event (valid: its own column's row `row` takes key nk) together with the partner's.
WT_DEV void wt_mw_step(const WtWalkCtx &w, WtMwState &st, int nt, int tid, bool valid, uint32_t row, uint32_t nk, uint32_t meta) {
    const uint32_t half = (uint32_t) (tid & 1);
    uint32_t *col = w.col + tid;
    const uint32_t a = col[row * (uint32_t) nt];
    if (valid) st.cov += ((meta & WT_WALK_INC) ? 1 : 0) - ((meta & WT_WALK_DEC) ? 1 : 0);
    if (!st.have) {                             // (before the stretch's state exists: the column only)
        if (valid) col[row * (uint32_t) nt] = nk;
        return;
    }
    if (valid && !half) col[row * (uint32_t) nt] = nk;          // set 0's event first: its lane scans a column that has it
    // the partner's event (0xffffffff: none)
    const uint32_t pa = wt_pair_xchg(valid ? a : 0xffffffffu, tid);
    const uint32_t pb = wt_pair_xchg(valid ? nk : 0xffffffffu, tid);
    const bool pvalid = pa != 0xffffffffu;
    int d_lt = 0, eq_a = 0, eq_b = 0;
    if (pvalid) wt_mw_scan2(w, nt, tid, pa, pb, d_lt, eq_a, eq_b);
    if (valid && half) col[row * (uint32_t) nt] = nk;           // ... set 1's afterwards: scanned for as it was
    if (valid) st.nn += (nk == WT_WALK_NANKEY ? 1 : 0) - (a == WT_WALK_NANKEY ? 1 : 0);
    // S: a set-0 element a -> b: + #{y < b} - #{y < a} (lane 1's scan); a set-1 element: + #{x > b} - #{x > a}
    //    = (#{x < a} + #{x == a}) - (#{x < b} + #{x == b}) (lane 0's)
    if (pvalid) st.Sp += half ? d_lt : eq_a - eq_b - d_lt;
    // what the partner counted for MY event
    const uint32_t res = wt_pair_xchg((uint32_t) eq_a | ((uint32_t) eq_b << 8), tid);
    if (st.ovf) return;
    // both lanes, the same updates: set 0's event, then set 1's
    const bool v0 = half ? pvalid : valid, v1 = half ? valid : pvalid;
    const uint32_t a0 = half ? pa : a, b0 = half ? pb : nk, a1 = half ? a : pa, b1 = half ? nk : pb;
    // (counts in the OTHER set: lane 1 counted set 1 for set 0's event, lane 0 set 0 for set 1's)
    const int e0a = half ? eq_a : (int) (res & 0xffu), e0b = half ? eq_b : (int) ((res >> 8) & 0xffu);
    const int e1a = half ? (int) (res & 0xffu) : eq_a, e1b = half ? (int) ((res >> 8) & 0xffu) : eq_b;
    if (v0 && a0 != b0) {
        wt_mw_shift(st, a0, b0);
        if (e0a > 0) wt_mw_leave(st, a0, 0);
        if (e0b > 0 && !wt_mw_add(st, b0, 0)) {
            // a new group (pair-uniform: both lanes hold the same groups): r0 = #{x' < b0} in set 0 as it is now, lane 0 counts
            int lt = 0, eq = 0;
            if (!half) wt_mw_scan1(w, nt, tid, b0, lt, eq);
            const uint32_t o = wt_pair_xchg((uint32_t) lt, tid);
            const int r0 = half ? (int) o : lt;
            wt_mw_open(st, b0, 1u | ((uint32_t) e0b << 8) | ((uint32_t) r0 << 16));
        }
    }
    if (v1 && a1 != b1 && !st.ovf) {
        if (e1a > 0) wt_mw_leave(st, a1, 1);
        if (e1b > 0 && !wt_mw_add(st, b1, 1)) {
            // a new group: c0 = #{x == b1} = e1b, r0 = #{x < b1} in set 0 (lane 0 counts; rare)
            int lt = 0, eq = 0;
            if (!half) wt_mw_scan1(w, nt, tid, b1, lt, eq);
            const uint32_t o = wt_pair_xchg((uint32_t) lt, tid);
            const int r0 = half ? (int) o : lt;
            wt_mw_open(st, b1, (uint32_t) e1b | 0x100u | ((uint32_t) r0 << 16));
        }
    }
}

// ---- the window's phases ----
// positions of the lane's stretch that have events, the stretch's first one (the counter words as the first pass left them)
WT_DEV void wt_mwalk_events(const WtParams &P, const WtCtx &c, WtWalkCtx &w, WtWalkLane &L, int tid, int nt) {
    const int S = w.S, q = tid >> 1, a = q * S;
    uint32_t evmask = 0;
    int fe = -1;
    for (int s = 0; s < S; s++) {
        if (!wt_walk_cnt_n(w, w.cnt[a + s])) continue;
        if (fe < 0) fe = a + s;
        evmask |= 1u << s;
    }
    w.fe[q] = fe;
    L.evmask = evmask;
    L.emitmask = 0;
}

// The steps of one position of the lane's stretch.  FIXED: step k pairs the k-th event of set 0 with the k-th of set 1 (each
// lane's own slots of the position), then come the overflow list's entries (one step each; most of them other positions':
// skipped).  !FIXED (fallback): the position's events in the sorted sequence, one per step, whichever set they belong to.
struct WtMwPos {
    uint32_t n_slot;        // steps that read slots (FIXED) / events of the position (!FIXED)
    uint32_t n_own;         // ... of which this lane has an event of its own (FIXED)
    uint32_t n_steps;       // all steps (FIXED with events beyond the slots: + the overflow list's entries)
    uint32_t from;          // slab index of the first one
};
template <bool FIXED>
WT_DEV WtMwPos wt_mw_pos(const WtWalkCtx &w, int a, int s, uint32_t half, uint32_t lcap, uint32_t novf, uint32_t ev0) {
    WtMwPos p;
    if (FIXED) {
        const uint32_t v = w.cnt[a + s];
        const uint32_t c0 = v & WT_WALK_PNMASK, c1 = (v >> 7) & WT_WALK_PNMASK;
        const uint32_t m0 = c0 < lcap ? c0 : lcap, m1 = c1 < lcap ? c1 : lcap;
        p.n_own = half ? m1 : m0;
        p.n_slot = m0 > m1 ? m0 : m1;
        p.n_steps = p.n_slot + ((c0 > lcap || c1 > lcap) ? novf : 0u);
        p.from = ((((uint32_t) (a + s)) << 1) | half) * lcap;
    } else {
        const uint32_t o0 = w.off[a + s], o1 = w.off[a + s + 1];
        p.n_slot = p.n_own = p.n_steps = o1 - o0;
        p.from = o0 - ev0;
    }
    return p;
}

// The lane's stretch (see the head of this file).  The result of an emitted position -- the table index |2 U1 - 2 mu|, or
// WT_MW_NANRES -- replaces its counter word (read for the last time when the pair's cursor reached the position).
template <bool FIXED>
WT_DEV void wt_mwalk_lane(const WtParams &P, const WtCtx &c, WtWalkCtx &w, WtWalkLane &L, uint32_t ev0, int tid, int nt) {
    const int n1 = P.n_set0, n2 = P.n_tracks - P.n_set0;
    const int S = w.S, a = (tid >> 1) * S;
    const uint32_t half = (uint32_t) (tid & 1);
    const bool strict0 = (P.flags & WT_STRICT_SET0) != 0, strict1 = (P.flags & WT_STRICT_SET1) != 0;
    const int32_t room = c.sh->emit_hi - (c.sh->w0 + a);         // positions of the stretch below the range end
    const int mu2 = 2 * (n1 * n2 / 2);                           // setComparisons.c:386 (C int division), doubled
    const uint32_t evmask = L.evmask;
    WtMwState st;
    wt_mw_clear(st);
    st.Sp = 0; st.nn = 0; st.have = false;
    st.cov = w.ncov[tid];
    uint32_t emitmask = 0;
    const uint32_t novf = FIXED ? (w.novf[0] < w.ov_cap ? w.novf[0] : w.ov_cap) : 0u;
    const uint32_t lcap = (uint32_t) w.capp >> 1;

    // the event of step k of a position (valid: this lane has one)
    auto event_at = [&](const WtMwPos &p, int s, uint32_t k, bool &valid, uint32_t &row, uint32_t &key, uint32_t &meta) {
        if (k < p.n_slot) {
            const WtWalkEvent e = w.slab[p.from + (FIXED ? (k < p.n_own ? k : 0u) : k)];
            meta = e.meta; key = e.key;
            valid = FIXED ? k < p.n_own : (meta & 1u) == half;
        } else {                                    // (FIXED) an entry of the overflow list: this position's, this lane's set?
            const WtWalkOvf q = w.ovf[k - p.n_slot];
            meta = q.meta; key = q.key;
            valid = q.pos == (uint32_t) (a + s) && (meta & 1u) == half;
        }
        row = valid ? (meta & 0xffffu) >> 1 : 0u;
    };
    // a position's events are all applied: emitted where both sets have a track in play (setComparisons.c:283-289; strict: all
    // of the set's); the value from S and the groups
    auto finish_position = [&](int s, bool at_end) {
        const int covo = (int) wt_pair_xchg((uint32_t) st.cov, tid);
        const int So = (int) wt_pair_xchg((uint32_t) st.Sp, tid);
        const int nno = (int) wt_pair_xchg((uint32_t) st.nn, tid);
        const int cov0 = half ? covo : st.cov, cov1 = half ? st.cov : covo;
        const bool emit = at_end && (strict0 ? cov0 == n1 : cov0 > 0) && (strict1 ? cov1 == n2 : cov1 > 0) && s < room;
        int C2 = wt_mw_eval(st, n1);
        if (wt_walk_any(at_end && st.ovf)) {        // (rare: more tie groups than slots somewhere in the wavefront)
            if (at_end && st.ovf) C2 = wt_mw_eval_slow(w, st, n1, nt, tid);
        }
        if (emit) {
            int k = 2 * (st.Sp + So) + C2 - mu2;
            k = k < 0 ? -k : k;
            emitmask |= 1u << s;
            w.cnt[a + s] = (st.nn + nno) ? WT_MW_NANRES : (uint32_t) k;
        }
    };

    if (evmask) {
        // the stretch's first position with events: its events into the columns, then the state from scratch
        int s = (int) wt_ctz64((uint64_t) evmask);
        WtMwPos p = wt_mw_pos<FIXED>(w, a, s, half, lcap, novf, ev0);
        for (uint32_t k = 0; k < p.n_steps; k++) {
            bool valid; uint32_t row, key, meta;
            event_at(p, s, k, valid, row, key, meta);
            wt_mw_step(w, st, nt, tid, valid, row, key, meta);
        }
    }
    // (every pair of the wavefront takes part: a stretch without events still has a partner lane to answer)
    if (evmask) wt_mw_init(w, st, n1, nt, tid);
    int s = evmask ? (int) wt_ctz64((uint64_t) evmask) : S;
    if (evmask) finish_position(s, true);
    // from here on every pair at its own pace: one step per turn, the position's result when its last event has been applied
    uint32_t rest = evmask ? evmask & ~((2u << s) - 1u) : 0u;           // positions with events after s  (s = 31: 2u << 31 = 0)
    if (s >= 31) rest = 0u;
    uint32_t k = 0;
    WtMwPos p{0u, 0u, 0u, 0u};
    bool active = rest != 0u;
    if (active) { s = (int) wt_ctz64((uint64_t) rest); rest &= rest - 1u; p = wt_mw_pos<FIXED>(w, a, s, half, lcap, novf, ev0); }
    while (wt_walk_any(active)) {
        if (active) {
            bool valid; uint32_t row, key, meta;
            event_at(p, s, k, valid, row, key, meta);
            wt_mw_step(w, st, nt, tid, valid, row, key, meta);
            k++;
            const bool at_end = k == p.n_steps;
            finish_position(s, at_end);
            if (at_end) {
                k = 0;
                active = rest != 0u;
                if (active) { s = (int) wt_ctz64((uint64_t) rest); rest &= rest - 1u; p = wt_mw_pos<FIXED>(w, a, s, half, lcap, novf, ev0); }
            }
        }
    }
    L.emitmask = emitmask;
}

// the stretch's runs (as wt_walk_write; the value comes from the table)
WT_DEV void wt_mwalk_write(const WtParams &P, WtCtx &c, const WtWalkCtx &w, const WtWalkLane &L, int tid, int nt) {
    if (!L.emitmask) return;
    const int S = w.S, q = tid >> 1, a = q * S, half = tid & 1;
    const int32_t w0 = c.sh->w0;
    long long idx = c.sh->goffset + (long long) w.base[q << 1];
    int32_t after = 0;
    bool have_after = false;
    unsigned long long bp = 0;
    int rank = 0;
    for (int s = 0; s < S; s++) {
        if (!((L.emitmask >> s) & 1u)) continue;
        const long long o = idx++;
        if ((rank++ & 1) != half) continue;
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
        const uint32_t k = w.cnt[a + s];
        P.o_start[o] = w0 + a + s;
        P.o_finish[o] = fin;
        P.o_value[o] = k == WT_MW_NANRES ? wt_nan() : P.mwu_table[k < (uint32_t) P.mwu_kmax ? k : (uint32_t) P.mwu_kmax];
    }
    if (bp) wt_lds_add64(&c.sh->bp_sum, bp);
}

#endif  // WT_MWALK_H_
