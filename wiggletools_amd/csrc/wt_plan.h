// wt_plan.h -- host-side planning shared by the HIP engine (wt_engine.hip) and
// the CPU emulator used by the no-GPU tests: window width / workgroup size /
// LDS carve for a given track count and reducer, and the per-chromosome window
// tables.  Pure C++, no HIP.
#ifndef WT_PLAN_H_
#define WT_PLAN_H_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "wt_core.h"

struct WtPlan {
    int W = 0;          // window width (bp), multiple of 64
    int T = 0;          // workgroup size (lanes), multiple of 64
    int n_words = 0;
    int spitch = 0, cpitch = 0, count_segs = 8;
    int chunk_tracks = 0, n_chunks = 1;   // tracks resident in LDS at a time / number of chunks
    int lanes_per_pos = 1;  // MWU: lanes of a workgroup sharing one window position (T = lanes_per_pos * W, K = 1)
    int off_acc = 0, off_ev = 0, off_ltv = 0, off_ltc = 0, off_gtv = 0, off_gtc = 0, off_tbase = 0, off_tpfx = 0, off_tfirst = 0, off_dsh = 0, off_tdef = 0;
    int off_qa = 0, off_ltq = 0, off_gtq = 0, delta_q = 0;
    int delta_ns = 1;       // difference-array plan: sets whose sums a position keeps (2: TTestReduction, wt_delta_scan3_tt)
    bool delta = false;     // difference-array plan (wt_delta.h)
    int off_S = 0, off_cnt = 0, off_segtot = 0, off_U = 0, off_cover = 0, off_E = 0, off_epfx = 0, off_nextw = 0, off_gbase = 0, off_scratch = 0, off_shared = 0;
    int off_dflt32 = 0;
    int lds_bytes = 0;
    int scratch_elem = 0;   // bytes per scratch element (0: op needs no scratch)
    long long scratch_slab = 0;   // > 0: scratch columns in global memory, this many bytes per workgroup
    long long attr_slab = 0;      // MWU: bytes of per-rank attributes per workgroup (always global)
    int ppt = 1;            // consecutive window positions per lane (1 or 4)
    int regcol = 0;         // median / MWU: register slots of the per-lane value column (0: LDS / global columns)
    int walk_S = 0;         // median by walking (wt_walk.h): positions per lane (0: not that plan)
    int walk_capp = 0;      // ... fixed event slots per position
    int walk_ov = 0;        // ... entries of the overflow list
    int walk_off_at = 0;    // ... where the fallback's offsets start in the workgroup's slab (bytes)
    int walk_pair = 0;      // ... 1: two lanes per stretch (half the column each)
    int walk_mwu = 0;       // ... 1: MWUReduction by walking (wt_mwalk.h): the two lanes of a stretch hold one SET each
    int off_wcol = 0, off_wcnt = 0, off_woff = 0, off_wtot = 0, off_wbase = 0, off_wgt = 0, off_wncov = 0, off_wfe = 0, off_wdk = 0, off_wguess = 0;
};

static inline int wt_align16(int x) { return (x + 15) & ~15; }

static inline bool wt_op_needs_scratch(int op) { return op == WT_OP_MEDIAN || op == WT_OP_MWU; }

// LDS bytes for a candidate (W, T)
// n_tracks: all tracks (scratch columns); chunk: tracks whose bitmaps are resident at a time
static inline void wt_carve(int n_tracks, int op, int W, int T, int scratch_elem, WtPlan &p, int chunk = 0,
                            bool scratch_global = false, int regcol = 0, int n_set0 = 0) {
    if (chunk <= 0 || chunk > n_tracks) chunk = n_tracks;
    p.chunk_tracks = chunk;
    p.n_chunks = (n_tracks + chunk - 1) / chunk;
    p.W = W; p.T = T; p.n_words = W / 64;
    p.spitch = W / 32 + 1;                    // {S,C} pairs per track, +1: rows start on different banks
    p.cpitch = (W / 32 + 2) & ~1;             // u16 rank prefix per 32-bit word + 1 spare flag entry, even count
    p.scratch_elem = scratch_elem;
    int o = 0;
    p.off_S = o;       o = wt_align16(o + chunk * p.spitch * 8);
    p.off_cnt = o;     o = wt_align16(o + chunk * p.cpitch * 2);
    p.count_segs = 8;                         // lanes per track in the count phase, >= 4 words each
    while (p.count_segs > 1 && (W / 32) / p.count_segs < 4) p.count_segs >>= 1;
    p.off_segtot = o;  o = wt_align16(o + chunk * p.count_segs * 4);
    p.off_U = o;       o = wt_align16(o + p.n_words * 8);
    p.off_cover = o;   o = wt_align16(o + 4 * p.n_words * 8);
    p.off_E = o;       o = wt_align16(o + p.n_words * 8);
    p.off_epfx = o;    o = wt_align16(o + (p.n_words + 1) * 4);
    p.off_nextw = o;   o = wt_align16(o + p.n_words * 2);
    p.off_gbase = o;   o = wt_align16(o + chunk * 8);
    p.off_dflt32 = o;  if (regcol) o = wt_align16(o + n_tracks * 4);
    p.off_scratch = o;
    long long scr_bytes = 0;
    if (op == WT_OP_MEDIAN || op == WT_OP_MWU) scr_bytes = (long long) n_tracks * W * scratch_elem;    // one column per position
    if (regcol) scr_bytes = op == WT_OP_MWU ? (long long) n_set0 * T * 4 : 0;       // register columns: MWU parks the sorted set 0, one column per LANE
    scr_bytes = (scr_bytes + 255) & ~255ll;
    p.scratch_slab = scratch_global ? scr_bytes : 0;
    // MWU: the per-rank attribute words (one u32 per set-0 track and lane, written once and read
    // once per run) always live in a global slab per workgroup: keeping them in LDS halved the lanes
    // per CU for the part that matters, the N^2 ranking over the value column
    p.attr_slab = (op == WT_OP_MWU && !regcol) ? (((long long) n_tracks * W * 4 + 255) & ~255ll) : 0;
    p.regcol = regcol;
    if (!scratch_global) o = (int) std::min<long long>(o + scr_bytes, 1 << 30);
    p.off_shared = o;  o = wt_align16(o + (int) sizeof(WtShared));
    p.lds_bytes = o;
}

// Difference-array plan (Sum / Mean over float tracks, wt_delta.h): 8 positions per lane; LDS
// does not depend on the track count.  WTAMD_DELTA_T overrides the workgroup size (tests).
// Measured on MI355X (scale 0.01 probes, kernel + index ms at T = 256 / 512): mean of 100 tracks
// 1.43+0.37 / 1.35+0.31, sum of 1000 tracks 9.2+3.5 / 8.4+2.8: the widest window wins here too.
static inline bool wt_op_is_var_family(int op) {
    return op == WT_OP_VAR || op == WT_OP_STDDEV || op == WT_OP_ENTROPY || op == WT_OP_CV;
}

// `squares`: the launch also accumulates the sum of squares (var / stddev / CV): two more u64 arrays
// per position: ~145 KB of LDS for the 4096-bp window, one workgroup per CU -- measured 21 %
// faster than 2048-bp windows (86 KB: also one workgroup per CU, but of 4 waves).
// `two_sets` (round 6, TTestReduction): sums, squares and coverage of TWO sets per position -- 56 bytes: a 2048-bp window (115 KB),
// 256 lanes for the scans, and like every launch with squares 768 for the passes over the runs.
static inline void wt_make_delta_plan(WtPlan &p, int n_tracks, bool squares = false, bool two_sets = false) {
    if (two_sets) squares = true;
    // Sum / Mean: 1024 lanes, an 8192-bp window, one workgroup of 16 waves per CU (round 3: 6 % faster than two
    // workgroups of 512 once pass 2 had shed its instructions -- the per-window chain of dependent round trips
    // is paid half as often; round 2 had measured +1.5 %).
    // With squares the window stays at 4096 bp (four 64-bit accumulators per position) but the workgroup is 768 lanes since
    // round 5: the passes over the runs -- 88 % of a window, each of VALU, LDS and loads ~40 % busy at two wavefronts per
    // SIMD -- are flat loops any number of wavefronts can share; the scans, which own 8 positions per lane, are run by the
    // first W / 8 = 512 lanes (wt_delta_kernel: `nts`).  768 and not 1024: three wavefronts per SIMD leave 168 registers per
    // lane, which the scans' 128-bit arithmetic fits into; at 128 they spilled 41 (C3: 57.5 against 55.5 ms; 512 lanes: 63.2).
    // WTAMD_DELTA_T=512: as before round 5; WTAMD_DELTA_SQ_T: any multiple of 64 from 512 to the launch bound.
    const char *eT = getenv("WTAMD_DELTA_T");
    const int T0 = squares ? WT_DELTA_SQ_T0 : 1024;
    int T = eT ? atoi(eT) : T0;
    (void) n_tracks;
    const int Tmax = squares ? WT_DELTA_SQ_T0 : WT_MAX_DELTA_T;
    if (T < 64 || T > Tmax || ((T & (T - 1)) && !(squares && T > 512 && T % 64 == 0))) T = T0;
    if (squares && !eT) {
        const char *eS = getenv("WTAMD_DELTA_SQ_T");
        if (eS && atoi(eS) >= 512 && atoi(eS) <= Tmax && atoi(eS) % 64 == 0) T = atoi(eS);
    }
    int TS = squares && T > 512 ? 512 : T;             // lanes of the scans: one per WT_DELTA_K positions
    const int ns = two_sets ? 2 : 1;
    if (two_sets) {
        const char *e2 = getenv("WTAMD_DELTA_TT_W");    // experiments: window of the two-sample launches (1024 / 2048)
        TS = (e2 && atoi(e2) == 1024) ? 128 : 256;
        if (T < 2 * TS) T = 2 * TS;         // (the scans split by set: 2 TS lanes)
    }
    p = WtPlan();
    p.delta = true;
    p.delta_ns = ns;
    p.T = T; p.ppt = WT_DELTA_K; p.W = WT_DELTA_K * TS; p.n_words = p.W / 64;
    p.chunk_tracks = 0; p.n_chunks = 1;
    int o = 0;
    p.off_acc = o;    o = wt_align16(o + (ns * p.W + p.W / 32) * 8);      // (+ W / 32: the staged runs' spare entries, WT_STAGE_AT)
    p.off_ev = o;     o = wt_align16(o + (ns * p.W + p.W / 32) * 4);
    p.off_U = o;      o = wt_align16(o + p.n_words * 8);
    p.off_E = o;      o = wt_align16(o + p.n_words * 8);
    p.off_epfx = o;   o = wt_align16(o + std::max(p.n_words + 1, 48) * 4);    // (device, Sum / Mean: [0, 16) the wavefronts' run counts, [32, 48) the sub-ranges' ranks)
    p.off_nextw = o;  o = wt_align16(o + p.n_words * 2);
    p.off_ltv = o;    o = wt_align16(o + ns * TS * 8);
    p.off_ltc = o;    o = wt_align16(o + std::max(T, ns * TS) * 4);     // (the tracks' run counts AND the scan lanes' totals)
    p.off_gtv = o;    o = wt_align16(o + ns * (TS / WT_DELTA_GROUP) * 8);
    p.off_gtc = o;    o = wt_align16(o + (std::max(T, ns * TS) / WT_DELTA_GROUP) * 4);
    p.off_tbase = o;  o = wt_align16(o + T * 8);
    p.off_tpfx = o;   o = wt_align16(o + (T + 1) * 4);
    p.off_tfirst = o; o = wt_align16(o + WT_DELTA_TF * 2);
    p.off_tdef = o;   o = wt_align16(o + (squares ? 0 : T * 4));        // (non-zero defaults: Sum / Mean only)
    p.off_dsh = o;    o = wt_align16(o + (int) sizeof(WtDeltaShared));
    p.delta_q = squares ? 1 : 0;
    if (squares) {
        p.off_qa = o;  o = wt_align16(o + 2 * ns * p.W * 8);
        p.off_ltq = o; o = wt_align16(o + 2 * ns * TS * 8);
        p.off_gtq = o; o = wt_align16(o + 2 * ns * (TS / WT_DELTA_GROUP) * 8);
    }
    p.off_shared = o; o = wt_align16(o + (int) sizeof(WtShared));
    p.lds_bytes = o;
}

static inline bool wt_defaults_fit_f32(const double *d, int n);

// the difference-array plan of reducer `op` (squares for the var family, two sets for the t-test)
static inline void wt_make_delta_plan_for(WtPlan &p, int n_tracks, int op) {
    wt_make_delta_plan(p, n_tracks, wt_op_is_var_family(op), op == WT_OP_TTEST);
}

// Sum / Mean over float tracks whose defaults are all zero can take the exact difference-array
// path (the kernel still verifies every window's exponent range).
static inline bool wt_delta_eligible(int op, bool value_f64, int n_tracks, const double *defaults) {
    if (getenv("WTAMD_NO_DELTA")) return false;
    const bool sq = wt_op_is_var_family(op);
    if (op == WT_OP_TTEST) {
        // TTestReduction (round 6): sums and squares of the values in play per set; defaults play no part (setComparisons.c:69-81)
        // (... but the general kernel that patches a window reads them into its float staging: defaults that are no floats keep
        //  the whole reduction on the general kernel's f64 staging, wt_launch_patch)
        return !getenv("WTAMD_NO_DELTA_TTEST") && !value_f64 && n_tracks >= 8 && n_tracks <= 32767 && wt_defaults_fit_f32(defaults, n_tracks);
    }
    const bool mm = op == WT_OP_MAX || op == WT_OP_MIN;        // round 6: range updates of a segment tree (wt_delta.h); zero defaults only
    if (mm && getenv("WTAMD_NO_DELTA_MINMAX")) return false;
    if (op != WT_OP_SUM && op != WT_OP_MEAN && !sq && !mm) return false;
    if (sq && (n_tracks < 8 || getenv("WTAMD_NO_DELTA_VAR"))) return false;     // (the split square accumulators need N >= 8)
    // a handful of tracks: nothing to gain over the general kernel
    // (measured: 10 tracks 0.77 general vs 0.53 ms difference array; 100 tracks 2.05 vs 1.33)
    const char *eM = getenv("WTAMD_DELTA_MIN_TRACKS");
    const int min_tracks = eM ? atoi(eM) : 4;
    if (value_f64 || n_tracks < min_tracks || n_tracks > 32767) return false;    // ev[] counts in 16-bit halves
    // Defaults: an absent track adds its default value (reducers.c:294-307, 375-401: every track's value or
    // default is summed).  Zero never changes a sum; a non-zero default is one more term of every sum, which
    // stays exact under the same condition as the values as long as it is itself a float (round 3: Sum / Mean
    // only -- the squares of the var family count the tracks IN PLAY).  WTAMD_NO_DELTA_DEFAULTS: zero only.
    for (int i = 0; i < n_tracks; i++) {
        if (defaults[i] == 0.0) continue;
        if (sq || mm || getenv("WTAMD_NO_DELTA_DEFAULTS")) return false;
        const float f = (float) defaults[i];
        if (!((double) f == defaults[i]) || !(f - f == 0.0f)) return false;       // not a float / NaN / Inf
    }
    return true;
}

// Difference-array launches: what the kernel needs to know about the defaults (after wt_plan_to_params).
static inline void wt_delta_defaults_params(const double *defaults, int n_tracks, WtParams &P) {
    P.delta_df = 0; P.def_emin = 255; P.def_emax = 0;
    if (P.op == WT_OP_TTEST) return;        // (setComparisons.c:69-81: only tracks in play are summed -- P.op is set before this call)
    for (int i = 0; i < n_tracks; i++) {
        if (defaults[i] == 0.0) continue;
        const float f = (float) defaults[i];
        uint32_t bits;
        memcpy(&bits, &f, 4);
        int e = (int) ((bits >> 23) & 0xffu);
        if (e == 0) e = 1;                  // denormal: exponent 1 without the hidden bit
        P.delta_df = 1;
        if (e < P.def_emin) P.def_emin = e;
        if (e > P.def_emax) P.def_emax = e;
    }
}

// Median by walking (wt_walk.h): T lanes x S consecutive positions each; the lanes' value columns (N x 4 bytes per
// lane) are what fills the LDS.  Same domain as the register columns of the bitmap kernel (wt_regcol_slots: float
// tracks, float-exact defaults, at most 128 of them); the window's runs are enumerated through the difference-array
// kernel's flat index space, one chunk: N <= T.
// WTAMD_WALK_T / WTAMD_WALK_S: experiments.  The slab of a workgroup: event slots per position (WTAMD_WALK_CAPP; sized
// from the data's density) + an overflow list of WTAMD_WALK_OV entries; a window with more than a few events beyond their
// slots (WT_WALK_OV_SCAN: a lane scans the whole list for every such position) falls back to sorting its events into the
// same memory, in as many rounds as it takes: at least the 2 N S events the stretch of one lane can hold fit.
// events_per_bp: what the data is expected to hold (the host's estimate from the run count and the covered span; <= 0: unknown).
// mwu_n_set0 >= 0: MWUReduction by walking (wt_mwalk.h) -- always two lanes per stretch, lane h of a pair holds the column of
// SET h (max(n1, n2) rows); 1 <= n1, n2 <= 64.
static inline bool wt_make_walk_plan(WtPlan &p, int n_tracks, int nr, double events_per_bp, int hard_limit = 160 * 1024, int mwu_n_set0 = -1) {
    // One lane per stretch: 256 lanes x 32 positions (an 8192-bp window, one workgroup per CU) when the columns fit, else fewer
    // positions, then fewer lanes.  Measured (MI355X, chromosome 21, 100 tracks): 256 x 32 11.8 ms, 64 x 32 11.9, 256 x 16 12.6,
    // 128 x 32 12.8, 128 x 16 14.3: the per-window chains of dependent loads are what the wider window saves.
    // Pair mode (the default; WTAMD_WALK_PAIR=0 for the above): two lanes per stretch, half the column each -- twice the lanes
    // per CU: 256 lanes = 128 stretches x 16 positions (a 2048-bp window, 72 KB, two workgroups per CU) 9.9-10.0 ms, 64 x 32
    // the same, 128 x 16 10.0, 256 x 8 10.6, 512 lanes (one workgroup of 8 waves) x 16 / 32 11.4-11.5.
    const char *eT = getenv("WTAMD_WALK_T"), *eS = getenv("WTAMD_WALK_S"), *eP = getenv("WTAMD_WALK_PAIR");
    const bool mwu = mwu_n_set0 >= 0;
    const int n_big = mwu ? std::max(mwu_n_set0, n_tracks - mwu_n_set0) : 0;
    if (mwu && (mwu_n_set0 < 1 || n_tracks - mwu_n_set0 < 1 || n_big > 64)) return false;
    const int pair = mwu ? 1 : (eP ? (atoi(eP) != 0 ? 1 : 0) : WT_WALK_PAIR_DEFAULT);
    const int wantT = eT ? atoi(eT) : 256, wantS = eS ? atoi(eS) : (pair ? 16 : 32);
    // (pair mode: twice through the candidates -- first only what leaves room for a second workgroup on the CU: two
    //  workgroups of 4 waves beat one of 8, their phases overlap)
    struct Cand { int T, S; bool half_cu; };
    for (const Cand cd : {Cand{wantT, wantS, pair != 0}, Cand{128, 16, pair != 0}, Cand{64, 32, pair != 0},
                          Cand{wantT, wantS, false}, Cand{wantT, 16, false}, Cand{256, 16, false}, Cand{128, 16, false}, Cand{64, 16, false}}) {
        const int T = cd.T, S = cd.S;
        if (cd.half_cu && (eT || eS) && (T != wantT || S != wantS)) continue;       // (an explicit request is tried as it is first)
        if (T != 64 && T != 128 && T != 256 && !(pair && T == 512)) continue;
        if (S != 4 && S != 8 && S != 16 && S != 32) continue;
        if (n_tracks > T) continue;
        WtPlan q;
        q.T = T; q.W = (T >> pair) * S; q.n_words = q.W / 64; q.ppt = S; q.walk_S = S; q.regcol = nr; q.walk_pair = pair; q.walk_mwu = mwu ? 1 : 0;
        if (q.W < 64) continue;
        q.chunk_tracks = n_tracks; q.n_chunks = 1;
        int o = 0;
        q.off_wcol = o;   o = wt_align16(o + (((mwu ? n_big : (pair ? (n_tracks + 1) / 2 : n_tracks)) + 7) & ~7) * T * 4);        // (WT_WALK_PAD rows)
        q.off_wcnt = o;   o = wt_align16(o + q.W * 4);
        q.off_woff = 0;                                                             // (the fallback's offsets live in the slab)
        q.off_wtot = o;   o = wt_align16(o + T * 4);
        q.off_wbase = o;  o = wt_align16(o + (T + 1) * 4);
        q.off_wgt = o;    o = wt_align16(o + (T / 64 + 1) * 4);
        q.off_wncov = o;  o = wt_align16(o + T * 4);
        q.off_wfe = o;    o = wt_align16(o + T * 4);
        q.off_wdk = o;    o = wt_align16(o + n_tracks * 4);
        q.off_wguess = o; o = wt_align16(o + 8);       // (+ the overflow counter)
        q.off_tbase = o;  o = wt_align16(o + T * 8);
        q.off_ltc = o;    o = wt_align16(o + T * 4);
        q.off_gtc = o;    o = wt_align16(o + (T / WT_DELTA_GROUP + 1) * 4);
        q.off_tpfx = o;   o = wt_align16(o + (T + 1) * 4);
        q.off_tfirst = o; o = wt_align16(o + WT_WALK_TF * 2);
        q.off_shared = o; o = wt_align16(o + (int) sizeof(WtShared));
        q.lds_bytes = o;
        if (q.lds_bytes > hard_limit - 1024) continue;
        if (cd.half_cu && q.lds_bytes > hard_limit / 2 - 512) continue;
        // slots per position: 2.5 x the expected events (a Poisson tail of 2e-4 per position at 6 events), 8 .. 64
        int capp = 16;
        if (events_per_bp > 0) {
            capp = 8;
            while (capp < 64 && capp < 2.5 * events_per_bp) capp <<= 1;
        }
        if (const char *eC = getenv("WTAMD_WALK_CAPP")) { const int c = atoi(eC); if (c >= 1 && c <= 64 && !(c & (c - 1))) capp = c; }
        long long ov = 2048;
        if (const char *eO = getenv("WTAMD_WALK_OV")) { const long long c = atoll(eO); if (c >= 0) ov = c; }
        q.walk_capp = capp;
        q.walk_ov = (int) ov;
        long long bytes = (long long) q.W * capp * 8 + (ov > 8 ? ov : 8) * 12;      // (the fixed fetch may read 8 events past the slots)
        const long long lane_max = 2ll * n_tracks * S * 8;      // (the events one stretch can hold)
        if (bytes < lane_max) bytes = lane_max;
        bytes = (bytes + 255) & ~255ll;
        q.walk_off_at = (int) bytes;
        q.scratch_slab = (bytes + (long long) (q.W + 1) * 4 + 255) & ~255ll;
        p = q;
        return true;
    }
    return false;
}

// Chooses (positions per lane, T, W = ppt*T): the widest window whose bitmaps (and scratch
// columns) fit one workgroup's LDS, 4 positions per lane when possible (instruction-level
// parallelism, shared bitmap reads).  Environment overrides (experiments only): WTAMD_PPT, WTAMD_T.
// Register-column slots for median / MWU (0: not applicable): float tracks with float-exact defaults
// (scratch_f32), at most 128 tracks -- MWU: at most 64 per set.
static inline int wt_regcol_slots(int n_tracks, int op, bool scratch_f32, int n_set0) {
    if (!scratch_f32 || getenv("WTAMD_NO_REGCOL")) return 0;
    if (getenv("WTAMD_CHUNK") || getenv("WTAMD_GLOBAL_SCRATCH") || getenv("WTAMD_PPT") || getenv("WTAMD_T") || getenv("WTAMD_MWU_LPP1"))
        return 0;       // experiments / tests that force one of the column plans
    int need = 0;
    if (op == WT_OP_MEDIAN) need = n_tracks;
    else if (op == WT_OP_MWU) need = 2 * std::max(n_set0, n_tracks - n_set0);
    else return 0;
    if (op == WT_OP_MWU && (n_set0 < 1 || n_set0 >= n_tracks)) return 0;
    for (int nr : {32, 64, 128})
        if (need <= nr) return nr;
    return 0;
}

static inline bool wt_make_plan(int n_tracks, int op, bool scratch_f32, WtPlan &out, std::string &err,
                                int soft_limit = 80 * 1024, int hard_limit = 160 * 1024, int n_set0 = 0) {
    const char *eP = getenv("WTAMD_PPT");
    if (const int nr = wt_regcol_slots(n_tracks, op, scratch_f32, n_set0)) {
        // 256 lanes, two positions per lane evaluated one after the other (window 512 bp): LDS holds the
        // bitmaps only (+ MWU's sorted set 0, one column per lane).  Measured on chromosome-sized
        // items (MI355X, 100 tracks): median 13.1 ms per 31 Mbp with 2 positions per lane, 44 ms with 4
        // (and 19 vs 18 ms on cache-resident 0.4 Mbp probes, which hide it), 30.4 ms with LDS columns.
        WtPlan p;
        const int kk = 2;
        wt_carve(n_tracks, op, 256 * kk, 256, 4, p, 0, false, nr, n_set0);
        p.ppt = kk;
        p.lanes_per_pos = 1;
        if (p.n_chunks == 1 && p.lds_bytes <= hard_limit / 2 - 512) { out = p; return true; }
    }
    const char *eT = getenv("WTAMD_T");
    const bool scr = wt_op_needs_scratch(op);
    const int scratch_elem = scr ? (scratch_f32 ? 4 : 8) : 0;
    struct Cand { int ppt, T, lpp; };
    std::vector<Cand> cands;
    if (eP || eT) {
        const int ppt = scr ? 1 : (eP ? atoi(eP) : 4);
        const int T = eT ? atoi(eT) : 256;
        if ((ppt == 1 || ppt == 4) && T >= 64 && T <= 512 && !(T & (T - 1))) cands.push_back({ppt, T, 1});
    }
    if (cands.empty()) {
        // MWU first tries two lanes per position (512 lanes on 256 positions): the value columns cap
        // the positions a CU can hold, the N^2 ranking is what needs the lanes
        if (op == WT_OP_MWU && !getenv("WTAMD_MWU_LPP1")) cands = {{1, 512, 2}, {1, 256, 2}, {1, 128, 2}, {1, 64, 1}};
        else if (scr) cands = {{1, 256, 1}, {1, 128, 1}, {1, 64, 1}};
        else cands = {{4, 512, 1}, {4, 256, 1}, {1, 512, 1}, {1, 256, 1}, {1, 128, 1}, {1, 64, 1}};
    }
    // Measured on MI355X (round 1):
    //  * the widest window wins: per-window fixed costs (header, scans, look-back, barriers)
    //    outweigh everything else (mean/200 tracks: W=2048 7.4 ms vs W=1024 8.5 ms);
    //  * two workgroups per CU beat one: with more tracks than fit in HALF the CU's LDS the
    //    tracks are visited in chunks whose bitmaps are rebuilt per chunk and pass
    //    (sum/1000: 28 ms chunked at 79 KB vs 48 ms at 140 KB; var/500: 25 vs 40; mean/200: 5.8 vs 7.4);
    //  * scratch-column ops (median, MWU) keep every track resident in the widest window that
    //    fits (median/100: 51 vs 59 ms), chunking only when nothing else fits.
    const int limit = hard_limit - 1024;
    const int half = hard_limit / 2 - 512;
    (void) soft_limit;
    const char *eC = getenv("WTAMD_CHUNK");     // experiments / tests: force a chunk size
    auto try_plan = [&](const Cand &cd, int chunk, int lim, bool glob = false) -> bool {
        WtPlan p;
        wt_carve(n_tracks, scr ? op : WT_OP_SUM, cd.ppt * cd.T / cd.lpp, cd.T, scratch_elem, p, chunk, glob);
        p.ppt = cd.ppt;
        p.lanes_per_pos = cd.lpp;
        if (p.lds_bytes > lim) return false;
        out = p;
        return true;
    };
    auto max_chunk = [&](const Cand &cd, int lim, bool glob = false) -> int {
        int lo = 0, hi = n_tracks;          // largest chunk whose plan fits `lim` (lds grows with chunk)
        while (lo < hi) {
            const int mid = (lo + hi + 1) / 2;
            WtPlan p;
            wt_carve(n_tracks, scr ? op : WT_OP_SUM, cd.ppt * cd.T / cd.lpp, cd.T, scratch_elem, p, mid, glob);
            if (p.lds_bytes <= lim) lo = mid; else hi = mid - 1;
        }
        return lo;
    };
    const bool eG = scr && getenv("WTAMD_GLOBAL_SCRATCH");   // experiments / tests: force global columns
    if (eC || eG) {
        for (const Cand &cd : cands)
            if (try_plan(cd, eC ? atoi(eC) : 0, limit, eG)) return true;
    } else if (!scr) {
        const Cand &cd = cands[0];
        if (try_plan(cd, 0, half)) return true;
        const int best = max_chunk(cd, half);
        if (best >= 16) {
            const int n_chunks = (n_tracks + best - 1) / best;
            if (try_plan(cd, (n_tracks + n_chunks - 1) / n_chunks, half)) return true;
        }
        for (const Cand &c2 : cands)
            if (try_plan(c2, 0, limit)) return true;
    } else {
        // MWU with few tracks: the columns of 256 positions fit half the LDS, two workgroups per CU
        // already fill the SIMDs -- one lane per position (mwu/20 tracks: 4.4 vs 9.1 ms)
        if (op == WT_OP_MWU && cands.size() > 1 && try_plan({1, 256, 1}, 0, half)) return true;
        for (const Cand &cd : cands)
            if (try_plan(cd, 0, limit)) return true;
        // more tracks than LDS columns hold: the columns move to a global slab per workgroup
        // (coalesced: element i of lane l at [i][l]); the bitmaps are chunked to half the LDS
        for (const Cand &cd : cands) {
            const int best = max_chunk(cd, half, true);
            if (best < 1) continue;
            const int n_chunks = (n_tracks + best - 1) / best;
            if (try_plan(cd, (n_tracks + n_chunks - 1) / n_chunks, half, true)) return true;
        }
    }
    err = "no LDS plan fits " + std::to_string(n_tracks) + " tracks (op " + std::to_string(op) + ")";
    return false;
}

// MWUReduction's last step as a table (WtParams::mwu_table): entry k = the reference's value for |U1 - mu| = k / 2, computed
// with the reference's own expression (setComparisons.c:361-366, mu and sigma from the constructor's C integer divisions,
// :386-387) and THIS host's erf; the table ends where erf has reached -1 exactly (every larger k reads the last entry).
// The constructor's products are C ints in the reference (n1 * n2 * (n1 + n2 + 1) overflows from ~1300 tracks per set on: undefined there);
// here they wrap as two's-complement integers do, so host, emulator and oracle agree with each other whatever the sizes.
static inline int wt_mwu_wrap_mul(int a, int b) { return (int) ((unsigned) a * (unsigned) b); }
// Returns false when no table stands in for erf (the cap was reached before erf got to -1: sets so large that sigma is huge): the
// caller leaves mwu_table NULL and the kernel calls erf itself.  sigma NaN (a product that wrapped negative) or 0: short tables
// with the reference's own NaN / -2 results.
static inline bool wt_mwu_make_table(int n1, int n2, std::vector<double> &t) {
    const double mu = (double) (wt_mwu_wrap_mul(n1, n2) / 2);
    const double sigma = sqrt((double) (wt_mwu_wrap_mul(wt_mwu_wrap_mul(n1, n2), (int) ((unsigned) n1 + (unsigned) n2 + 1u)) / 12));
    t.clear();
    if (sigma != sigma) { t.push_back(sigma); return true; }        // every run NaN: erf(x / NaN)
    const int cap = 1 << 20;
    for (int k = 0; k < cap; k++) {
        const double U1 = mu + 0.5 * (double) k;            // (U1 > mu: the first branch; U1 < mu gives the same argument)
        const double v = k == 0 ? 2 * erf((U1 - mu) / sigma) : 2 * erf((mu - U1) / sigma);
        t.push_back(v);
        if (v == -2.0) return true;                         // (sigma == 0: entry 0 is NaN -- 0 / 0 -- and entry 1 already -2)
    }
    t.clear();
    return false;
}

static inline void wt_plan_to_params(const WtPlan &p, WtParams &P) {
    P.W = p.W; P.n_words = p.n_words; P.spitch = p.spitch; P.cpitch = p.cpitch; P.count_segs = p.count_segs;
    P.chunk_tracks = p.chunk_tracks; P.n_chunks = p.n_chunks;
    P.g_scratch = nullptr; P.g_scratch_slab = p.scratch_slab; P.g_attr_slab = p.attr_slab;
    P.logW = 0;
    while ((1 << P.logW) < p.W) P.logW++;
    P.off_S = p.off_S; P.off_cnt = p.off_cnt; P.off_segtot = p.off_segtot; P.off_U = p.off_U; P.off_cover = p.off_cover; P.off_E = p.off_E;
    P.off_epfx = p.off_epfx; P.off_nextw = p.off_nextw; P.off_gbase = p.off_gbase; P.off_scratch = p.off_scratch;
    P.off_dflt32 = p.off_dflt32;
    P.off_acc = p.off_acc; P.off_ev = p.off_ev; P.off_ltv = p.off_ltv; P.off_ltc = p.off_ltc;
    P.off_gtv = p.off_gtv; P.off_gtc = p.off_gtc; P.off_tbase = p.off_tbase; P.off_tpfx = p.off_tpfx; P.off_tfirst = p.off_tfirst; P.off_dsh = p.off_dsh; P.off_tdef = p.off_tdef;
    P.delta_df = 0; P.def_emin = 255; P.def_emax = 0;
    P.off_qa = p.off_qa; P.off_ltq = p.off_ltq; P.off_gtq = p.off_gtq; P.delta_q = p.delta_q; P.delta_ns = p.delta_ns;
    P.off_wcol = p.off_wcol; P.off_wcnt = p.off_wcnt; P.off_woff = p.off_woff; P.off_wtot = p.off_wtot; P.off_wbase = p.off_wbase;
    P.off_wgt = p.off_wgt; P.off_wncov = p.off_wncov; P.off_wfe = p.off_wfe; P.off_wdk = p.off_wdk; P.walk_S = p.walk_S;
    P.off_wguess = p.off_wguess; P.walk_capp = p.walk_capp; P.walk_ov = p.walk_ov; P.walk_off_at = p.walk_off_at; P.walk_pair = p.walk_pair;
    P.walk_mwu = p.walk_mwu;
    P.off_shared = p.off_shared; P.lds_bytes = p.lds_bytes;
}

struct WtWindowTables {
    std::vector<int32_t> cbase, c_nwin, c_hi, win_chrom;
    std::vector<int64_t> c_first_win;   // n_chrom + 1
    std::vector<int> empty_chrom;       // chromosomes whose range excludes all data
    int64_t n_windows = 0;
    int64_t n_rows = 0;                 // n_windows + n_chrom
    int64_t span_bp = 0;                // sum over chromosomes of (last finish - first start)
};

// first_start / last_finish: per (chrom, track) segment, only meaningful when the
// segment is non-empty (seg_off[s+1] > seg_off[s]).
static inline void wt_make_windows(int n_chrom, int n_tracks, const int64_t *seg_off,
                                   const int32_t *first_start, const int32_t *last_finish, int W,
                                   WtWindowTables &t, const int32_t *range_lo = nullptr,
                                   const int32_t *range_hi = nullptr) {
    t.cbase.assign(n_chrom, 0);
    t.c_nwin.assign(n_chrom, 1);
    t.c_hi.assign(n_chrom, INT32_MAX);
    t.c_first_win.assign(n_chrom + 1, 0);
    t.win_chrom.clear();
    t.empty_chrom.clear();
    t.span_bp = 0;
    for (int c = 0; c < n_chrom; c++) {
        int64_t lo = INT64_MAX, hi = INT64_MIN;
        for (int i = 0; i < n_tracks; i++) {
            const int64_t s = (int64_t) c * n_tracks + i;
            if (seg_off[s + 1] > seg_off[s]) {
                lo = std::min<int64_t>(lo, first_start[s]);
                hi = std::max<int64_t>(hi, last_finish[s]);
            }
        }
        int64_t nw = 1;
        if (lo <= hi) {
            // optional run-start range [range_lo, range_hi): the window grid starts at range_lo
            // and stops at range_hi (a multiple of every window width away, see the header)
            const int64_t data_hi = hi;
            if (range_lo && range_lo[c] > lo) lo = range_lo[c];
            if (range_hi && range_hi[c] != INT32_MAX && range_hi[c] < hi) hi = range_hi[c];
            if (lo < hi) {
                t.cbase[c] = (int32_t) lo;
                nw = std::max<int64_t>(1, (hi - lo + W - 1) / W);
                t.span_bp += hi - lo;
            } else {
                // the range excludes every run start: park one window past the data (emits nothing)
                t.cbase[c] = (int32_t) std::min<int64_t>(data_hi, INT32_MAX - 65536);
                t.empty_chrom.push_back(c);
            }
        }
        t.c_nwin[c] = (int32_t) nw;
        if (range_hi) t.c_hi[c] = range_hi[c];
        t.c_first_win[c + 1] = t.c_first_win[c] + nw;
        for (int64_t m = 0; m < nw; m++) t.win_chrom.push_back(c);
    }
    t.n_windows = t.c_first_win[n_chrom];
    t.n_rows = t.n_windows + n_chrom;
}

// Template dispatch over (op, value type, scratch type).  F must provide
//   template <int OP, class ValT, class ScrT, int K, bool MULTI> void run();
// MULTI: the tracks are visited in more than one chunk (bitmaps rebuilt per chunk and pass).
// Streaming ops ignore ScrT (ScrT = ValT keeps the instantiation count down).
template <int OP, int K, class F>
static inline void wt_dispatch_types2(bool value_f64, bool scratch_f32, bool multi, F &f) {
    // ScrT == float <=> float tracks whose defaults are float-exact (f32 select / f32 scratch)
    if (multi) {
        if (value_f64) f.template run<OP, double, double, K, true>();
        else if (!scratch_f32) f.template run<OP, float, double, K, true>();
        else f.template run<OP, float, float, K, true>();
    } else {
        if (value_f64) f.template run<OP, double, double, K, false>();
        else if (!scratch_f32) f.template run<OP, float, double, K, false>();
        else f.template run<OP, float, float, K, false>();
    }
}

template <int OP, class F>
static inline void wt_dispatch_types(bool value_f64, bool scratch_f32, int ppt, bool multi, F &f) {
    if (wt_op_needs_scratch(OP) || ppt == 1) wt_dispatch_types2<OP, 1>(value_f64, scratch_f32, multi, f);
    else wt_dispatch_types2<OP, 4>(value_f64, scratch_f32, multi, f);
}

template <class F>
static inline bool wt_dispatch(int op, bool value_f64, bool scratch_f32, int ppt, bool multi, F &f, int regcol = 0) {
    if (regcol && (op == WT_OP_MEDIAN || op == WT_OP_MWU) && !value_f64 && scratch_f32 && !multi) {
        if (op == WT_OP_MEDIAN) {
            if (regcol == 32) f.template run<WT_OP_MEDIAN, float, float, 2, false, 32>();
            else if (regcol == 64) f.template run<WT_OP_MEDIAN, float, float, 2, false, 64>();
            else f.template run<WT_OP_MEDIAN, float, float, 2, false, 128>();
        } else {
            if (regcol == 32) f.template run<WT_OP_MWU, float, float, 2, false, 32>();
            else if (regcol == 64) f.template run<WT_OP_MWU, float, float, 2, false, 64>();
            else f.template run<WT_OP_MWU, float, float, 2, false, 128>();
        }
        return true;
    }
    switch (op) {
    case WT_OP_SUM: wt_dispatch_types<WT_OP_SUM>(value_f64, scratch_f32, ppt, multi, f); return true;
    case WT_OP_PRODUCT: wt_dispatch_types<WT_OP_PRODUCT>(value_f64, scratch_f32, ppt, multi, f); return true;
    case WT_OP_MEAN: wt_dispatch_types<WT_OP_MEAN>(value_f64, scratch_f32, ppt, multi, f); return true;
    case WT_OP_VAR: wt_dispatch_types<WT_OP_VAR>(value_f64, scratch_f32, ppt, multi, f); return true;
    case WT_OP_STDDEV: case WT_OP_ENTROPY:   // reference reducers.c:665: entropy runs the stddev pop
        wt_dispatch_types<WT_OP_STDDEV>(value_f64, scratch_f32, ppt, multi, f); return true;
    case WT_OP_CV: wt_dispatch_types<WT_OP_CV>(value_f64, scratch_f32, ppt, multi, f); return true;
    case WT_OP_MIN: wt_dispatch_types<WT_OP_MIN>(value_f64, scratch_f32, ppt, multi, f); return true;
    case WT_OP_MAX: wt_dispatch_types<WT_OP_MAX>(value_f64, scratch_f32, ppt, multi, f); return true;
    case WT_OP_MEDIAN: wt_dispatch_types<WT_OP_MEDIAN>(value_f64, scratch_f32, ppt, multi, f); return true;
    case WT_OP_TTEST: wt_dispatch_types<WT_OP_TTEST>(value_f64, scratch_f32, ppt, multi, f); return true;
    case WT_OP_MWU: wt_dispatch_types<WT_OP_MWU>(value_f64, scratch_f32, ppt, multi, f); return true;
    case WT_OP_MULTIPLEX: wt_dispatch_types<WT_OP_MULTIPLEX>(value_f64, scratch_f32, ppt, multi, f); return true;
    default: return false;
    }
}

// True iff every default survives a round trip through float (then float
// tracks may keep their median / MWU scratch column in 4-byte elements).
static inline bool wt_defaults_fit_f32(const double *d, int n) {
    for (int i = 0; i < n; i++) {
        if (d[i] != d[i]) continue;
        if ((double) (float) d[i] != d[i]) return false;
    }
    return true;
}

#endif  // WT_PLAN_H_
