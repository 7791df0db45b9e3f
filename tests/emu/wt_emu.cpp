// wt_emu.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Phase-by-phase CPU emulation of the HIP kernels in
// wiggletools_amd/csrc/wt_core.h: every __syncthreads()-delimited phase is run
// for tid = 0..T-1 in turn, windows are executed in ticket order.  It exists so
// that the alignment / reducer / look-back logic can be checked against the
// oracle in the build container, which has no GPU.  Built by tests/emu/build.py
// into tests/emu/libwt_emu.so.
#define WT_EMU 1
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <thread>

#include "../../wiggletools_amd/csrc/wt_core.h"
#include "../../wiggletools_amd/csrc/wt_plan.h"

namespace {

struct EmuRun {
    WtParams P;
    WtPlan plan;
    std::vector<char> lds;
    int T_lanes = 0;
    long long n_redo = 0;
    // patch mode (wt_patch_kernel): only these windows, run offsets given instead of looked back
    struct PatchGroup { long long goff; std::vector<long long> wins; };
    bool patch = false;
    std::vector<PatchGroup> groups;

    template <int OP, class ValT, class ScrT, int K, bool MULTI, int NR = 0>
    void run() {
        WtCtx c;
        wt_ctx_init(c, P, lds.data());
        std::vector<char> slab((size_t) (P.g_scratch_slab + P.g_attr_slab));
        if (P.g_scratch_slab) c.scratch = slab.data();      // one emulated workgroup: one slab
        if (P.g_attr_slab) c.attr = slab.data() + P.g_scratch_slab;
        const int T = plan.T;
        if (NR > 0)
            for (int i = 0; i < P.n_tracks; i++) c.dflt32[i] = (float) P.defaults[i];
        std::vector<WtLane<K>> lanes(T);
        size_t gi = 0, wi = 0;
        long long patch_goff = 0;
        for (;;) {
            long long k;
            if (patch) {
                while (gi < groups.size() && wi >= groups[gi].wins.size()) { gi++; wi = 0; }
                if (gi >= groups.size()) break;
                if (wi == 0) patch_goff = groups[gi].goff;
                k = groups[gi].wins[wi++];
            } else {
                k = (long long) wt_glb_add64(&P.counters[WT_CTR_TICKET], 1ull);
                if (k >= P.n_windows) break;
            }
            constexpr bool two = (OP == WT_OP_TTEST || OP == WT_OP_MWU);
            constexpr int npass = wt_eval_passes(OP);
            const int N = P.n_tracks, NC = P.chunk_tracks;
            constexpr bool multi = MULTI;
            std::vector<WtAcc<K, NR>> acc(T);
            constexpr bool fuse = MULTI && OP != WT_OP_MULTIPLEX;      // as in wt_reduce_kernel
            wt_phase_header(P, c, k);
            for (int t = 0; t < T; t++) wt_phase_zero(P, c, true, t, T);
            for (int t = 0; t < T; t++) wt_eval_init<OP, K>(acc[t]);
            for (int ch = 0; ch < P.n_chunks; ch++) {
                const int t_lo = ch * NC, t_hi = std::min(N, t_lo + NC);
                if (ch > 0) for (int t = 0; t < T; t++) wt_phase_zero(P, c, false, t, T);
                for (int t = 0; t < T; t++) wt_phase_load<ValT>(P, c, t_lo, t_hi, true, t, T);
                for (int t = 0; t < T; t++) wt_phase_count_a(P, c, t_lo, t_hi, t, T);
                for (int t = 0; t < T; t++) wt_phase_count_b(P, c, t_lo, t_hi, t, T);
                if (fuse) for (int t = 0; t < T; t++) wt_phase_eval_chunk<OP, ValT, ScrT, K>(P, c, acc[t], 0, t_lo, t_hi, true, t, T);
            }
            if (fuse && npass == 2) for (int t = 0; t < T; t++) wt_eval_mid<OP, K>(P, acc[t]);
            for (int t = 0; t < T; t++) wt_phase_emask(P, c, two, t, T);
            for (int t = 0; t < T; t++) wt_phase_escan(P, c, t, T);
            if (patch) {                    // the difference-array kernel already placed this window's runs
                c.sh->n_emit = (int32_t) c.epfx[P.n_words];
                c.sh->goffset = patch_goff;
                patch_goff += c.sh->n_emit;
            } else {
                wt_phase_lookback(P, c, k);     // sequential emulation: the offset is known at once
            }
            for (int pass = fuse ? 1 : 0; pass < npass; pass++) {
                for (int ch = 0; ch < P.n_chunks; ch++) {
                    const int t_lo = ch * NC, t_hi = std::min(N, t_lo + NC);
                    if (multi) {
                        for (int t = 0; t < T; t++) wt_phase_zero(P, c, false, t, T);
                        for (int t = 0; t < T; t++) wt_phase_load<ValT>(P, c, t_lo, t_hi, false, t, T);
                        for (int t = 0; t < T; t++) wt_phase_count_a(P, c, t_lo, t_hi, t, T);
                        for (int t = 0; t < T; t++) wt_phase_count_b(P, c, t_lo, t_hi, t, T);
                    }
                    for (int t = 0; t < T; t++) wt_phase_eval_chunk<OP, ValT, ScrT, K>(P, c, acc[t], pass, t_lo, t_hi, false, t, T);
                }
                if (pass == 0 && npass == 2) for (int t = 0; t < T; t++) wt_eval_mid<OP, K>(P, acc[t]);
            }
            for (int t = 0; t < T; t++) wt_phase_eval_finish<OP, ValT, ScrT, K>(P, c, acc[t], lanes[t], t, T);
            if (OP == WT_OP_MWU && NR == 0) {
                for (int t = 0; t < T; t++) wt_phase_mwu_rank<ScrT>(P, c, t, T);
                for (int t = 0; t < T; t++) wt_phase_mwu_tail<K>(P, c, acc[t], lanes[t], t, T);
            }
            for (int t = 0; t < T; t++) wt_phase_write<OP, ValT, K>(P, c, lanes[t], t, T);
            if (!patch) wt_window_stats(P, c);
        }
    }

    // wt_walk_kernel, phase by phase
    void run_walk() {
        WtCtx c{};
        c.sh = (WtShared *) (lds.data() + P.off_shared);
        WtDeltaCtx d;
        wt_delta_ctx_init(d, P, lds.data());
        std::vector<char> slab((size_t) P.g_scratch_slab + 64);
        WtWalkCtx w;
        wt_walk_ctx_init(w, P, lds.data(), slab.data());
        const int T = plan.T;
        for (int t = 0; t < T; t++) wt_walk_defaults(P, w, t, T);
        std::vector<WtWalkLane> L(T);
        for (;;) {
            const long long k = (long long) wt_glb_add64(&P.counters[WT_CTR_TICKET], 1ull);
            if (k >= P.n_windows) break;
            wt_phase_header(P, c, k);
            for (int t = 0; t < T; t++) wt_walk_zero(P, c, w, t, T);
            for (int t = 0; t < T; t++) wt_delta_ranges1(P, c, d, 0, t, T);
            for (int t = 0; t < T; t++) wt_walk_ranges3(d, t, T);
            for (int t = 0; t < T; t++) wt_walk_pass<false>(P, c, w, d, 0u, 0u, t, T);
            for (int t = 0; t < T; t++) wt_walk_emits(P, c, w, L[t], t, T);
            for (int t = 0; t < T; t++) wt_walk_scan_a(w, wt_walk_emit_count(w, L[t], t), t, T);
            for (int t = 0; t < T; t++) wt_walk_scan_b(w, t, T);
            const unsigned long long mine = w.base[T];
            // pair mode: the two lanes of a stretch run side by side (they exchange values: wt_pair_xchg), one pair at a time
            auto walk_lanes = [&](int l0, int l1, uint32_t ev0, bool fixed) {
                if (!w.pair) {
                    for (int t = l0; t < l1; t++) { if (fixed) wt_walk_lane<true, false>(P, c, w, L[t], ev0, t, T); else wt_walk_lane<false, false>(P, c, w, L[t], ev0, t, T); }
                    return;
                }
                for (int q = l0; q < l1; q++) {
                    std::thread odd([&, q] { if (fixed) wt_walk_lane<true, true>(P, c, w, L[2 * q + 1], ev0, 2 * q + 1, T); else wt_walk_lane<false, true>(P, c, w, L[2 * q + 1], ev0, 2 * q + 1, T); });
                    if (fixed) wt_walk_lane<true, true>(P, c, w, L[2 * q], ev0, 2 * q, T); else wt_walk_lane<false, true>(P, c, w, L[2 * q], ev0, 2 * q, T);
                    odd.join();
                }
            };
            if (w.novf[0] <= w.ov_cap) {
                n_rounds++;
                walk_lanes(0, w.nstr, 0u, true);
            } else {
                n_fallback++;
                for (int t = 0; t < T; t++) wt_walk_offsets1(P, w, t, T);
                for (int t = 0; t < T; t++) wt_walk_scan_b(w, t, T);
                for (int t = 0; t < T; t++) wt_walk_offsets2(P, w, t, T);
                for (int l0 = 0; l0 < w.nstr;) {
                    const int l1 = wt_walk_round_end(w, l0, T);
                    const uint32_t ev0 = w.base[l0 << w.pair], ev1 = w.base[l1 << w.pair];
                    if (ev1 > ev0) {
                        n_rounds++;
                        for (int t = 0; t < T; t++) wt_walk_pass<true>(P, c, w, d, ev0, ev1, t, T);
                        walk_lanes(l0, l1, ev0, false);
                    }
                    l0 = l1;
                }
                for (int t = 0; t < T; t++) wt_walk_scan_a(w, wt_walk_emit_count(w, L[t], t), t, T);
                for (int t = 0; t < T; t++) wt_walk_scan_b(w, t, T);
            }
            if ((unsigned long long) w.base[T] != mine) { fprintf(stderr, "wtemu: walking kernel: run count changed\n"); abort(); }
            wt_phase_lookback(P, c, k, (unsigned long long) w.base[T]);
            for (int t = 0; t < T; t++) wt_walk_write(P, c, w, L[t], t, T);
            wt_window_stats(P, c);
        }
    }
    // wt_mwalk_kernel, phase by phase (the two lanes of a stretch side by side: they exchange values, wt_pair_xchg)
    void run_mwalk() {
        WtCtx c{};
        c.sh = (WtShared *) (lds.data() + P.off_shared);
        WtDeltaCtx d;
        wt_delta_ctx_init(d, P, lds.data());
        std::vector<char> slab((size_t) P.g_scratch_slab + 64);
        WtWalkCtx w;
        wt_walk_ctx_init(w, P, lds.data(), slab.data());
        const int T = plan.T;
        for (int t = 0; t < T; t++) wt_walk_defaults(P, w, t, T);
        std::vector<WtWalkLane> L(T);
        for (;;) {
            const long long k = (long long) wt_glb_add64(&P.counters[WT_CTR_TICKET], 1ull);
            if (k >= P.n_windows) break;
            wt_phase_header(P, c, k);
            for (int t = 0; t < T; t++) wt_walk_zero(P, c, w, t, T);
            for (int t = 0; t < T; t++) wt_delta_ranges1(P, c, d, 0, t, T);
            for (int t = 0; t < T; t++) wt_walk_ranges3(d, t, T);
            for (int t = 0; t < T; t++) wt_walk_pass<false>(P, c, w, d, 0u, 0u, t, T);
            for (int t = 0; t < T; t++) wt_mwalk_events(P, c, w, L[t], t, T);
            auto walk_lanes = [&](int l0, int l1, uint32_t ev0, bool fixed) {
                for (int q = l0; q < l1; q++) {
                    std::thread odd([&, q] { if (fixed) wt_mwalk_lane<true>(P, c, w, L[2 * q + 1], ev0, 2 * q + 1, T); else wt_mwalk_lane<false>(P, c, w, L[2 * q + 1], ev0, 2 * q + 1, T); });
                    if (fixed) wt_mwalk_lane<true>(P, c, w, L[2 * q], ev0, 2 * q, T); else wt_mwalk_lane<false>(P, c, w, L[2 * q], ev0, 2 * q, T);
                    odd.join();
                }
            };
            if (w.novf[0] <= w.ov_cap) {
                n_rounds++;
                walk_lanes(0, w.nstr, 0u, true);
            } else {
                n_fallback++;
                for (int t = 0; t < T; t++) wt_walk_offsets1(P, w, t, T);
                for (int t = 0; t < T; t++) wt_walk_scan_b(w, t, T);
                for (int t = 0; t < T; t++) wt_walk_offsets2(P, w, t, T);
                for (int l0 = 0; l0 < w.nstr;) {
                    const int l1 = wt_walk_round_end(w, l0, T);
                    const uint32_t ev0 = w.base[l0 << 1], ev1 = w.base[l1 << 1];
                    if (ev1 > ev0) {
                        n_rounds++;
                        for (int t = 0; t < T; t++) wt_walk_pass<true>(P, c, w, d, ev0, ev1, t, T);
                        walk_lanes(l0, l1, ev0, false);
                    }
                    l0 = l1;
                }
            }
            for (int t = 0; t < T; t++) wt_walk_scan_a(w, wt_walk_emit_count(w, L[t], t), t, T);
            for (int t = 0; t < T; t++) wt_walk_scan_b(w, t, T);
            wt_phase_lookback(P, c, k, (unsigned long long) w.base[T]);
            for (int t = 0; t < T; t++) wt_mwalk_write(P, c, w, L[t], t, T);
            wt_window_stats(P, c);
        }
    }
    long long n_rounds = 0, n_fallback = 0;

    template <int OP, bool DF = false>
    void run_delta() {
        constexpr bool TT = OP == WT_OP_TTEST;      // two sets per position (wt_delta_scan3_tt)
        constexpr bool MM = OP == WT_OP_MAX || OP == WT_OP_MIN;     // range updates of a segment tree (wt_delta_apply_mm)
        constexpr bool QQ = OP == WT_OP_VAR || OP == WT_OP_STDDEV || OP == WT_OP_ENTROPY || OP == WT_OP_CV || TT;
        WtCtx c;
        wt_ctx_init(c, P, lds.data());
        WtDeltaCtx d;
        wt_delta_ctx_init(d, P, lds.data());
        const int T = plan.T;
        std::vector<WtDeltaLane> dl(T);
        std::vector<WtDeltaLane2> dl2(T);
        std::vector<int32_t> wc_mm(T, 0);
        int guess = 0;      // the workgroup's unit exponent (0: none yet)
        std::vector<WtLane<WT_DELTA_K>> lanes(T);
        for (;;) {
            const long long k = (long long) wt_glb_add64(&P.counters[WT_CTR_TICKET], 1ull);
            if (k >= P.n_windows) break;
            wt_phase_header(P, c, k);
            if constexpr (MM) { for (int t = 0; t < T; t++) wt_delta_zero_mm<OP == WT_OP_MAX>(P, c, d, t, T); }
            else for (int t = 0; t < T; t++) wt_delta_zero<QQ, TT>(P, c, d, t, T);
            const int nchunks = (P.n_tracks + T - 1) / T;
            auto ranges = [&](int ch) {
                for (int t = 0; t < T; t++) wt_delta_ranges1(P, c, d, ch * T, t, T);
                for (int t = 0; t < T; t++) wt_delta_ranges2(P, c, d, t, T);
                for (int t = 0; t < T; t++) wt_delta_ranges3(P, c, d, t, T);
            };
            int scale = 1;
            if constexpr (MM) {
                for (int ch = 0; ch < nchunks; ch++) {
                    ranges(ch);
                    for (int t = 0; t < T; t++) wt_delta_pass_mm<OP == WT_OP_MAX>(P, c, d, t, T);
                }
                if (d.dsh->bad) wt_delta_mark_bad(P, c, k);
            } else if (guess == 0) {       // no unit exponent known yet: range pass, then the delta pass
                for (int ch = 0; ch < nchunks; ch++) {
                    ranges(ch);
                    for (int t = 0; t < T; t++) wt_delta_pass1(P, c, d, t, T);
                }
                const bool any = d.dsh->emin <= d.dsh->emax;
                const bool ok = wt_delta_verdict(P, d, scale);
                if (!ok) wt_delta_mark_bad(P, c, k);
                for (int ch = 0; ch < nchunks; ch++) {
                    if (nchunks > 1) ranges(ch);
                    for (int t = 0; t < T; t++) wt_delta_pass2<QQ, DF, TT>(P, c, d, scale, ok, false, true, t, T, std::min(T, P.n_tracks - ch * T), ch * T);
                }
                if (any && ok) guess = scale;
            } else {                // speculative single pass with the workgroup's unit
                for (int ch = 0; ch < nchunks; ch++) {
                    ranges(ch);
                    for (int t = 0; t < T; t++) wt_delta_pass2<QQ, DF, TT>(P, c, d, guess, true, true, true, t, T, std::min(T, P.n_tracks - ch * T), ch * T);
                }
                int lo; bool ok;
                scale = guess;
                if (!wt_delta_window_verdict(P, d, guess, lo, ok)) {
                    if (!ok) {
                        wt_delta_mark_bad(P, c, k);         // (values rewritten by the patch; the structure is in place)
                    } else {
                        n_redo++;
                        for (int t = 0; t < T; t++) wt_delta_rezero<QQ, TT>(P, c, d, t, T);
                        for (int ch = 0; ch < nchunks; ch++) {
                            if (nchunks > 1) ranges(ch);
                            for (int t = 0; t < T; t++) wt_delta_pass2<QQ, DF, TT>(P, c, d, lo, ok, false, false, t, T, std::min(T, P.n_tracks - ch * T), ch * T);
                        }
                        scale = lo;
                        guess = lo;
                    }
                }
            }
            const int TS = P.W / WT_DELTA_K;        // the scans' lanes (wt_delta_kernel: nts)
            if constexpr (TT) {
                for (int t = 0; t < 2 * TS; t++) wt_delta_scan1_tt(P, c, d, dl2[t], t, TS);
                for (int t = 0; t < 2 * TS; t++) wt_delta_scan2_tt(P, c, d, t, TS);
                for (int t = 0; t < 2 * TS; t++) wt_delta_scan3_tt(P, c, d, dl2[t], scale, t, TS);
                for (int t = 0; t < T; t++) wt_delta_combine_tt(P, c, d, t, T);
                for (int t = 0; t < T; t++) wt_delta_tail_tt(P, d, t, T);
                if (d.dsh->risk && c.sh->bad_slot < 0) wt_delta_mark_bad(P, c, k);
            } else if constexpr (MM) {
                for (int t = 0; t < TS; t++) wt_delta_scan1_mm(P, c, d, wc_mm[t], t, TS);
                for (int t = 0; t < TS; t++) wt_delta_scan2_mm(P, c, d, t, TS);
                for (int t = 0; t < TS; t++) wt_delta_scan3_mm<OP == WT_OP_MAX>(P, c, d, 0, lanes[t], t, TS);
            } else {
                for (int t = 0; t < TS; t++) wt_delta_scan1<QQ>(P, c, d, dl[t], t, TS);
                for (int t = 0; t < TS; t++) wt_delta_scan2<QQ>(P, c, d, t, TS);
                for (int t = 0; t < TS; t++) wt_delta_scan3<OP>(P, c, d, dl[t], lanes[t], scale, t, TS);
            }
            for (int t = 0; t < T; t++) wt_delta_nextw(P, c, t, T);
            for (int t = 0; t < T; t++) wt_phase_escan(P, c, t, T);
            wt_phase_lookback(P, c, k);
            for (int t = 0; t < WT_BAD_SUB; t++) wt_delta_note_offset(P, c, t);
            if constexpr (TT) for (int t = 0; t < TS; t++) wt_delta_load_res_tt(P, d, lanes[t], t);
            for (int t = 0; t < TS; t++) wt_delta_stage<OP>(P, c, d, lanes[t], t, TS);
            for (int t = 0; t < T; t++) wt_delta_copy_out(P, c, d, t, T);
            wt_window_stats(P, c);
        }
    }
};

}  // namespace

extern "C" {

// Returns number of runs (>= 0) or a negative error.  All pointers are host.
// info[0..3] receive W, T, lds_bytes, n_windows.
long long wtemu_reduce(int n_chrom, int n_tracks, const int64_t *seg_off, const int32_t *start,
                       const int32_t *finish, const void *value, int value_is_f64, const double *defaults,
                       int op, unsigned flags, int n_set0, long long capacity,
                       int32_t *o_start, int32_t *o_finish, double *o_value, int64_t *chrom_run_off,
                       double *o_tile, uint8_t *o_inplay, long long *info,
                       const int32_t *range_lo, const int32_t *range_hi) {
    const bool s32 = !value_is_f64 && wt_defaults_fit_f32(defaults, n_tracks);
    const int64_t n_seg = (int64_t) n_chrom * n_tracks;
    std::vector<int32_t> fs(n_seg, 0), lf(n_seg, 0);
    for (int64_t s = 0; s < n_seg; s++)
        if (seg_off[s + 1] > seg_off[s]) { fs[s] = start[seg_off[s]]; lf[s] = finish[seg_off[s + 1] - 1]; }
    const int64_t total = seg_off[n_seg];
    std::vector<unsigned long long> counters(WT_CTR_N, 0);
    long long used_delta = 0, delta_bad = 0, n_redo_total = 0, patched = 0, walk_used = 0, walk_rounds = 0, walk_fallback = 0;
    std::vector<int32_t> bad_list;
    std::vector<long long> bad_goff;
    WtWindowTables delta_tab;
    int delta_W = 0;
    std::vector<unsigned long long> delta_counters;
    // the engine's policy: exact difference-array path first when eligible, the general kernel
    // if any window had to give up
    for (int attempt = 0; attempt < 2; attempt++) {
        const bool delta = attempt == 0 && o_tile == nullptr && wt_delta_eligible(op, value_is_f64 != 0, n_tracks, defaults);
        if (attempt == 0 && !delta) continue;
        EmuRun R;
        std::string err;
        if (delta) wt_make_delta_plan_for(R.plan, n_tracks, op);
        else {
            // the engine's wt_pick_plan: the median over float tracks walks
            bool walk = false;
            if (op == WT_OP_MEDIAN && !value_is_f64 && !o_tile && !getenv("WTAMD_NO_WALK"))
                if (const int nr = wt_regcol_slots(n_tracks, op, s32, n_set0)) walk = wt_make_walk_plan(R.plan, n_tracks, nr, 0.0);
            // ... and so does MWU when asked to (wt_mwalk.h; WTAMD_MWALK=1: the engine's switch)
            if (op == WT_OP_MWU && !value_is_f64 && !o_tile && getenv("WTAMD_MWALK") && atoi(getenv("WTAMD_MWALK")) != 0 && !getenv("WTAMD_NO_WALK"))
                if (const int nr = wt_regcol_slots(n_tracks, op, s32, n_set0)) walk = wt_make_walk_plan(R.plan, n_tracks, nr, 0.0, 160 * 1024, n_set0);
            if (!walk && !wt_make_plan(n_tracks, op, s32, R.plan, err, 80 * 1024, 160 * 1024, n_set0)) { fprintf(stderr, "wtemu: %s\n", err.c_str()); return -10; }
        }
        WtWindowTables tab;
        wt_make_windows(n_chrom, n_tracks, seg_off, fs.data(), lf.data(), R.plan.W, tab, range_lo, range_hi);
        std::vector<uint32_t> widx((size_t) tab.n_rows * n_tracks, 0);
        std::vector<unsigned long long> status(tab.n_windows, 0);
        counters.assign(WT_CTR_N, 0);
        if (delta) { bad_list.assign(tab.n_windows + 1, 0); bad_goff.assign((tab.n_windows + 1) * WT_BAD_SUB, 0); }

        WtParams &P = R.P;
        memset(&P, 0, sizeof(P));
        P.start = start; P.finish = finish; P.value = value; P.seg_off = seg_off; P.defaults = defaults;
        P.n_chrom = n_chrom; P.n_tracks = n_tracks;
        P.cbase = tab.cbase.data(); P.c_nwin = tab.c_nwin.data(); P.c_hi = tab.c_hi.data(); P.c_first_win = tab.c_first_win.data();
        P.n_windows = tab.n_windows; P.win_chrom = tab.win_chrom.data(); P.widx = widx.data();
        P.op = op; P.flags = flags; P.n_set0 = n_set0;
        P.status = status.data(); P.counters = counters.data();
        P.capacity = capacity; P.o_start = o_start; P.o_finish = o_finish; P.o_value = o_value;
        P.chrom_run_off = chrom_run_off; P.o_tile = o_tile; P.o_inplay = o_inplay;
        wt_plan_to_params(R.plan, P);
        std::vector<double> mwu_table;
        if (op == WT_OP_MWU && !getenv("WTAMD_MWU_DEVICE_ERF")) {
            if (wt_mwu_make_table(n_set0, n_tracks - n_set0, mwu_table)) { P.mwu_table = mwu_table.data(); P.mwu_kmax = (int) mwu_table.size() - 1; }
        }
        if (delta) { P.bad_list = bad_list.data(); P.bad_goff = bad_goff.data(); wt_delta_defaults_params(defaults, n_tracks, P); }
        // few inexact windows: the general kernel rewrites the values of just those (the engine's
        // wt_patch_kernel); many: it redoes everything
        const bool patching = !delta && attempt == 1 && delta_bad > 0 && delta_bad * 4 <= (long long) delta_tab.n_windows &&
                              delta_W >= R.plan.W && delta_W % R.plan.W == 0 && delta_W / R.plan.W <= WT_BAD_SUB;
        if (patching) {
            const int ratio = delta_W / R.plan.W;
            R.patch = true;
            for (long long j = 0; j < delta_bad; j++) {
                const long long kd = bad_list[j];
                const int ch = delta_tab.win_chrom[kd];
                const long long m = kd - delta_tab.c_first_win[ch];
                // (as the patch kernel: every narrower window is an item of its own, at the offset the difference-array
                //  kernel recorded for its sub-range)
                for (int h = 0; h < ratio; h++) {
                    const long long mg = m * ratio + h;
                    if (mg >= tab.c_nwin[ch]) continue;
                    EmuRun::PatchGroup g;
                    g.goff = bad_goff[j * WT_BAD_SUB + h * (WT_BAD_SUB / ratio)];
                    g.wins.push_back(tab.c_first_win[ch] + mg);
                    R.groups.push_back(g);
                }
            }
            patched = delta_bad;
        }

        // window index "kernel"
        P.n_total = total;
        if (total > 0) {
            WtIndexCursor cur;
            wt_index_cursor_set(P, cur, wt_index_find_segment(P, 0));
            for (int64_t g = 0; g < total; g++) wt_index_apply(P, cur, g, finish[g], g > 0 ? finish[g - 1] : 0);
        }

        R.lds.assign((size_t) R.plan.lds_bytes + 64, 0);
        if (total > 0) {
            if (delta) {
                switch (op) {
                case WT_OP_SUM: if (P.delta_df) R.run_delta<WT_OP_SUM, true>(); else R.run_delta<WT_OP_SUM>(); break;
                case WT_OP_MEAN: if (P.delta_df) R.run_delta<WT_OP_MEAN, true>(); else R.run_delta<WT_OP_MEAN>(); break;
                case WT_OP_VAR: R.run_delta<WT_OP_VAR>(); break;
                case WT_OP_CV: R.run_delta<WT_OP_CV>(); break;
                case WT_OP_TTEST: R.run_delta<WT_OP_TTEST>(); break;
                case WT_OP_MAX: R.run_delta<WT_OP_MAX>(); break;
                case WT_OP_MIN: R.run_delta<WT_OP_MIN>(); break;
                default: R.run_delta<WT_OP_STDDEV>(); break;
                }
            } else if (R.plan.walk_S && R.plan.walk_mwu) {
                R.run_mwalk();
            } else if (R.plan.walk_S) {
                R.run_walk();
            } else if (!wt_dispatch(op, value_is_f64 != 0, s32, R.plan.ppt, R.plan.n_chunks > 1 || R.plan.scratch_slab > 0, R, R.plan.regcol)) {
                return -11;
            }
        }
        if (R.plan.walk_S) { walk_used = 1; walk_rounds = R.n_rounds; walk_fallback = R.n_fallback; }
        if (info && !patching) { info[0] = R.plan.W; info[1] = R.plan.T; info[2] = R.plan.lds_bytes; info[3] = tab.n_windows; info[6] = R.plan.n_chunks; info[7] = R.plan.scratch_slab;
                    info[4] = (long long) counters[WT_CTR_BP]; info[5] = (long long) counters[WT_CTR_INTERVALS]; }
        if (delta) {
            n_redo_total = R.n_redo;
            delta_bad = (long long) counters[WT_CTR_DELTA_BAD];
            used_delta = delta_bad == 0;
            if (used_delta) break;
            delta_tab = tab;
            delta_W = R.plan.W;
            delta_counters = counters;
        } else if (patching) {
            counters = delta_counters;          // run count, covered bp, ... are the difference-array launch's
            used_delta = 1;
        }
    }
    if (info) { info[8] = used_delta; info[9] = delta_bad; info[10] = n_redo_total; info[11] = patched; }
    if (info) { info[12] = walk_used; info[13] = walk_rounds; info[14] = walk_fallback; }
    if (counters[WT_CTR_ERROR] & WT_ERR_CAPACITY) return -1;
    if (counters[WT_CTR_ERROR]) return -2;
    return (long long) counters[WT_CTR_RUNS];
}

// the device's form of the t-test's tail (wt_core.h wt_tdist_2Q_fast: the device reducers' wt_ttest_tail) and the oracle's form
// beside it, for tests/test_tdist_fast.py
double wtemu_tdist_2q_fast(double t, double nu) { return wt_tdist_2Q_fast(t, nu); }
double wtemu_tdist_2q(double t, double nu) { return 2 * wt_tdist_Q(t, nu); }
double wtemu_lgamma_half_diff(double a) { return wt_lgamma_half_diff(a); }
// wt_div_n (csrc/wt_delta.h) against the division it stands in for: `cases` quotients per count n in [n_lo, n_hi], chosen to sit on and beside
// the doubles' rounding boundaries (s = RN(n (m + h ulp)) for h in {0, 1/2} -+ a few ulp) and at random; returns the mismatches
long long wtemu_div_n_mismatches(int n_lo, int n_hi, int cases, unsigned long long seed) {
    long long bad = 0;
    unsigned long long x = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (int n = n_lo; n <= n_hi; n++) {
        const double dn = (double) n, y = 1.0 / dn;
        for (int c = 0; c < cases; c++) {
            const unsigned long long r = rnd();
            const unsigned long long mant = (r & ((1ull << 52) - 1)) | (c % 7 == 0 ? ((1ull << 52) - 1) & ~((r >> 52) & 0xff) : 0ull) ;
            const int e = (int) (rnd() % 300) - 150;
            double m = ldexp(1.0 + (double) (mant & ((1ull << 52) - 1)) * 0x1p-52, e);
            double s;
            switch (c % 4) {
            case 0: s = m; break;                                                   // any double
            case 1: s = m * dn; break;                                              // a quotient next to a double
            case 2: s = (m + ldexp(1.0, e - 53)) * dn; break;                       // ... next to a midpoint
            default: s = (double) (long long) (rnd() >> 11) * ldexp(1.0, e - 60); break;    // an integer sum times a unit
            }
            for (int d = -2; d <= 2; d++) {
                double t = s;
                for (int q = 0; q < (d < 0 ? -d : d); q++) t = nextafter(t, d < 0 ? -INFINITY : INFINITY);
                for (int sg = 0; sg < 2; sg++) {
                    const double v = sg ? -t : t;
                    const double a = wt_div_n(v, dn, y), b = v / dn;
                    if (!(a == b) || std::signbit(a) != std::signbit(b)) bad++;
                }
            }
        }
    }
    return bad;
}

}  // extern "C"
