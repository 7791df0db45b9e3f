// wt_engine.hip -- gfx950 kernels + the bulk C ABI (wtamd_*) of
// include/wiggletools_amd.h.  Compiled only by hipcc --offload-arch=gfx950.
//
// Kernels
//   wt_index_kernel     window index (widx): contiguous span of input runs per block, 4 per lane
//   wt_reduce_kernel    persistent workgroups, one alignment window per ticket:
//                       bitmap multiplexer + per-run reducer + ordered output
//                       (logic in wt_core.h) -- every reducer, any track count
//   wt_delta_kernel     Sum / Mean over float tracks: exact difference array, O(input runs)
//                       (logic in wt_delta.h)
//   wt_patch_kernel     the bitmap multiplexer over just the windows wt_delta_kernel could not
//                       prove exact
//   wt_extents_kernel   first start / last finish per (chrom, track) segment
//   wt_validate_kernel  input contract (sorted, non-overlapping, positive length)
//   wt_auc_kernel       sum (finish-start)*value (+ span) over a run list (AUC, meanI)
//   wt_pearson_kernel   Pearson of two tracks over their Multiplexer tile
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <thread>
#include <atomic>
#include <sys/mman.h>
#include <unistd.h>
#include <cstring>
#include <map>
#include <tuple>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/wiggletools_amd.h"
#include "wt_core.h"
#include "wt_plan.h"
#include "wt_devscope.h"
#include "wt_bwdev_core.h"

#define WT_MAX_BLOCK 512
// minimum waves per SIMD the register allocator must leave room for (MI355X_MICROARCH:
// w = k*T/256).  Measured on MI355X: the K=4 kernels sit at 129 VGPRs unconstrained -- one
// register over the limit for two 512-lane workgroups per CU -- so they are held to 128
// (w = 4: 2.35 vs 3.14 ms on the bench kernel); the K=1 kernels fit anyway and schedule
// better unconstrained (var/500 tracks: 71 vs 93 ms).
#ifndef WT_MIN_WAVES
#define WT_MIN_WAVES(K) ((K) == 4 ? 4 : 3)
#endif

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
template <int OP, class ValT, class ScrT, int K, bool MULTI, int NR>
// (register columns, NR > 0: 256 lanes; the column + the exchange network's temporaries need more
//  than the 168 VGPRs three waves per SIMD leave -- with that bound the compiler spilled 250
//  registers into the middle of the network -- so NR = 128 runs two waves per SIMD, NR = 64 three)
__global__ void __launch_bounds__(NR > 0 ? 256 : WT_MAX_BLOCK, NR > 0 ? (NR > 64 ? 2 : (NR > 32 ? 3 : 4)) : WT_MIN_WAVES(K)) wt_reduce_kernel(const WtParams P) {
    extern __shared__ __attribute__((aligned(16))) char wt_lds[];
    WtCtx c;
    wt_ctx_init(c, P, wt_lds);
    // global slab of this workgroup: [value columns, if they do not fit LDS][MWU attributes]
    const size_t slab = (size_t) P.g_scratch_slab + (size_t) P.g_attr_slab;
    if (MULTI && (OP == WT_OP_MEDIAN || OP == WT_OP_MWU) && P.g_scratch_slab)
        c.scratch = P.g_scratch + (size_t) blockIdx.x * slab;
    if (OP == WT_OP_MWU) c.attr = P.g_scratch + (size_t) blockIdx.x * slab + (size_t) P.g_scratch_slab;
    WtLane<K> L;
    const int tid = threadIdx.x, nt = blockDim.x;
#ifdef WT_MARK_ONLY
#define WT_MARK(x) do { if ((x) == WT_MARK_ONLY && (tid & 63) == 0) { P.debug[2 + (tid >> 6)] = (unsigned long long) (x); __threadfence_system(); } } while (0)
#elif defined(WT_DEBUG_MARK)
#define WT_MARK(x) do { if ((tid & 63) == 0) { if (tid == 0) { P.debug[0] = (unsigned long long) (x); P.debug[1] = (unsigned long long) k_dbg; } P.debug[2 + (tid >> 6)] = (unsigned long long) (x); __threadfence_system(); } } while (0)
#else
#define WT_MARK(x) do { } while (0)
#endif
#ifdef WT_PROFILE
#define WT_TICK(slot) do { if (tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); \
        prof[slot] += t_ - t_last; t_last = t_; } } while (0)
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_readcyclecounter();
#else
#define WT_TICK(slot) do { } while (0)
#endif
    long long k_dbg = -1;
    (void) k_dbg;
    // Window tickets.  The lane-0 work at the end of one iteration (statistics) and at the start of
    // the next (ticket) must NOT be left adjacent across the loop back-edge: hipcc (ROCm 7.2) merges
    // the two `tid == 0` regions into a divergent exit of an inner loop, whose header -- including
    // its s_barrier -- the other 63 lanes of wave 0 then re-enter before lane 0 has fetched the next
    // ticket: every wave re-reads the stale ticket and the workgroup never terminates (observed on
    // MI355X; any instruction between the two regions hides it).  So the next ticket is taken in the
    // same lane-0 block as the statistics, followed by the barrier that publishes it.
    if (tid == 0) c.sh->ticket = (long long) wt_glb_add64(&P.counters[WT_CTR_TICKET], 1ull);
    if (NR > 0)
        for (int i = tid; i < P.n_tracks; i += nt) c.dflt32[i] = (float) P.defaults[i];
    __syncthreads();
    for (;;) {
        WT_MARK(1);
        const long long k = c.sh->ticket;
        k_dbg = k;
        WT_MARK(2);
        if (k >= P.n_windows) break;
        if (tid == 0) wt_phase_header(P, c, k);
        wt_phase_zero(P, c, true, tid, nt);
        __syncthreads();
        WT_TICK(0);
        // pass A: every track's breakpoints and coverage enter U / cover[]; with one chunk the
        // per-track bitmaps stay resident for the evaluation
        const int N = P.n_tracks, NC = MULTI ? P.chunk_tracks : N, n_chunks = MULTI ? P.n_chunks : 1;
        // Chunked tracks: every sweep over the chunks rebuilds their bitmaps, so the first
        // evaluation pass is fused into the sweep that builds U / the coverage summaries (one
        // sweep saved: sum-like ops 2 -> 1, var / stddev / CV 3 -> 2).  Not for the Multiplexer
        // tile, whose rows need the look-back offset first.
        constexpr bool FUSE = MULTI && OP != WT_OP_MULTIPLEX;
        constexpr int npass = wt_eval_passes(OP);
        WtAcc<K, NR> A;
        wt_eval_init<OP, K>(A);
        for (int ch = 0; ch < n_chunks; ch++) {
            const int t_lo = ch * NC, t_hi = (t_lo + NC < N) ? t_lo + NC : N;
            if (MULTI && ch > 0) {
                wt_phase_zero(P, c, false, tid, nt);
                __syncthreads();
            }
            WT_MARK(3);
            wt_phase_load<ValT>(P, c, t_lo, t_hi, true, tid, nt);
            __syncthreads();
            WT_TICK(1);
            WT_MARK(4);
            wt_phase_count_a(P, c, t_lo, t_hi, tid, nt);
            __syncthreads();
            WT_MARK(5);
            wt_phase_count_b(P, c, t_lo, t_hi, tid, nt);
            __syncthreads();
            WT_TICK(2);
            if (FUSE) {     // first evaluation pass rides on this sweep (every position: E is not known yet)
                wt_phase_eval_chunk<OP, ValT, ScrT, K>(P, c, A, 0, t_lo, t_hi, true, tid, nt);
                __syncthreads();
                WT_TICK(4);
            }
        }
        if (FUSE && npass == 2) wt_eval_mid<OP, K>(P, A);
        WT_MARK(6);
        wt_phase_emask(P, c, OP == WT_OP_TTEST || OP == WT_OP_MWU, tid, nt);
        __syncthreads();
        WT_MARK(7);
        wt_phase_escan(P, c, tid, nt);
        __syncthreads();
        WT_TICK(3);
        // the window's run count is known before the reducers run: publish it now, so that no
        // successor ever waits for our evaluation
        WT_MARK(8);
        if (tid == 0) wt_lookback_publish(P, c, k);
        if (OP == WT_OP_MULTIPLEX) {     // the tile rows are written by the evaluation: offset first
            if (tid < 64) wt_lookback_complete(P, c, k, tid);
            __syncthreads();
        }
        WT_MARK(9);
#pragma unroll
        for (int pass = FUSE ? 1 : 0; pass < npass; pass++) {
            for (int ch = 0; ch < n_chunks; ch++) {
                const int t_lo = ch * NC, t_hi = (t_lo + NC < N) ? t_lo + NC : N;
                if (MULTI) {
                    wt_phase_zero(P, c, false, tid, nt);
                    __syncthreads();
                    wt_phase_load<ValT>(P, c, t_lo, t_hi, false, tid, nt);
                    __syncthreads();
                    wt_phase_count_a(P, c, t_lo, t_hi, tid, nt);
                    __syncthreads();
                    wt_phase_count_b(P, c, t_lo, t_hi, tid, nt);
                    __syncthreads();
                }
                wt_phase_eval_chunk<OP, ValT, ScrT, K>(P, c, A, pass, t_lo, t_hi, false, tid, nt);
                if (MULTI) __syncthreads();     // the next chunk overwrites the bitmaps
            }
            if (pass == 0 && npass == 2) wt_eval_mid<OP, K>(P, A);
        }
        wt_phase_eval_finish<OP, ValT, ScrT, K>(P, c, A, L, tid, nt);
        if (OP == WT_OP_MWU && NR == 0) {      // the value columns are complete: rank with every lane, then the tie scan
            __syncthreads();
            wt_phase_mwu_rank<ScrT>(P, c, tid, nt);
            __syncthreads();
            wt_phase_mwu_tail<K>(P, c, A, L, tid, nt);
        }
        WT_TICK(4);
        WT_MARK(10);
        if (OP != WT_OP_MULTIPLEX && tid < 64) wt_lookback_complete(P, c, k, tid);
        __syncthreads();
        WT_TICK(5);
        WT_MARK(11);
        wt_phase_write<OP, ValT, K>(P, c, L, tid, nt);
        __syncthreads();
        WT_MARK(12);
        if (tid == 0) {
            wt_window_stats(P, c);
            c.sh->ticket = (long long) wt_glb_add64(&P.counters[WT_CTR_TICKET], 1ull);
        }
        __syncthreads();
        WT_TICK(6);
    }
#ifdef WT_PROFILE
    if (tid == 0)
        for (int q = 0; q < 8; q++) wt_glb_add64(&P.counters[WT_CTR_PROF + q], prof[q]);
#endif
}

// Exact difference-array path for Sum / Mean over float tracks (wt_delta.h): O(intervals) work
// instead of O(tracks x runs); LDS independent of the track count.
// (var / stddev / CV also accumulate the sum of squares: 512 lanes, one workgroup per CU)
#define WT_DELTA_SQ(OP) ((OP) == WT_OP_VAR || (OP) == WT_OP_STDDEV || (OP) == WT_OP_ENTROPY || (OP) == WT_OP_CV || (OP) == WT_OP_TTEST)
#ifndef WT_DELTA_MIN_WAVES
#define WT_DELTA_MIN_WAVES 4     // waves per SIMD the register allocation aims at (experiments: 6 spills, see DESIGN A.1)
#endif
#ifndef WT_DELTA_SQ_BLOCK
#define WT_DELTA_SQ_BLOCK WT_DELTA_SQ_T0   // workgroup of the launches that also accumulate squares (768: three wavefronts per SIMD, 168 registers; the scans: the first 512 lanes, see wt_make_delta_plan)
#endif
#ifndef WT_DELTA_ZERO_EARLY
#define WT_DELTA_ZERO_EARLY 1      // the accumulators are zeroed beside lane 0's ticket + header chain (0: at the window's start, rounds 1-5): -2 % at every density
#endif
#ifndef WT_DELTA_EARLY_PUBLISH
#define WT_DELTA_EARLY_PUBLISH 1
#endif
#ifndef WT_DELTA_BLOCK
#define WT_DELTA_BLOCK 1024     // (launch bound; the plan's default, see wt_make_delta_plan)
#endif
// DF: some track's default is non-zero (Sum / Mean; P.delta_df)
// U: runs per lane and tile of the pass (round 6: 4, or 2 for launches whose windows hold few tiles per wavefront -- wt_launch_delta)
template <int OP, bool DF = false, int U = WT_DELTA_U>
__global__ void __launch_bounds__(WT_DELTA_SQ(OP) ? WT_DELTA_SQ_BLOCK : WT_DELTA_BLOCK, WT_DELTA_SQ(OP) ? 3 : WT_DELTA_MIN_WAVES) wt_delta_kernel(const WtParams P) {
    extern __shared__ __attribute__((aligned(16))) char wt_lds[];
    WtCtx c;
    wt_ctx_init(c, P, wt_lds);
    WtDeltaCtx d;
    wt_delta_ctx_init(d, P, wt_lds);
    constexpr bool QQ = WT_DELTA_SQ(OP);
    constexpr bool TT = OP == WT_OP_TTEST;      // two sets per position (wt_delta_scan3_tt)
    constexpr bool MM = OP == WT_OP_MAX || OP == WT_OP_MIN;     // range updates of a segment tree (wt_delta_apply_mm)
    constexpr bool EP = WT_DELTA_EARLY_PUBLISH && (OP == WT_OP_SUM || OP == WT_OP_MEAN);       // the run count is published before the values are computed (wt_delta_scan3_cov / _val)
    WtDeltaLane DL;
    WtDeltaLane2 DL2;
    (void) DL; (void) DL2;
    WtLane<WT_DELTA_K> L;
    uint32_t ep_rank = 0, ep_em = 0;        // EP: the lane's first rank among the window's emitted runs, its emitted byte
    (void) ep_rank; (void) ep_em;
    const int tid = threadIdx.x, nt = blockDim.x;
    // lanes of the scans and the staging (8 positions each): all of them -- or, with squares, the first 512 of 1024.  (Sum / Mean must not
    // see a run-time bound here: the guard alone cost wt_delta_kernel<mean> 31 more spilled registers and a quarter more HBM traffic.)
    const int nts = QQ ? P.W / WT_DELTA_K : nt;
#define WT_SCAN_LANE (!QQ || tid < nts)
    int guess = 0;              // the workgroup's unit exponent (0: none yet); uniform across the lanes
    long long k_dbg = -1;
    (void) k_dbg;
#ifdef WT_PROFILE
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_readcyclecounter();
#endif
    // ticket handling: see wt_reduce_kernel; lane 0 also prepares the next window's header there,
    // so that the first phase of a window needs no barrier of its own
    if (tid == 0) {
        const long long k0 = (long long) wt_glb_add64(&P.counters[WT_CTR_TICKET], 1ull);
        c.sh->ticket = k0;
        if (k0 < P.n_windows) wt_phase_header(P, c, k0);
    }
#if WT_DELTA_ZERO_EARLY
    // (experiment: the accumulators are zeroed beside lane 0's ticket + header chain, before the barrier that publishes the header)
    if constexpr (MM) wt_delta_zero_mm<OP == WT_OP_MAX>(P, c, d, tid, nt);
    else wt_delta_zero<QQ, TT>(P, c, d, tid, nt);
#endif
    __syncthreads();
    for (;;) {
        WT_MARK(101);
        const long long k = c.sh->ticket;
        k_dbg = k;
        if (k >= P.n_windows) break;
        const int nchunks = (P.n_tracks + nt - 1) / nt;
        auto ntr = [&](int ch) { const int r = P.n_tracks - ch * nt; return r < nt ? r : nt; };     // tracks of chunk ch
#if !WT_DELTA_ZERO_EARLY
        if constexpr (MM) wt_delta_zero_mm<OP == WT_OP_MAX>(P, c, d, tid, nt);
        else wt_delta_zero<QQ, TT>(P, c, d, tid, nt);
#endif
        WT_TICK(0);
        WT_MARK(102);
        int scale = 1;
        if constexpr (MM) {
            // one pass, no unit exponent: a float's order-preserving key needs none
            for (int ch = 0; ch < nchunks; ch++) {
                wt_delta_ranges_w1(P, c, d, ch * nt, tid, nt);
                __syncthreads();
                wt_delta_ranges_w2(P, c, d, tid, nt, 64u * U);
                __syncthreads();
                WT_TICK(1);
                wt_delta_pass_mm<OP == WT_OP_MAX>(P, c, d, tid, nt);
                __syncthreads();
                WT_TICK(3);
            }
            if (tid == 0 && d.dsh->bad) wt_delta_mark_bad(P, c, k);     // a NaN or a -0.0: the general kernel's window
        } else if (guess == 0) {
            // no unit exponent known to this workgroup yet: exponent-range pass, then the delta pass
            for (int ch = 0; ch < nchunks; ch++) {
                wt_delta_ranges_w1(P, c, d, ch * nt, tid, nt);
                __syncthreads();
                wt_delta_ranges_w2(P, c, d, tid, nt, 64u * U);
                __syncthreads();
                WT_TICK(1);
                wt_delta_pass1<U>(P, c, d, tid, nt);
                __syncthreads();
                WT_TICK(2);
            }
            const bool any = d.dsh->emin <= d.dsh->emax;
            const bool ok = wt_delta_verdict(P, d, scale);
            if (!ok && tid == 0) wt_delta_mark_bad(P, c, k);
            for (int ch = 0; ch < nchunks; ch++) {
                if (nchunks > 1) {
                    wt_delta_ranges_w1(P, c, d, ch * nt, tid, nt);
                    __syncthreads();
                    wt_delta_ranges_w2(P, c, d, tid, nt, 64u * U);
                    __syncthreads();
                }
                wt_delta_pass2<QQ, DF, TT, U>(P, c, d, scale, ok, false, true, tid, nt, ntr(ch), ch * nt);
                __syncthreads();
                WT_TICK(3);
            }
            if (any && ok) guess = scale;
        } else {
            // speculative single pass with the workgroup's unit (wt_delta_window_verdict)
            for (int ch = 0; ch < nchunks; ch++) {
                wt_delta_ranges_w1(P, c, d, ch * nt, tid, nt);
                __syncthreads();
                wt_delta_ranges_w2(P, c, d, tid, nt, 64u * U);
                __syncthreads();
                WT_TICK(1);
                wt_delta_pass2<QQ, DF, TT, U>(P, c, d, guess, true, true, true, tid, nt, ntr(ch), ch * nt);
                __syncthreads();
                WT_TICK(3);
            }
            int lo;
            bool ok;
            scale = guess;
            if (!wt_delta_window_verdict(P, d, guess, lo, ok)) {      // workgroup-uniform
              if (!ok) {
                // not provably exact whatever the unit: the patch kernel rewrites this window's values, and everything
                // else about it -- breakpoints, coverage, run count -- is in place after the speculative pass.  (It used
                // to be redone like the windows below: twice the time of a window, during which every later window sat
                // in its look-back; 5 % such windows cost the kernel a third more time, round 4.)
                if (tid == 0) wt_delta_mark_bad(P, c, k);
              } else {
                __syncthreads();            // every lane has read the verdict fields
                wt_delta_rezero<QQ, TT>(P, c, d, tid, nt);
                __syncthreads();
                for (int ch = 0; ch < nchunks; ch++) {
                    if (nchunks > 1) {
                        wt_delta_ranges_w1(P, c, d, ch * nt, tid, nt);
                        __syncthreads();
                        wt_delta_ranges_w2(P, c, d, tid, nt, 64u * U);
                        __syncthreads();
                    }
                    wt_delta_pass2<QQ, DF, TT, U>(P, c, d, lo, ok, false, false, tid, nt, ntr(ch), ch * nt);
                    __syncthreads();
                }
                scale = lo;
                guess = lo;
              }
            }
        }
        WT_MARK(105);
        if constexpr (TT) {
            // the scans split by set over 2 nts lanes, then one lane per position (wt_delta.h: phases A - C)
            if (tid < 2 * nts) wt_delta_scan_w1_tt(P, c, d, DL2, tid, nts);
            __syncthreads();
            WT_MARK(107);
            if (tid < 2 * nts) wt_delta_scan3_tt(P, c, d, DL2, scale, tid, nts);
            __syncthreads();
            wt_delta_combine_tt(P, c, d, tid, nt);
            __syncthreads();
            // a position whose variance cancels too much for the exact sums (wt_delta_scan3_tt): the window's values are the general kernel's
            if (tid == 0 && d.dsh->risk && c.sh->bad_slot < 0) wt_delta_mark_bad(P, c, k);
        } else if constexpr (MM) {
            int32_t wc_mm = 0;
            wt_delta_scan_w1_mm(P, c, d, wc_mm, tid, nt);
            __syncthreads();
            WT_MARK(107);
            wt_delta_scan3_mm<OP == WT_OP_MAX>(P, c, d, wc_mm, L, tid, nt);
            __syncthreads();
        } else if constexpr (EP) {
            // Sum / Mean: the bytes of the breakpoint / emitted bitmaps first ...
            wt_delta_scan_w1<QQ>(P, c, d, DL, tid, nts);
            __syncthreads();
            WT_MARK(107);
            ep_rank = wt_delta_scan3_cov(P, c, d, DL, ep_em, tid, nts);
            __syncthreads();
        } else {
            if (WT_SCAN_LANE) wt_delta_scan_w1<QQ>(P, c, d, DL, tid, nts);
            __syncthreads();
            WT_MARK(107);
            if (WT_SCAN_LANE) wt_delta_scan3<OP>(P, c, d, DL, L, scale, tid, nts);
            __syncthreads();
        }
        WT_TICK(4);
        WT_MARK(108);
        // wave 0: run-count scan and look-back back to back (it owns the counts); the last lanes
        // build the breakpoint jump table meanwhile
        unsigned long long mine = 0;
        if constexpr (EP) {
            // the wavefronts' run counts are in epfx[0 .. nwaves): this one's first rank, and -- wave 0 -- the window's count, published at once
            const int lane_ = tid & 63;
            if (tid < 64) {
                mine = wt_waves_before32(c.epfx, 0, nt >> 6, lane_);
                WT_TICK(5);
                if (tid == 0) wt_lookback_publish(P, c, k, mine);
            }
            ep_rank += wt_waves_before32(c.epfx, 0, tid >> 6, lane_);
            // ... the count is out; now the values (wt_delta_scan3_val: nobody waits for them but this window's own staging)
            wt_delta_scan3_val<OP>(P, c, d, DL, L, scale, tid, nts);
        } else if (tid < 64) {
            mine = wt_delta_escan_wave(P, c, tid);
            WT_TICK(5);
            if (tid == 0) wt_lookback_publish(P, c, k, mine);
        }
        wt_delta_nextw(P, c, tid, nt);
        __syncthreads();
        WT_MARK(110);
        // the look-back's round trips to the status words overlap the staging of the other waves
        // (and the predecessors get that much longer to publish)
        if (tid < 64) {
            wt_lookback_complete(P, c, k, tid, mine);
            if constexpr (!EP) wt_delta_note_offset(P, c, tid);     // (lane 0 set the offset in the look-back: same wave, LDS in order)
        }
        WT_TICK(6);
        if constexpr (TT) {
            // two-sample launches: the Student tail of every emitted position is what the look-back of wave 0 overlaps -- the other
            // wavefronts share the window's positions (the staging needs their results: one more barrier)
            if (tid >= 64) wt_delta_tail_tt(P, d, tid - 64, nt - 64);
            __syncthreads();
            WT_TICK(2);             // (profile builds: the tail, less the look-back, in the slot of the exponent-range pass)
            if (tid < nts) wt_delta_load_res_tt(P, d, L, tid);
        }
        if constexpr (EP) wt_delta_stage_ep<OP>(P, c, d, L, ep_em, DL.evmask, ep_rank, tid, nts);
        else if (WT_SCAN_LANE) wt_delta_stage<OP>(P, c, d, L, tid, nts);
        __syncthreads();
        if constexpr (EP) { if (tid < 64) wt_delta_note_offset_ep(P, c, tid); }
#ifdef WT_PROFILE_TAIL
        WT_TICK(2);                 // (experiment: the tail of a window apart -- staging here, copy-out in "write", ticket + header in "zero")
#endif
        wt_delta_copy_out(P, c, d, tid, nt);
        __syncthreads();
#ifdef WT_PROFILE_TAIL
        WT_TICK(7);
#endif
        WT_MARK(111);
        if (tid == 0) {
            wt_window_stats(P, c);
            const long long kn = (long long) wt_glb_add64(&P.counters[WT_CTR_TICKET], 1ull);
            c.sh->ticket = kn;
            if (kn < P.n_windows) wt_phase_header(P, c, kn);
        }
#if WT_DELTA_ZERO_EARLY
        if constexpr (MM) wt_delta_zero_mm<OP == WT_OP_MAX>(P, c, d, tid, nt);
        else wt_delta_zero<QQ, TT>(P, c, d, tid, nt);
#endif
        __syncthreads();
#ifdef WT_PROFILE_TAIL
        WT_TICK(0);
#else
        WT_TICK(7);
#endif
    }
#ifdef WT_PROFILE
    if (tid == 0)
        for (int q = 0; q < 8; q++) wt_glb_add64(&P.counters[WT_CTR_PROF + q], prof[q]);
#endif
}

#undef WT_SCAN_LANE

// Patch kernel: the general bitmap multiplexer over just the windows the difference-array kernel
// could not prove exact (a NaN, an Inf, too wide a dynamic range).  That kernel has already emitted
// those windows' runs -- coordinates, run count, position in the output -- so this one only has
// to recompute their values in the reference's own summation order and store them at the recorded
// offsets: no ticket, no look-back, no statistics.  One difference-array window (8192 bp) is
// `ratio` general windows; the difference-array kernel recorded the run offset of 16 sub-ranges of
// every such window, so every (window, sub-window) pair is a work item of its own.
struct WtPatchArgs {
    const int32_t *bad_list;            // difference-array window ids (slot order)
    const long long *bad_goff;          // first run of each
    const unsigned long long *n_bad;    // how many (device counter of the difference-array launch)
    const int32_t *d_win_chrom;         // the difference-array launch's window tables
    const int64_t *d_c_first_win;
    int ratio;                          // its window width / this launch's
};

__device__ __forceinline__ long long wt_lane_lower_bound(const int32_t *fin, long long lo, long long hi, long long g, long long b);

// The narrow-window index rows the patch kernel is going to read, and only those: for every window the
// difference-array kernel recorded, the ratio + 1 boundaries inside it, per track, by binary search.  (Round 3 built the
// WHOLE index at the patch kernel's window width whenever a launch had a window to patch -- 4 x the rows of the
// 8192-bp index, a third of a millisecond per chromosome for a few hundred windows.)
__global__ void __launch_bounds__(256) wt_patch_index_kernel(const WtParams P, const WtPatchArgs Q) {
    const long long n_bad = (long long) *Q.n_bad;
    const int N = P.n_tracks, R1 = Q.ratio + 1;
    const long long total = n_bad * R1 * N;
    for (long long t = (long long) blockIdx.x * 256 + threadIdx.x; t < total; t += (long long) gridDim.x * 256) {
        const long long j = t / ((long long) R1 * N);
        const int rem = (int) (t - j * R1 * N), r = rem / N, i = rem - r * N;
        const long long kd = Q.bad_list[j];
        const int ch = Q.d_win_chrom[kd];
        const long long m = (kd - Q.d_c_first_win[ch]) * Q.ratio + r;
        if (m > P.c_nwin[ch]) continue;             // (row c_nwin is the chromosome's last boundary)
        const long long seg = (long long) ch * N + i;
        const long long s0 = P.seg_off[seg], n = P.seg_off[seg + 1] - s0;
        const long long b = (long long) P.cbase[ch] + (m << P.logW);
        P.widx[(P.c_first_win[ch] + ch + m) * N + i] = (uint32_t) wt_lane_lower_bound(P.finish + s0, 0, n, n >> 1, b);
    }
}

template <int OP, int K, bool MULTI>
__global__ void __launch_bounds__(WT_MAX_BLOCK, WT_MIN_WAVES(K)) wt_patch_kernel(const WtParams P, const WtPatchArgs Q) {
    typedef float ValT;
    typedef float ScrT;
    extern __shared__ __attribute__((aligned(16))) char wt_lds[];
    WtCtx c;
    wt_ctx_init(c, P, wt_lds);
    WtLane<K> L;
    const int tid = threadIdx.x, nt = blockDim.x;
    const long long n_bad = (long long) *Q.n_bad;
    const int N = P.n_tracks, NC = MULTI ? P.chunk_tracks : N, n_chunks = MULTI ? P.n_chunks : 1;
    // work items = (window the difference-array kernel recorded, narrower window h inside it): independent of one
    // another -- the recording kernel left the run offset of every sub-range (wt_delta_note_offset)
    const long long n_items = n_bad * Q.ratio;
    for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
        const long long j = item / Q.ratio;
        const int h = (int) (item - j * Q.ratio);
        const long long kd = Q.bad_list[j];
        const int ch = Q.d_win_chrom[kd];
        const long long m = kd - Q.d_c_first_win[ch];
        const long long goff = Q.bad_goff[j * WT_BAD_SUB + h * (WT_BAD_SUB / Q.ratio)];
        {
            const long long mg = m * Q.ratio + h;
            if (mg >= P.c_nwin[ch]) continue;               // workgroup-uniform
            const long long k = P.c_first_win[ch] + mg;
            __syncthreads();                                // the previous window is done with the shared block
            if (tid == 0) wt_phase_header(P, c, k);
            wt_phase_zero(P, c, true, tid, nt);
            __syncthreads();
            // same sweeps as wt_reduce_kernel (chunked tracks: the first evaluation pass rides on the
            // sweep that builds the bitmaps; var / stddev / CV take a second one)
            constexpr int npass = wt_eval_passes(OP);
            WtAcc<K> A;
            wt_eval_init<OP, K>(A);
            for (int cc = 0; cc < n_chunks; cc++) {
                const int t_lo = cc * NC, t_hi = (t_lo + NC < N) ? t_lo + NC : N;
                if (MULTI && cc > 0) {
                    wt_phase_zero(P, c, false, tid, nt);
                    __syncthreads();
                }
                wt_phase_load<ValT>(P, c, t_lo, t_hi, false, tid, nt);
                __syncthreads();
                wt_phase_count_a(P, c, t_lo, t_hi, tid, nt);
                __syncthreads();
                wt_phase_count_b(P, c, t_lo, t_hi, tid, nt);
                __syncthreads();
                if (MULTI) {
                    wt_phase_eval_chunk<OP, ValT, ScrT, K>(P, c, A, 0, t_lo, t_hi, true, tid, nt);
                    __syncthreads();
                }
            }
            if (MULTI && npass == 2) wt_eval_mid<OP, K>(P, A);
            wt_phase_emask(P, c, OP == WT_OP_TTEST, tid, nt);
            __syncthreads();
            wt_phase_escan(P, c, tid, nt);
            __syncthreads();
            const long long n_emit = (long long) c.epfx[P.n_words];
            if (tid == 0) { c.sh->n_emit = (int32_t) n_emit; c.sh->goffset = goff; }
#pragma unroll
            for (int pass = MULTI ? 1 : 0; pass < npass; pass++) {
                for (int cc = 0; cc < n_chunks; cc++) {
                    const int t_lo = cc * NC, t_hi = (t_lo + NC < N) ? t_lo + NC : N;
                    if (MULTI) {
                        wt_phase_zero(P, c, false, tid, nt);
                        __syncthreads();
                        wt_phase_load<ValT>(P, c, t_lo, t_hi, false, tid, nt);
                        __syncthreads();
                        wt_phase_count_a(P, c, t_lo, t_hi, tid, nt);
                        __syncthreads();
                        wt_phase_count_b(P, c, t_lo, t_hi, tid, nt);
                        __syncthreads();
                    }
                    wt_phase_eval_chunk<OP, ValT, ScrT, K>(P, c, A, pass, t_lo, t_hi, false, tid, nt);
                    if (MULTI) __syncthreads();
                }
                if (pass == 0 && npass == 2) wt_eval_mid<OP, K>(P, A);
            }
            wt_phase_eval_finish<OP, ValT, ScrT, K>(P, c, A, L, tid, nt);
            __syncthreads();
            wt_phase_write<OP, ValT, K>(P, c, L, tid, nt);
        }
    }
}

// Each block owns ONE contiguous span of intervals: a single binary search finds the
// (chrom,track) segment of its first interval, every lane then walks its cursor forward.  The
// finish[] reads of the next sub-chunk are in flight while the current one is applied.
#ifndef WT_INDEX_UNROLL
#define WT_INDEX_UNROLL 2
#endif
#define WT_INDEX_CHUNK (256 * 4 * WT_INDEX_UNROLL)
__global__ void __launch_bounds__(256) wt_index_kernel(const WtParams P, long long total, long long span) {
    // a batch whose run lists were compacted on device (operator chains that drop runs) holds fewer
    // intervals than the host counted: the device's seg_off[] is the authority
    const long long dev_total = P.seg_off[(long long) P.n_chrom * P.n_tracks];
    if (total > dev_total) total = dev_total;
    const long long begin0 = (long long) blockIdx.x * span;
    long long end0 = begin0 + span;
    if (end0 > total) end0 = total;
    if (begin0 >= end0) return;
    WtIndexCursor cur;
    wt_index_cursor_set(P, cur, wt_index_find_segment(P, begin0));
    // Every lane takes 4 CONSECUTIVE runs per sub-chunk: one 16-byte load of finish[] plus the
    // predecessor's finish (the kernel was instruction-bound on per-element 64-bit addressing).
    // begin0 is a multiple of WT_INDEX_CHUNK, so the vectors never straddle `total` unguarded.
    struct __attribute__((packed, aligned(4))) V4 { int32_t x[4]; };
    V4 f[WT_INDEX_UNROLL], nf[WT_INDEX_UNROLL];
    int32_t pf[WT_INDEX_UNROLL], npf[WT_INDEX_UNROLL];
    // FULL sub-chunks: unconditional loads.  (Round 3, read off the ISA: with `if (more) fetch(next)` and the
    // per-lane tail guards around the loads, the compiler could not count the loads in flight and waited for
    // ALL of them before applying the current sub-chunk -- the prefetch overlapped nothing, and 1, 2 or 4
    // vectors per lane made no difference.)  A prefetch past the last full sub-chunk re-reads that one.
    const long long n_full = (end0 - begin0) / WT_INDEX_CHUNK;
    auto fetch = [&](long long base, V4 (&v)[WT_INDEX_UNROLL], int32_t (&p)[WT_INDEX_UNROLL]) {
#pragma unroll
        for (int u = 0; u < WT_INDEX_UNROLL; u++) {
            const long long g = base + 4ll * (threadIdx.x + 256 * u);
            v[u] = *(const V4 *) (P.finish + g);
            p[u] = P.finish[g > 0 ? g - 1 : 0];
        }
    };
    if (n_full > 0) {
        const long long last_full = begin0 + (n_full - 1) * WT_INDEX_CHUNK;
        fetch(begin0, f, pf);
        for (long long begin = begin0; begin <= last_full; begin += WT_INDEX_CHUNK) {
            const long long nb = begin + WT_INDEX_CHUNK;
            fetch(nb <= last_full ? nb : last_full, nf, npf);
#pragma unroll
            for (int u = 0; u < WT_INDEX_UNROLL; u++) {
                const long long g = begin + 4ll * (threadIdx.x + 256 * u);
                int32_t prev = pf[u];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    wt_index_apply(P, cur, g + q, f[u].x[q], prev);
                    prev = f[u].x[q];
                }
            }
#pragma unroll
            for (int u = 0; u < WT_INDEX_UNROLL; u++) { f[u] = nf[u]; pf[u] = npf[u]; }
        }
    }
    // the partial sub-chunk at the end of the span (only the last block of a launch has one)
    const long long begin = begin0 + n_full * WT_INDEX_CHUNK;
    if (begin < end0) {
#pragma unroll
        for (int u = 0; u < WT_INDEX_UNROLL; u++) {
            const long long g = begin + 4ll * (threadIdx.x + 256 * u);
            int32_t prev = (g < end0 && g > 0) ? P.finish[g - 1] : 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (g + q < end0) {
                    const int32_t fq = P.finish[g + q];
                    wt_index_apply(P, cur, g + q, fq, prev);
                    prev = fq;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Window index by SEARCH (round 3).  widx[row(b)][track] is the lower bound of the boundary b = cbase + m * W
// in the track's finish[] (wt_index_apply's claims, restated), so nothing obliges the kernel to read every
// finish: at a mean run of 16 bp the scan above reads 256 entries (1 KB) per boundary and stays near
// 3.3 TB/s whatever its loads look like (1, 2 or 4 vectors in flight, long or short spans per block).
//   coarse  every 64th row of every track by a plain binary search over the track's whole segment
//           (one lane each; ~1 % of the boundaries, the upper levels of the search stay in L2);
//   fine    a wave owns the 64 rows between two coarse rows of ONE track: each lane interpolates its boundary
//           between the two coarse answers and gallops / bisects from the guess -- a few probes, almost all
//           inside one or two 128-byte lines; the 16 waves of a workgroup (16 neighbouring tracks, same
//           rows) exchange through LDS so that a row is written as one 64-byte piece.
// (A first version found the bracket with a cooperative 64-ary search per wave: 17.7 ms against the scan's
// 22-27 -- its 64 probes per step were 64 cache lines per step.)  Same result as the scan by construction;
// WTAMD_INDEX=scan selects the scan, WTAMD_INDEX_CHECK=1 runs both and compares (tests).
// ---------------------------------------------------------------------------
#define WT_ISEARCH_ROWS 64
#define WT_ISEARCH_TRACKS 16

// chromosome of an index row: the rows of chromosome ch start at c_first_win[ch] + ch
__device__ __forceinline__ int wt_index_row_chrom(const WtParams &P, long long row) {
    int lo = 0;
    for (int hi = P.n_chrom; hi - lo > 1;) {
        const int mid = (lo + hi) >> 1;
        if (P.c_first_win[mid] + mid <= row) lo = mid; else hi = mid;
    }
    return lo;
}

// first x in [lo, hi) with fin[x] >= b (hi if none), starting from a guess g in [lo, hi)
__device__ __forceinline__ long long wt_lane_lower_bound(const int32_t *fin, long long lo, long long hi, long long g, long long b) {
    if (lo >= hi) return lo;
    if ((long long) fin[g] >= b) {
        hi = g;
        for (long long d = 1;; d <<= 1) {
            const long long q = hi - d;
            if (q < lo) break;
            if ((long long) fin[q] < b) { lo = q + 1; break; }
            hi = q;
        }
    } else {
        lo = g + 1;
        for (long long d = 1;; d <<= 1) {
            const long long q = lo + d - 1;
            if (q >= hi) break;
            if ((long long) fin[q] >= b) { hi = q; break; }
            lo = q + 1;
        }
    }
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if ((long long) fin[mid] < b) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// coarse[strip][track]: the index entry of row strip * 64
__global__ void __launch_bounds__(256) wt_index_coarse_kernel(const WtParams P, long long n_strips, uint32_t *coarse) {
    const long long t = (long long) blockIdx.x * 256 + threadIdx.x;
    const int N = P.n_tracks;
    if (t >= n_strips * N) return;
    const long long strip = t / N;
    const int i = (int) (t - strip * N);
    const long long row = strip * WT_ISEARCH_ROWS;
    const int ch = wt_index_row_chrom(P, row);
    const long long m = row - (P.c_first_win[ch] + ch);
    const long long seg = (long long) ch * N + i;
    const long long s0 = P.seg_off[seg], n = P.seg_off[seg + 1] - s0;
    const long long b = (long long) P.cbase[ch] + (m << P.logW);
    coarse[t] = (uint32_t) wt_lane_lower_bound(P.finish + s0, 0, n, n >> 1, b);
}

__global__ void __launch_bounds__(WT_ISEARCH_ROWS * WT_ISEARCH_TRACKS) wt_index_search_kernel(const WtParams P, long long n_rows, long long n_strips,
                                                                                              const uint32_t *coarse) {
    __shared__ uint32_t out[WT_ISEARCH_ROWS][WT_ISEARCH_TRACKS + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = P.n_tracks;
    const long long strip = blockIdx.x;
    const long long r0 = strip * WT_ISEARCH_ROWS;
    const int i = (int) blockIdx.y * WT_ISEARCH_TRACKS + wave;
    if (i < N) {                                    // (wave-uniform)
        long long row = r0 + lane;
        if (row > n_rows - 1) row = n_rows - 1;     // the last strip: duplicates of the last row, not stored
        const int ch = wt_index_row_chrom(P, row);
        const long long m = row - (P.c_first_win[ch] + ch);
        const long long seg = (long long) ch * N + i;
        const long long s0 = P.seg_off[seg], n = P.seg_off[seg + 1] - s0;
        const long long b = (long long) P.cbase[ch] + (m << P.logW);
        const int32_t *fin = P.finish + s0;
        // the bracket: the coarse answers of this strip's and the next strip's first rows, where they belong
        // to the lane's chromosome (a strip may straddle a chromosome edge)
        const int ch_first = __shfl(ch, 0);
        const bool have_next = strip + 1 < n_strips;
        const int ch_next = have_next ? wt_index_row_chrom(P, r0 + WT_ISEARCH_ROWS) : -1;
        const bool real_a = ch == ch_first, real_b = ch == ch_next;
        const long long A = real_a ? (long long) coarse[strip * N + i] : 0;
        const long long B = real_b ? (long long) coarse[(strip + 1) * N + i] : n;
        long long g = (real_a && real_b) ? A + (((B - A) * lane) >> 6) : (A + B) >> 1;
        if (g > B - 1) g = B - 1;
        out[lane][wave] = (uint32_t) wt_lane_lower_bound(fin, A, B, g, b);
    }
    __syncthreads();
    const int orow = tid / WT_ISEARCH_TRACKS, ocol = tid % WT_ISEARCH_TRACKS;
    const long long row = r0 + orow;
    const int track = (int) blockIdx.y * WT_ISEARCH_TRACKS + ocol;
    if (row < n_rows && track < N) P.widx[(size_t) row * N + track] = out[orow][ocol];
}

// check mode: number of entries in which two indices differ
__global__ void __launch_bounds__(256) wt_index_compare_kernel(const uint32_t *a, const uint32_t *b, long long n, unsigned long long *diff) {
    const long long t = (long long) blockIdx.x * 256 + threadIdx.x;
    if (t < n && a[t] != b[t]) atomicAdd(diff, 1ull);
}

__global__ void __launch_bounds__(256) wt_extents_kernel(const int64_t *seg_off, const int32_t *start,
                                                          const int32_t *finish, long long n_seg,
                                                          int32_t *first_start, int32_t *last_finish) {
    const long long s = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const long long lo = seg_off[s], hi = seg_off[s + 1];
    first_start[s] = hi > lo ? start[lo] : 0;
    last_finish[s] = hi > lo ? finish[hi - 1] : 0;
}

// AUC: statistics.c:103-120.  Deterministic two-level sum.
__global__ void __launch_bounds__(256) wt_auc_kernel(const int32_t *start, const int32_t *finish, const double *value,
                                                      long long n, double *partial, double *partial_span,
                                                      const unsigned long long *n_dev = nullptr) {
    __shared__ double red[256];
    double acc = 0, span = 0;
    if (n_dev && (long long) *n_dev < n) n = (long long) *n_dev;       // (pipeline: the run count only exists on the device)
    const long long stride = (long long) gridDim.x * blockDim.x;
    for (long long r = (long long) blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
        const double v = value[r];
        if (v == v) {                                   // NaN runs are skipped (statistics.c:78, 110)
            const double len = (double) (finish[r] - start[r]);
            acc += len * v;
            span += len;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int) threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
    if (partial_span) {                                 // MeanIntegrator also needs the non-NaN span
        __syncthreads();
        red[threadIdx.x] = span;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int) threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial_span[blockIdx.x] = red[0];
    }
}

__global__ void wt_auc_final_kernel(const double *partial, int n, double *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double acc = 0;
        for (int i = 0; i < n; i++) acc += partial[i];
        *out = acc;
    }
}

// ---------------------------------------------------------------------------
// Pearson correlation of two tracks over the Multiplexer tile (reference PearsonIntegrator,
// statistics.c:414-465).  The reference updates {count, sum, T_XX, T_XY, T_YY} run by run with the
// weighted Welford / Chan step (its `new_mean` is old sum / new count; expanding
// n*L/(n+L) * (X - mean)^2 gives exactly its expression).  Here every lane applies that very step
// to a contiguous slice of runs, and slices are merged pairwise in genome order with the same
// formula for two aggregates -- mathematically identical, rounding differs in the last bits
// (tests: 1e-9 relative against the oracle; the reference prints 6 decimals).
// ---------------------------------------------------------------------------
struct WtMoments {
    double n, sx, sy, txx, txy, tyy;
};

__device__ inline void wt_moments_add_run(WtMoments &m, double X, double Y, double L) {
    if (m.n > 0) {
        const double nn = m.n + L;
        const double old_mx = m.sx / m.n, new_mx = m.sx / nn;
        const double old_my = m.sy / m.n, new_my = m.sy / nn;
        const double ratio = m.n / nn;
        m.txy += (new_mx * old_my + ratio * X * Y - new_mx * Y - new_my * X) * L;
        m.txx += (new_mx * (old_mx - 2 * X) + ratio * X * X) * L;
        m.tyy += (new_my * (old_my - 2 * Y) + ratio * Y * Y) * L;
    }
    m.n += L;
    m.sx += X * L;
    m.sy += Y * L;
}

// a := a (+) b, b following a in genome order
__device__ inline void wt_moments_merge(WtMoments &a, const WtMoments &b) {
    if (b.n == 0) return;
    if (a.n == 0) { a = b; return; }
    const double n = a.n + b.n;
    const double dx = b.sx / b.n - a.sx / a.n, dy = b.sy / b.n - a.sy / a.n;
    const double w = a.n * b.n / n;
    a.txx += b.txx + dx * dx * w;
    a.txy += b.txy + dx * dy * w;
    a.tyy += b.tyy + dy * dy * w;
    a.n = n;
    a.sx += b.sx;
    a.sy += b.sy;
}

__global__ void __launch_bounds__(256) wt_pearson_kernel(const int32_t *start, const int32_t *finish, const double *tile,
                                                          const uint8_t *inplay, double dx, double dy, long long n,
                                                          WtMoments *partial, const unsigned long long *n_dev = nullptr) {
    __shared__ WtMoments red[256];
    if (n_dev && (long long) *n_dev < n) n = (long long) *n_dev;
    const long long total_lanes = (long long) gridDim.x * blockDim.x;
    const long long per = (n + total_lanes - 1) / total_lanes;          // contiguous slice per lane
    const long long lane_id = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    long long lo = lane_id * per, hi = lo + per;
    if (hi > n) hi = n;
    WtMoments m = {0, 0, 0, 0, 0, 0};
    for (long long r = lo; r < hi; r++) {
        const double X = inplay[2 * r] ? tile[2 * r] : dx;
        const double Y = inplay[2 * r + 1] ? tile[2 * r + 1] : dy;
        wt_moments_add_run(m, X, Y, (double) (finish[r] - start[r]));
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 1; s < 256; s <<= 1) {                                 // ordered pairwise merge
        if ((threadIdx.x & (2 * s - 1)) == 0) wt_moments_merge(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void wt_pearson_final_kernel(const WtMoments *partial, int n, double *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        WtMoments m = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n; i++) wt_moments_merge(m, partial[i]);
        double txx = m.txx, tyy = m.tyy;                               // (constant track: see wtamd_pearson_finish)
        if (m.n > 0) {
            const double mx = m.sx / m.n, my = m.sy / m.n;
            if (txx <= m.n * mx * mx * 1e-14) txx = 0;
            if (tyy <= m.n * my * my * 1e-14) tyy = 0;
        }
        const double den = txx * tyy;
        out[0] = den ? m.txy / sqrt(den) : __builtin_nan("");          // statistics.c:421-423
        out[1] = m.n; out[2] = m.sx; out[3] = m.sy; out[4] = m.txx; out[5] = m.txy; out[6] = m.tyy;
    }
}


// Input contract check (sorted, non-overlapping, positive-length runs inside every (chrom, track)
// segment -- anything else is undefined behaviour in the reference's Multiplexer too,
// multiplexer.c:76-96, and bedReader.c:46-49 checks it for text input).  One lane per run.
__global__ void __launch_bounds__(256) wt_validate_kernel(const int32_t *start, const int32_t *finish, const int64_t *seg_off,
                                                           long long n_seg, long long total, unsigned long long *out) {
    const long long g = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    bool bad = finish[g] <= start[g];
    if (!bad && g > 0 && start[g] < finish[g - 1]) {
        long long lo = 0, hi = n_seg;                   // is g the first run of its segment?
        while (hi - lo > 1) {
            const long long mid = (lo + hi) >> 1;
            if (seg_off[mid] <= g) lo = mid; else hi = mid;
        }
        bad = seg_off[lo] != g;
    }
    if (bad) {
        atomicAdd(&out[0], 1ull);
        atomicMin(&out[1], (unsigned long long) g);
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int wt_fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}
int wt_fail_ext(int code, const std::string &msg) { return wt_fail(code, msg); }   // for the other translation units

#define WT_HIP(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return wt_fail(WTAMD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));  \
    } while (0)

struct WtWindows {
    WtPlan plan_geom;        // only W matters for the tables
    WtWindowTables tab;
    int32_t *d_cbase = nullptr, *d_cnwin = nullptr, *d_chi = nullptr, *d_win_chrom = nullptr;
    int64_t *d_cfirst = nullptr;
    uint32_t *d_widx = nullptr;
    uint32_t *d_cidx = nullptr;         // coarse index of the searched window index (every 64th row)
    unsigned long long *d_status = nullptr;
    int32_t *d_bad_list = nullptr;      // difference-array launches: windows not provably exact ...
    long long *d_bad_goff = nullptr;    // ... and where their runs start (both [n_windows])
    bool indexed = false;
    // capacities of the device tables (entries); the tables are reused and only ever grow
    int64_t cap_chrom = 0, cap_win = 0, cap_widx = 0, cap_bad = 0, cap_cidx = 0;
    bool tab_valid = false;             // tab / device tables describe the track set's current data
    int64_t *h_tab = nullptr;           // pinned staging of the per-chromosome tables (pipeline slots: asynchronous upload)
    // round 6: cbase | cnwin | chi | cfirst | win_chrom live in ONE allocation (d_tabs) and travel in ONE copy -- a NEW track set's first
    // index paid five hipMallocs and five blocking copies for them, 0.1 ms of host time per chromosome of a resident pass
    char *d_tabs = nullptr;
    int64_t cap_tabs = 0;               // bytes
};

struct wtamd_trackset {
    int n_chrom = 0, n_tracks = 0;
    bool value_f64 = false;
    bool owns = false;
    int64_t n_intervals = 0;
    std::vector<int64_t> seg_off;
    std::vector<double> defaults;
    std::vector<int32_t> first_start, last_finish;
    std::vector<int32_t> range_lo, range_hi;     // optional run-start ranges (empty = none)
    int32_t *d_start = nullptr, *d_finish = nullptr;
    void *d_value = nullptr;
    int64_t *d_seg_off = nullptr;
    double *d_defaults = nullptr;
    unsigned long long *d_counters = nullptr;
    unsigned long long *h_counters = nullptr;   // pinned
    unsigned long long *h_debug = nullptr;      // pinned, device-visible (debug builds)
    int64_t *d_chrom_run_off = nullptr;         // scratch when the caller passes none
    char *d_gscratch = nullptr;                 // median / MWU columns of very many tracks (grown on demand)
    size_t gscratch_bytes = 0;
    double *d_mwu_table = nullptr;              // MWUReduction's last step as a table (wt_mwu_make_table), for set sizes mwu_n1 / mwu_n2
    int mwu_n1 = -1, mwu_n2 = -1, mwu_kmax = 0;
    int mwu_few_ties = -1;                      // MWUReduction's kernel by the data (wt_mwu_few_ties): -1 not looked at yet, 1 walk (wt_mwalk.h), 0 register columns
    std::vector<double *> mwu_retired;          // tables of earlier set sizes (freed with the track set)
    std::map<int, WtWindows> windows;           // keyed by W
    hipEvent_t ev_i0 = nullptr, ev_i1 = nullptr, ev_r0 = nullptr, ev_r1 = nullptr;
    bool have_index_time = false, have_reduce_time = false;
    wtamd_stats stats{};
    int device = 0;
    int num_cu = 256;
    bool scratch_f32 = false;
    // Sum / Mean over float tracks: what a completed difference-array launch found out about this data
    // the difference-array launches' verdict on this data, per class of reducer -- [0] Sum / Mean / the var family (exponent range
    // of a window), [1] TTestReduction (its own, narrower windows, and positions whose variance cancels: wt_delta_scan3_tt)
    // [2] Max / Min (only a NaN or a -0.0 sends a window to the general kernel)
    bool delta_failed_[3] = {false, false, false};      // many windows are not provably exact: the class uses the general kernel
    bool delta_verified_[3] = {false, false, false};    // verdict known: delta_n_bad windows (few) get patched by the general kernel
    long long delta_n_bad_[3] = {0, 0, 0};
    // pipeline slot (wt_pipe.h): the run lists are rebound per batch, device tables are reused,
    // every upload is asynchronous on the launch stream from pinned staging
    bool pipe_mode = false;
};

template <class T>
static hipError_t wt_grow(T **p, int64_t *cap, int64_t need) {
    if (*cap >= need && *p) return hipSuccess;
    int64_t c = *cap * 2;
    if (c < need) c = need;
    if (c < 1) c = 1;
    (void) hipFree(*p);
    *p = nullptr; *cap = 0;
    const hipError_t e = hipMalloc((void **) p, sizeof(T) * (size_t) c);
    if (e == hipSuccess) *cap = c;
    return e;
}

static void wt_free_windows(WtWindows &w) {
    (void) hipFree(w.d_tabs);          // (d_cbase, d_cnwin, d_chi, d_cfirst, d_win_chrom point into it)
    (void) hipFree(w.d_widx); (void) hipFree(w.d_cidx); (void) hipFree(w.d_status); (void) hipFree(w.d_bad_list); (void) hipFree(w.d_bad_goff);
    if (w.h_tab) (void) hipHostFree(w.h_tab);
    w = WtWindows();
}

extern "C" {

const char *wtamd_last_error(void) { return g_last_error.c_str(); }
const char *wtamd_version(void) { return "wiggletools_amd 0.1 (gfx950)"; }

int wtamd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The first HIP call of a process brings the runtime up (device discovery, the library's code object, the first hardware
// queues): 170 ms between newMultiplexer and the first batch in the trace of a fresh `mean *.bw` process (round 5,
// tools/cli_cold.py) -- spent AFTER the 100 files had been opened, although neither needs the other.  The callers that
// know GPU work is coming (wtamd_BigWiggleReaders: it opens the files on worker threads) start it on a helper thread
// first; wtamd_pipe_create waits for the helper.  The helper works on the default device (a process that selects another
// one has called into HIP already: the runtime is up and this costs nothing).
__global__ void wt_warm_kernel(int *p) { if (p && threadIdx.x == 4096) *p = 0; }

static std::mutex g_warm_mu;
static std::thread *g_warm_thread = nullptr;
static bool g_warm_started = false;

static void wt_warmup_join() {
    std::thread *t = nullptr;
    { std::lock_guard<std::mutex> lk(g_warm_mu); t = g_warm_thread; g_warm_thread = nullptr; }
    if (t) { t->join(); delete t; }
}

// The helper is a new thread: HIP's current device is per thread, so it is told which device to warm -- the one the caller
// selected (wtamd_set_device, or the calling thread's current device when that was set through the runtime, e.g. by
// torch.cuda.set_device in a one-process-per-GPU job).  With several devices visible and none selected the warm-up is skipped:
// a context on GPU 0 from every rank would help nobody.
static std::atomic<int> g_device_chosen{-1};

void wtamd_warmup_async(void) {
    std::lock_guard<std::mutex> lk(g_warm_mu);
    if (g_warm_started || getenv("WTAMD_NO_WARMUP")) return;
    int n = 0, cur = -1;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void) hipGetLastError(); return; }
    int ordinal = g_device_chosen.load();
    if (ordinal < 0 && hipGetDevice(&cur) == hipSuccess && (n == 1 || cur > 0)) ordinal = cur;     // (cur == 0 of several: nobody chose yet)
    if (ordinal < 0 || ordinal >= n) return;                    // (not started: a later call, after wtamd_set_device, may)
    g_warm_started = true;
    g_warm_thread = new std::thread([ordinal] {
        if (hipSetDevice(ordinal) != hipSuccess) { (void) hipGetLastError(); return; }
        hipStream_t st[3] = {nullptr, nullptr, nullptr};
        for (auto &s : st)
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
        for (auto &s : st)
            if (s) hipLaunchKernelGGL(wt_warm_kernel, dim3(1), dim3(64), 0, s, (int *) nullptr);
        for (auto &s : st)
            if (s) { (void) hipStreamSynchronize(s); (void) hipStreamDestroy(s); }
        (void) hipGetLastError();
    });
    atexit(wt_warmup_join);         // (a process that exits before it ever built a pipe must not leave the helper inside the runtime)
}

int wtamd_set_device(int ordinal) {
    wt_warmup_join();               // (a helper still warming another device finishes first)
    WT_HIP(hipSetDevice(ordinal));
    g_device_chosen.store(ordinal);
    return WTAMD_OK;
}

int wtamd_current_device(void) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void) hipGetLastError(); return -1; }
    return d;
}

static int wt_trackset_common(const wtamd_tracks *t, wtamd_trackset *ts) {
    if (!t || t->n_chrom < 0 || t->n_tracks <= 0 || !t->seg_off || !t->defaults)
        return wt_fail(WTAMD_ERR_ARG, "wtamd_trackset_create: bad tracks descriptor");
    if (wtamd_device_count() <= 0) return wt_fail(WTAMD_ERR_NODEVICE, "no HIP device visible");
    ts->n_chrom = t->n_chrom;
    ts->n_tracks = t->n_tracks;
    ts->value_f64 = t->value_is_f64 != 0;
    const int64_t n_seg = (int64_t) t->n_chrom * t->n_tracks;
    ts->seg_off.assign(t->seg_off, t->seg_off + n_seg + 1);
    for (int64_t s = 0; s < n_seg; s++)
        if (ts->seg_off[s + 1] < ts->seg_off[s]) return wt_fail(WTAMD_ERR_ARG, "seg_off not monotone");
    ts->n_intervals = ts->seg_off[n_seg] - ts->seg_off[0];
    if (ts->seg_off[0] != 0) return wt_fail(WTAMD_ERR_ARG, "seg_off[0] must be 0");
    ts->defaults.assign(t->defaults, t->defaults + t->n_tracks);
    if (t->range_lo && t->range_hi) {
        ts->range_lo.assign(t->range_lo, t->range_lo + t->n_chrom);
        ts->range_hi.assign(t->range_hi, t->range_hi + t->n_chrom);
    }
    ts->scratch_f32 = !ts->value_f64 && wt_defaults_fit_f32(ts->defaults.data(), ts->n_tracks);
    WT_HIP(hipGetDevice(&ts->device));
    hipDeviceProp_t prop;
    WT_HIP(hipGetDeviceProperties(&prop, ts->device));
    ts->num_cu = prop.multiProcessorCount;
    WT_HIP(hipMalloc(&ts->d_seg_off, sizeof(int64_t) * (n_seg + 1)));
    WT_HIP(hipMemcpy(ts->d_seg_off, ts->seg_off.data(), sizeof(int64_t) * (n_seg + 1), hipMemcpyHostToDevice));
    WT_HIP(hipMalloc(&ts->d_defaults, sizeof(double) * t->n_tracks));
    WT_HIP(hipMemcpy(ts->d_defaults, ts->defaults.data(), sizeof(double) * t->n_tracks, hipMemcpyHostToDevice));
    WT_HIP(hipMalloc(&ts->d_counters, sizeof(unsigned long long) * WT_CTR_N));
    WT_HIP(hipHostMalloc(&ts->h_counters, sizeof(unsigned long long) * WT_CTR_N));
    WT_HIP(hipHostMalloc(&ts->h_debug, sizeof(unsigned long long) * 16));
    memset(ts->h_debug, 0, sizeof(unsigned long long) * 16);
    WT_HIP(hipMalloc(&ts->d_chrom_run_off, sizeof(int64_t) * (t->n_chrom + 1)));
    WT_HIP(hipEventCreate(&ts->ev_i0));
    WT_HIP(hipEventCreate(&ts->ev_i1));
    WT_HIP(hipEventCreate(&ts->ev_r0));
    WT_HIP(hipEventCreate(&ts->ev_r1));
    return WTAMD_OK;
}

// The window arithmetic (w0 + W, look-ahead sentinels) is 32-bit: keep a margin below INT32_MAX.
// Loud, at creation -- a run that close to 2^31 would otherwise be silently lost.
static int wt_check_extents(wtamd_trackset *ts) {
    for (size_t q = 0; q < ts->last_finish.size(); q++)
        if (ts->seg_off[q + 1] > ts->seg_off[q] && ts->last_finish[q] > WTAMD_MAX_COORD) {
            const int64_t bad = ts->last_finish[q];
            return wt_fail(WTAMD_ERR_ARG, "run finish " + std::to_string(bad) + " above the supported maximum " +
                           std::to_string((long long) WTAMD_MAX_COORD) + " (2^31 - 65537)");
        }
    return WTAMD_OK;
}

// Device-side (first start, last finish) of every (chrom, track) segment into the host copies.
static int wt_refresh_extents_device(wtamd_trackset *ts) {
    const int64_t n_seg = (int64_t) ts->n_chrom * ts->n_tracks;
    ts->first_start.assign(n_seg, 0);
    ts->last_finish.assign(n_seg, 0);
    if (n_seg > 0 && ts->n_intervals > 0) {
        int32_t *d_fs = nullptr, *d_lf = nullptr;
        WtDevScope scope;
        WT_HIP(scope.alloc(&d_fs, sizeof(int32_t) * n_seg));
        WT_HIP(scope.alloc(&d_lf, sizeof(int32_t) * n_seg));
        hipLaunchKernelGGL(wt_extents_kernel, dim3((unsigned) ((n_seg + 255) / 256)), dim3(256), 0, 0,
                           ts->d_seg_off, ts->d_start, ts->d_finish, (long long) n_seg, d_fs, d_lf);
        WT_HIP(hipGetLastError());
        WT_HIP(hipMemcpy(ts->first_start.data(), d_fs, sizeof(int32_t) * n_seg, hipMemcpyDeviceToHost));
        WT_HIP(hipMemcpy(ts->last_finish.data(), d_lf, sizeof(int32_t) * n_seg, hipMemcpyDeviceToHost));
    }
    return WTAMD_OK;
}

static int wt_create_host_impl(const wtamd_tracks *t, wtamd_trackset *ts) {
    int rc = wt_trackset_common(t, ts);
    if (rc != WTAMD_OK) return rc;
    ts->owns = true;
    const int64_t n = ts->n_intervals;
    const size_t vsz = ts->value_f64 ? 8 : 4;
    const int64_t n_alloc = n > 0 ? n : 1;
    WT_HIP(hipMalloc(&ts->d_start, sizeof(int32_t) * n_alloc));
    WT_HIP(hipMalloc(&ts->d_finish, sizeof(int32_t) * n_alloc));
    WT_HIP(hipMalloc(&ts->d_value, vsz * n_alloc));
    if (n > 0) {
        if (!t->start || !t->finish || !t->value) return wt_fail(WTAMD_ERR_ARG, "NULL arrays");
        WT_HIP(hipMemcpy(ts->d_start, t->start, sizeof(int32_t) * n, hipMemcpyHostToDevice));
        WT_HIP(hipMemcpy(ts->d_finish, t->finish, sizeof(int32_t) * n, hipMemcpyHostToDevice));
        WT_HIP(hipMemcpy(ts->d_value, t->value, vsz * n, hipMemcpyHostToDevice));
    }
    const int64_t n_seg = (int64_t) ts->n_chrom * ts->n_tracks;
    ts->first_start.assign(n_seg, 0);
    ts->last_finish.assign(n_seg, 0);
    for (int64_t s = 0; s < n_seg; s++)
        if (ts->seg_off[s + 1] > ts->seg_off[s]) {
            ts->first_start[s] = t->start[ts->seg_off[s]];
            ts->last_finish[s] = t->finish[ts->seg_off[s + 1] - 1];
        }
    return wt_check_extents(ts);
}

int wtamd_trackset_create_host(const wtamd_tracks *t, wtamd_trackset **out) {
    if (!out) return wt_fail(WTAMD_ERR_ARG, "out == NULL");
    wtamd_trackset *ts = new wtamd_trackset();
    const int rc = wt_create_host_impl(t, ts);
    if (rc != WTAMD_OK) { wtamd_trackset_destroy(ts); return rc; }     // every failing path releases what was allocated
    *out = ts;
    return WTAMD_OK;
}

static int wt_create_device_impl(const wtamd_tracks *t, wtamd_trackset *ts) {
    int rc = wt_trackset_common(t, ts);
    if (rc != WTAMD_OK) return rc;
    ts->owns = false;
    ts->d_start = const_cast<int32_t *>(t->start);
    ts->d_finish = const_cast<int32_t *>(t->finish);
    ts->d_value = const_cast<void *>(t->value);
    rc = wt_refresh_extents_device(ts);
    if (rc != WTAMD_OK) return rc;
    return wt_check_extents(ts);
}

int wtamd_trackset_create_device(const wtamd_tracks *t, wtamd_trackset **out) {
    if (!out) return wt_fail(WTAMD_ERR_ARG, "out == NULL");
    wtamd_trackset *ts = new wtamd_trackset();
    const int rc = wt_create_device_impl(t, ts);
    if (rc != WTAMD_OK) { wtamd_trackset_destroy(ts); return rc; }
    *out = ts;
    return WTAMD_OK;
}

void wtamd_trackset_destroy(wtamd_trackset *ts) {
    if (!ts) return;
    if (ts->owns) { (void) hipFree(ts->d_start); (void) hipFree(ts->d_finish); (void) hipFree(ts->d_value); }
    (void) hipFree(ts->d_seg_off); (void) hipFree(ts->d_defaults); (void) hipFree(ts->d_counters); (void) hipFree(ts->d_chrom_run_off); (void) hipFree(ts->d_gscratch); (void) hipFree(ts->d_mwu_table);
    for (double *q : ts->mwu_retired) (void) hipFree(q);
    if (ts->h_counters) (void) hipHostFree(ts->h_counters);
    if (ts->h_debug) (void) hipHostFree(ts->h_debug);
    for (auto &kv : ts->windows) wt_free_windows(kv.second);
    if (ts->ev_i0) (void) hipEventDestroy(ts->ev_i0);
    if (ts->ev_i1) (void) hipEventDestroy(ts->ev_i1);
    if (ts->ev_r0) (void) hipEventDestroy(ts->ev_r0);
    if (ts->ev_r1) (void) hipEventDestroy(ts->ev_r1);
    delete ts;
}

static int64_t wt_span(const wtamd_trackset *ts) {
    // sum over chromosomes of (max finish - min start)
    int64_t span = 0;
    for (int c = 0; c < ts->n_chrom; c++) {
        int64_t lo = INT64_MAX, hi = INT64_MIN;
        for (int i = 0; i < ts->n_tracks; i++) {
            const int64_t s = (int64_t) c * ts->n_tracks + i;
            if (ts->seg_off[s + 1] > ts->seg_off[s]) {
                lo = std::min<int64_t>(lo, ts->first_start[s]);
                hi = std::max<int64_t>(hi, ts->last_finish[s]);
            }
        }
        if (lo <= hi) span += hi - lo;
    }
    return span;
}

int64_t wtamd_trackset_max_runs(const wtamd_trackset *ts) {
    if (!ts) return 0;
    // every run starts at an interval start or finish, and no two runs start at the same position
    return std::min<int64_t>(2 * ts->n_intervals, wt_span(ts));
}

// Window tables of width W for the track set's current data.  The device tables are kept and
// reused (they only ever grow); a pipeline slot uploads them asynchronously on `s` from pinned
// staging (its per-batch data is one chromosome: four scalars and an all-zero win_chrom[]).
static int wt_get_windows(wtamd_trackset *ts, int W, WtWindows **out, hipStream_t s = nullptr) {
    WtWindows &w = ts->windows[W];
    *out = &w;
    if (w.tab_valid) return WTAMD_OK;
    wt_make_windows(ts->n_chrom, ts->n_tracks, ts->seg_off.data(), ts->first_start.data(), ts->last_finish.data(), W, w.tab,
                    ts->range_lo.empty() ? nullptr : ts->range_lo.data(),
                    ts->range_hi.empty() ? nullptr : ts->range_hi.data());
    w.indexed = false;
    const int64_t nc = ts->n_chrom > 0 ? ts->n_chrom : 1;
    const int64_t nwin = w.tab.n_windows > 0 ? w.tab.n_windows : 1;
    const int64_t nwidx = (w.tab.n_rows > 0 ? w.tab.n_rows : 1) * (int64_t) ts->n_tracks;
    // layout of the combined table allocation (8-byte aligned pieces)
    auto up8 = [](int64_t x) { return (x + 7) & ~(int64_t) 7; };
    if (w.cap_chrom < nc || w.cap_win < nwin || !w.d_tabs) {
        // (growing: twice the old capacity at least, as wt_grow does -- a pipeline slot's batches creep up, and every hipFree waits for the device)
        const int64_t cc = w.cap_chrom < nc ? std::max<int64_t>(nc, 2 * w.cap_chrom) : w.cap_chrom;
        const int64_t cw = w.cap_win < nwin ? std::max<int64_t>(nwin, 2 * w.cap_win) : w.cap_win;
        const int64_t o_cnwin = up8(4 * cc), o_chi = o_cnwin + up8(4 * cc), o_cfirst = o_chi + up8(4 * cc), o_win = o_cfirst + 8 * (cc + 1);
        const int64_t total = o_win + up8(4 * cw);
        (void) hipFree(w.d_tabs);
        w.d_tabs = nullptr; w.cap_tabs = 0;
        w.d_cbase = w.d_cnwin = w.d_chi = w.d_win_chrom = nullptr; w.d_cfirst = nullptr;
        if (w.cap_win < cw) { (void) hipFree(w.d_status); w.d_status = nullptr; }      // (cleared before every launch: its own allocation)
        w.cap_chrom = 0; w.cap_win = 0;
        WT_HIP(hipMalloc((void **) &w.d_tabs, (size_t) total));
        if (!w.d_status) WT_HIP(hipMalloc((void **) &w.d_status, sizeof(unsigned long long) * (size_t) cw));
        w.cap_tabs = total;
        w.d_cbase = (int32_t *) w.d_tabs; w.d_cnwin = (int32_t *) (w.d_tabs + o_cnwin); w.d_chi = (int32_t *) (w.d_tabs + o_chi);
        w.d_cfirst = (int64_t *) (w.d_tabs + o_cfirst); w.d_win_chrom = (int32_t *) (w.d_tabs + o_win);
        w.cap_chrom = cc; w.cap_win = cw;
    }
    WT_HIP(wt_grow(&w.d_widx, &w.cap_widx, nwidx));
    WT_HIP(wt_grow(&w.d_cidx, &w.cap_cidx, ((w.tab.n_rows > 0 ? w.tab.n_rows : 1) + WT_ISEARCH_ROWS - 1) / WT_ISEARCH_ROWS * (int64_t) ts->n_tracks));
    if (ts->n_chrom > 0) {
        if (ts->pipe_mode) {
            if (ts->n_chrom != 1) return wt_fail(WTAMD_ERR_INTERNAL, "pipeline slots hold one chromosome");
            if (!w.h_tab) WT_HIP(hipHostMalloc((void **) &w.h_tab, 64, hipHostMallocDefault));
            int32_t *h32 = (int32_t *) w.h_tab;
            h32[0] = w.tab.cbase[0]; h32[1] = w.tab.c_nwin[0]; h32[2] = w.tab.c_hi[0];
            w.h_tab[2] = w.tab.c_first_win[0]; w.h_tab[3] = w.tab.c_first_win[1];
            WT_HIP(hipMemcpyAsync(w.d_cbase, h32 + 0, sizeof(int32_t), hipMemcpyHostToDevice, s));
            WT_HIP(hipMemcpyAsync(w.d_cnwin, h32 + 1, sizeof(int32_t), hipMemcpyHostToDevice, s));
            WT_HIP(hipMemcpyAsync(w.d_chi, h32 + 2, sizeof(int32_t), hipMemcpyHostToDevice, s));
            WT_HIP(hipMemcpyAsync(w.d_cfirst, w.h_tab + 2, 2 * sizeof(int64_t), hipMemcpyHostToDevice, s));
            WT_HIP(hipMemsetAsync(w.d_win_chrom, 0, sizeof(int32_t) * (size_t) nwin, s));
        } else {
            // one (blocking) copy of the five tables, packed as they lie on the device
            const int64_t used = ((char *) w.d_win_chrom - w.d_tabs) + (int64_t) sizeof(int32_t) * std::max<int64_t>(w.tab.n_windows, 0);
            std::vector<char> pack((size_t) used, 0);
            memcpy(pack.data() + ((char *) w.d_cbase - w.d_tabs), w.tab.cbase.data(), sizeof(int32_t) * ts->n_chrom);
            memcpy(pack.data() + ((char *) w.d_cnwin - w.d_tabs), w.tab.c_nwin.data(), sizeof(int32_t) * ts->n_chrom);
            memcpy(pack.data() + ((char *) w.d_chi - w.d_tabs), w.tab.c_hi.data(), sizeof(int32_t) * ts->n_chrom);
            memcpy(pack.data() + ((char *) w.d_cfirst - w.d_tabs), w.tab.c_first_win.data(), sizeof(int64_t) * (ts->n_chrom + 1));
            if (w.tab.n_windows > 0) memcpy(pack.data() + ((char *) w.d_win_chrom - w.d_tabs), w.tab.win_chrom.data(), sizeof(int32_t) * w.tab.n_windows);
            WT_HIP(hipMemcpy(w.d_tabs, pack.data(), (size_t) used, hipMemcpyHostToDevice));
        }
    }
    w.tab_valid = true;
    return WTAMD_OK;
}

static void wt_fill_params(const wtamd_trackset *ts, const WtWindows *w, const WtPlan &plan, WtParams &P) {
    memset(&P, 0, sizeof(P));
    P.start = ts->d_start; P.finish = ts->d_finish; P.value = ts->d_value;
    P.seg_off = ts->d_seg_off; P.defaults = ts->d_defaults;
    P.n_chrom = ts->n_chrom; P.n_tracks = ts->n_tracks; P.n_total = ts->n_intervals;
    P.cbase = w->d_cbase; P.c_nwin = w->d_cnwin; P.c_hi = w->d_chi; P.c_first_win = w->d_cfirst;
    P.n_windows = w->tab.n_windows; P.win_chrom = w->d_win_chrom; P.widx = w->d_widx;
    P.status = w->d_status; P.counters = ts->d_counters; P.debug = ts->h_debug;
    wt_plan_to_params(plan, P);
}

static int wt_build_index(wtamd_trackset *ts, WtWindows *w, const WtPlan &plan, hipStream_t s) {
    WtParams P;
    wt_fill_params(ts, w, plan, P);
    WT_HIP(hipEventRecord(ts->ev_i0, s));
    static const int mode = [] {            // 0: search (default), 1: scan, 2: both + compare
        const char *e = getenv("WTAMD_INDEX"), *c = getenv("WTAMD_INDEX_CHECK");
        if (c && atoi(c) > 0) return 2;
        return (e && !strcmp(e, "scan")) ? 1 : 0;
    }();
    const size_t n_widx = (size_t) w->tab.n_rows * ts->n_tracks;
    auto scan = [&](uint32_t *dst) -> int {
        WtParams Q = P;
        Q.widx = dst;
        WT_HIP(hipMemsetAsync(dst, 0, sizeof(uint32_t) * n_widx, s));
        if (ts->n_intervals > 0) {
            const long long total = ts->n_intervals;
            long long blocks = (total + WT_INDEX_CHUNK - 1) / WT_INDEX_CHUNK;
            const long long cap = (long long) ts->num_cu * 8;      // 8 x 256 lanes = a full CU
            if (blocks > cap) blocks = cap;
            // contiguous span per block, a whole number of sub-chunks
            long long span = (total + blocks - 1) / blocks;
            span = (span + WT_INDEX_CHUNK - 1) / WT_INDEX_CHUNK * WT_INDEX_CHUNK;
            blocks = (total + span - 1) / span;
            hipLaunchKernelGGL(wt_index_kernel, dim3((unsigned) blocks), dim3(256), 0, s, Q, total, span);
            WT_HIP(hipGetLastError());
        }
        return WTAMD_OK;
    };
    auto search = [&](uint32_t *dst) -> int {
        WtParams Q = P;
        Q.widx = dst;
        const long long n_rows = w->tab.n_rows;
        if (n_rows > 0 && ts->n_tracks > 0) {
            const long long n_strips = (n_rows + WT_ISEARCH_ROWS - 1) / WT_ISEARCH_ROWS;
            if (w->cap_cidx < n_strips * ts->n_tracks) return wt_fail(WTAMD_ERR_INTERNAL, "coarse window index not allocated");
            hipLaunchKernelGGL(wt_index_coarse_kernel, dim3((unsigned) ((n_strips * ts->n_tracks + 255) / 256)), dim3(256), 0, s, Q, n_strips, w->d_cidx);
            WT_HIP(hipGetLastError());
            const dim3 grid((unsigned) n_strips, (unsigned) ((ts->n_tracks + WT_ISEARCH_TRACKS - 1) / WT_ISEARCH_TRACKS));
            hipLaunchKernelGGL(wt_index_search_kernel, grid, dim3(WT_ISEARCH_ROWS * WT_ISEARCH_TRACKS), 0, s, Q, n_rows, n_strips, (const uint32_t *) w->d_cidx);
            WT_HIP(hipGetLastError());
        }
        return WTAMD_OK;
    };
    if (mode == 1) {
        if (int rc = scan(w->d_widx)) return rc;
    } else {
        if (int rc = search(w->d_widx)) return rc;
    }
    if (mode == 2 && n_widx > 0) {
        uint32_t *other = nullptr;
        unsigned long long *d_diff = nullptr, h_diff = 0;
        WT_HIP(hipMalloc(&other, sizeof(uint32_t) * n_widx));
        WT_HIP(hipMalloc(&d_diff, sizeof(unsigned long long)));
        WT_HIP(hipMemsetAsync(d_diff, 0, sizeof(unsigned long long), s));
        if (int rc = scan(other)) return rc;
        hipLaunchKernelGGL(wt_index_compare_kernel, dim3((unsigned) ((n_widx + 255) / 256)), dim3(256), 0, s, w->d_widx, other, (long long) n_widx, d_diff);
        WT_HIP(hipMemcpyAsync(&h_diff, d_diff, sizeof(h_diff), hipMemcpyDeviceToHost, s));
        WT_HIP(hipStreamSynchronize(s));
        (void) hipFree(other); (void) hipFree(d_diff);
        if (h_diff) {
            fprintf(stderr, "wiggletools_amd: WTAMD_INDEX_CHECK: the searched window index differs from the scanned one in %llu of %zu entries\n", h_diff, n_widx);
            return wt_fail(WTAMD_ERR_INTERNAL, "window index check failed");
        }
    }
    WT_HIP(hipEventRecord(ts->ev_i1, s));
    ts->have_index_time = true;
    w->indexed = true;
    return WTAMD_OK;
}

// The plan a reduction of `op` will run with first: the exact difference-array plan for Sum /
// Mean over float tracks with zero defaults (until a window of this data proved inexact), else the
// general bitmap plan.
static inline int wt_delta_class(int op) { return op == WT_OP_TTEST ? 1 : ((op == WT_OP_MAX || op == WT_OP_MIN) ? 2 : 0); }
static bool wt_wants_delta(const wtamd_trackset *ts, int op) {
    return !ts->delta_failed_[wt_delta_class(op)] && wt_delta_eligible(op, ts->value_f64, ts->n_tracks, ts->defaults.data());
}

// Events (run starts, plus a finish wherever a gap follows) per base pair the track set is expected to hold: the walking
// kernel sizes its per-position slots from it.  Contiguous runs assumed for a quarter of them to be followed by a gap.
static double wt_events_per_bp(const wtamd_trackset *ts) {
    const int N = ts->n_tracks;
    double span = 0;
    for (int c = 0; c < ts->n_chrom; c++) {
        int64_t lo = INT64_MAX, hi = INT64_MIN;
        for (int i = 0; i < N; i++) {
            const size_t s = (size_t) c * N + i;
            if (s + 1 >= ts->seg_off.size() || ts->seg_off[s + 1] <= ts->seg_off[s] || s >= ts->first_start.size()) continue;
            lo = std::min<int64_t>(lo, ts->first_start[s]);
            hi = std::max<int64_t>(hi, ts->last_finish[s]);
        }
        if (hi > lo) span += (double) (hi - lo);
    }
    return span > 0 ? 1.25 * (double) ts->n_intervals / span : 0.0;
}

// MWUReduction has two kernels, and which one is faster is a matter of the VALUES (round 6; chromosome 21, 50 v 50): the register
// columns of the bitmap kernel sort every position from scratch, 35.8 ms whatever the values are; the walking kernel (wt_mwalk.h)
// keeps its state from position to position and pays for every group of equal values across the two sets -- 28.5 ms on
// full-mantissa values (no position has one), 43.8 ms on the generator's 800 levels (nearly every position has one).  So the
// track set's values are SAMPLED once (4096 of them, evenly strided over the value column; one strided copy and a stream
// synchronisation per track set and index epoch): when nearly all of the sample's values are distinct, equal values at one position are
// rare and the walk is taken.  Real coverage signal with its many exact zeros and small integers stays on the register columns.
// WTAMD_MWALK=1 / 0: always / never.  Both kernels are bit-identical on every input (tests/test_gpu_parity.py::test_gpu_mwu_walk_paths).
static bool wt_mwu_few_ties(wtamd_trackset *ts, hipStream_t stream) {
    if (ts->mwu_few_ties >= 0) return ts->mwu_few_ties != 0;
    ts->mwu_few_ties = 0;
    if (hipStreamSynchronize(stream) != hipSuccess) { (void) hipGetLastError(); return false; }      // (the values may still be on their way on this stream)
    const int64_t n = ts->n_intervals;
    if (n < 1024 || ts->value_f64 || !ts->d_value) return false;      // (a small sample says little: the register columns)
    const int64_t S = std::min<int64_t>(4096, n), stride = n / S;
    std::vector<uint32_t> v((size_t) S);
    if (hipMemcpy2D(v.data(), 4, ts->d_value, (size_t) stride * 4, 4, (size_t) S, hipMemcpyDeviceToHost) != hipSuccess) { (void) hipGetLastError(); return false; }
    std::sort(v.begin(), v.end());
    const int64_t distinct = (int64_t) (std::unique(v.begin(), v.end()) - v.begin());
    ts->mwu_few_ties = distinct * 100 >= S * 97 ? 1 : 0;
    return ts->mwu_few_ties != 0;
}

// The general (non difference-array) plan: MedianReduction over float tracks walks (wt_walk.h), everything else
// -- and the median with WTAMD_NO_WALK=1, or when a Multiplexer tile is wanted -- takes the bitmap kernel.
static bool wt_pick_plan(wtamd_trackset *ts, int op, int n_set0, WtPlan &plan, std::string &err, hipStream_t stream = nullptr) {
    if (op == WT_OP_MEDIAN && !ts->value_f64 && !getenv("WTAMD_NO_WALK"))
        if (const int nr = wt_regcol_slots(ts->n_tracks, op, ts->scratch_f32, n_set0))
            if (wt_make_walk_plan(plan, ts->n_tracks, nr, wt_events_per_bp(ts))) return true;
    // MWUReduction by walking (wt_mwalk.h): when the values say so (wt_mwu_few_ties) -- on the generator's 800 levels it needs
    // 284 wave-wide VALU instructions per output run where the bitmap kernel's register columns need 253 (chromosome 21:
    // 43.9 against 35.6 ms; profiles/r05_mwu_walk_vs_bitmap.json)
    // (same domain as the register columns, wt_regcol_slots: float tracks, float-exact defaults, at most 64 per set).
    static const int mwalk = getenv("WTAMD_MWALK") ? (atoi(getenv("WTAMD_MWALK")) != 0 ? 1 : 0) : -1;      // (-1: by the data)
    if (op == WT_OP_MWU && mwalk != 0 && !ts->value_f64 && !getenv("WTAMD_NO_WALK") && (mwalk == 1 || wt_mwu_few_ties(ts, stream)))
        if (const int nr = wt_regcol_slots(ts->n_tracks, op, ts->scratch_f32, n_set0))
            if (wt_make_walk_plan(plan, ts->n_tracks, nr, wt_events_per_bp(ts), 160 * 1024, n_set0)) return true;
    return wt_make_plan(ts->n_tracks, op, ts->scratch_f32, plan, err, 80 * 1024, 160 * 1024, n_set0);
}

int wtamd_trackset_index(wtamd_trackset *ts, int op, void *stream) {
    if (!ts) return wt_fail(WTAMD_ERR_ARG, "ts == NULL");
    // an explicit re-index means the run lists may have been rewritten in place (zero-copy track
    // sets): what was learnt about their values is void, the next Sum / Mean verifies again
    for (int q = 0; q < 3; q++) { ts->delta_verified_[q] = false; ts->delta_failed_[q] = false; ts->delta_n_bad_[q] = 0; }
    ts->mwu_few_ties = -1;
    for (auto &kv : ts->windows) kv.second.indexed = false;     // every width's index describes the old data
    if (!ts->owns && !ts->pipe_mode) {
        // zero-copy track set rewritten in place: its runs may start earlier / end later than
        // before, so the extents, their check and every width's window tables are rebuilt too
        int rce = wt_refresh_extents_device(ts);
        if (rce != WTAMD_OK) return rce;
        rce = wt_check_extents(ts);
        if (rce != WTAMD_OK) return rce;
        for (auto &kv : ts->windows) kv.second.tab_valid = false;
    }
    WtPlan plan;
    std::string err;
    if (wt_wants_delta(ts, op)) wt_make_delta_plan_for(plan, ts->n_tracks, op);
    // (two-sample ops: the index only depends on the window width; the usual even split is assumed)
    else if (!wt_pick_plan(ts, op, ts->n_tracks / 2, plan, err, (hipStream_t) stream)) return wt_fail(WTAMD_ERR_ARG, err);
    WtWindows *w = nullptr;
    int rc = wt_get_windows(ts, plan.W, &w, (hipStream_t) stream);
    if (rc != WTAMD_OK) return rc;
    return wt_build_index(ts, w, plan, (hipStream_t) stream);
}

}  // extern "C"

// Launch functor for wt_dispatch
struct WtLaunch {
    WtParams P;
    int T = 0, lds = 0, grid = 0;
    bool small_tiles = false;           // difference-array Sum / Mean: the pass in 128-run tiles (wt_launch_delta)
    hipStream_t stream = nullptr;
    int num_cu = 256;
    char **gscratch = nullptr;
    size_t *gscratch_bytes = nullptr;
    hipError_t err = hipSuccess;

    template <int OP, class ValT, class ScrT, int K, bool MULTI, int NR = 0>
    void run() {
        auto kern = wt_reduce_kernel<OP, ValT, ScrT, K, MULTI, NR>;
        if (lds > 48 * 1024) {
            err = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (err != hipSuccess) return;
        }
        int per_cu = 0;
        err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, T, (size_t) lds);
        if (err != hipSuccess) return;
        if (per_cu < 1) per_cu = 1;
        long long g = (long long) num_cu * per_cu;
        if (g > P.n_windows) g = P.n_windows;
        if (g < 1) g = 1;
        if (P.g_scratch_slab || P.g_attr_slab) {    // one global slab per resident workgroup
            if (P.g_scratch_slab && g > 2ll * num_cu) g = 2ll * num_cu;
            const size_t need = (size_t) g * (size_t) (P.g_scratch_slab + P.g_attr_slab);
            if (*gscratch_bytes < need) {
                (void) hipFree(*gscratch);      // synchronises with earlier launches
                *gscratch = nullptr; *gscratch_bytes = 0;
                err = hipMalloc((void **) gscratch, need);
                if (err != hipSuccess) return;
                *gscratch_bytes = need;
            }
            P.g_scratch = *gscratch;
        }
        grid = (int) g;
        hipLaunchKernelGGL(kern, dim3((unsigned) grid), dim3((unsigned) T), (size_t) lds, stream, P);
        err = hipGetLastError();
    }
};

template <int OP, int K, bool MULTI>
static hipError_t wt_launch_patch_t(const WtParams &P, const WtPatchArgs &Q, int T, int lds, int num_cu, long long n_bad,
                                    hipStream_t s) {
    auto kern = wt_patch_kernel<OP, K, MULTI>;
    hipError_t e = hipSuccess;
    if (lds > 48 * 1024) {
        e = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, T, (size_t) lds);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    long long g = (long long) num_cu * per_cu;
    if (g > n_bad * Q.ratio) g = n_bad * Q.ratio;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned) g), dim3((unsigned) T), (size_t) lds, s, P, Q);
    return hipGetLastError();
}

// csrc/wt_walk.hip (its own translation unit: the three instantiations of the kernel take minutes to compile)
hipError_t wt_walk_launch(WtParams &P, int nr, int T, int lds, int num_cu, char **gscratch, size_t *gscratch_bytes, hipStream_t s, int *grid);

template <int OP, bool DF = false>
static void wt_launch_delta(WtLaunch &L) {
    // 128-run tiles when a window holds fewer than 8 of the 256-run ones per wavefront (round 6: the last round of tiles
    // leaves wavefronts idle -- mean run 64: 50 tiles over 16 wavefronts, -6.5 % with the small ones; mean run 200 -2.5 %; mean run 16,
    // 12.5 per wavefront: +1 %, so the large ones stay there).  WTAMD_DELTA_U=2 / 4 forces one.
    // (the t-test's 2048-bp windows: 50 tiles over 12 wavefronts, -6 % with the small ones; the variance family measured +-0 at mean run 16
    //  and +2.5 % at 200 with them and keeps the large ones; Max / Min have a pass of their own, wt_delta_pass_mm)
    constexpr bool TWO = OP == WT_OP_SUM || OP == WT_OP_MEAN || OP == WT_OP_TTEST;
    auto kern = (TWO && L.small_tiles) ? wt_delta_kernel<OP, DF, TWO ? 2 : WT_DELTA_U> : wt_delta_kernel<OP, DF, WT_DELTA_U>;
    // (the attribute and the occupancy query once per instantiation, device and launch shape: they are host calls of 50-150 us each,
    //  and they sat between the event that starts the reduction's clock and the launch -- round 6: the bench's events read 0.12-0.28 ms
    //  more per launch than rocprofv3's kernel durations)
    static std::mutex mu;
    static std::map<std::tuple<int, int, int>, int> known;
    int dev = 0;
    (void) hipGetDevice(&dev);
    int per_cu = 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = known.find(std::make_tuple(dev, L.T + (L.small_tiles ? 1 : 0), L.lds));
        if (it != known.end()) per_cu = it->second;
    }
    if (per_cu == 0) {
        if (L.lds > 48 * 1024) {
            L.err = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, L.lds);
            if (L.err != hipSuccess) return;
        }
        L.err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, L.T, (size_t) L.lds);
        if (L.err != hipSuccess) return;
        if (per_cu < 1) per_cu = 1;
        std::lock_guard<std::mutex> lk(mu);
        known[std::make_tuple(dev, L.T + (L.small_tiles ? 1 : 0), L.lds)] = per_cu;
    }
    long long g = (long long) L.num_cu * per_cu;
    if (g > L.P.n_windows) g = L.P.n_windows;
    if (g < 1) g = 1;
    L.grid = (int) g;
    hipLaunchKernelGGL(kern, dim3((unsigned) L.grid), dim3((unsigned) L.T), (size_t) L.lds, L.stream, L.P);
    L.err = hipGetLastError();
}

extern "C" {

static int wt_check_desc(const wtamd_trackset *ts, const wtamd_reduce_desc *d) {
    if (!ts || !d) return wt_fail(WTAMD_ERR_ARG, "NULL argument");
    if (d->op < 0 || d->op >= WTAMD_OP_COUNT_) return wt_fail(WTAMD_ERR_ARG, "unknown op");
    if (d->op == WTAMD_OP_TTEST) {
        // message + precondition of reference setComparisons.c:123-128
        if (d->n_set0 < 3 || ts->n_tracks - d->n_set0 < 3)
            return wt_fail(WTAMD_ERR_ARG, "The t-test function only works for two sets with enough elements to compute variance");
    } else if (d->op == WTAMD_OP_MWU) {
        // reference setComparisons.c:374-377
        if (d->n_set0 < 1 || ts->n_tracks - d->n_set0 < 1)
            return wt_fail(WTAMD_ERR_ARG, "The Mann-Whitney U function only works for two non-empty sets");
    }
    return WTAMD_OK;
}

static int wt_reduce_plan(wtamd_trackset *ts, const WtPlan &plan, int op, uint32_t flags, int n_set0, wtamd_runs *runs,
                          double *d_tile, uint8_t *d_inplay, int64_t *n_runs, hipStream_t s);

// The general kernel over the `n_bad` windows the difference-array launch (window width delta_W,
// just finished or still running on `s`) recorded as not provably exact.
static int wt_launch_patch(wtamd_trackset *ts, int delta_W, int op, uint32_t flags, int n_set0, wtamd_runs *runs, long long n_bad,
                           hipStream_t s) {
    WtPlan plan;
    std::string err;
    if (!wt_make_plan(ts->n_tracks, op, ts->scratch_f32, plan, err)) return wt_fail(WTAMD_ERR_INTERNAL, err);
    if (plan.scratch_slab > 0 || plan.W > delta_W || delta_W % plan.W != 0 || delta_W / plan.W > WT_BAD_SUB || !ts->scratch_f32 || ts->value_f64)
        return wt_fail(WTAMD_ERR_INTERNAL, "no general plan compatible with the difference-array windows (general W " + std::to_string(plan.W) + ", difference-array W " +
                       std::to_string(delta_W) + ", slab " + std::to_string((long long) plan.scratch_slab) + ", float staging " + std::to_string((int) ts->scratch_f32) + ", f64 values " +
                       std::to_string((int) ts->value_f64) + ")");
    WtWindows *dw = nullptr, *w = nullptr;
    int rc = wt_get_windows(ts, delta_W, &dw, s);
    if (rc != WTAMD_OK) return rc;
    rc = wt_get_windows(ts, plan.W, &w, s);
    if (rc != WTAMD_OK) return rc;
    WtParams P;
    wt_fill_params(ts, w, plan, P);
    P.op = op; P.flags = flags; P.n_set0 = op == WT_OP_TTEST ? n_set0 : 0;
    P.capacity = runs->capacity;
    P.o_start = runs->start; P.o_finish = runs->finish; P.o_value = runs->value;
    P.chrom_run_off = runs->chrom_run_off ? runs->chrom_run_off : ts->d_chrom_run_off;
    WtPatchArgs Q;
    Q.bad_list = dw->d_bad_list; Q.bad_goff = dw->d_bad_goff;
    Q.n_bad = ts->d_counters + WT_CTR_DELTA_BAD;
    Q.d_win_chrom = dw->d_win_chrom; Q.d_c_first_win = dw->d_cfirst;
    Q.ratio = delta_W / plan.W;
    if (!w->indexed) {
        // no index at this window width yet: only the rows of the recorded windows are filled in (w->indexed stays false)
        long long blocks = (n_bad * (Q.ratio + 1) * ts->n_tracks + 255) / 256;
        if (blocks > 4ll * ts->num_cu) blocks = 4ll * ts->num_cu;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(wt_patch_index_kernel, dim3((unsigned) blocks), dim3(256), 0, s, P, Q);
        WT_HIP(hipGetLastError());
    }
    const bool multi = plan.n_chunks > 1;
    hipError_t e;
#define WT_PATCH_GO(OPC, KK, MM) e = wt_launch_patch_t<OPC, KK, MM>(P, Q, plan.T, plan.lds_bytes, ts->num_cu, n_bad, s)
    if (op == WT_OP_SUM) {
        if (plan.ppt == 4) { if (multi) WT_PATCH_GO(WT_OP_SUM, 4, true); else WT_PATCH_GO(WT_OP_SUM, 4, false); }
        else { if (multi) WT_PATCH_GO(WT_OP_SUM, 1, true); else WT_PATCH_GO(WT_OP_SUM, 1, false); }
    } else if (op == WT_OP_MEAN) {
        if (plan.ppt == 4) { if (multi) WT_PATCH_GO(WT_OP_MEAN, 4, true); else WT_PATCH_GO(WT_OP_MEAN, 4, false); }
        else { if (multi) WT_PATCH_GO(WT_OP_MEAN, 1, true); else WT_PATCH_GO(WT_OP_MEAN, 1, false); }
    } else if (op == WT_OP_MAX || op == WT_OP_MIN) {
        if (plan.ppt != 4) return wt_fail(WTAMD_ERR_INTERNAL, "no general plan compatible with the difference-array windows");
        if (op == WT_OP_MAX) { if (multi) WT_PATCH_GO(WT_OP_MAX, 4, true); else WT_PATCH_GO(WT_OP_MAX, 4, false); }
        else { if (multi) WT_PATCH_GO(WT_OP_MIN, 4, true); else WT_PATCH_GO(WT_OP_MIN, 4, false); }
    } else if (op == WT_OP_TTEST) {
        if (plan.ppt != 4) return wt_fail(WTAMD_ERR_INTERNAL, "no general plan compatible with the difference-array windows");
        if (multi) WT_PATCH_GO(WT_OP_TTEST, 4, true); else WT_PATCH_GO(WT_OP_TTEST, 4, false);
    } else {
        // var / stddev / entropy / CV: 4 positions per lane only (what the plans pick unless forced)
        if (plan.ppt != 4) return wt_fail(WTAMD_ERR_INTERNAL, "no general plan compatible with the difference-array windows");
        if (op == WT_OP_VAR) { if (multi) WT_PATCH_GO(WT_OP_VAR, 4, true); else WT_PATCH_GO(WT_OP_VAR, 4, false); }
        else if (op == WT_OP_CV) { if (multi) WT_PATCH_GO(WT_OP_CV, 4, true); else WT_PATCH_GO(WT_OP_CV, 4, false); }
        else { if (multi) WT_PATCH_GO(WT_OP_STDDEV, 4, true); else WT_PATCH_GO(WT_OP_STDDEV, 4, false); }
    }
#undef WT_PATCH_GO
    if (e != hipSuccess) return wt_fail(WTAMD_ERR_HIP, std::string("patch kernel launch: ") + hipGetErrorString(e));
    WT_HIP(hipEventRecord(ts->ev_r1, s));       // the reduction's time includes its patches
    ts->stats.patched_windows = (int32_t) n_bad;
    return WTAMD_OK;
}

static int wt_reduce_impl(wtamd_trackset *ts, int op, uint32_t flags, int n_set0, wtamd_runs *runs,
                          double *d_tile, uint8_t *d_inplay, int64_t *n_runs, hipStream_t s) {
    if (!runs || !runs->start || !runs->finish || !runs->value)
        return wt_fail(WTAMD_ERR_ARG, "wtamd_reduce: output arrays missing");
    WtPlan plan;
    std::string err;
    // Sum / Mean: exact difference-array kernel first.  It verifies every window.  A few windows
    // it cannot prove exact (NaN, Inf, too wide a dynamic range) keep their coordinates and get
    // their values from the general kernel restricted to them (wt_patch_kernel); if they are many
    // the whole launch is redone by the general kernel and this data stays on it.  The verdict
    // depends on the data and the windows only (not on the op or its flags), so it is established
    // once per track set -- that first launch is waited for even when the caller asked for an
    // asynchronous one -- and later launches (patch included) need no host round trip.
    if (!d_tile && wt_wants_delta(ts, op)) {
        wt_make_delta_plan_for(plan, ts->n_tracks, op);
        const int dc = wt_delta_class(op);
        int64_t n_probe = 0;
        const bool probe = !ts->delta_verified_[dc];
        const int rc = wt_reduce_plan(ts, plan, op, flags, n_set0, runs, d_tile, d_inplay,
                                      (probe && !n_runs) ? &n_probe : n_runs, s);
        if (!probe) {
            if (ts->delta_n_bad_[dc] > 0 && (rc == WTAMD_OK || rc == WTAMD_ERR_CAPACITY)) {
                const int rp = wt_launch_patch(ts, plan.W, op, flags, n_set0, runs, ts->delta_n_bad_[dc], s);
                if (rp != WTAMD_OK) return rp;
                if (n_runs) WT_HIP(hipStreamSynchronize(s));
            }
            return rc;
        }
        if (rc != WTAMD_OK && rc != WTAMD_ERR_CAPACITY) return rc;
        const long long n_bad = (long long) ts->h_counters[WT_CTR_DELTA_BAD];
        if (n_bad == 0) { ts->delta_verified_[dc] = true; ts->delta_n_bad_[dc] = 0; return rc; }
        if (n_bad * 4 <= (long long) ts->stats.n_windows && !getenv("WTAMD_NO_PATCH")) {
            const int rp = wt_launch_patch(ts, plan.W, op, flags, n_set0, runs, n_bad, s);
            if (rp == WTAMD_OK) {
                WT_HIP(hipStreamSynchronize(s));
                ts->delta_verified_[dc] = true;
                ts->delta_n_bad_[dc] = n_bad;
                return rc;
            }
            if (rp != WTAMD_ERR_INTERNAL) return rp;        // INTERNAL: no compatible general plan -> full redo
        }
        ts->delta_failed_[dc] = true;
    }
    if (!wt_pick_plan(ts, op, n_set0, plan, err, s)) return wt_fail(WTAMD_ERR_ARG, err);
    return wt_reduce_plan(ts, plan, op, flags, n_set0, runs, d_tile, d_inplay, n_runs, s);
}

static int wt_reduce_plan(wtamd_trackset *ts, const WtPlan &plan, int op, uint32_t flags, int n_set0, wtamd_runs *runs,
                          double *d_tile, uint8_t *d_inplay, int64_t *n_runs, hipStream_t s) {
    if (plan.T > (plan.delta ? ((wt_op_is_var_family(op) || op == WT_OP_TTEST) ? WT_DELTA_SQ_BLOCK : WT_DELTA_BLOCK) : WT_MAX_BLOCK)) return wt_fail(WTAMD_ERR_ARG, "workgroup size above the kernel's launch bound");
    WtWindows *w = nullptr;
    int rc = wt_get_windows(ts, plan.W, &w, s);
    if (rc != WTAMD_OK) return rc;
    if (!w->indexed) {
        rc = wt_build_index(ts, w, plan, s);
        if (rc != WTAMD_OK) return rc;
    }
    WtLaunch L;
    wt_fill_params(ts, w, plan, L.P);
    L.P.op = op; L.P.flags = flags; L.P.n_set0 = n_set0;
    L.P.capacity = runs->capacity;
    L.P.o_start = runs->start; L.P.o_finish = runs->finish; L.P.o_value = runs->value;
    L.P.chrom_run_off = runs->chrom_run_off ? runs->chrom_run_off : ts->d_chrom_run_off;
    L.P.o_tile = d_tile; L.P.o_inplay = d_inplay;
    L.T = plan.T; L.lds = plan.lds_bytes; L.stream = s; L.num_cu = ts->num_cu;
    L.gscratch = &ts->d_gscratch; L.gscratch_bytes = &ts->gscratch_bytes;
    if (op == WT_OP_MWU && !getenv("WTAMD_MWU_DEVICE_ERF")) {
        const int n1 = n_set0, n2 = ts->n_tracks - n_set0;
        if (ts->mwu_n1 != n1 || ts->mwu_n2 != n2) {
            // (a table per pair of set sizes; the previous pair's table may still be read by a launch in flight on ANY stream of
            //  this track set: it is retired, not freed -- wtamd_trackset_destroy frees them all)
            std::vector<double> t;
            const bool have = wt_mwu_make_table(n1, n2, t);
            if (ts->d_mwu_table) { ts->mwu_retired.push_back(ts->d_mwu_table); ts->d_mwu_table = nullptr; }
            if (have) {
                WT_HIP(hipMalloc((void **) &ts->d_mwu_table, sizeof(double) * t.size()));
                WT_HIP(hipMemcpyAsync(ts->d_mwu_table, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice, s));
                WT_HIP(hipStreamSynchronize(s));                // (t is a local)
            }
            ts->mwu_n1 = n1; ts->mwu_n2 = n2; ts->mwu_kmax = have ? (int) t.size() - 1 : 0;
        }
        L.P.mwu_table = ts->d_mwu_table;
        L.P.mwu_kmax = ts->mwu_kmax;
    }
    if (plan.delta) {
        const int64_t nwin = w->tab.n_windows > 0 ? w->tab.n_windows : 1;
        if (w->cap_bad < nwin) {
            int64_t c1 = w->cap_bad, c2 = w->cap_bad;
            WT_HIP(wt_grow(&w->d_bad_list, &c1, nwin));
            WT_HIP(wt_grow(&w->d_bad_goff, &c2, nwin * WT_BAD_SUB));
            w->cap_bad = c1 < c2 / WT_BAD_SUB ? c1 : c2 / WT_BAD_SUB;
        }
        L.P.bad_list = w->d_bad_list;
        L.P.bad_goff = w->d_bad_goff;
        wt_delta_defaults_params(ts->defaults.data(), ts->n_tracks, L.P);
    }

    WT_HIP(hipMemsetAsync(ts->d_counters, 0, sizeof(unsigned long long) * WT_CTR_N, s));
    WT_HIP(hipMemsetAsync(L.P.chrom_run_off, 0, sizeof(int64_t) * (ts->n_chrom + 1), s));
    if (w->tab.n_windows > 0 && ts->n_intervals > 0) {
        WT_HIP(hipMemsetAsync(w->d_status, 0, sizeof(unsigned long long) * w->tab.n_windows, s));
        if (plan.delta) {
            // (tiles of 256 runs per wavefront and window, on average)
            static const int force_u = getenv("WTAMD_DELTA_U") ? atoi(getenv("WTAMD_DELTA_U")) : 0;
            const double tiles_per_wave = (double) ts->n_intervals / (double) w->tab.n_windows / 256.0 / (double) std::max(1, L.T / 64);
            L.small_tiles = force_u == 2 || (force_u != 4 && tiles_per_wave < 8.0);
        }
        WT_HIP(hipEventRecord(ts->ev_r0, s));
        if (plan.delta) {
            switch (op) {
            case WT_OP_SUM: if (L.P.delta_df) wt_launch_delta<WT_OP_SUM, true>(L); else wt_launch_delta<WT_OP_SUM>(L); break;
            case WT_OP_MEAN: if (L.P.delta_df) wt_launch_delta<WT_OP_MEAN, true>(L); else wt_launch_delta<WT_OP_MEAN>(L); break;
            case WT_OP_VAR: wt_launch_delta<WT_OP_VAR>(L); break;
            case WT_OP_CV: wt_launch_delta<WT_OP_CV>(L); break;
            case WT_OP_TTEST: wt_launch_delta<WT_OP_TTEST>(L); break;
            case WT_OP_MAX: wt_launch_delta<WT_OP_MAX>(L); break;
            case WT_OP_MIN: wt_launch_delta<WT_OP_MIN>(L); break;
            default: wt_launch_delta<WT_OP_STDDEV>(L); break;      // stddev, entropy (reducers.c:665)
            }
        } else if (plan.walk_S) {
            L.err = wt_walk_launch(L.P, plan.regcol, L.T, L.lds, L.num_cu, L.gscratch, L.gscratch_bytes, L.stream, &L.grid);
        } else if (!wt_dispatch(op, ts->value_f64, ts->scratch_f32, plan.ppt, plan.n_chunks > 1 || plan.scratch_slab > 0, L, plan.regcol)) {
            return wt_fail(WTAMD_ERR_ARG, "op not dispatchable");
        }
        if (L.err != hipSuccess) return wt_fail(WTAMD_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(L.err));
        WT_HIP(hipEventRecord(ts->ev_r1, s));
        ts->have_reduce_time = true;
    }
    ts->stats.n_windows = w->tab.n_windows;
    ts->stats.window_bp = plan.W;
    ts->stats.lds_bytes = plan.lds_bytes;
    ts->stats.kernel = plan.delta ? 1 : (plan.walk_S ? (plan.walk_mwu ? 3 : 2) : 0);     // (2: median by walking, csrc/wt_walk.h; 3: MWU by walking, csrc/wt_mwalk.h)
    ts->stats.patched_windows = 0;
    if (n_runs) {
        WT_HIP(hipMemcpyAsync(ts->h_counters, ts->d_counters, sizeof(unsigned long long) * WT_CTR_N, hipMemcpyDeviceToHost, s));
        {
            // bounded wait: a kernel that does not finish is reported, never waited for forever
            const double limit_s = getenv("WTAMD_TIMEOUT_S") ? atof(getenv("WTAMD_TIMEOUT_S")) : 120.0;
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                const hipError_t q = hipStreamQuery(s);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) return wt_fail(WTAMD_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (el > limit_s) {
                    char buf[320];
                    snprintf(buf, sizeof buf, "kernel did not finish within %.0f s (debug marker %llu, window %lld; per wave %llu %llu %llu %llu %llu %llu %llu %llu)",
                             limit_s, ts->h_debug[0], (long long) ts->h_debug[1], ts->h_debug[2], ts->h_debug[3], ts->h_debug[4],
                             ts->h_debug[5], ts->h_debug[6], ts->h_debug[7], ts->h_debug[8], ts->h_debug[9]);
                    // a kernel that never finishes cannot be cancelled and every later HIP call of this
                    // process (even hipFree) would block behind it: report and terminate the process
                    fprintf(stderr, "wiggletools_amd: FATAL: %s\n", buf);
                    fflush(stderr);
                    _exit(70);
                }
                if (el > 0.002) std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
        }
        ts->stats.n_runs = (int64_t) ts->h_counters[WT_CTR_RUNS];
        ts->stats.covered_bp = (int64_t) ts->h_counters[WT_CTR_BP];
        ts->stats.n_intervals = (int64_t) ts->h_counters[WT_CTR_INTERVALS];
        *n_runs = ts->stats.n_runs;
        if (plan.walk_S) {      // (-DWT_PROFILE builds of wt_walk.hip: cycles of wave 0 per phase, summed over the workgroups)
            static const bool show = getenv("WTAMD_WALK_PROF") != nullptr;
            if (show) {
                static const char *names_w[8] = {"zero+ranges", "count", "offsets", "scatter", "events", "first-median", "moves", "rest"};
                unsigned long long tot = 0;
                for (int q = 0; q < 8; q++) tot += ts->h_counters[WT_CTR_PROF + q];
                fprintf(stderr, "[wt_walk_profile]");
                for (int q = 0; q < 8; q++) fprintf(stderr, " %s %.1f%%", names_w[q], tot ? 100.0 * ts->h_counters[WT_CTR_PROF + q] / tot : 0.0);
                fprintf(stderr, " (total %.3g cycles)\n", (double) tot);
            }
        }
#ifdef WT_PROFILE
        {
            static const char *names_g[8] = {"zero", "load", "count", "emask+escan", "eval", "lookback", "write", "-"};
            static const char *names_d[8] = {"zero", "ranges", "pass1", "pass2", "scan", "escan", "lookback", "write"};
            const char **names = plan.delta ? names_d : names_g;
            unsigned long long tot = 0;
            for (int q = 0; q < 8; q++) tot += ts->h_counters[WT_CTR_PROF + q];
            fprintf(stderr, "[wt_profile] op %d:", op);
            for (int q = 0; q < 8; q++)
                fprintf(stderr, " %s %.1f%%", names[q], tot ? 100.0 * ts->h_counters[WT_CTR_PROF + q] / tot : 0.0);
            fprintf(stderr, " (total %.3g cycles over all workgroups)\n", (double) tot);
            unsigned long long p2[8] = {0};
            if (hipMemcpyFromSymbol(p2, HIP_SYMBOL(wt_prof2), sizeof p2) == hipSuccess && (p2[0] | p2[4])) {
                fprintf(stderr, "[wt_profile] register column, cycles summed over waves: gather %.3g  sort/park %.3g  count loop %.3g  erf %.3g  median sort+select %.3g\n",
                        (double) p2[0], (double) p2[1], (double) p2[2], (double) p2[3], (double) p2[4]);
                unsigned long long z[8] = {0};
                (void) hipMemcpyToSymbol(HIP_SYMBOL(wt_prof2), z, sizeof z);
            }
        }
#endif
        if (ts->h_counters[WT_CTR_ERROR] & WT_ERR_LOOKBACK) return wt_fail(WTAMD_ERR_INTERNAL, "look-back timed out");
        if (ts->h_counters[WT_CTR_ERROR] & WT_ERR_CAPACITY) return wt_fail(WTAMD_ERR_CAPACITY, "output capacity too small");
    }
    return WTAMD_OK;
}

int wtamd_reduce(wtamd_trackset *ts, const wtamd_reduce_desc *desc, wtamd_runs *runs, int64_t *n_runs, void *stream) {
    int rc = wt_check_desc(ts, desc);
    if (rc != WTAMD_OK) return rc;
    return wt_reduce_impl(ts, desc->op, desc->flags, desc->n_set0, runs, nullptr, nullptr, n_runs, (hipStream_t) stream);
}

static int wt_reduce_host_impl(wtamd_trackset *ts, int op, uint32_t flags, int n_set0, wtamd_runs *runs,
                               double *h_tile, uint8_t *h_inplay, int64_t *n_runs) {
    if (!runs) return wt_fail(WTAMD_ERR_ARG, "runs == NULL");
    int64_t cap = wtamd_trackset_max_runs(ts);
    if (cap > runs->capacity) cap = runs->capacity;     // never write past the caller's arrays
    const int64_t alloc = cap > 0 ? cap : 1;
    const int N = ts->n_tracks;
    WtDevScope scope;
    wtamd_runs d{};
    d.capacity = cap;
    double *d_tile = nullptr;
    uint8_t *d_inplay = nullptr;
    WT_HIP(scope.alloc(&d.start, sizeof(int32_t) * alloc));
    WT_HIP(scope.alloc(&d.finish, sizeof(int32_t) * alloc));
    WT_HIP(scope.alloc(&d.value, sizeof(double) * alloc));
    WT_HIP(scope.alloc(&d.chrom_run_off, sizeof(int64_t) * (ts->n_chrom + 1)));
    if (op == WT_OP_MULTIPLEX) {
        WT_HIP(scope.alloc(&d_tile, sizeof(double) * alloc * N));
        WT_HIP(scope.alloc(&d_inplay, sizeof(uint8_t) * alloc * N));
    }
    int64_t n = 0;
    int rc = wt_reduce_impl(ts, op, flags, n_set0, &d, d_tile, d_inplay, &n, nullptr);
    if (rc == WTAMD_OK) {
        if (n > 0) {
            WT_HIP(hipMemcpy(runs->start, d.start, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
            WT_HIP(hipMemcpy(runs->finish, d.finish, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
            if (runs->value) WT_HIP(hipMemcpy(runs->value, d.value, sizeof(double) * n, hipMemcpyDeviceToHost));
            if (h_tile) WT_HIP(hipMemcpy(h_tile, d_tile, sizeof(double) * n * N, hipMemcpyDeviceToHost));
            if (h_inplay) WT_HIP(hipMemcpy(h_inplay, d_inplay, sizeof(uint8_t) * n * N, hipMemcpyDeviceToHost));
        }
        if (runs->chrom_run_off)
            WT_HIP(hipMemcpy(runs->chrom_run_off, d.chrom_run_off, sizeof(int64_t) * (ts->n_chrom + 1), hipMemcpyDeviceToHost));
        if (n_runs) *n_runs = n;
    }
    return rc;
}

int wtamd_reduce_host(wtamd_trackset *ts, const wtamd_reduce_desc *desc, wtamd_runs *runs, int64_t *n_runs) {
    int rc = wt_check_desc(ts, desc);
    if (rc != WTAMD_OK) return rc;
    return wt_reduce_host_impl(ts, desc->op, desc->flags, desc->n_set0, runs, nullptr, nullptr, n_runs);
}

int wtamd_multiplex_host(wtamd_trackset *ts, uint32_t flags, wtamd_runs *runs, double *values, uint8_t *inplay,
                         int64_t *n_runs) {
    if (!ts || !values || !inplay) return wt_fail(WTAMD_ERR_ARG, "NULL argument");
    return wt_reduce_host_impl(ts, WT_OP_MULTIPLEX, flags, 0, runs, values, inplay, n_runs);
}

static int wt_runs_auc_span(const wtamd_runs *runs, int64_t n_runs, double *auc, double *span, void *stream) {
    hipStream_t s = (hipStream_t) stream;
    const int blocks = 512;
    double *d_partial = nullptr, h[2] = {0, 0};
    WtDevScope scope;
    WT_HIP(scope.alloc(&d_partial, sizeof(double) * (2 * blocks + 2)));
    hipLaunchKernelGGL(wt_auc_kernel, dim3(blocks), dim3(256), 0, s, runs->start, runs->finish, runs->value,
                       (long long) n_runs, d_partial, span ? d_partial + blocks : nullptr);
    hipLaunchKernelGGL(wt_auc_final_kernel, dim3(1), dim3(64), 0, s, d_partial, blocks, d_partial + 2 * blocks);
    if (span) hipLaunchKernelGGL(wt_auc_final_kernel, dim3(1), dim3(64), 0, s, d_partial + blocks, blocks, d_partial + 2 * blocks + 1);
    WT_HIP(hipGetLastError());
    WT_HIP(hipMemcpyAsync(h, d_partial + 2 * blocks, sizeof(double) * (span ? 2 : 1), hipMemcpyDeviceToHost, s));
    WT_HIP(hipStreamSynchronize(s));
    *auc = h[0];
    if (span) *span = h[1];
    return WTAMD_OK;
}

int wtamd_runs_auc(const wtamd_runs *runs, int64_t n_runs, double *auc, void *stream) {
    if (!runs || !auc) return wt_fail(WTAMD_ERR_ARG, "NULL argument");
    return wt_runs_auc_span(runs, n_runs, auc, nullptr, stream);
}

int wtamd_runs_mean(const wtamd_runs *runs, int64_t n_runs, double *mean, void *stream) {
    if (!runs || !mean) return wt_fail(WTAMD_ERR_ARG, "NULL argument");
    double sum = 0, span = 0;
    const int rc = wt_runs_auc_span(runs, n_runs, &sum, &span, stream);
    if (rc != WTAMD_OK) return rc;
    *mean = span > 0 ? sum / span : __builtin_nan("");      // statistics.c:66-68, res initialised to NAN :98
    return WTAMD_OK;
}

// result[0] = correlation, result[1..6] = {n, sum_X, sum_Y, T_XX, T_XY, T_YY}
static int wt_pearson_impl(wtamd_trackset *ts, double *result7) {
    if (ts->n_tracks != 2) return wt_fail(WTAMD_ERR_ARG, "wtamd_pearson: the track set must hold exactly two tracks");
    int64_t cap = wtamd_trackset_max_runs(ts);
    const int64_t alloc = cap > 0 ? cap : 1;
    wtamd_runs d{};
    d.capacity = cap;
    double *d_tile = nullptr;
    uint8_t *d_inplay = nullptr;
    WtMoments *d_partial = nullptr;
    double *d_out = nullptr;
    const int blocks = 256;
    WtDevScope scope;
    WT_HIP(scope.alloc(&d.start, sizeof(int32_t) * alloc));
    WT_HIP(scope.alloc(&d.finish, sizeof(int32_t) * alloc));
    WT_HIP(scope.alloc(&d.value, sizeof(double) * alloc));
    WT_HIP(scope.alloc(&d.chrom_run_off, sizeof(int64_t) * (ts->n_chrom + 1)));
    WT_HIP(scope.alloc(&d_tile, sizeof(double) * alloc * 2));
    WT_HIP(scope.alloc(&d_inplay, sizeof(uint8_t) * alloc * 2));
    WT_HIP(scope.alloc(&d_partial, sizeof(WtMoments) * blocks));
    WT_HIP(scope.alloc(&d_out, sizeof(double) * 7));
    int64_t n = 0;
    int rc = wt_reduce_impl(ts, WT_OP_MULTIPLEX, 0, 0, &d, d_tile, d_inplay, &n, nullptr);
    if (rc == WTAMD_OK) {
        hipLaunchKernelGGL(wt_pearson_kernel, dim3(blocks), dim3(256), 0, nullptr, d.start, d.finish, d_tile, d_inplay,
                           ts->defaults[0], ts->defaults[1], (long long) n, d_partial);
        hipLaunchKernelGGL(wt_pearson_final_kernel, dim3(1), dim3(64), 0, nullptr, d_partial, blocks, d_out);
        if (hipGetLastError() != hipSuccess || hipMemcpy(result7, d_out, sizeof(double) * 7, hipMemcpyDeviceToHost) != hipSuccess)
            rc = wt_fail(WTAMD_ERR_HIP, "wtamd_pearson: kernel launch / copy failed");
    }
    return rc;
}

int wtamd_pearson(wtamd_trackset *ts, double *result) {
    if (!ts || !result) return wt_fail(WTAMD_ERR_ARG, "NULL argument");
    double r[7];
    const int rc = wt_pearson_impl(ts, r);
    if (rc == WTAMD_OK) *result = r[0];
    return rc;
}

int wtamd_pearson_moments(wtamd_trackset *ts, double *moments) {
    if (!ts || !moments) return wt_fail(WTAMD_ERR_ARG, "NULL argument");
    double r[7];
    const int rc = wt_pearson_impl(ts, r);
    if (rc == WTAMD_OK) memcpy(moments, r + 1, sizeof(double) * 6);
    return rc;
}

// (wtamd_pearson_merge / wtamd_pearson_finish: host-only, csrc/wt_defaults.cpp)

int wtamd_trackset_validate(wtamd_trackset *ts, int64_t *n_bad, int64_t *first_bad) {
    if (!ts || !n_bad) return wt_fail(WTAMD_ERR_ARG, "NULL argument");
    *n_bad = 0;
    if (first_bad) *first_bad = -1;
    if (ts->n_intervals <= 0) return WTAMD_OK;
    unsigned long long h[2] = {0ull, ~0ull}, *d = nullptr;
    WtDevScope scope;
    WT_HIP(scope.alloc(&d, sizeof h));
    WT_HIP(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
    const long long total = ts->n_intervals;
    hipLaunchKernelGGL(wt_validate_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, nullptr, ts->d_start,
                       ts->d_finish, ts->d_seg_off, (long long) ts->n_chrom * ts->n_tracks, total, d);
    const hipError_t e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return wt_fail(WTAMD_ERR_HIP, "wtamd_trackset_validate: kernel / copy failed");
    *n_bad = (int64_t) h[0];
    if (first_bad && h[0]) *first_bad = (int64_t) h[1];
    return WTAMD_OK;
}

int wtamd_get_stats(const wtamd_trackset *ts_c, wtamd_stats *out) {
    if (!ts_c || !out) return wt_fail(WTAMD_ERR_ARG, "NULL argument");
    wtamd_trackset *ts = const_cast<wtamd_trackset *>(ts_c);
    if (ts->have_index_time && hipEventSynchronize(ts->ev_i1) == hipSuccess)
        (void) hipEventElapsedTime(&ts->stats.index_ms, ts->ev_i0, ts->ev_i1);
    if (ts->have_reduce_time && hipEventSynchronize(ts->ev_r1) == hipSuccess)
        (void) hipEventElapsedTime(&ts->stats.reduce_ms, ts->ev_r0, ts->ev_r1);
    *out = ts->stats;
    return WTAMD_OK;
}

}  // extern "C"

#include "wt_pipe.h"
