// wt_walk.hip -- gfx950 kernel of MedianReduction by walking (logic in wt_walk.h) and its launcher, a translation unit of
// its own next to wt_engine.hip.  Compiled only by hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>

#include "../../include/wiggletools_amd.h"
#include "wt_core.h"

#define WT_MARK(x) do { } while (0)
#ifdef WT_PROFILE
#define WT_TICK(slot) do { if (tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); \
        prof[slot] += t_ - t_last; t_last = t_; } } while (0)
#else
#define WT_TICK(slot) do { } while (0)
#endif

// MedianReduction by walking (wt_walk.h): a lane carries its column of current values from one position to the next.
// T lanes (128: two workgroups per CU at 100 tracks) x S positions each; persistent workgroups, window tickets, ordered
// output through the look-back chain like the other kernels.
template <int MAXT, bool PAIR>
__global__ void __launch_bounds__(MAXT, 1) wt_walk_kernel(const WtParams P) {
    extern __shared__ __attribute__((aligned(16))) char wt_lds[];
    WtCtx c{};
    c.sh = (WtShared *) (wt_lds + P.off_shared);
    WtDeltaCtx d;
    wt_delta_ctx_init(d, P, wt_lds);
    WtWalkCtx w;
    wt_walk_ctx_init(w, P, wt_lds, P.g_scratch + (size_t) blockIdx.x * (size_t) P.g_scratch_slab);
    w.pair = PAIR ? 1 : 0;                      // (== P.walk_pair: a constant of this instantiation from here on)
    w.mwu = 0;                                  // (... and so is this: the MWU branches of the shared phases fold away)
    const int tid = threadIdx.x, nt = blockDim.x;
    long long k_dbg = -1;
    (void) k_dbg;
#ifdef WT_PROFILE
    // cycles of wave 0: 0 zero + ranges, 1 count pass, 2 offsets, 3 scatter pass, 4 events, 5 first median, 6 moves of
    // the median (4-6: lane 0's view inside the walk), 7 everything else (run-count scan, look-back, write, ticket)
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_readcyclecounter();
#endif
    if (tid == 0) {
        const long long k0 = (long long) wt_glb_add64(&P.counters[WT_CTR_TICKET], 1ull);
        c.sh->ticket = k0;
        if (k0 < P.n_windows) wt_phase_header(P, c, k0);
    }
    wt_walk_defaults(P, w, tid, nt);
    __syncthreads();
    for (;;) {
        const long long k = c.sh->ticket;
        k_dbg = k;
        if (k >= P.n_windows) break;
        WT_MARK(201);
        wt_walk_zero(P, c, w, tid, nt);
        wt_delta_ranges_w1(P, c, d, 0, tid, nt);
        __syncthreads();
        wt_walk_ranges_w2(d, tid, nt);
        __syncthreads();
        WT_TICK(0);
        WT_MARK(202);
        wt_walk_pass<false>(P, c, w, d, 0u, 0u, tid, nt);
        __syncthreads();                        // the events are in the slab, the counts in cnt[]
        WT_TICK(1);
        WtWalkLane L;
        // positions with events / emitted runs per lane, the window's run count: published before the walk
        wt_walk_emits(P, c, w, L, tid, nt);
        wt_walk_scan_a(w, wt_walk_emit_count(w, L, tid), tid, nt);
        __syncthreads();
        wt_walk_scan_b(w, tid, nt);
        __syncthreads();
        const unsigned long long mine = w.base[nt];
        if (tid == 0) wt_lookback_publish(P, c, k, mine);
        if (w.novf[0] <= w.ov_cap) {            // (uniform) every position's events fit its slots + the overflow list
#ifdef WT_PROFILE
            wt_walk_lane<true, PAIR>(P, c, w, L, 0u, tid, nt, tid == 0 ? prof : nullptr);
            if (tid == 0) t_last = __builtin_readcyclecounter();
#else
            wt_walk_lane<true, PAIR>(P, c, w, L, 0u, tid, nt);
#endif
            __syncthreads();
            WT_TICK(7);
        } else {
            // a window denser than that: its events sorted by position into the same memory (a second pass over the
            // runs), as many lanes' worth at a time as fit
            __syncthreads();                    // (base[] is about to hold the lanes' event offsets)
            wt_walk_offsets1(P, w, tid, nt);
            __syncthreads();
            wt_walk_scan_b(w, tid, nt);
            __syncthreads();
            wt_walk_offsets2(P, w, tid, nt);
            __syncthreads();
            WT_TICK(2);
            WT_MARK(203);
            for (int l0 = 0; l0 < w.nstr;) {        // (stretches)
                const int l1 = wt_walk_round_end(w, l0, nt);
                const uint32_t ev0 = w.base[l0 << w.pair], ev1 = w.base[l1 << w.pair];
                if (ev1 > ev0) {                    // (uniform)
                    wt_walk_pass<true>(P, c, w, d, ev0, ev1, tid, nt);
                    __syncthreads();
                    WT_TICK(3);
                    if ((tid >> w.pair) >= l0 && (tid >> w.pair) < l1) wt_walk_lane<false, PAIR>(P, c, w, L, ev0, tid, nt);
                    __syncthreads();                // before the next round reuses the slab
                    WT_TICK(7);
                }
                l0 = l1;
            }
            // the lanes' run offsets again (base[] held the event offsets meanwhile)
            wt_walk_scan_a(w, wt_walk_emit_count(w, L, tid), tid, nt);
            __syncthreads();
            wt_walk_scan_b(w, tid, nt);
            __syncthreads();
        }
        WT_MARK(204);
        WT_TICK(7);
        if (tid < 64) wt_lookback_complete(P, c, k, tid, mine);
        __syncthreads();
#ifdef WT_PROFILE_LB
        WT_TICK(2);                             // (experiment: the look-back's wait alone, in the "offsets" slot)
#endif
        WT_MARK(205);
        wt_walk_write(P, c, w, L, tid, nt);
        __syncthreads();
#ifdef WT_PROFILE_LB
        WT_TICK(3);                             // (experiment: the write, in the "scatter" slot)
#endif
        if (tid == 0) {
            wt_window_stats(P, c);
            const long long kn = (long long) wt_glb_add64(&P.counters[WT_CTR_TICKET], 1ull);
            c.sh->ticket = kn;
            if (kn < P.n_windows) wt_phase_header(P, c, kn);
        }
        __syncthreads();
        WT_TICK(7);
    }
#ifdef WT_PROFILE
    if (tid == 0)
        for (int q = 0; q < 8; q++) wt_glb_add64(&P.counters[WT_CTR_PROF + q], prof[q]);
#endif
}

// MWUReduction by walking (wt_mwalk.h): the same window machinery; the two lanes of a stretch hold one SET each.  Which
// positions emit a run is only known once the events have been applied (tracks in play per set), so the window's run count
// goes to the look-back chain after the walk.
template <int MAXT>
__global__ void __launch_bounds__(MAXT, 1) wt_mwalk_kernel(const WtParams P) {
    extern __shared__ __attribute__((aligned(16))) char wt_lds[];
    WtCtx c{};
    c.sh = (WtShared *) (wt_lds + P.off_shared);
    WtDeltaCtx d;
    wt_delta_ctx_init(d, P, wt_lds);
    WtWalkCtx w;
    wt_walk_ctx_init(w, P, wt_lds, P.g_scratch + (size_t) blockIdx.x * (size_t) P.g_scratch_slab);
    w.pair = 1; w.mwu = 1;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) {
        const long long k0 = (long long) wt_glb_add64(&P.counters[WT_CTR_TICKET], 1ull);
        c.sh->ticket = k0;
        if (k0 < P.n_windows) wt_phase_header(P, c, k0);
    }
    wt_walk_defaults(P, w, tid, nt);
    __syncthreads();
    for (;;) {
        const long long k = c.sh->ticket;
        if (k >= P.n_windows) break;
        wt_walk_zero(P, c, w, tid, nt);
        wt_delta_ranges_w1(P, c, d, 0, tid, nt);
        __syncthreads();
        wt_walk_ranges_w2(d, tid, nt);
        __syncthreads();
        wt_walk_pass<false>(P, c, w, d, 0u, 0u, tid, nt);
        __syncthreads();                        // the events are in the slab, the counts in cnt[]
        WtWalkLane L;
        wt_mwalk_events(P, c, w, L, tid, nt);
        if (w.novf[0] <= w.ov_cap) {            // (uniform) every position's events fit its slots + the overflow list
            wt_mwalk_lane<true>(P, c, w, L, 0u, tid, nt);
            __syncthreads();
        } else {
            // a window denser than that: its events sorted by position into the same memory, as many stretches at a time as fit
            __syncthreads();
            wt_walk_offsets1(P, w, tid, nt);
            __syncthreads();
            wt_walk_scan_b(w, tid, nt);
            __syncthreads();
            wt_walk_offsets2(P, w, tid, nt);
            __syncthreads();
            for (int l0 = 0; l0 < w.nstr;) {
                const int l1 = wt_walk_round_end(w, l0, nt);
                const uint32_t ev0 = w.base[l0 << 1], ev1 = w.base[l1 << 1];
                if (ev1 > ev0) {                    // (uniform)
                    wt_walk_pass<true>(P, c, w, d, ev0, ev1, tid, nt);
                    __syncthreads();
                    if ((tid >> 1) >= l0 && (tid >> 1) < l1) wt_mwalk_lane<false>(P, c, w, L, ev0, tid, nt);
                    __syncthreads();                // before the next round reuses the slab
                }
                l0 = l1;
            }
        }
        // the lanes' run offsets, the window's run count
        wt_walk_scan_a(w, wt_walk_emit_count(w, L, tid), tid, nt);
        __syncthreads();
        wt_walk_scan_b(w, tid, nt);
        __syncthreads();
        const unsigned long long mine = w.base[nt];
        if (tid == 0) wt_lookback_publish(P, c, k, mine);
        if (tid < 64) wt_lookback_complete(P, c, k, tid, mine);
        __syncthreads();
        wt_mwalk_write(P, c, w, L, tid, nt);
        __syncthreads();
        if (tid == 0) {
            wt_window_stats(P, c);
            const long long kn = (long long) wt_glb_add64(&P.counters[WT_CTR_TICKET], 1ull);
            c.sh->ticket = kn;
            if (kn < P.n_windows) wt_phase_header(P, c, kn);
        }
        __syncthreads();
    }
}

// (nr: the register-column slots the bitmap kernel would use for this track count -- eligibility only)
hipError_t wt_walk_launch(WtParams &P, int nr, int T, int lds, int num_cu, char **gscratch, size_t *gscratch_bytes, hipStream_t s, int *grid) {
    (void) nr;
    auto kern = P.walk_mwu ? wt_mwalk_kernel<256> : (P.walk_pair ? (T > 256 ? wt_walk_kernel<512, true> : wt_walk_kernel<256, true>) : wt_walk_kernel<256, false>);
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
    if (g > P.n_windows) g = P.n_windows;
    if (g < 1) g = 1;
    const size_t need = (size_t) g * (size_t) P.g_scratch_slab;        // one slab of events per resident workgroup
    if (*gscratch_bytes < need) {
        (void) hipFree(*gscratch);          // synchronises with earlier launches
        *gscratch = nullptr; *gscratch_bytes = 0;
        e = hipMalloc((void **) gscratch, need);
        if (e != hipSuccess) return e;
        *gscratch_bytes = need;
    }
    P.g_scratch = *gscratch;
    *grid = (int) g;
    hipLaunchKernelGGL(kern, dim3((unsigned) g), dim3((unsigned) T), (size_t) lds, s, P);
    return hipGetLastError();
}
