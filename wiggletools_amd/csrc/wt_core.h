// wt_core.h -- the "bitmap multiplexer": window-local breakpoint alignment +
// per-run reducers.  This header is the single source of the kernel logic.  It
// is compiled
//   * by hipcc for gfx950 inside wt_engine.hip (the product), and
//   * by g++ with -DWT_EMU inside tests/emu/wt_emu.cpp, a phase-by-phase CPU
//     emulator of ONE workgroup used only by the `-m "not gpu"` tests to check
//     the algorithm against the oracle in a container that has no GPU.
//     The emulator is test infrastructure; the product library never contains it.
//
// What replaces what (reference = WiggleTools v1.2.11, /root/reference/src):
//   multiplexer.c:37-128  (two Fibonacci heaps, 2 heap ops / track / interval)
//        -> per-window bitmaps in LDS.  A window owns the run STARTS (breakpoints)
//           in [w0, w1).  U = union bitmap of all true breakpoints, S_i = bitmap
//           of (clipped) interval starts of track i.  For a breakpoint p the
//           interval of track i covering p is found in O(1):
//               rank = cnt_i[p/32] + popc(S_i[p/32] & mask(p));  idx = gbase_i + rank
//           and `covered` = bit p of the coverage bitmap C_i (prefix-XOR of the start / finish
//           toggles): the run's finish is never read during evaluation.
//   reducers.c / setComparisons.c ...ReductionPop
//        -> wt_eval_chunk / wt_eval_finish<OP>: one lane per 4 (or 1) positions, tracks visited in
//           index order i = 0..N-1 in f64, i.e. the reference's own summation order.
//           (Sum / Mean over float tracks: wt_delta.h, an exact O(input runs) formulation.)
//   output ordering (strcmp(chrom), start)
//        -> windows are handed out by an atomic ticket in genome order and the
//           global run offset comes from a decoupled look-back over 64-bit
//           {flag,count} status words.
#ifndef WT_CORE_H_
#define WT_CORE_H_

#include <stdint.h>
#include <math.h>
#include <type_traits>

#ifdef WT_EMU
#define WT_DEV inline
#define WT_RESTRICT
#else
#include <hip/hip_runtime.h>
#define WT_DEV __device__ __forceinline__
#define WT_RESTRICT __restrict__
#endif

// op codes == enum wtamd_op in include/wiggletools_amd.h
enum {
    WT_OP_SUM = 0, WT_OP_PRODUCT = 1, WT_OP_MEAN = 2, WT_OP_VAR = 3, WT_OP_STDDEV = 4,
    WT_OP_ENTROPY = 5, WT_OP_CV = 6, WT_OP_MIN = 7, WT_OP_MAX = 8, WT_OP_MEDIAN = 9,
    WT_OP_TTEST = 10, WT_OP_MWU = 11,
    WT_OP_MULTIPLEX = 12   // internal: emit the values[]/inplay[] tile
};
#define WT_STRICT_SET0 1u
#define WT_STRICT_SET1 2u

// consecutive window positions evaluated by one lane (template parameter K): 4 or 1; W == K * blockDim
#define WT_FLAG_AGG  (1ull << 62)
#define WT_FLAG_PFX  (2ull << 62)
#define WT_VAL_MASK  ((1ull << 62) - 1)

// counters[] slots (device, 64-bit each)
enum { WT_CTR_TICKET = 0, WT_CTR_RUNS = 1, WT_CTR_BP = 2, WT_CTR_INTERVALS = 3, WT_CTR_ERROR = 4,
       WT_CTR_DELTA_BAD = 5 /* windows the exact difference-array path had to give up on */,
       WT_CTR_PROF = 8 /* 8 phase cycle counters, -DWT_PROFILE builds only */, WT_CTR_N = 16 };
// error bits
#define WT_ERR_CAPACITY 1ull
#define WT_ERR_LOOKBACK 2ull

struct WtParams {
    // ---- tracks (run lists, see include/wiggletools_amd.h) ----
    const int32_t *start;
    const int32_t *finish;
    const void *value;            // float* or double*
    const int64_t *seg_off;       // [n_chrom*n_tracks+1]
    const double *defaults;       // [n_tracks]
    int32_t n_chrom, n_tracks;
    long long n_total;            // total number of intervals (>= 1 when a kernel is launched)
    // ---- chromosome tables ----
    const int32_t *cbase;         // [n_chrom] position of window 0 of the chromosome
    const int32_t *c_nwin;        // [n_chrom] number of windows (>= 1)
    const int32_t *c_hi;          // [n_chrom] runs starting at or beyond this position are not produced
    const int64_t *c_first_win;   // [n_chrom+1] first global window index
    // ---- windows ----
    int32_t W;                    // window width in bp, power of two >= 64
    int32_t logW;                 // log2(W)
    int32_t n_words;              // W / 64 (64-bit words of U / E)
    int64_t n_windows;
    const int32_t *win_chrom;     // [n_windows]
    uint32_t *widx;               // [(n_windows + n_chrom) * n_tracks] first interval with finish >= boundary
    // ---- operation ----
    int32_t op;
    uint32_t flags;
    int32_t n_set0;
    int32_t pad0;
    // ---- ordering / bookkeeping ----
    unsigned long long *status;   // [n_windows] look-back words, zeroed before each launch
    unsigned long long *counters; // [WT_CTR_N], zeroed before each launch
    unsigned long long *debug;    // host-visible progress markers (-DWT_DEBUG_MARK builds only)
    // ---- output ----
    int64_t capacity;
    int32_t *o_start;
    int32_t *o_finish;
    double *o_value;
    int64_t *chrom_run_off;       // [n_chrom+1]
    double *o_tile;               // WT_OP_MULTIPLEX only: [capacity * n_tracks]
    uint8_t *o_inplay;            // WT_OP_MULTIPLEX only
    int32_t *bad_list;            // difference-array kernel: windows it could not prove exact (slot order arbitrary)
    long long *bad_goff;          //   global index of the first run of each of them, same slots
    char *g_scratch;              // median / MWU with more tracks than LDS columns hold: one slab per workgroup
    long long g_scratch_slab;     // bytes per workgroup (0: the columns live in LDS)
    long long g_attr_slab;        // MWU: bytes of per-rank attributes per workgroup, after the columns' slab
    // ---- LDS carve (bytes from the dynamic LDS base; all multiples of 16) ----
    int32_t chunk_tracks;         // tracks whose bitmaps are resident in LDS at a time (== n_tracks: one chunk)
    int32_t n_chunks;             // ceil(n_tracks / chunk_tracks)
    int32_t count_segs;           // lanes per track in the count phase (1, 2, 4 or 8; >= 4 words each)
    int32_t spitch;               // u64 {S,C} pairs per track row (W/32 + 1: bank spread)
    int32_t cpitch;               // u16 entries per cnt_i row (W/32, even)
    int32_t off_S, off_cnt, off_segtot, off_U, off_cover, off_E, off_epfx, off_nextw, off_gbase, off_scratch, off_shared;
    int32_t off_acc, off_ev, off_ltv, off_ltc, off_gtv, off_gtc, off_tbase, off_tpfx, off_tfirst, off_dsh;   // difference-array path (wt_delta.h)
    int32_t off_qa, off_ltq, off_gtq;   // ... its sum-of-squares accumulators (var / stddev / CV)
    int32_t delta_q;                    // != 0: the launch accumulates squares too
    int32_t off_tdef;                   // ... its per-track default bits (delta_df)
    int32_t delta_df;                   // != 0: some default is non-zero (Sum / Mean): absent tracks add their defaults
    int32_t def_emin, def_emax;         // exponent range of the non-zero defaults (255 / 0: none)
    int32_t off_dflt32;           // register-column median / MWU: float copy of defaults[] in LDS (filled once per workgroup)
    int32_t off_wcol, off_wcnt, off_woff, off_wtot, off_wbase, off_wgt, off_wncov, off_wfe, off_wdk, off_wguess;   // median by walking (wt_walk.h)
    int32_t walk_S;               // ... positions per lane (0: not a walking launch)
    int32_t walk_capp;            // ... fixed event slots per position
    int32_t walk_ov;              // ... entries of the overflow list behind them
    int32_t walk_pair;            // ... 1: two lanes per stretch
    int32_t walk_off_at;          // ... byte offset, in the workgroup's slab, of the fallback's offsets off[W + 1]
    int32_t walk_mwu;             // ... 1: the walking launch is MWUReduction's (wt_mwalk.h): lane h of a pair holds SET h's column
    int32_t lds_bytes;
    // Mann-Whitney: 2 erf(-|U1 - mu| / sigma) for |U1 - mu| = k / 2, k = 0 .. mwu_kmax (the last entry: erf has reached -1 for good);
    // filled on the host with the host's libm (wt_mwu_make_table, wt_plan.h).  NULL: the device's erf.
    const double *mwu_table;
    int32_t mwu_kmax;
    int32_t delta_ns;             // difference-array launches: sets per position (0 / 1: one; 2: TTestReduction, wt_delta.h)
};

// Block-shared scalars (live in LDS at off_shared)
struct WtShared {
    long long ticket;
    long long goffset;            // global index of this window's first emitted run
    int32_t chrom, w0, w1, emit_hi; // emit_hi: first run start NOT produced (range end)
    long long row;                // widx row of w0
    int32_t next_bp;              // first breakpoint >= w1 (INT32_MAX if none)
    int32_t n_emit;               // runs emitted by this window
    unsigned long long bp_sum;    // covered bp of this window
    unsigned long long n_intervals;
    int32_t bad_slot;             // difference-array kernel: slot of this window in bad_list, or -1
};

// Per-lane state that lives across phases (registers on the GPU)
template <int K>
struct WtLane {
    double res[K];                // reducer values of the lane's K positions (registers across the barrier)
};

// LDS views
struct WtCtx {
    uint64_t *SC;       // [n_tracks * spitch] low half: start bits S, high half: toggles -> coverage C
    uint16_t *cnt;      // [n_tracks * cpitch] start-bit rank prefix per 32-bit word
    uint32_t *segtot;   // [n_tracks * count_segs] count-phase segment totals
    uint64_t *U;        // [n_words] true breakpoints
    uint32_t *cover;    // [4][2*n_words] coverage summaries over tracks: any(set0), all(set0), any(set1), all(set1)
    uint64_t *E;        // [n_words] emitted run starts
    uint32_t *epfx;     // [n_words + 1]
    int16_t *nextw;     // [n_words] index of the next non-empty word of U after w, or -1
    long long *gbase;   // [n_tracks] global index of (first covering interval) - 1
    char *scratch;      // per-lane column scratch for median / MWU
    char *attr;         // MWU: per-rank attribute words [n_set0][lanes] (global slab of this workgroup)
    float *dflt32;      // register-column kernels: defaults[] as float (LDS)
    WtShared *sh;
};

WT_DEV void wt_ctx_init(WtCtx &c, const WtParams &P, char *lds) {
    c.SC = (uint64_t *) (lds + P.off_S);
    c.cnt = (uint16_t *) (lds + P.off_cnt);
    c.segtot = (uint32_t *) (lds + P.off_segtot);
    c.U = (uint64_t *) (lds + P.off_U);
    c.cover = (uint32_t *) (lds + P.off_cover);
    c.E = (uint64_t *) (lds + P.off_E);
    c.epfx = (uint32_t *) (lds + P.off_epfx);
    c.nextw = (int16_t *) (lds + P.off_nextw);
    c.gbase = (long long *) (lds + P.off_gbase);
    c.scratch = lds + P.off_scratch;
    c.attr = nullptr;
    c.dflt32 = (float *) (lds + P.off_dflt32);
    c.sh = (WtShared *) (lds + P.off_shared);
}

// ---------------------------------------------------------------------------
// portability shims (device vs emulator)
// ---------------------------------------------------------------------------
#ifdef WT_EMU
WT_DEV int wt_popc64(uint64_t x) { return __builtin_popcountll(x); }
WT_DEV int wt_popc32(uint32_t x) { return __builtin_popcount(x); }
WT_DEV int wt_ctz64(uint64_t x) { return __builtin_ctzll(x); }
WT_DEV long long wt_uniform64(long long x) { return x; }
WT_DEV void wt_lds_or64(uint64_t *p, uint64_t v) { *p |= v; }
WT_DEV void wt_lds_xor64(uint64_t *p, uint64_t v) { *p ^= v; }
WT_DEV void wt_lds_or32(uint32_t *p, uint32_t v) { *p |= v; }
WT_DEV void wt_lds_and32(uint32_t *p, uint32_t v) { *p &= v; }
WT_DEV void wt_lds_min32(int32_t *p, int32_t v) { if (v < *p) *p = v; }
WT_DEV void wt_lds_umax32(uint32_t *p, uint32_t v) { if (v > *p) *p = v; }
WT_DEV void wt_lds_umin32(uint32_t *p, uint32_t v) { if (v < *p) *p = v; }
WT_DEV void wt_lds_add64(unsigned long long *p, unsigned long long v) { *p += v; }
WT_DEV void wt_lds_sub64(unsigned long long *p, unsigned long long v) { *p -= v; }
WT_DEV int32_t wt_uniform32(int32_t x) { return x; }
// wave-wide reductions: the emulator runs one lane at a time, every lane is its own wave and its leader
WT_DEV int32_t wt_wave_min_i32(int32_t x) { return x; }
WT_DEV uint32_t wt_wave_min_u32(uint32_t x) { return x; }
WT_DEV uint32_t wt_wave_max_u32(uint32_t x) { return x; }
WT_DEV unsigned long long wt_wave_sum_u64(unsigned long long x) { return x; }
WT_DEV bool wt_wave_leader(int lane) { return true; }
WT_DEV unsigned long long wt_glb_add64(unsigned long long *p, unsigned long long v) {
    unsigned long long o = *p; *p += v; return o;
}
WT_DEV void wt_glb_or64(unsigned long long *p, unsigned long long v) { *p |= v; }
WT_DEV unsigned long long wt_status_load(unsigned long long *p) { return *p; }
WT_DEV void wt_status_store(unsigned long long *p, unsigned long long v) { *p = v; }
WT_DEV void wt_backoff() {}
template <class T> WT_DEV void wt_keep_alive(T) {}
#else
WT_DEV int wt_popc64(uint64_t x) { return __popcll(x); }
WT_DEV int wt_popc32(uint32_t x) { return __popc(x); }
WT_DEV int wt_ctz64(uint64_t x) { return __ffsll((unsigned long long) x) - 1; }
// value known to be identical in every lane of the wave -> keep it in SGPRs
WT_DEV long long wt_uniform64(long long x) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) (unsigned long long) x);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned) ((unsigned long long) x >> 32));
    return (long long) (((unsigned long long) hi << 32) | lo);
}
WT_DEV void wt_lds_or64(uint64_t *p, uint64_t v) { atomicOr((unsigned long long *) p, (unsigned long long) v); }
WT_DEV void wt_lds_xor64(uint64_t *p, uint64_t v) { atomicXor((unsigned long long *) p, (unsigned long long) v); }
WT_DEV void wt_lds_or32(uint32_t *p, uint32_t v) { atomicOr((unsigned int *) p, (unsigned int) v); }
WT_DEV void wt_lds_and32(uint32_t *p, uint32_t v) { atomicAnd((unsigned int *) p, (unsigned int) v); }
WT_DEV void wt_lds_min32(int32_t *p, int32_t v) { atomicMin(p, v); }
WT_DEV void wt_lds_umax32(uint32_t *p, uint32_t v) { atomicMax((unsigned int *) p, (unsigned int) v); }
WT_DEV void wt_lds_umin32(uint32_t *p, uint32_t v) { atomicMin((unsigned int *) p, (unsigned int) v); }
WT_DEV void wt_lds_add64(unsigned long long *p, unsigned long long v) { atomicAdd(p, v); }
WT_DEV void wt_lds_sub64(unsigned long long *p, unsigned long long v) {      // ds_sub_u64: no negation in registers
    __hip_atomic_fetch_sub(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
WT_DEV int32_t wt_uniform32(int32_t x) { return (int32_t) __builtin_amdgcn_readfirstlane((unsigned) x); }
// Wave-wide scans and reductions on DPP row shifts / row broadcasts (GFX9 has both): 6 cross-lane steps of one VALU
// instruction each.  (Rounds 1-5 used __shfl_up / __shfl_xor: ds_bpermute_b32 -- an LDS-queue round trip per step, behind whatever
// LDS traffic the wavefront had outstanding; 110 of them in wt_delta_kernel<mean>.  An LDS atomic issued by all 64 lanes is
// turned by hipcc into a scalar loop over the active lanes -- ~450 SALU instructions per atomic -- hence reductions at all.)
// All 64 lanes must be active.  `ID`: the operation's identity (what a lane without a source keeps).
#define WT_DPP_STEP(x, id, ctrl, rows) (uint32_t) __builtin_amdgcn_update_dpp((int) (id), (int) (x), ctrl, rows, 0xf, false)
template <class F>
WT_DEV uint32_t wt_wave_scan32(uint32_t x, uint32_t id, F f) {     // inclusive
    x = f(x, WT_DPP_STEP(x, id, 0x111, 0xf));        // row_shr:1
    x = f(x, WT_DPP_STEP(x, id, 0x112, 0xf));        // row_shr:2
    x = f(x, WT_DPP_STEP(x, id, 0x114, 0xf));        // row_shr:4
    x = f(x, WT_DPP_STEP(x, id, 0x118, 0xf));        // row_shr:8
    x = f(x, WT_DPP_STEP(x, id, 0x142, 0xa));        // row_bcast:15 -> rows 1, 3
    x = f(x, WT_DPP_STEP(x, id, 0x143, 0xc));        // row_bcast:31 -> rows 2, 3
    return x;
}
WT_DEV unsigned long long wt_wave_scan_add64(unsigned long long x) {     // inclusive
    auto step = [](unsigned long long v, int ctrl_rows) {
        uint32_t lo = (uint32_t) v, hi = (uint32_t) (v >> 32), tl, th;
        switch (ctrl_rows) {
        case 0: tl = WT_DPP_STEP(lo, 0, 0x111, 0xf); th = WT_DPP_STEP(hi, 0, 0x111, 0xf); break;
        case 1: tl = WT_DPP_STEP(lo, 0, 0x112, 0xf); th = WT_DPP_STEP(hi, 0, 0x112, 0xf); break;
        case 2: tl = WT_DPP_STEP(lo, 0, 0x114, 0xf); th = WT_DPP_STEP(hi, 0, 0x114, 0xf); break;
        case 3: tl = WT_DPP_STEP(lo, 0, 0x118, 0xf); th = WT_DPP_STEP(hi, 0, 0x118, 0xf); break;
        case 4: tl = WT_DPP_STEP(lo, 0, 0x142, 0xa); th = WT_DPP_STEP(hi, 0, 0x142, 0xa); break;
        default: tl = WT_DPP_STEP(lo, 0, 0x143, 0xc); th = WT_DPP_STEP(hi, 0, 0x143, 0xc); break;
        }
        return v + (((unsigned long long) th << 32) | tl);
    };
#pragma unroll
    for (int q = 0; q < 6; q++) x = step(x, q);
    return x;
}
WT_DEV uint32_t wt_wave_last32(uint32_t x) { return (uint32_t) __builtin_amdgcn_readlane((int) x, 63); }
WT_DEV int32_t wt_wave_min_i32(int32_t x) {
    return (int32_t) wt_wave_last32(wt_wave_scan32((uint32_t) x, 0x7fffffffu, [](uint32_t a, uint32_t b) { return (uint32_t) ((int32_t) b < (int32_t) a ? (int32_t) b : (int32_t) a); }));
}
WT_DEV uint32_t wt_wave_min_u32(uint32_t x) {
    return wt_wave_last32(wt_wave_scan32(x, 0xffffffffu, [](uint32_t a, uint32_t b) { return b < a ? b : a; }));
}
WT_DEV uint32_t wt_wave_max_u32(uint32_t x) {
    return wt_wave_last32(wt_wave_scan32(x, 0u, [](uint32_t a, uint32_t b) { return b > a ? b : a; }));
}
WT_DEV unsigned long long wt_wave_sum_u64(unsigned long long x) {
    x = wt_wave_scan_add64(x);
    return ((unsigned long long) wt_wave_last32((uint32_t) (x >> 32)) << 32) | wt_wave_last32((uint32_t) x);
}
// Sum of arr[first .. wave) for a workgroup's per-wavefront totals (at most 16 wavefronts: one row of lanes; `first`, `wave` uniform):
// one LDS read per lane and four row shifts instead of a loop of up to fifteen dependent LDS round trips on the last wavefront
// (round 6: that loop sat in every cross-wavefront prefix of the difference-array kernels' scans).  All 64 lanes active.
WT_DEV uint32_t wt_waves_before32(const uint32_t *arr, int first, int wave, int lane) {
    uint32_t v = (lane >= first && lane < wave) ? arr[lane & 15] : 0u;
    v += WT_DPP_STEP(v, 0, 0x111, 0xf);
    v += WT_DPP_STEP(v, 0, 0x112, 0xf);
    v += WT_DPP_STEP(v, 0, 0x114, 0xf);
    v += WT_DPP_STEP(v, 0, 0x118, 0xf);
    return (uint32_t) __builtin_amdgcn_readlane((int) v, 15);
}
WT_DEV unsigned long long wt_waves_before64(const unsigned long long *arr, int first, int wave, int lane) {
    unsigned long long v = (lane >= first && lane < wave) ? arr[lane & 15] : 0ull;
#define WT_ROW64(ctrl) do { const uint32_t tl_ = WT_DPP_STEP((uint32_t) v, 0, ctrl, 0xf), th_ = WT_DPP_STEP((uint32_t) (v >> 32), 0, ctrl, 0xf); \
                            v += ((unsigned long long) th_ << 32) | tl_; } while (0)
    WT_ROW64(0x111); WT_ROW64(0x112); WT_ROW64(0x114); WT_ROW64(0x118);
#undef WT_ROW64
    return ((unsigned long long) (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (v >> 32), 15) << 32) | (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) v, 15);
}
WT_DEV bool wt_wave_leader(int lane) { return lane == 0; }
WT_DEV unsigned long long wt_glb_add64(unsigned long long *p, unsigned long long v) { return atomicAdd(p, v); }
WT_DEV void wt_glb_or64(unsigned long long *p, unsigned long long v) { atomicOr(p, v); }
// Look-back words are 8-byte granules whose payload IS the flag: relaxed
// agent-scope atomics (sc1 load/store on gfx950) suffice, no fence.
WT_DEV unsigned long long wt_status_load(unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
WT_DEV void wt_status_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
WT_DEV void wt_backoff() { __builtin_amdgcn_s_sleep(8); }
// forces a loaded value to materialise (used to warm L2 with data a later phase gathers)
WT_DEV void wt_keep_alive(float v) { asm volatile("" :: "v"(v)); }
WT_DEV void wt_keep_alive(double v) { asm volatile("" :: "v"(v)); }
#endif

WT_DEV bool wt_isnan(double x) { return x != x; }
WT_DEV bool wt_isnanf(float x) { return x != x; }
WT_DEV double wt_nan() { return __builtin_nan(""); }

// mask of bits [0..b] of a 64-bit word
WT_DEV uint64_t wt_mask_incl(int b) { return (b >= 63) ? ~0ull : ((2ull << b) - 1ull); }

// ---------------------------------------------------------------------------
// Student-t upper tail 2*Q(t;nu) for the t-test (stands in for GSL's
// gsl_cdf_tdist_Q, reference setComparisons.c:117; see DESIGN.md "unpinned").
// Q(t;nu) = I_x(nu/2, 1/2) / 2, x = nu/(nu+t^2); Lentz continued fraction.
// ---------------------------------------------------------------------------
WT_DEV double wt_betacf(double a, double b, double x) {
    const double tiny = 1e-300, eps = 1e-16;
    double qab = a + b, qap = a + 1, qam = a - 1;
    double c = 1, d = 1 - qab * x / qap;
    if (fabs(d) < tiny) d = tiny;
    d = 1 / d;
    double h = d;
    for (int m = 1; m <= 10000; m++) {
        int m2 = 2 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1 / d; h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1 / d;
        double del = d * c;
        h *= del;
        if (fabs(del - 1) < eps) break;
    }
    return h;
}

// (y = 1 - x, formed by the caller without the cancellation: oracle/wt_oracle.c inc_beta)
WT_DEV double wt_inc_beta(double a, double b, double x, double y) {
    if (x <= 0) return 0;
    if (y <= 0) return 1;
    // the t-test always comes with b = 1/2 (or a = 1/2 after the symmetry swap): lgamma(1/2) = ln sqrt(pi)
    const double lg_half = 0.57236494292470008707;
    const double lga = (a == 0.5) ? lg_half : lgamma(a), lgb = (b == 0.5) ? lg_half : lgamma(b);
    double lnfront = lgamma(a + b) - lga - lgb + a * log(x) + b * log(y);
    // one continued-fraction evaluation with the arguments chosen per lane (a divergent if / else
    // made every wave run both); same arithmetic per lane as the two-branch form
    const bool flip = !(x < (a + 1) / (a + b + 2));
    const double a2 = flip ? b : a, b2 = flip ? a : b, x2 = flip ? y : x;
    const double r = exp(lnfront) * wt_betacf(a2, b2, x2) / a2;
    return flip ? 1 - r : r;
}

WT_DEV double wt_tdist_Q(double t, double nu) {
    if (wt_isnan(t) || wt_isnan(nu) || nu <= 0) return wt_nan();
    if (isinf(t)) return t > 0 ? 0.0 : 1.0;
    double x = nu / (nu + t * t), y = t * t / (nu + t * t);
    double tail = 0.5 * wt_inc_beta(nu / 2, 0.5, x, y);
    return t >= 0 ? tail : 1 - tail;
}

// ---------------------------------------------------------------------------
// The same tail, written for the device (round 6): 2 Q(t; nu) = I_x(nu/2, 1/2), x = nu / (nu + t^2), t >= 0.
// wt_tdist_Q above costs ~2000 wave instructions per call -- two lgamma, and six f64 divisions per round of the modified
// Lentz continued fraction -- and a launch evaluates it once per output run.  Here:
//   * lgamma(a + 1/2) - lgamma(a) is ONE asymptotic series, 1/2 ln a + sum_k (2^(1-k) - 2) B_k / (k (k - 1) a^(k-1)) over even k
//     (Bernoulli polynomials at 1/2: B_k(1/2) = (2^(1-k) - 1) B_k), through k = 12 at a >= 16 (next term 3e-18); a smaller a is
//     shifted up: D(a) = D(a + j) - ln prod_{i<j} (a + i + 1/2) / (a + i).  One log, one or two divisions;
//   * the continued fraction 1 / (1 + d1 / (1 + d2 / ...)), d_n = p_n / q_n (the coefficients of wt_betacf), is run as the
//     equivalent fraction with partial numerators p_n q_(n-1) and denominators q_n by the forward recurrence
//     A_n = q_n A_(n-1) + p_n q_(n-1) A_(n-2) (same for B): no division until the very end; the four running terms are scaled
//     back by a power of two now and then; converged when |B_n A_(n-1) - B_(n-1) A_n| < eps |A_n B_(n-1)|, i.e. when Lentz's
//     ratio of successive approximants is within eps of 1.
// Same mathematical quantity as wt_tdist_Q, rounded differently (~1e-15 relative apart; tests/test_tdist_fast.py holds it to 1e-12
// against the oracle's over the (t, nu) plane).  The device's reducers call wt_ttest_tail; the emulator keeps the oracle's form.
// ---------------------------------------------------------------------------
// exp(lgamma(a + 1/2) - lgamma(a)) = sqrt(a') exp(S(a')) den / num with a' = a + j >= 16: returns S(a') (|S| < 1 / 128), a', num / den
WT_DEV double wt_gamma_half_ratio_parts(double a, double &a_shifted, double &num_over_den) {
    double num = 1.0, den = 1.0;
    bool shifted = false;
    while (a < 16.0) { num *= a + 0.5; den *= a; a += 1.0; shifted = true; }
    num_over_den = shifted ? num / den : 1.0;
    a_shifted = a;
    const double r = 1.0 / a, r2 = r * r;
    double sres = 691.0 / 180224.0;
    sres = sres * r2 - 31.0 / 18432.0;
    sres = sres * r2 + 17.0 / 14336.0;
    sres = sres * r2 - 1.0 / 640.0;
    sres = sres * r2 + 1.0 / 192.0;
    sres = sres * r2 - 1.0 / 8.0;
    return sres * r;
}
WT_DEV double wt_lgamma_half_diff(double a) {
    // lgamma(a + 1/2) - lgamma(a), a > 0 (the tests' view of the series)
    double as, nd;
    const double sres = wt_gamma_half_ratio_parts(a, as, nd);
    return 0.5 * log(as) + sres - (nd != 1.0 ? log(nd) : 0.0);
}

// h = 1 / (1 + d1 / (1 + d2 / ...)) with wt_betacf's coefficients
// (returned as the pair A, B with h = B / A: the caller has more to divide by)
WT_DEV void wt_betacf_wallis(double a, double b, double x, double &A_out, double &B_out) {
    const double eps = 1e-16;
    // n = 0: A_0 = 1, B_0 = 1 (f_0 = 1), A_-1 = 1, B_-1 = 0; q_0 = 1
    double A2 = 1.0, B2 = 0.0, A1 = 1.0, B1 = 1.0, qprev = 1.0;
    // n = 1 (the odd formula at m = 0): p = -(a + b) x a, q = a (a + 1) -- reduced by a: p = -(a + b) x, q = a + 1
    {
        const double p = -(a + b) * x, q = a + 1.0;
        const double c = p * qprev;
        const double An = q * A1 + c * A2, Bn = q * B1 + c * B2;
        A2 = A1; B2 = B1; A1 = An; B1 = Bn; qprev = q;
    }
    for (int m = 1; m <= 10000; m++) {
        const double m2 = 2.0 * m, am2 = a + m2;
        // even step n = 2 m
        {
            const double p = m * (b - m) * x, q = (am2 - 1.0) * am2;
            const double c = p * qprev;
            const double An = q * A1 + c * A2, Bn = q * B1 + c * B2;
            A2 = A1; B2 = B1; A1 = An; B1 = Bn; qprev = q;
        }
        // odd step n = 2 m + 1
        {
            const double p = -(a + m) * (a + b + m) * x, q = am2 * (am2 + 1.0);
            const double c = p * qprev;
            const double An = q * A1 + c * A2, Bn = q * B1 + c * B2;
            A2 = A1; B2 = B1; A1 = An; B1 = Bn; qprev = q;
        }
        // h_n / h_(n-1) - 1 over the last step (wt_betacf's `del`)
        const double lhs = fabs(B1 * A2 - B2 * A1), rhs = fabs(A1 * B2);
        if (lhs < eps * rhs) break;
        // scale by a power of two: the terms grow by ~q per step
        if (fabs(A1) > 1e60) {
            const double sc = 1e-60;
            A1 *= sc; B1 *= sc; A2 *= sc; B2 *= sc;
        }
    }
    A_out = A1; B_out = B1;
}

WT_DEV double wt_tdist_2Q_fast(double t, double nu) {
    if (wt_isnan(t) || wt_isnan(nu) || nu <= 0) return wt_nan();
    if (isinf(t)) return t > 0 ? 0.0 : 2.0;
    const double a = nu / 2, b = 0.5;
    const double t2 = t * t, inv = 1.0 / (nu + t2);
    const double x = nu * inv, y = t2 * inv;            // y = 1 - x, to full relative accuracy
    double I;
    if (x <= 0) I = 0;
    else if (y <= 0) I = 1;
    else {
        // the front factor x^a (1 - x)^b / B(a, b) with b = 1/2: ONE log and ONE exponential --
        //   Gamma(a + 1/2) / Gamma(a) = sqrt(a') exp(S(a')) den / num,  Gamma(1/2) = sqrt(pi),  (1 - x)^(1/2) = sqrt(y)
        double as, nd;
        const double sres = wt_gamma_half_ratio_parts(a, as, nd);
        const double front = exp(a * log(x) + sres) * sqrt(as * y * 0.31830988618379067154);       // 1 / pi
        // which side: the textbook rule x >= (a + 1) / (a + b + 2) -- and every t < 3: just below that rule's switch the fraction
        // in x needs 30-40 rounds (nu = 98: 38 at t = 1.8, 20 at t = 3) where the one in y needs 8-12, and the 64 lanes of a
        // wavefront wait for the slowest.  The price: r ~ 1 carries the front factor's a * 1e-16, so 1 - r >= 0.0027 is good to
        // a * 4e-14 relative there (nu = 98: 2e-12) -- hence only up to nu = 2000
        const bool flip = !(x * (a + b + 2) < a + 1) || (t2 < 9.0 && a <= 1000.0);
        const double a2 = flip ? b : a, b2 = flip ? a : b, x2 = flip ? y : x;
        double A, B;
        wt_betacf_wallis(a2, b2, x2, A, B);
        const double r = front * B / (A * a2 * nd);
        I = flip ? 1 - r : r;
    }
    return t >= 0 ? I : 2 - I;
}

// what the two-sample t-test's reducers store (setComparisons.c:117: 2 * gsl_cdf_tdist_Q(t, nu))
WT_DEV double wt_ttest_tail(double t, double nu) {
#ifdef WT_EMU
    return 2 * wt_tdist_Q(t, nu);           // the oracle's form, operation for operation
#else
    return wt_tdist_2Q_fast(t, nu);
#endif
}

// ---------------------------------------------------------------------------
// Phase 0: window header (one lane)
// ---------------------------------------------------------------------------
WT_DEV void wt_phase_header(const WtParams &P, WtCtx &c, long long k) {
    WtShared *sh = c.sh;
    int ch = P.win_chrom[k];
    long long m = k - P.c_first_win[ch];
    sh->chrom = ch;
    sh->w0 = P.cbase[ch] + (int32_t) m * P.W;
    sh->w1 = sh->w0 + P.W;
    sh->emit_hi = P.c_hi[ch];
    sh->row = k + ch;                 // one extra boundary row per chromosome
    sh->next_bp = 0x7fffffff;
    sh->n_emit = 0;
    sh->bp_sum = 0;
    sh->n_intervals = 0;
    sh->goffset = 0;
    sh->bad_slot = -1;
}

// ---------------------------------------------------------------------------
// Phase 1: clear the bitmaps
// ---------------------------------------------------------------------------
// `first`: also clear the window-wide bitmaps (once per window); the per-track bitmaps are cleared
// for every chunk of tracks.
WT_DEV void wt_phase_zero(const WtParams &P, WtCtx &c, bool first, int tid, int nt) {
    const int nS = P.chunk_tracks * P.spitch;
    for (int x = tid; x < nS; x += nt) c.SC[x] = 0;
    // spare cnt entry of every track: "an interval of this track spans w0" (its start bit at
    // position 0 is a clipping artefact, not a breakpoint)
    for (int i = tid; i < P.chunk_tracks; i += nt) c.cnt[(size_t) i * P.cpitch + P.n_words * 2] = 0;
    if (!first) return;
    for (int x = tid; x < P.n_words; x += nt) { c.U[x] = 0; c.E[x] = 0; }
    const int nw32z = P.n_words * 2;
    for (int x = tid; x < nw32z; x += nt) {
        c.cover[x] = 0;                      // any(set 0)
        c.cover[nw32z + x] = 0xffffffffu;    // all(set 0)
        c.cover[2 * nw32z + x] = 0;          // any(set 1)
        c.cover[3 * nw32z + x] = 0xffffffffu;// all(set 1)
    }
}

// ---------------------------------------------------------------------------
// Phase 2: stream the window's intervals; build, per track i, the paired
// 32-bit bitmaps SC_i[w] = {S (low half): clipped interval starts,
//                           C (high half): coverage TOGGLES (start ^ finish)}
// and the shared union bitmap U of true breakpoints.
// One WT_LOAD_GROUP-lane group per track (coalesced reads of start[]/finish[]); the loads
// of the group's NEXT track are issued before the LDS atomics of the current
// one so that one global round trip is hidden per track.
// Interval classes relative to the window [w0,w1):
//   f == w0   : true breakpoint at w0, covers nothing here
//   s >= w1   : first interval beyond the window: candidate for next_bp
//   otherwise : covers part of the window: start bit + toggle at the clipped
//               start, toggle at f if f < w1, true start/finish bits in U,
//               f >= w1 -> candidate for next_bp
// A start bit is set exactly once, so S can share one 64-bit XOR with the toggle.
// ---------------------------------------------------------------------------
#define WT_LOAD_GROUP 32         // lanes per track in the load phase
#define WT_LOAD_UNROLL 4        // intervals per lane fetched ahead
template <class ValT>
struct WtLoadBatch {
    long long off, lo, hi;
    int32_t s[WT_LOAD_UNROLL], f[WT_LOAD_UNROLL];
    ValT v[WT_LOAD_UNROLL];
};

template <class ValT>
WT_DEV void wt_load_fetch(const WtParams &P, const WtCtx &c, int i, int lane, WtLoadBatch<ValT> &b) {
    const int N = P.n_tracks;
    const uint32_t *row0 = P.widx + (size_t) c.sh->row * N;
    const long long seg = (long long) c.sh->chrom * N + i;
    b.off = P.seg_off[seg];
    const long long n = P.seg_off[seg + 1] - b.off;
    b.lo = row0[i];
    b.hi = row0[N + i];
    if (b.hi >= n) b.hi = n - 1;
#pragma unroll
    // unconditional loads (an index past the track's range reads interval 0 of the batch and is never
    // applied): with a predicate -- or a branch -- around them the compiler cannot count the loads in
    // flight and makes the consumer of the previous batch wait for all of them (see wt_delta_fetch)
    for (int u = 0; u < WT_LOAD_UNROLL; u++) {
        const long long jr = b.lo + lane + WT_LOAD_GROUP * u;
        const long long at = jr <= b.hi ? b.off + jr : 0;
        b.s[u] = P.start[at];
        b.f[u] = P.finish[at];
        // the value is not needed here: the load (coalesced, many in flight) warms L2 with the
        // very cache lines the eval phase gathers from, so those gathers stop paying HBM latency
        b.v[u] = ((const ValT *) P.value)[at];
    }
}

WT_DEV void wt_load_apply(const WtParams &P, WtCtx &c, uint64_t *SCi, uint16_t *pseudo, int32_t s, int32_t f) {
    const int32_t w0 = c.sh->w0, w1 = c.sh->w1;
    if (f == w0) { wt_lds_or64(&c.U[0], 1ull); return; }
    if (s >= w1) { wt_lds_min32(&c.sh->next_bp, s); return; }
    const int cs = s > w0 ? s - w0 : 0;
    if (s < w0) *pseudo = 1;                  // at most one such interval per track and window
    const uint64_t sbit = 1ull << (cs & 31);
    uint64_t x = sbit | (sbit << 32);         // start bit (low half) + coverage toggle (high half)
    if (f < w1) {
        const int cf = f - w0;
        const uint64_t fbit = (1ull << (cf & 31)) << 32;
        if ((cf >> 5) == (cs >> 5)) x ^= fbit;            // same word: one atomic for both
        else wt_lds_xor64(&SCi[cf >> 5], fbit);
    } else {
        wt_lds_min32(&c.sh->next_bp, f);
    }
    wt_lds_xor64(&SCi[cs >> 5], x);
    // true breakpoints (U) are derived from S | toggles in the count phase: no atomics here
}

// Tracks [t_lo, t_hi) are loaded into the LDS rows 0 .. t_hi-t_lo-1.  `stats`: count the
// examined intervals (first pass over the chunks only).
template <class ValT>
WT_DEV void wt_phase_load(const WtParams &P, WtCtx &c, int t_lo, int t_hi, bool stats, int tid, int nt) {
    const int group = tid / WT_LOAD_GROUP, lane = tid % WT_LOAD_GROUP, ngroups = nt / WT_LOAD_GROUP;
    if (t_lo + group >= t_hi) return;
    WtLoadBatch<ValT> cur;
    wt_load_fetch<ValT>(P, c, t_lo + group, lane, cur);
    for (int g = t_lo + group; g < t_hi; g += ngroups) {
        WtLoadBatch<ValT> nxt;
        const int gnext = g + ngroups;
        wt_load_fetch<ValT>(P, c, gnext < t_hi ? gnext : t_hi - 1, lane, nxt);     // (past the end: the last track again, unused)
        const int i = g - t_lo;
        uint64_t *SCi = c.SC + (size_t) i * P.spitch;
        uint16_t *pseudo = c.cnt + (size_t) i * P.cpitch + P.n_words * 2;
        if (lane == 0) {
            // gbase + r = global index of the r-th interval that has a start bit in S_i;
            // clamped so that gbase + 1 is always a valid address (r == 0 lookups load it blindly)
            long long gb = -1;
            if (cur.hi >= cur.lo) {
                gb = cur.off + cur.lo - 1 + (cur.f[0] == c.sh->w0 ? 1 : 0);
                if (stats) wt_lds_add64(&c.sh->n_intervals, (unsigned long long) (cur.hi - cur.lo + 1));
            }
            if (gb > P.n_total - 2) gb = P.n_total - 2;
            c.gbase[i] = gb;
        }
#pragma unroll
        for (int u = 0; u < WT_LOAD_UNROLL; u++)
            if (cur.lo + lane + WT_LOAD_GROUP * u <= cur.hi) {
                wt_load_apply(P, c, SCi, pseudo, cur.s[u], cur.f[u]);
                wt_keep_alive(cur.v[u]);
            }
        for (long long jr = cur.lo + lane + WT_LOAD_GROUP * WT_LOAD_UNROLL; jr <= cur.hi; jr += WT_LOAD_GROUP)
            wt_load_apply(P, c, SCi, pseudo, P.start[cur.off + jr], P.finish[cur.off + jr]);
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------
// Phase 3: per track, over the 32-bit words of the window:
//   cnt_i[w]  = number of start bits in words < w       (rank prefix)
//   C_i[w]    = coverage bits = running XOR of the toggles (prefix-xor inside
//               the word by shifts, carry = last bit of the previous word)
// ---------------------------------------------------------------------------
// Two sub-phases (a barrier in between) so that count_segs lanes share a track:
//   3a  every lane totals its own segment of words (start bits, toggle parity)
//   3b  every lane prefixes the totals of the segments before it, then rewrites
//       its own words in place (toggles -> coverage) and fills cnt.
WT_DEV void wt_phase_count_a(const WtParams &P, WtCtx &c, int t_lo, int t_hi, int tid, int nt) {
    const int nw32 = P.n_words * 2;
    const int seg_words = (nw32 + P.count_segs - 1) / P.count_segs;
    const int items = (t_hi - t_lo) * P.count_segs;
    for (int it = tid; it < items; it += nt) {
        const int i = it / P.count_segs, q = it % P.count_segs;
        const uint64_t *SCi = c.SC + (size_t) i * P.spitch;
        const int w_lo = q * seg_words;
        int w_hi = w_lo + seg_words;
        if (w_hi > nw32) w_hi = nw32;
        unsigned run = 0, par = 0;
        for (int w = w_lo; w < w_hi; w++) {
            const uint64_t sc = SCi[w];
            run += (unsigned) wt_popc32((uint32_t) sc);
            par ^= (unsigned) wt_popc32((uint32_t) (sc >> 32));
        }
        c.segtot[it] = (run & 0xffffu) | ((par & 1u) << 31);
    }
}

WT_DEV void wt_phase_count_b(const WtParams &P, WtCtx &c, int t_lo, int t_hi, int tid, int nt) {
    const int nw32 = P.n_words * 2;
    const int seg_words = (nw32 + P.count_segs - 1) / P.count_segs;
    const int items = (t_hi - t_lo) * P.count_segs;
    for (int it = tid; it < items; it += nt) {
        const int i = it / P.count_segs, q = it % P.count_segs;
        uint64_t *SCi = c.SC + (size_t) i * P.spitch;
        uint16_t *ci = c.cnt + (size_t) i * P.cpitch;
        const int w_lo = q * seg_words;
        int w_hi = w_lo + seg_words;
        if (w_hi > nw32) w_hi = nw32;
        unsigned run = 0, par = 0;
        for (int x = 0; x < q; x++) {
            const uint32_t st = c.segtot[i * P.count_segs + x];
            run += st & 0xffffu;
            par ^= st >> 31;
        }
        uint32_t carry = par ? 0xffffffffu : 0u;
        uint32_t *U32 = (uint32_t *) c.U;
        const bool pseudo = ci[nw32] != 0;
        const int set1 = (P.n_set0 > 0 && t_lo + i >= P.n_set0) ? 1 : 0;
        uint32_t *cov_any = c.cover + (size_t) (2 * set1) * nw32;
        uint32_t *cov_all = cov_any + nw32;
        const bool strict_set = (P.flags & (set1 ? WT_STRICT_SET1 : WT_STRICT_SET0)) != 0;
        for (int w = w_lo; w < w_hi; w++) {
            const uint64_t sc = SCi[w];
            const uint32_t sbits = (uint32_t) sc;
            uint32_t t = (uint32_t) (sc >> 32);
            // true breakpoints of this track: starts | finishes, and finishes == toggles ^ starts
            // (a finish meeting the next start cancels in the toggles but shows in the starts);
            // the clipped start of an interval spanning w0 is not a breakpoint
            uint32_t u = sbits | t;
            if (w == 0 && pseudo) u &= ~1u;
            if (u) wt_lds_or32(&U32[w], u);
            ci[w] = (uint16_t) run;
            run += (unsigned) wt_popc32(sbits);
            t ^= t << 1; t ^= t << 2; t ^= t << 4; t ^= t << 8; t ^= t << 16;
            t ^= carry;
            carry = (t >> 31) ? 0xffffffffu : 0u;
            SCi[w] = ((uint64_t) t << 32) | sbits;
            // coverage summaries for the emission predicate (multiplexer.c:120,125;
            // setComparisons.c:48-54): any = OR over the set's tracks, all = AND (strict only)
            if (t) wt_lds_or32(&cov_any[w], t);
            if (strict_set && t != 0xffffffffu) wt_lds_and32(&cov_all[w], t);
        }
    }
}

// ---------------------------------------------------------------------------
// Track lookup for a GROUP of K consecutive window positions p0..p0+K-1 that
// share one 32-bit bitmap word (K divides 32, p0 % K == 0).  Branch-free:
//   rank_k = cnt_i[w] + popc(S & mask_k)     -> interval index gbase_i + rank_k
//   cov_k  = bit (p0+k) of C_i[w]
// The value gather is issued unconditionally (index clamped to >= 1, which is
// always a valid address) so K independent loads are in flight per lane; the
// interval's finish is never read here.
// ---------------------------------------------------------------------------
// Track-loop unrolling of the streaming reducers: gathers of WT_TRACK_UNROLL
// consecutive tracks are in flight at once (the adds still happen in index order).
#ifndef WT_TRACK_UNROLL
#define WT_TRACK_UNROLL 2
#endif
#ifndef WT_PIPE
#define WT_PIPE 3
#endif
#ifndef WT_MEDIAN_BITS
#define WT_MEDIAN_BITS 2  // key bits decided per sweep of the median's bitwise selection
#endif
#ifndef WT_MWU_RB
#define WT_MWU_RB 8      // column values read ahead per batch in the MWU ranking sweeps
#endif
#ifndef WT_MWU_EB
#define WT_MWU_EB 4      // set-0 elements ranked per sweep over a lane's value column
#endif
#define WT_PRAGMA(x) _Pragma(#x)
#define WT_UNROLL_TRACKS WT_PRAGMA(unroll WT_TRACK_UNROLL)

template <int K>
struct WtFetchK {
    bool cov[K];
    double x[K];      // default-substituted value
    uint32_t cbits;   // coverage word of the track (bit b0+k <=> cov[k])
};

// Raw result of the gathers of one track (issued, not yet consumed): keeping several of these
// alive is what keeps many global loads in flight per lane.
template <class ValT, int K>
struct WtRawK {
    ValT v[K];
    uint32_t cbits;
    double dflt;
};

// compile-time loop: f(std::integral_constant<int, k>) for k = B..E-1
template <int B, int E, class F>
WT_DEV void wt_static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        wt_static_for<B + 1, E>(f);
    }
}

// Three VALU idioms the optimiser undoes when they are written in C (it turns the masks back
// into compare + select pairs and the bit add into shift/and/add); spelled out so the
// evaluation loop stays at its minimum instruction count.
//   wt_bit_mask<k>(x)        all-ones if bit k of x is set, else 0          v_bfe_i32
//   wt_bfi(m, a, b)          (a & m) | (b & ~m)                             v_bfi_b32
//   wt_add_bit_shl<k,SH>(o,x)  o + (bit k of x << SH)                       v_bfe_u32 + v_lshl_add_u32
template <int k>
WT_DEV uint32_t wt_bit_mask(uint32_t x) {
#if defined(WT_EMU) || defined(WT_NO_ASM)
    return 0u - ((x >> k) & 1u);
#else
    uint32_t m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(x), "n"(k));
    return m;
#endif
}
WT_DEV uint32_t wt_bfi(uint32_t m, uint32_t a, uint32_t b) {
#if defined(WT_EMU) || defined(WT_NO_ASM)
    return (a & m) | (b & ~m);
#else
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
#endif
}
template <int k, unsigned SH>
WT_DEV unsigned wt_add_bit_shl(unsigned o, uint32_t x) {
#if defined(WT_EMU) || defined(WT_NO_ASM)
    return o + (((x >> k) & 1u) << SH);
#else
    uint32_t t, r;
    asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(t) : "v"(x), "n"(k));
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(t), "n"(SH), "v"(o));
    return r;
#endif
}

template <class ValT, int K>
WT_DEV void wt_fetch_issue(const WtParams &P, const WtCtx &c, int i, int w32, int b0, const uint32_t (&mask)[K],
                           double dflt, WtRawK<ValT, K> &raw) {
    const uint64_t sc = c.SC[(size_t) i * P.spitch + w32];
    const uint32_t sbits = (uint32_t) sc;
    const unsigned cn = (unsigned) c.cnt[(size_t) i * P.cpitch + w32];
    raw.cbits = (uint32_t) (sc >> 32);
    raw.dflt = dflt;
#ifdef WT_EMU
    const char *vp = (const char *) ((const ValT *) P.value + wt_uniform64(c.gbase[i]));
#pragma unroll
    for (int k = 0; k < K; k++) {
        unsigned r = cn + (unsigned) wt_popc32(sbits & mask[k]);
        r = r ? r : 1u;
        raw.v[k] = *(const ValT *) (vp + (uint32_t) (r * (unsigned) sizeof(ValT)));
    }
#else
    // The eval phase is VALU-bound (measured), so the index arithmetic is kept minimal: a buffer
    // descriptor per track (scalar), byte offset of position 0 from one popcount, the following
    // positions add their own start bit.  rank 0 (nothing started yet: the position is not
    // covered, the value unused) gives offset -sizeof(ValT), which the descriptor's range check
    // turns into a zero result instead of an access.
    constexpr unsigned SH = sizeof(ValT) == 4 ? 2u : 3u;
    const ValT *tb = (const ValT *) P.value + (wt_uniform64(c.gbase[i]) + 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *) tb, 0, 1 << 30, 0x00020000);
    unsigned off = ((unsigned) wt_popc32(sbits & mask[0]) << SH) + ((cn << SH) - (unsigned) sizeof(ValT));
    const unsigned sk = sbits >> b0;
    wt_static_for<0, K>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k > 0) off = wt_add_bit_shl<k, SH>(off, sk);
        if constexpr (sizeof(ValT) == 4) raw.v[k] = __builtin_bit_cast(ValT, (uint32_t) __builtin_amdgcn_raw_buffer_load_b32(rs, (int) off, 0, 0));
        else raw.v[k] = __builtin_bit_cast(ValT, __builtin_amdgcn_raw_buffer_load_b64(rs, (int) off, 0, 0));
    });
#endif
}

template <class ValT, class ScrT, int K>
WT_DEV void wt_fetch_finish(const WtRawK<ValT, K> &raw, int b0, WtFetchK<K> &out) {
    out.cbits = raw.cbits;
    const uint32_t ck = raw.cbits >> b0;
    wt_static_for<0, K>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        out.cov[k] = ((ck >> k) & 1u) != 0;
        // branch-free select through a sign mask: one bit-field insert per 32 bits
        const uint32_t m = wt_bit_mask<k>(ck);
        if (sizeof(ValT) == 4 && sizeof(ScrT) == 4) {
            // float tracks whose defaults are float-exact: select in f32, widen once
            const uint32_t vb = __builtin_bit_cast(uint32_t, (float) raw.v[k]);
            const uint32_t db = __builtin_bit_cast(uint32_t, (float) raw.dflt);
            out.x[k] = (double) __builtin_bit_cast(float, wt_bfi(m, vb, db));
        } else {
            const uint64_t vb = __builtin_bit_cast(uint64_t, (double) raw.v[k]);
            const uint64_t db = __builtin_bit_cast(uint64_t, raw.dflt);
            const uint32_t lo = wt_bfi(m, (uint32_t) vb, (uint32_t) db);
            const uint32_t hi = wt_bfi(m, (uint32_t) (vb >> 32), (uint32_t) (db >> 32));
            out.x[k] = __builtin_bit_cast(double, ((uint64_t) hi << 32) | lo);
        }
    });
}

template <class ValT, class ScrT, int K>
WT_DEV void wt_fetch_group(const WtParams &P, const WtCtx &c, int i, int w32, int b0, const uint32_t (&mask)[K],
                           double dflt, WtFetchK<K> &out) {
    WtRawK<ValT, K> raw;
    wt_fetch_issue<ValT, K>(P, c, i, w32, b0, mask, dflt, raw);
    wt_fetch_finish<ValT, ScrT, K>(raw, b0, out);
}

// order-preserving integer keys for selection
WT_DEV uint32_t wt_key32(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
WT_DEV float wt_unkey32(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __builtin_bit_cast(float, u);
}
WT_DEV uint64_t wt_key64(double f) {
    uint64_t u = __builtin_bit_cast(uint64_t, f);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
WT_DEV double wt_unkey64(uint64_t k) {
    uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __builtin_bit_cast(double, u);
}

// Visits the (global) tracks lo..hi-1 in index order, two tracks' gathers in flight at a time.
// The tracks' bitmaps are the LDS rows (track - base) of the resident chunk.
// body(track, F) is called in increasing track order (the reference's summation order).
template <class ValT, class ScrT, int K, class Body>
WT_DEV void wt_for_tracks(const WtParams &P, const WtCtx &c, int base, int lo, int hi, int w32, int b0,
                          const uint32_t (&mask)[K], bool use_defaults, Body body) {
    const double *dflt = P.defaults;
    // The loop is bound by the gather round trip (LDS word -> global value), not by VALU work
    // (measured: 40 % fewer VALU instructions changed nothing), so WT_PIPE tracks' gathers are
    // kept in flight: track t + WT_PIPE is issued right after track t is consumed.
    constexpr int D = WT_PIPE;
    WtRawK<ValT, K> R[D];
#pragma unroll
    for (int u = 0; u < D; u++)
        if (lo + u < hi) wt_fetch_issue<ValT, K>(P, c, lo + u - base, w32, b0, mask, use_defaults ? dflt[lo + u] : 0.0, R[u]);
    for (int j = lo; j < hi; j += D) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            const int t = j + u;
            if (t < hi) {                       // wave-uniform
                WtFetchK<K> F;
                wt_fetch_finish<ValT, ScrT, K>(R[u], b0, F);
                body(t, F);
                if (t + D < hi) wt_fetch_issue<ValT, K>(P, c, t + D - base, w32, b0, mask, use_defaults ? dflt[t + D] : 0.0, R[u]);
            }
        }
    }
}

// Same contract, but WT_DEEP tracks' gathers are in flight per lane.  Used by median / MWU (one
// position per lane, no accumulators in registers): their gather phase is pure memory latency,
// and with only a few waves per CU (the LDS scratch columns limit the workgroup size) depth is
// what keeps enough loads outstanding.
#ifndef WT_DEEP
#define WT_DEEP 8
#endif
template <class ValT, class ScrT, int K, class Body>
WT_DEV void wt_for_tracks_deep(const WtParams &P, const WtCtx &c, int base, int lo, int hi, int w32, int b0,
                               const uint32_t (&mask)[K], Body body) {
    const double *dflt = P.defaults;
    constexpr int D = WT_DEEP;
    WtRawK<ValT, K> R[D];
#pragma unroll
    for (int u = 0; u < D; u++)
        if (lo + u < hi) wt_fetch_issue<ValT, K>(P, c, lo + u - base, w32, b0, mask, dflt[lo + u], R[u]);
    for (int j = lo; j < hi; j += D) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            const int t = j + u;
            if (t < hi) {                       // wave-uniform
                WtFetchK<K> F;
                wt_fetch_finish<ValT, ScrT, K>(R[u], b0, F);
                body(t, F);
                if (t + D < hi) wt_fetch_issue<ValT, K>(P, c, t + D - base, w32, b0, mask, dflt[t + D], R[u]);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Per-run reducers.  One lane evaluates K consecutive window positions; for every position the
// tracks are visited in index order i = 0..N-1 in f64, exactly the reference's summation order
// (bit-identical sums).  The tracks arrive in CHUNKS (the ones whose bitmaps are resident in
// LDS), so a reducer is init / add(chunk) [/ second pass for var, stddev, CV] / finish, with its
// accumulators living in registers across the chunk loop.  With few enough tracks there is one
// chunk and this degenerates to one loop over all tracks.
// Positions of the group that do not start an emitted run are computed too (cheaper than
// diverging) and discarded by the caller.
// NaN: the reference tests isnan() per value and yields NaN; for sum / product / mean / var /
// stddev / CV IEEE propagation through the accumulator gives the same answer (NaN in -> NaN
// out), so no flag is carried; min / max / median / MWU carry an explicit flag because
// comparisons swallow NaN.  Median / MWU use K == 1 and this lane's LDS scratch column.
// ---------------------------------------------------------------------------
// NR > 0 (median / MWU, K == 1, float tracks with float-exact defaults, at most NR tracks): the
// position's value column lives in REGISTERS -- NR 32-bit slots with compile-time indices -- instead
// of an LDS column.  The LDS columns (N * 4 B per lane) capped a CU at 4 waves, one per SIMD, and
// both reducers were bound by dependent-instruction latency with nothing to interleave; the
// register column leaves LDS to the bitmaps alone (3 workgroups of 256 lanes per CU at NR = 128).
template <int K, int NR = 0>
struct WtAcc {
    double a[K], b[K];        // sum|product|best|mean|s1 , squares|q1
    double c2[K], d[K];       // t-test: s2, q2
    bool nan[K];
    // two half columns (median: lower / upper half of the padded column; MWU: set 0 / set 1).  Two
    // arrays on purpose: as ONE 128-entry array the compiler kept the upper half in scratch memory
    // and ran its sorting network through load / exchange / store round trips.
    uint32_t col[NR > 0 ? NR / 2 : 1], col2[NR > 0 ? NR / 2 : 1];
};

WT_DEV constexpr int wt_eval_passes(int op) {
    return (op == WT_OP_VAR || op == WT_OP_STDDEV || op == WT_OP_ENTROPY || op == WT_OP_CV) ? 2 : 1;
}

template <int OP, int K, int NR>
WT_DEV void wt_eval_init(WtAcc<K, NR> &A) {
#pragma unroll
    for (int k = 0; k < K; k++) {
        A.a[k] = (OP == WT_OP_PRODUCT) ? 1.0 : 0.0;
        A.b[k] = 0; A.c2[k] = 0; A.d[k] = 0;
        A.nan[k] = false;
    }
}

// Adds the tracks [t_lo, t_hi) (resident chunk, LDS rows relative to t_lo) in pass `pass`.
// goff_run0: MULTIPLEX only -- global index of the run at the lane's first emitted position.
// Compare-exchange network over registers (bitonic sort, compile-time indices): v[0..NR) ascending
// (DESC: descending) as unsigned keys.  NR * log2(NR) * (log2(NR) + 1) / 4 exchanges, two VALU each.
template <int NR, bool DESC>
WT_DEV void wt_sort_regs(uint32_t *v) {
#pragma unroll
    for (int k = 2; k <= NR; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < NR; i++) {
                const int l = i ^ j;
                if (l > i) {
                    const uint32_t a = v[i], b = v[l];
                    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
                    const bool up = ((i & k) == 0) != DESC;
                    v[i] = up ? lo : hi;
                    v[l] = up ? hi : lo;
                }
            }
        }
    }
}

// Register column: the default-substituted values of the position at compile-time slots (median:
// order-preserving keys, real tracks at slots pad_lo + i; MWU: float bits, set 0 at [0, NR/2),
// set 1 at [NR/2, NR)).  The tracks are fetched by a ROLLED loop over blocks of WT_REGCOL_BLOCK
// slots -- a fully unrolled gather made the compiler hoist every track's scalar state (buffer
// descriptor, default, first-interval index) and spill hundreds of SGPRs and VGPRs -- and every
// block is moved to its registers through a switch on the (uniform) block number.
#ifndef WT_REGCOL_BLOCK
#define WT_REGCOL_BLOCK 8
#endif
template <int NR>
WT_DEV void wt_regcol_store(uint32_t *col, uint32_t *col2, int blk, const uint32_t (&tmp)[WT_REGCOL_BLOCK]) {
    // independent uniform tests (no else-chain, no switch): every store keeps a constant index
    wt_static_for<0, NR / WT_REGCOL_BLOCK>([&](auto bc) {
        constexpr int B0 = decltype(bc)::value;
        if (blk == B0) {
#pragma unroll
            for (int q = 0; q < WT_REGCOL_BLOCK; q++) {
                constexpr int H = NR / 2;
                if (B0 * WT_REGCOL_BLOCK < H) col[B0 * WT_REGCOL_BLOCK + q] = tmp[q];
                else col2[B0 * WT_REGCOL_BLOCK + q - H] = tmp[q];
            }
        }
    });
}

#ifdef WT_REGCOL_FLAT
// Variant: both passes fully unrolled, every load of the position in flight at once.
//   A  per track: bitmap word + rank prefix + first-interval index from LDS, the value's address in
//      VECTOR registers (no per-track buffer descriptor: no scalar state to hoist and spill), the
//      global load issued straight into the slot's register; the coverage bit goes into a bit mask;
//   B  per slot: default (LDS table, filled once per workgroup) where the track is absent, NaN
//      test, key.
template <int OP, int NR>
WT_DEV int wt_regcol_track(int s, int N, int na, int pad_lo) {      // track held by slot s, or -1 (pad)
    if (OP == WT_OP_MEDIAN) { const int i = s - pad_lo; return (i < 0 || i >= N) ? -1 : i; }
    if (s < NR / 2) return s < na ? s : -1;
    const int i = na + (s - NR / 2);
    return i < N ? i : -1;
}

template <int OP, class ValT, int NR>
WT_DEV bool wt_gather_regs(const WtParams &P, const WtCtx &c, int p0, uint32_t *col, uint32_t *col2) {
    constexpr int H = NR / 2;
    const int w32 = p0 >> 5, b0 = p0 & 31;
    const uint32_t m0 = (2u << b0) - 1u;
    const int N = P.n_tracks, na = P.n_set0;
    const int pad_lo = NR / 2 - N / 2;
    const uint32_t *vals = (const uint32_t *) P.value;
    uint32_t cov[NR / 32];
#pragma unroll
    for (int q = 0; q < NR / 32; q++) cov[q] = 0;
    wt_static_for<0, NR>([&](auto sc) {         // A: issue every load
        constexpr int s = decltype(sc)::value;
        const int i = wt_regcol_track<OP, NR>(s, N, na, pad_lo);
        uint32_t raw = 0;
        if (i >= 0) {                           // workgroup-uniform
            const uint64_t scw = c.SC[(size_t) i * P.spitch + w32];
            unsigned r = (unsigned) c.cnt[(size_t) i * P.cpitch + w32] + (unsigned) wt_popc32((uint32_t) scw & m0);
            r = r ? r : 1u;                     // nothing started yet: not covered, the value is unused
            raw = vals[c.gbase[i] + (long long) r];
            cov[s >> 5] |= (((uint32_t) (scw >> 32) >> b0) & 1u) << (s & 31);
        }
        if constexpr (s < H) col[s] = raw; else col2[s - H] = raw;
    });
    bool nan = false;
    wt_static_for<0, NR>([&](auto sc) {         // B: defaults, NaN, keys
        constexpr int s = decltype(sc)::value;
        const int i = wt_regcol_track<OP, NR>(s, N, na, pad_lo);
        uint32_t v = OP == WT_OP_MEDIAN ? (s < pad_lo ? 0u : 0xffffffffu) : 0x7fc00000u;
        if (i >= 0) {
            uint32_t raw;
            if constexpr (s < H) raw = col[s]; else raw = col2[s - H];
            const uint32_t m = 0u - ((cov[s >> 5] >> (s & 31)) & 1u);
            const uint32_t xb = (raw & m) | (__builtin_bit_cast(uint32_t, c.dflt32[i]) & ~m);
            const float x = __builtin_bit_cast(float, xb);
            nan |= wt_isnanf(x);
            v = OP == WT_OP_MEDIAN ? wt_key32(x) : xb;
        }
        if constexpr (s < H) col[s] = v; else col2[s - H] = v;
    });
    return nan;
}
#else
template <int OP, class ValT, int NR>
WT_DEV bool wt_gather_regs(const WtParams &P, const WtCtx &c, int p0, uint32_t *col, uint32_t *col2) {
    const int w32 = p0 >> 5, b0 = p0 & 31;
    const uint32_t mask[1] = { (2u << b0) - 1u };
    const int N = P.n_tracks, na = P.n_set0;
    bool nan = false;
    // Median: pads below and above place the wanted order statistic vals[N/2] (reducers.c:810) at
    // the FIXED slot NR/2 of the sorted column: N/2 of the real keys and NR/2 - N/2 low pads lie
    // below it.  MWU: pads are NaN (they compare false with everything).
    const int pad_lo = NR / 2 - N / 2;
    // The issue half is BRANCH-FREE: a slot without a track gathers track 0 and is overridden when it is
    // consumed, and the defaults come from the LDS copy.  (With a uniform `if (track >= 0)` around every
    // gather and defaults[i] read from global memory, hipcc placed `s_waitcnt vmcnt(0)` in front of every
    // track's issue: the block's eight loads went out one L2 round trip after the other -- measured
    // as 39 % of the whole kernel.)
#pragma unroll 1
    for (int blk = 0; blk < NR / WT_REGCOL_BLOCK; blk++) {
        uint32_t tmp[WT_REGCOL_BLOCK];
        int trk[WT_REGCOL_BLOCK];
        WtRawK<ValT, 1> raw[WT_REGCOL_BLOCK];
#pragma unroll
        for (int q = 0; q < WT_REGCOL_BLOCK; q++) {       // issue the block's gathers ...
            const int s = blk * WT_REGCOL_BLOCK + q;
            int i;
            if (OP == WT_OP_MEDIAN) { i = s - pad_lo; if (i < 0 || i >= N) i = -1; }
            else { i = s < NR / 2 ? (s < na ? s : -1) : (na + (s - NR / 2) < N ? na + (s - NR / 2) : -1); }
            trk[q] = i;
            wt_fetch_issue<ValT, 1>(P, c, i < 0 ? 0 : i, w32, b0, mask, 0.0, raw[q]);
        }
#pragma unroll
        for (int q = 0; q < WT_REGCOL_BLOCK; q++) {       // ... then consume them
            const int s = blk * WT_REGCOL_BLOCK + q;
            const int i = trk[q];
            const bool covered = ((raw[q].cbits >> b0) & 1u) != 0;
            const uint32_t db = __builtin_bit_cast(uint32_t, c.dflt32[i < 0 ? 0 : i]);
            const uint32_t xb = covered ? __builtin_bit_cast(uint32_t, (float) raw[q].v[0]) : db;
            const bool real = i >= 0;                     // workgroup-uniform
            nan |= real && ((xb & 0x7fffffffu) > 0x7f800000u);
            const uint32_t pad = OP == WT_OP_MEDIAN ? (s < pad_lo ? 0u : 0xffffffffu) : 0x7fc00000u;
            tmp[q] = real ? (OP == WT_OP_MEDIAN ? wt_key32(__builtin_bit_cast(float, xb)) : xb) : pad;
        }
        wt_regcol_store<NR>(col, col2, blk, tmp);
    }
    return nan;
}

#endif  // WT_REGCOL_FLAT

#if defined(WT_PROFILE) && !defined(WT_EMU)
// -DWT_PROFILE builds: cycles of the register-column reducers' sub-phases (lane 0 of every wave)
__device__ unsigned long long wt_prof2[8];
#define WT_SUBTICK(slot) do { if ((threadIdx.x & 63) == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); \
        atomicAdd(&wt_prof2[slot], t_ - wt_sub_t); wt_sub_t = t_; } } while (0)
#define WT_SUBTICK_BEGIN unsigned long long wt_sub_t = __builtin_readcyclecounter()
#else
#define WT_SUBTICK(slot) do { } while (0)
#define WT_SUBTICK_BEGIN do { } while (0)
#endif
// (Round 2 also tried: both sets sorted, set 1 parked, two branch-free binary searches per set-0 value with
// the set-0 keys in a register vector indexed in GPR-index mode.  No faster -- the kernel is bound by
// instruction issue and the searches compiled to ~190 instructions per value, as many as the 2 x 64
// compare / add-carry pairs below -- and the build was not stable at 128 slots on MI355X (sporadic
// corrupted run counts at chromosome size), so it was dropped.  DESIGN A.1.)
// MWU over a register column (wt_gather_regs): set 0 sorted by a register network and parked in
// this lane's LDS column (n_set0 * 4 B -- the only LDS the reducer needs); then ONE pass over the
// sorted set-0 values e, each compared with the set-1 registers (compile-time indices):
//   L_e = #set-1 values < x_e,   t_e = #set-1 values == x_e,   last_e = next set-0 value differs,
// feeding the reference's tie state machine (setComparisons.c:335-359) in the reference's order --
// the table's stable sort puts set-0 entries first inside a tie group, and tied set-0 entries are
// interchangeable (same L, same t; `last` is positional).  No attribute slab, no second phase.
// setComparisons.c:361-366 for U1 (a multiple of 1/2, whatever the tie state machine did): mu is an integer (:386, C int division),
// so |U1 - mu| = k / 2 exactly and the reference's value is a function of the integer k alone -- read from a table the HOST
// filled with the reference's own expression and the host's erf (bit-identical to what the reference prints on this
// platform; the device's erf agrees to ~1e-16 and costs ~3 000 instructions per position, a fifth of the register-column
// kernel).
WT_DEV double wt_mwu_value(const WtParams &P, double U1, double mu, double sigma) {
    if (P.mwu_table) {
        const double d = U1 > mu ? U1 - mu : mu - U1;
        const double k2 = d * 2.0;
        const int k = k2 >= (double) P.mwu_kmax ? P.mwu_kmax : (int) k2;
        return P.mwu_table[k];
    }
    return (U1 > mu) ? 2 * erf((mu - U1) / sigma) : 2 * erf((U1 - mu) / sigma);
}

#ifndef WT_MWU_SKIP_PADS
#define WT_MWU_SKIP_PADS 0      // (measured, round 5: SLOWER -- 36.8 against 35.6 ms for 50 v 50, 34.6 against 32.1 for 40 v 60: the scalar branches cost the schedule more than 32 VALU instructions per skipped block)
#endif
template <int NR>
WT_DEV double wt_mwu_regs(const WtParams &P, uint32_t *col, const uint32_t *col2, float *xs, int colstride) {
    constexpr int H = NR / 2;
    const int na = P.n_set0, nb = P.n_tracks - P.n_set0;
    // sort set 0 as order-preserving keys (pads: 0xffffffff, they end up last)
#pragma unroll
    for (int s = 0; s < H; s++) col[s] = s < na ? wt_key32(__builtin_bit_cast(float, col[s])) : 0xffffffffu;
    wt_sort_regs<H, false>(col);
#pragma unroll
    for (int s = 0; s < H; s++)
        if (s < na) xs[(size_t) s * colstride] = wt_unkey32(col[s]);
    float y[H];
#pragma unroll
    for (int s = 0; s < H; s++) y[s] = __builtin_bit_cast(float, col2[s]);
    const double mu = (double) (na * nb / 2);                               // :386 int division
    const double sigma = sqrt((double) (na * nb * (na + nb + 1) / 12));     // :387 int division
    double U1 = 0;
    int ties = 0, prevTies = 0;
    float x = xs[0];
    for (int e = 0; e < na; e++) {
        const float xn = e + 1 < na ? xs[(size_t) (e + 1) * colstride] : x;
        int L = 0, t = 0;
        // (blocks of 8 set-1 slots; the blocks past set 1 hold NaN pads that compare false -- skipped: nb is uniform, so the test
        //  is a scalar branch, and 50 tracks per set leave the last of the eight blocks of a 64-slot column out: round 5)
#pragma unroll
        for (int s0 = 0; s0 < H; s0 += 8) {
#if WT_MWU_SKIP_PADS
            if (s0 < nb)
#endif
            {
#pragma unroll
                for (int s = s0; s < s0 + 8; s++) { L += (y[s] < x); t += (y[s] == x); }
            }
        }
        const bool last = (e + 1 == na) || !(xn == x);
        U1 += L;                                      // :336  U1 += index - prev
        if (ties) {                                   // :337-346
            if (last) prevTies += t;
            U1 -= prevTies / 2.0;
            U1 += (ties - prevTies) / 2.0;
            if (prevTies == ties) prevTies = ties = 0;
        } else {                                      // :347-354
            ties += t;
            if (ties) U1 += ties / 2.0;
        }
        x = xn;
    }
    return wt_mwu_value(P, U1, mu, sigma);
}

template <int OP, class ValT, class ScrT, int K, int NR>
WT_DEV void wt_eval_chunk(const WtParams &P, const WtCtx &c, int p0, WtAcc<K, NR> &A, int pass, int t_lo, int t_hi,
                          char *scratch, int lane_col, int colstride, unsigned emit_bits, long long run0) {
    const int w32 = p0 >> 5, b0 = p0 & 31;
    uint32_t mask[K];
#pragma unroll
    for (int k = 0; k < K; k++) mask[k] = (2u << (b0 + k)) - 1u;
    if constexpr (NR > 0 && (OP == WT_OP_MEDIAN || OP == WT_OP_MWU)) {
        // Register columns: the lane's K positions one after the other (the column is reused), each
        // evaluated to the end -- gather, exchange network, result.  K = 4 positions per lane keep
        // the window at 1024 bp for 256 lanes: the per-window costs (bitmaps, scans, look-back) are
        // shared by four times the positions a one-position-per-lane window would hold.
#pragma unroll 1
        for (int k = 0; k < K; k++) {
            if (!((emit_bits >> k) & 1u)) continue;
            WT_SUBTICK_BEGIN;
            const bool nan = wt_gather_regs<OP, ValT, NR>(P, c, p0 + k, A.col, A.col2);
            WT_SUBTICK(0);
            double v;
            if constexpr (OP == WT_OP_MEDIAN) {
                // vals[N/2] (reducers.c:810) sits at slot NR/2 of the sorted padded column.  Lower half
                // ascending, upper half descending: the whole is bitonic, and after the first stage of
                // its merge (slot i keeps min, slot i + NR/2 max) every upper element is >= every lower
                // one -- slot NR/2 of the sorted column is the MINIMUM of the upper half: NR/2 max +
                // NR/2 - 1 min instead of the full merge.
                constexpr int H = NR / 2;
                wt_sort_regs<H, false>(A.col);
                wt_sort_regs<H, true>(A.col2);
                uint32_t m = 0xffffffffu;
#pragma unroll
                for (int i = 0; i < H; i++) {
                    const uint32_t hi = A.col[i] > A.col2[i] ? A.col[i] : A.col2[i];
                    m = hi < m ? hi : m;
                }
                v = (double) wt_unkey32(m);
                WT_SUBTICK(4);
            } else {
                v = wt_mwu_regs<NR>(P, A.col, A.col2, (float *) scratch + lane_col, P.W / K);
            }
            A.a[k] = nan ? wt_nan() : v;
        }
        return;
    }

    if (OP == WT_OP_SUM || OP == WT_OP_PRODUCT || OP == WT_OP_MEAN) {
        // reducers.c:259-292, 313-346, 367-402
        wt_for_tracks<ValT, ScrT, K>(P, c, t_lo, t_lo, t_hi, w32, b0, mask, true, [&](int, const WtFetchK<K> &F) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (OP == WT_OP_PRODUCT) A.a[k] *= F.x[k]; else A.a[k] += F.x[k];
            }
        });
        return;
    }
    if (OP == WT_OP_MIN || OP == WT_OP_MAX) {
        // reducers.c:125-168, 192-235: seed is 0 (not the default) when track 0 is absent
        int lo = t_lo;
        if (t_lo == 0) {
            WtFetchK<K> F;
            wt_fetch_group<ValT, ScrT, K>(P, c, 0, w32, b0, mask, 0.0, F);
#pragma unroll
            for (int k = 0; k < K; k++) { A.a[k] = F.x[k]; A.nan[k] = wt_isnan(F.x[k]); }
            lo = 1;
        }
        wt_for_tracks<ValT, ScrT, K>(P, c, t_lo, lo, t_hi, w32, b0, mask, true, [&](int, const WtFetchK<K> &F) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const double x = F.x[k];
                A.nan[k] |= wt_isnan(x);
                if (OP == WT_OP_MAX) { if (A.a[k] < x) A.a[k] = x; } else { if (A.a[k] > x) A.a[k] = x; }
            }
        });
        return;
    }
    if (OP == WT_OP_VAR || OP == WT_OP_STDDEV || OP == WT_OP_ENTROPY || OP == WT_OP_CV) {
        // reducers.c:428-479 (var), 511-563 (stddev; entropy installs the same pop, :665),
        // 672-725 (CV).  Pass 0 (mean) rounds every value through `float`; pass 1 sums squares.
        if (pass == 0) {
            wt_for_tracks<ValT, ScrT, K>(P, c, t_lo, t_lo, t_hi, w32, b0, mask, true, [&](int, const WtFetchK<K> &F) {
#pragma unroll
                for (int k = 0; k < K; k++) A.a[k] += (double) (float) F.x[k];
            });
        } else {
            wt_for_tracks<ValT, ScrT, K>(P, c, t_lo, t_lo, t_hi, w32, b0, mask, true, [&](int, const WtFetchK<K> &F) {
#pragma unroll
                for (int k = 0; k < K; k++) {
                    // var ignores absent tracks in pass 2 (:470-475); stddev / CV use their default
                    double diff = A.a[k] - F.x[k];
                    if (OP == WT_OP_VAR) diff = F.cov[k] ? diff : 0.0;
                    A.b[k] += diff * diff;
                }
            });
        }
        return;
    }
    if (OP == WT_OP_TTEST) {
        // setComparisons.c:60-117: sums over in-play tracks, counts over all tracks
        const int na = P.n_set0;
        const int mid = na < t_lo ? t_lo : (na > t_hi ? t_hi : na);
        wt_for_tracks<ValT, ScrT, K>(P, c, t_lo, t_lo, mid, w32, b0, mask, false, [&](int, const WtFetchK<K> &F) {
#pragma unroll
            for (int k = 0; k < K; k++)
                if (F.cov[k]) { A.a[k] += F.x[k]; A.b[k] += F.x[k] * F.x[k]; }
        });
        wt_for_tracks<ValT, ScrT, K>(P, c, t_lo, mid, t_hi, w32, b0, mask, false, [&](int, const WtFetchK<K> &F) {
#pragma unroll
            for (int k = 0; k < K; k++)
                if (F.cov[k]) { A.c2[k] += F.x[k]; A.d[k] += F.x[k] * F.x[k]; }
        });
        return;
    }
    if (OP == WT_OP_MEDIAN) {
        // gather: default-substituted values -> order-preserving keys in this lane's LDS column
        typedef typename std::conditional<sizeof(ScrT) == 4, uint32_t, uint64_t>::type KeyT;
        KeyT *col = (KeyT *) scratch + lane_col;
        bool nan = A.nan[0];
        wt_for_tracks_deep<ValT, ScrT, K>(P, c, t_lo, t_lo, t_hi, w32, b0, mask, [&](int i, const WtFetchK<K> &F) {
            const double x = F.x[0];
            nan |= wt_isnan(x);
            if (sizeof(ScrT) == 4) col[(size_t) i * colstride] = (KeyT) wt_key32((float) x);
            else col[(size_t) i * colstride] = (KeyT) wt_key64(x);
        });
        A.nan[0] = nan;
        return;
    }
    if (OP == WT_OP_MWU) {
        ScrT *val = (ScrT *) scratch + lane_col;                                   // [N][colstride]
        bool nan = A.nan[0];
        wt_for_tracks_deep<ValT, ScrT, K>(P, c, t_lo, t_lo, t_hi, w32, b0, mask, [&](int i, const WtFetchK<K> &F) {
            nan |= wt_isnan(F.x[0]);
            val[(size_t) i * colstride] = (ScrT) F.x[0];
        });
        A.nan[0] = nan;
        return;
    }
    if (OP == WT_OP_MULTIPLEX) {
        // the per-run value of the materialised Multiplexer is its inplay_count (multiplexer.h:27);
        // the values[] / inplay[] columns of this chunk's tracks go straight to the tile
        const int N = P.n_tracks;
        for (int g = t_lo; g < t_hi; g++) {
            WtFetchK<K> F;
            wt_fetch_group<ValT, double, K>(P, c, g - t_lo, w32, b0, mask, P.defaults[g], F);
            long long o = run0;
#pragma unroll
            for (int k = 0; k < K; k++) {
                A.a[k] += F.cov[k] ? 1.0 : 0.0;
                if ((emit_bits >> k) & 1u) {
                    if (o < P.capacity) {
                        P.o_tile[o * N + g] = F.x[k];
                        P.o_inplay[o * N + g] = (uint8_t) F.cov[k];
                    }
                    o++;
                }
            }
        }
        return;
    }
}

// Between the two passes of var / stddev / CV: the mean is final.
template <int OP, int K, int NR>
WT_DEV void wt_eval_mid(const WtParams &P, WtAcc<K, NR> &A) {
#pragma unroll
    for (int k = 0; k < K; k++) A.a[k] /= P.n_tracks;
}

// ---------------------------------------------------------------------------
// Mann-Whitney U (setComparisons.c:293-366) without sorting, in two steps so that SEVERAL lanes can
// share one run: the N^2 ranking is split over `nparts` lanes (they all read the run's value
// column; the set-0 blocks are dealt round robin), the short sequential tie scan is done by one.
// The value columns cap the lanes a CU can hold (N * 4 B per run), and with one wave per SIMD the
// ranking was issue-bound: two lanes per run double the waves for the same LDS.
// The reference sorts the n1+n2 (value,set)
// pairs stably (set-0 entries first inside a tie group) and scans them; every quantity the
// scan uses for a set-0 element e is a COUNT:
//   index - prev            = L_e  = #set-1 values <  x_e
//   set-1 entries tied to e = t_e  = #set-1 values == x_e, all of them after e in the table
//   "contiguous set-1 successors" (:338-341) = t_e if e is the last set-0 entry of its tie
//                             group, else 0 (the next entry is another set-0 one)
// and the scan visits the set-0 elements in (value, index) order, i.e. at rank
//   r_e = #set-0 values < x_e + #set-0 values == x_e with a smaller index.
// All counts come from two branch-free n1*n2 / n1*n1 loops over this lane's LDS column
// (independent reads: no data-dependent trip counts, no divergence); (L,t,last) are
// scattered to rank r_e and the reference's tie state machine then runs over n1 entries,
// performing the same double additions in the same order.
// ---------------------------------------------------------------------------
template <class ScrT>
WT_DEV void wt_mwu_rank(const WtParams &P, char *scratch, char *attr_base, int col, int colstride, int part, int nparts) {
    const int N = P.n_tracks;
    const int na = P.n_set0;
    const ScrT *val = (const ScrT *) scratch + col;                            // [N][colstride]
    uint32_t *attr = (uint32_t *) attr_base + col;                             // [na][colstride]
    // WT_MWU_EB set-0 elements per sweep over the column: every LDS read serves EB comparisons
    constexpr int EB = WT_MWU_EB;
    for (int e0 = part * EB; e0 < na; e0 += EB * nparts) {
        ScrT x[EB];
        int L[EB], t[EB], r[EB], later[EB];
#pragma unroll
        for (int q = 0; q < EB; q++) {
            x[q] = val[(size_t) (e0 + q < na ? e0 + q : na - 1) * colstride];
            L[q] = 0; t[q] = 0; r[q] = 0; later[q] = 0;
        }
        // Sweeps over [lo, hi) of the column in batches of WT_MWU_RB values that are read first and
        // consumed afterwards: the LDS reads of a batch are in flight together (with one wave
        // per SIMD their latency was fully exposed -- ~60 cycles per value).
        auto sweep = [&](int lo, int hi, auto body) {
            int j = lo;
            for (; j + WT_MWU_RB <= hi; j += WT_MWU_RB) {
                ScrT y[WT_MWU_RB];
#pragma unroll
                for (int u = 0; u < WT_MWU_RB; u++) y[u] = val[(size_t) (j + u) * colstride];
#pragma unroll
                for (int u = 0; u < WT_MWU_RB; u++) body(y[u], j + u);
            }
            for (; j < hi; j++) body(val[(size_t) j * colstride], j);
        };
        sweep(na, N, [&](ScrT y, int) {
#pragma unroll
            for (int q = 0; q < EB; q++) { L[q] += (y < x[q]); t[q] += (y == x[q]); }
        });
        // set-0 entries before the block only need (y <= x), the ones after it (y < x) and
        // (y == x); the index comparisons matter inside the block alone
        sweep(0, e0, [&](ScrT y, int) {
#pragma unroll
            for (int q = 0; q < EB; q++) r[q] += (y <= x[q]);
        });
        const int e1 = e0 + EB < na ? e0 + EB : na;
        for (int j = e0; j < e1; j++) {
            const ScrT y = val[(size_t) j * colstride];
#pragma unroll
            for (int q = 0; q < EB; q++) {
                r[q] += (y < x[q]) | ((y == x[q]) & (j < e0 + q));
                later[q] += (y == x[q]) & (j > e0 + q);
            }
        }
        sweep(e1, na, [&](ScrT y, int) {
#pragma unroll
            for (int q = 0; q < EB; q++) { r[q] += (y < x[q]); later[q] += (y == x[q]); }
        });
#pragma unroll
        for (int q = 0; q < EB; q++)
            if (e0 + q < na)
                attr[(size_t) r[q] * colstride] = ((uint32_t) L[q] << 17) | ((uint32_t) t[q] << 1) | (later[q] == 0 ? 1u : 0u);
    }
}

WT_DEV double wt_mwu_tail(const WtParams &P, char *attr_base, int col, int colstride) {
    const int na = P.n_set0, nb = P.n_tracks - P.n_set0;
    const uint32_t *attr = (const uint32_t *) attr_base + col;
    // (:386-387: C int products and divisions; wrapping like wt_mwu_make_table's)
    const int nab = (int) ((unsigned) na * (unsigned) nb);
    const double mu = (double) (nab / 2);
    const double sigma = sqrt((double) ((int) ((unsigned) nab * ((unsigned) na + (unsigned) nb + 1u)) / 12));
    double U1 = 0;
    int ties = 0, prevTies = 0;
    for (int q = 0; q < na; q++) {
        const uint32_t a = attr[(size_t) q * colstride];
        const int L = (int) (a >> 17), t = (int) ((a >> 1) & 0xffffu);
        U1 += L;                                      // :336  U1 += index - prev
        if (ties) {                                   // :337-346
            if (a & 1u) prevTies += t;
            U1 -= prevTies / 2.0;
            U1 += (ties - prevTies) / 2.0;
            if (prevTies == ties) prevTies = ties = 0;
        } else {                                      // :347-354
            ties += t;
            if (ties) U1 += ties / 2.0;
        }
    }
    return wt_mwu_value(P, U1, mu, sigma);
}

template <int OP, class ValT, class ScrT, int K, int NR>
WT_DEV void wt_eval_finish(const WtParams &P, WtAcc<K, NR> &A, double (&res)[K], char *scratch, char *attr_base,
                           int lane_col, int colstride, unsigned emit_bits) {
    const int N = P.n_tracks;
    if constexpr (NR > 0 && (OP == WT_OP_MEDIAN || OP == WT_OP_MWU)) {      // evaluated to the end by wt_eval_chunk
#pragma unroll
        for (int k = 0; k < K; k++) res[k] = A.a[k];
        return;
    }
    if (OP == WT_OP_SUM || OP == WT_OP_PRODUCT || OP == WT_OP_MEAN) {
#pragma unroll
        for (int k = 0; k < K; k++) res[k] = (OP == WT_OP_MEAN) ? A.a[k] / N : A.a[k];
        return;
    }
    if (OP == WT_OP_MIN || OP == WT_OP_MAX) {
#pragma unroll
        for (int k = 0; k < K; k++) res[k] = A.nan[k] ? wt_nan() : A.a[k];
        return;
    }
    if (OP == WT_OP_VAR || OP == WT_OP_STDDEV || OP == WT_OP_ENTROPY || OP == WT_OP_CV) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const double mean = A.a[k];
            double a = A.b[k] / N;
            bool bad = wt_isnan(mean);
            if (OP == WT_OP_VAR) { if (N < 2) bad = true; }
            else {
                a = sqrt(a);
                if (OP == WT_OP_CV) { if (mean == 0) bad = true; a /= mean; }
            }
            res[k] = bad ? wt_nan() : a;
        }
        return;
    }
    if (OP == WT_OP_TTEST) {
        const int na = P.n_set0, nb = N - P.n_set0;
        // statistic and degrees of freedom of all K positions first (the accumulators die here),
        // then ONE rolled loop around the Student-t tail: K inlined copies of it (lgamma x3, log,
        // exp, a continued fraction) made the kernel spill 600+ bytes per lane
        double tt[K], nn[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const double m1 = A.a[k] / na, m2 = A.c2[k] / nb;
            const double msq1 = A.b[k] / na, msq2 = A.d[k] / nb;
            const double var1 = msq1 - m1 * m1, var2 = msq2 - m2 * m2;
            double t = (m1 - m2) / sqrt(var1 / na + var2 / nb);
            if (t < 0) t = -t;
            const double den = var1 / na + var2 / nb;
            const double c1 = (double) ((long long) na * na * (na - 1));
            const double c2 = (double) ((long long) nb * nb * (nb - 1));
            nn[k] = den * den / ((var1 * var1) / c1 + (var2 * var2) / c2);
            tt[k] = (var1 + var2 == 0) ? wt_nan() : t;          // setComparisons.c:98 -> NaN
        }
        // (positions that start no emitted run are skipped: the tail is the expensive part)
#pragma unroll 1
        for (int k = 0; k < K; k++)
            res[k] = (wt_isnan(tt[k]) || !((emit_bits >> k) & 1u)) ? wt_nan() : wt_ttest_tail(tt[k], nn[k]);
        return;
    }
    if (OP == WT_OP_MEDIAN) {
        // reducers.c:780-813: upper median.  Selection by bitwise binary search over the keys of
        // this lane's LDS column (no divergence, no writes).
        typedef typename std::conditional<sizeof(ScrT) == 4, uint32_t, uint64_t>::type KeyT;
        const KeyT *col = (const KeyT *) scratch + lane_col;
        const int kth = N / 2;       // 0-based rank of vals[N/2]
        // largest key Kk such that count(keys < Kk) <= kth  ==  the kth smallest key
        // WT_MEDIAN_BITS key bits are decided per sweep over the column (2^B - 1 trial keys counted
        // at once): fewer sweeps, each LDS read serves several comparisons (the loop is bound by
        // LDS latency at the few waves per CU the columns leave room for).
        constexpr int B = WT_MEDIAN_BITS, NT = (1 << B) - 1, KB = (int) sizeof(KeyT) * 8;
        static_assert(KB % B == 0, "WT_MEDIAN_BITS must divide the key width");
        // One sweep first: bits on which ALL keys agree are already decided (they are the median's
        // bits too), and B-bit groups without a disputed bit need no counting sweep.  Signal
        // tracks are mostly counts or coarsely rounded values, whose low mantissa bits are all
        // zero: typically half of the sweeps go away.
        KeyT all_and = ~(KeyT) 0, all_or = 0;
        // (column sweeps read WT_MWU_RB keys ahead, then consume them: LDS latency overlapped)
        auto sweep = [&](auto body) {
            int i = 0;
            for (; i + WT_MWU_RB <= N; i += WT_MWU_RB) {
                KeyT k[WT_MWU_RB];
#pragma unroll
                for (int u = 0; u < WT_MWU_RB; u++) k[u] = col[(size_t) (i + u) * colstride];
#pragma unroll
                for (int u = 0; u < WT_MWU_RB; u++) body(k[u]);
            }
            for (; i < N; i++) body(col[(size_t) i * colstride]);
        };
        sweep([&](KeyT key) { all_and &= key; all_or |= key; });
        const KeyT disputed = all_and ^ all_or;
        KeyT Kk = 0;
        for (int b = KB - B; b >= 0; b -= B) {
            const KeyT group = (KeyT) NT << b;
            if (!(disputed & group)) { Kk |= all_and & group; continue; }
            KeyT trial[NT];
            int below[NT];
#pragma unroll
            for (int q = 0; q < NT; q++) { trial[q] = Kk | ((KeyT) (q + 1) << b); below[q] = 0; }
            sweep([&](KeyT key) {
#pragma unroll
                for (int q = 0; q < NT; q++) below[q] += (key < trial[q]);
            });
#pragma unroll
            for (int q = 0; q < NT; q++)
                if (below[q] <= kth) Kk = trial[q];      // below[] is non-decreasing in q: the last hit wins
        }
        const double m = (sizeof(ScrT) == 4) ? (double) wt_unkey32((uint32_t) Kk) : wt_unkey64((uint64_t) Kk);
        res[0] = A.nan[0] ? wt_nan() : m;
        return;
    }
    if (OP == WT_OP_MWU) return;    // ranking and tie scan are their own phases (wt_phase_mwu_rank / _tail)
    if (OP == WT_OP_MULTIPLEX) {
#pragma unroll
        for (int k = 0; k < K; k++) res[k] = A.a[k];
        return;
    }
}

// First true breakpoint after window position p (absolute coordinate).  Loop-free: the
// emask phase records for every word of U the next non-empty word (nextw).
WT_DEV int32_t wt_next_breakpoint(const WtParams &P, const WtCtx &c, int p) {
    const int w = p >> 6;
    const uint64_t here = c.U[w] & ~wt_mask_incl(p & 63);
    const int w2 = here ? w : (int) c.nextw[w];
    if (w2 < 0) return c.sh->next_bp;
    const uint64_t bits = here ? here : c.U[w2];
    return c.sh->w0 + w2 * 64 + wt_ctz64(bits);
}

// ---------------------------------------------------------------------------
// Phase 4: emitted-run bitmap E = true breakpoints & emission predicate & range.
// The predicate comes from the coverage summaries of the count phase, so the run
// count of the window is known BEFORE the reducers run: the look-back can overlap
// the evaluation instead of following it.
// ---------------------------------------------------------------------------
WT_DEV void wt_phase_emask(const WtParams &P, WtCtx &c, bool two, int tid, int nt) {
    const int nw32 = P.n_words * 2;
    const uint64_t *any0 = (const uint64_t *) c.cover;
    const uint64_t *all0 = (const uint64_t *) (c.cover + nw32);
    const uint64_t *any1 = (const uint64_t *) (c.cover + 2 * nw32);
    const uint64_t *all1 = (const uint64_t *) (c.cover + 3 * nw32);
    for (int w = tid; w < P.n_words; w += nt) {
        uint64_t emit = (P.flags & WT_STRICT_SET0) ? all0[w] : any0[w];     // multiplexer.c:120,125
        if (two) emit &= (P.flags & WT_STRICT_SET1) ? all1[w] : any1[w];    // setComparisons.c:48-54
        // run starts at or beyond the range end belong to the next batch / shard
        const long long room = (long long) c.sh->emit_hi - ((long long) c.sh->w0 + (long long) w * 64);
        if (room < 64) emit &= (room <= 0) ? 0ull : ((1ull << room) - 1ull);
        c.E[w] = c.U[w] & emit;
    }
    if (tid == nt - 1) {            // tiny backward scan (n_words <= 512)
        int last = -1;
        for (int w = P.n_words - 1; w >= 0; w--) {
            c.nextw[w] = (int16_t) last;
            if (c.U[w]) last = w;
        }
    }
}

// ---------------------------------------------------------------------------
// Phase 5: exclusive popcount prefix over E
// ---------------------------------------------------------------------------
WT_DEV void wt_phase_escan(const WtParams &P, WtCtx &c, int tid, int nt) {
    for (int w = tid; w <= P.n_words; w += nt) {
        unsigned s = 0;
        for (int x = 0; x < w; x++) s += (unsigned) wt_popc64(c.E[x]);
        c.epfx[w] = s;
    }
}

// ---------------------------------------------------------------------------
// Phase 6: evaluate the reducer at every emitted run start.  Lane `tid` owns the K consecutive
// positions [tid*K, tid*K+K); its accumulators (WtAcc) live in registers across the chunk loop.
// ---------------------------------------------------------------------------
template <int K>
WT_DEV unsigned wt_lane_emit_bits(const WtParams &P, const WtCtx &c, int tid) {
    const int p0 = tid * K;
    if (p0 >= P.W) return 0u;
    return (unsigned) ((c.E[p0 >> 6] >> (p0 & 63)) & ((1ull << K) - 1ull));
}

template <int OP, class ValT, class ScrT, int K, int NR>
// `all`: the emitted bitmap is not known yet (first pass fused into the chunk sweep that builds it,
// see the kernel): evaluate every position of every lane.
WT_DEV void wt_phase_eval_chunk(const WtParams &P, WtCtx &c, WtAcc<K, NR> &A, int pass, int t_lo, int t_hi, bool all,
                                int tid, int nt) {
    const int p0 = tid * K;
    if (p0 >= P.W) return;
    const unsigned emit_bits = all ? ((1u << K) - 1u) : wt_lane_emit_bits<K>(P, c, tid);
    if (!emit_bits) return;
    long long run0 = 0;
    if (OP == WT_OP_MULTIPLEX) {    // global index of the lane's first emitted run (look-back already done)
        const int w = p0 >> 6, b0 = p0 & 63;
        const uint64_t below0 = b0 ? wt_mask_incl(b0 - 1) : 0ull;
        run0 = c.sh->goffset + c.epfx[w] + wt_popc64(c.E[w] & below0);
    }
    wt_eval_chunk<OP, ValT, ScrT, K>(P, c, p0, A, pass, t_lo, t_hi, c.scratch, tid, P.W, emit_bits, run0);
}

template <int OP, class ValT, class ScrT, int K, int NR>
WT_DEV void wt_phase_eval_finish(const WtParams &P, WtCtx &c, WtAcc<K, NR> &A, WtLane<K> &L, int tid, int nt) {
    const unsigned emit_bits = wt_lane_emit_bits<K>(P, c, tid);
    if (!emit_bits) return;
    wt_eval_finish<OP, ValT, ScrT, K>(P, A, L.res, c.scratch, c.attr, tid, P.W, emit_bits);
}

// MWU: lanes tid, tid + W, ... share the run at window position tid % W (K == 1; the plan may give a
// workgroup lanes_per_pos * W lanes)
template <class ScrT>
WT_DEV void wt_phase_mwu_rank(const WtParams &P, WtCtx &c, int tid, int nt) {
    const int p = tid % P.W, part = tid / P.W, nparts = nt / P.W;
    if (!((c.E[p >> 6] >> (p & 63)) & 1ull)) return;
    wt_mwu_rank<ScrT>(P, c.scratch, c.attr, p, P.W, part, nparts);
}
template <int K, int NR>
WT_DEV void wt_phase_mwu_tail(const WtParams &P, WtCtx &c, const WtAcc<K, NR> &A, WtLane<K> &L, int tid, int nt) {
    if (tid >= P.W || !((c.E[tid >> 6] >> (tid & 63)) & 1ull)) return;
    L.res[0] = A.nan[0] ? wt_nan() : wt_mwu_tail(P, c.attr, tid, P.W);
}

// ---------------------------------------------------------------------------
// Look-back: global offset of the window's first run.  Decoupled look-back over
// 64-bit {flag,count} status words:
//   status[k] = AGG|count once window k knows its own count,
//               PFX|inclusive_prefix once it also knows everything before it.
// Windows are handed out in order by the ticket, so every predecessor has
// started; the spin is bounded and reports WT_ERR_LOOKBACK instead of hanging.
// ---------------------------------------------------------------------------
// Bookkeeping shared by both look-back flavours once `excl` is known (one lane).
WT_DEV void wt_lookback_finish(const WtParams &P, WtCtx &c, long long k, unsigned long long excl,
                               unsigned long long mine) {
    WtShared *sh = c.sh;
    wt_status_store(&P.status[k], WT_FLAG_PFX | (excl + mine));
    sh->goffset = (long long) excl;
    const int ch = sh->chrom;
    if (k == P.c_first_win[ch]) P.chrom_run_off[ch] = (long long) excl;
    if (k == P.n_windows - 1) {
        P.chrom_run_off[P.n_chrom] = (long long) (excl + mine);
        P.counters[WT_CTR_RUNS] = excl + mine;
    }
    if ((long long) (excl + mine) > P.capacity) wt_glb_or64(&P.counters[WT_CTR_ERROR], WT_ERR_CAPACITY);
}

// Sequential flavour (one lane): used by the CPU emulator, kept as the plain statement
// of the protocol.
WT_DEV void wt_phase_lookback(const WtParams &P, WtCtx &c, long long k, unsigned long long mine);
WT_DEV void wt_phase_lookback(const WtParams &P, WtCtx &c, long long k) {
    wt_phase_lookback(P, c, k, (unsigned long long) c.epfx[P.n_words]);
}
WT_DEV void wt_phase_lookback(const WtParams &P, WtCtx &c, long long k, unsigned long long mine) {
    c.sh->n_emit = (int32_t) mine;
    unsigned long long excl = 0;
    if (k > 0) {
        wt_status_store(&P.status[k], WT_FLAG_AGG | mine);
        long long j = k - 1;
        for (;;) {
            unsigned long long v = wt_status_load(&P.status[j]);
            unsigned spins = 0;
            while (v == 0) {
                wt_backoff();
                v = wt_status_load(&P.status[j]);
                if (++spins > (1u << 24)) {
                    wt_glb_or64(&P.counters[WT_CTR_ERROR], WT_ERR_LOOKBACK);
                    v = WT_FLAG_PFX;    // give up: offsets are garbage, error is reported
                }
            }
            excl += v & WT_VAL_MASK;
            if (v & WT_FLAG_PFX) break;
            j--;
        }
    }
    wt_lookback_finish(P, c, k, excl, mine);
}

#ifndef WT_EMU
// Wave flavour (the 64 lanes of wave 0), split in two around the evaluation phase:
//   publish  (before eval)  AGG|count of this window -- successors never wait for our reducers
//   complete (after eval)   lane L inspects window base-L: 64 predecessors per round trip; by
//                           now they have almost always published, so there is no spinning.
WT_DEV void wt_lookback_publish(const WtParams &P, WtCtx &c, long long k, unsigned long long mine) {
    c.sh->n_emit = (int32_t) mine;
    if (k > 0) wt_status_store(&P.status[k], WT_FLAG_AGG | mine);
}
WT_DEV void wt_lookback_publish(const WtParams &P, WtCtx &c, long long k) {
    wt_lookback_publish(P, c, k, (unsigned long long) c.epfx[P.n_words]);
}

WT_DEV void wt_lookback_complete(const WtParams &P, WtCtx &c, long long k, int lane, unsigned long long mine);
WT_DEV void wt_lookback_complete(const WtParams &P, WtCtx &c, long long k, int lane) {
    wt_lookback_complete(P, c, k, lane, (unsigned long long) c.epfx[P.n_words]);
}
#ifndef WT_LOOKBACK_DEPTH
#define WT_LOOKBACK_DEPTH 1     // status words per lane fetched at once (4: one round trip covers 256 predecessors -- measured, round 5: no effect)
#endif
WT_DEV void wt_lookback_complete(const WtParams &P, WtCtx &c, long long k, int lane, unsigned long long mine) {
    unsigned long long excl = 0;
    long long base = k - 1;
    // A persistent grid keeps ~one window per workgroup in flight (256 on this GPU, 512 with two workgroups per CU), and a window
    // that has just published its count typically finds that many predecessors still without an inclusive prefix: at 64 status
    // words per round trip that is up to four DEPENDENT round trips.  With WT_LOOKBACK_DEPTH > 1 the words of the next rounds are
    // requested together (a round whose words were not all ready polls as before).  Measured on MI355X (round 5, tools/r5_ab.sh,
    // depth 4 against 1 on one box, twice): C2 67.96 / 68.10 against 67.94 / 67.71 ms, mean run 200 33.03 / 32.94 against 32.92 /
    // 32.83, C3 70.63 / 70.60 against 70.43 / 70.41 -- the look-back's 8-11 % of a window is waiting for the predecessors to
    // PUBLISH, not for the round trips that fetch what they published.  Default 1.
    unsigned long long ahead[WT_LOOKBACK_DEPTH];
    int have = 0;               // rounds of ahead[] not yet consumed (ahead[WT_LOOKBACK_DEPTH - have] is the next)
    while (base >= 0) {
        const long long j = base - lane;
        if (have == 0) {
#pragma unroll
            for (int r = 0; r < WT_LOOKBACK_DEPTH; r++) {
                const long long jr = j - 64ll * r;
                // windows before the first one behave like a published prefix of 0
                ahead[r] = (jr >= 0) ? wt_status_load(&P.status[jr]) : WT_FLAG_PFX;
            }
            have = WT_LOOKBACK_DEPTH;
        }
        unsigned long long v = ahead[0];
#pragma unroll
        for (int r = 1; r < WT_LOOKBACK_DEPTH; r++) v = (WT_LOOKBACK_DEPTH - have == r) ? ahead[r] : v;
        have--;
        unsigned long long pfx;
        unsigned spins = 0;
        for (;;) {
            const unsigned long long ready = __ballot(v != 0);
            pfx = __ballot((v & WT_FLAG_PFX) != 0);
            // lanes 0..p must be ready, p = nearest lane holding a prefix (all 64 if none)
            const unsigned long long need = pfx ? (((pfx & (0ull - pfx)) << 1) - 1ull) : ~0ull;
            if ((ready & need) == need) break;
            if (++spins > (1u << 22)) {
                if (lane == 0) wt_glb_or64(&P.counters[WT_CTR_ERROR], WT_ERR_LOOKBACK);
                pfx = 1ull;         // give up: offsets are garbage, error is reported
                break;
            }
            if (v == 0) { wt_backoff(); v = wt_status_load(&P.status[j]); }
        }
        const int p = pfx ? (int) wt_ctz64(pfx) : 63;
        unsigned long long part = (lane <= p) ? (v & WT_VAL_MASK) : 0ull;
        excl += wt_wave_sum_u64(part);
        if (pfx) break;
        base -= 64;
    }
    if (lane == 0) wt_lookback_finish(P, c, k, excl, mine);
}
#endif

// ---------------------------------------------------------------------------
// Phase 7: write the emitted runs at their global positions
// ---------------------------------------------------------------------------
template <int OP, class ValT, int K>
WT_DEV void wt_phase_write(const WtParams &P, WtCtx &c, const WtLane<K> &L, int tid, int nt) {
    const int p0 = tid * K;
    unsigned emit_bits = 0;
    if (p0 < P.W) emit_bits = (unsigned) ((c.E[p0 >> 6] >> (p0 & 63)) & ((1ull << K) - 1ull));
    unsigned long long bp = 0;
    if (emit_bits) {
        const long long goff = c.sh->goffset;
        const int32_t w0 = c.sh->w0;
        const int w = p0 >> 6, b0 = p0 & 63;
        const uint64_t below0 = b0 ? wt_mask_incl(b0 - 1) : 0ull;
        long long idx = goff + c.epfx[w] + wt_popc64(c.E[w] & below0);
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (!((emit_bits >> k) & 1u)) continue;
            const long long o = idx++;
            const int p = p0 + k;
            const int32_t fin = wt_next_breakpoint(P, c, p);
            bp += (unsigned long long) (fin - (w0 + p));
            if (o >= P.capacity) continue;
            P.o_start[o] = w0 + p;
            P.o_finish[o] = fin;
            P.o_value[o] = L.res[k];
        }
        wt_lds_add64(&c.sh->bp_sum, bp);
    }
}

// Per-window statistics -> global counters (one lane, after every lane's contribution is in).
WT_DEV void wt_window_stats(const WtParams &P, WtCtx &c) {
    if (c.sh->bp_sum) wt_glb_add64(&P.counters[WT_CTR_BP], c.sh->bp_sum);
    if (c.sh->n_intervals) wt_glb_add64(&P.counters[WT_CTR_INTERVALS], c.sh->n_intervals);
}

// ---------------------------------------------------------------------------
// Window index: one lane per input interval; interval jr of (chrom, track)
// claims every window boundary b with finish[jr-1] < b <= finish[jr]:
//   widx[row(b)][track] = jr  == first interval with finish >= b.
// Rows past the last interval get n (= none).  Empty (chrom,track) segments
// rely on the rows having been zeroed (0 == n).
// `seg` is an in/out hint: the segment of the previous interval handled by
// this lane (intervals are visited in increasing g, so it only moves forward).
// ---------------------------------------------------------------------------
WT_DEV long long wt_index_find_segment(const WtParams &P, long long g) {
    long long lo = 0, hi = (long long) P.n_chrom * P.n_tracks;   // seg_off[lo] <= g < seg_off[hi]
    while (hi - lo > 1) {
        const long long mid = (lo + hi) >> 1;
        if (P.seg_off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

// Per-lane cursor: the (chrom, track) segment the lane is in and its constants.
struct WtIndexCursor {
    long long seg, s0, s1;      // segment index, its first interval, one past its last
    long long nw, rowbase;
    int32_t cb;
    int i;
};

WT_DEV void wt_index_cursor_set(const WtParams &P, WtIndexCursor &c, long long seg) {
    const int N = P.n_tracks;
    c.seg = seg;
    c.s0 = P.seg_off[seg];
    c.s1 = P.seg_off[seg + 1];
    const int ch = (int) (seg / N);
    c.i = (int) (seg - (long long) ch * N);
    c.cb = P.cbase[ch];
    c.nw = P.c_nwin[ch];
    c.rowbase = P.c_first_win[ch] + ch;
}

// f = finish[g], pf = finish[g-1] (ignored for the first interval of a segment)
WT_DEV void wt_index_apply(const WtParams &P, WtIndexCursor &c, long long g, int32_t f, int32_t pf) {
    while (g >= c.s1) wt_index_cursor_set(P, c, c.seg + 1);      // empty segments are skipped too
    // by far the most common case, tested first in 32-bit arithmetic: an interior interval of its
    // segment whose finish lies in the same window as its predecessor's claims nothing
    if (g > c.s0 && g + 1 < c.s1 && pf >= c.cb &&
        ((uint32_t) (pf - c.cb) >> P.logW) == ((uint32_t) (f - c.cb) >> P.logW)) return;
    const long long jr = g - c.s0;
    const int32_t cb = c.cb;
    // cbase may be a range start above the data start: intervals ending before it claim nothing
    long long m_lo = 0;
    if (jr > 0 && pf >= cb) m_lo = (long long) ((uint32_t) (pf - cb) >> P.logW) + 1;
    long long m_hi = (f >= cb) ? (long long) ((uint32_t) (f - cb) >> P.logW) : -1;
    const bool last = (g + 1 == c.s1);
    if (m_lo > m_hi && !last) return;                    // common case: no boundary inside this interval
    const int N = P.n_tracks;
    if (m_hi > c.nw) m_hi = c.nw;
    for (long long m = m_lo; m <= m_hi; m++) P.widx[(size_t) (c.rowbase + m) * N + c.i] = (uint32_t) jr;
    if (last)
        for (long long m = m_hi + 1; m <= c.nw; m++) P.widx[(size_t) (c.rowbase + m) * N + c.i] = (uint32_t) (jr + 1);
}

#include "wt_delta.h"
#include "wt_walk.h"
#include "wt_mwalk.h"

#endif  // WT_CORE_H_
