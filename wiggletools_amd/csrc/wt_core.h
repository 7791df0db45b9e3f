// wt_core.h -- the "bitmap multiplexer": window-local breakpoint alignment +
// per-run reducers.  This header is the single source of the kernel logic.  It
// is compiled
//   * by hipcc for gfx950 inside wt_kernels.hip (the product), and
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
//               rank = cnt_i[p/64] + popc(S_i[p/64] & mask(p));  idx = base_i + rank - 1
//           and `covered` = p < finish[idx].
//   reducers.c / setComparisons.c ...ReductionPop
//        -> eval_position<OP>(): one lane per run, tracks visited in index order
//           i = 0..N-1 in f64, i.e. the reference's own summation order.
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

#define WT_MAX_ITERS 4           // positions per lane per window: W <= WT_MAX_ITERS * blockDim
#define WT_FLAG_AGG  (1ull << 62)
#define WT_FLAG_PFX  (2ull << 62)
#define WT_VAL_MASK  ((1ull << 62) - 1)

// counters[] slots (device, 64-bit each)
enum { WT_CTR_TICKET = 0, WT_CTR_RUNS = 1, WT_CTR_BP = 2, WT_CTR_INTERVALS = 3, WT_CTR_ERROR = 4, WT_CTR_N = 8 };
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
    // ---- chromosome tables ----
    const int32_t *cbase;         // [n_chrom] position of window 0 of the chromosome
    const int32_t *c_nwin;        // [n_chrom] number of windows (>= 1)
    const int64_t *c_first_win;   // [n_chrom+1] first global window index
    // ---- windows ----
    int32_t W;                    // window width in bp, multiple of 64
    int32_t n_words;              // W / 64
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
    // ---- output ----
    int64_t capacity;
    int32_t *o_start;
    int32_t *o_finish;
    double *o_value;
    int64_t *chrom_run_off;       // [n_chrom+1]
    double *o_tile;               // WT_OP_MULTIPLEX only: [capacity * n_tracks]
    uint8_t *o_inplay;            // WT_OP_MULTIPLEX only
    // ---- LDS carve (bytes from the dynamic LDS base; all multiples of 16) ----
    int32_t spitch;               // u64 words per S_i row (n_words + 1: bank spread)
    int32_t cpitch;               // u16 entries per cnt_i row
    int32_t off_S, off_cnt, off_U, off_E, off_epfx, off_gbase, off_scratch, off_shared;
    int32_t lds_bytes;
};

// Block-shared scalars (live in LDS at off_shared)
struct WtShared {
    long long ticket;
    long long goffset;            // global index of this window's first emitted run
    int32_t chrom, w0, w1, nbits; // nbits = w1 - w0
    long long row;                // widx row of w0
    int32_t next_bp;              // first breakpoint >= w1 (INT32_MAX if none)
    int32_t n_emit;               // runs emitted by this window
    unsigned long long bp_sum;    // covered bp of this window
    unsigned long long n_intervals;
};

// Per-lane state that lives across phases (registers on the GPU)
struct WtLane {
    double res[WT_MAX_ITERS];
    int32_t fin[WT_MAX_ITERS];
    uint32_t emit;                // bit `it` set <=> position it*T+tid starts an emitted run
};

// LDS views
struct WtCtx {
    uint64_t *S;        // [n_tracks * spitch]
    uint16_t *cnt;      // [n_tracks * cpitch]
    uint64_t *U;        // [n_words] true breakpoints
    uint64_t *E;        // [n_words] emitted run starts
    uint32_t *epfx;     // [n_words + 1]
    long long *gbase;   // [n_tracks] global index of (first covering interval) - 1
    char *scratch;      // per-lane column scratch for median / MWU
    WtShared *sh;
};

WT_DEV void wt_ctx_init(WtCtx &c, const WtParams &P, char *lds) {
    c.S = (uint64_t *) (lds + P.off_S);
    c.cnt = (uint16_t *) (lds + P.off_cnt);
    c.U = (uint64_t *) (lds + P.off_U);
    c.E = (uint64_t *) (lds + P.off_E);
    c.epfx = (uint32_t *) (lds + P.off_epfx);
    c.gbase = (long long *) (lds + P.off_gbase);
    c.scratch = lds + P.off_scratch;
    c.sh = (WtShared *) (lds + P.off_shared);
}

// ---------------------------------------------------------------------------
// portability shims (device vs emulator)
// ---------------------------------------------------------------------------
#ifdef WT_EMU
WT_DEV int wt_popc64(uint64_t x) { return __builtin_popcountll(x); }
WT_DEV int wt_ctz64(uint64_t x) { return __builtin_ctzll(x); }
WT_DEV void wt_lds_or64(uint64_t *p, uint64_t v) { *p |= v; }
WT_DEV void wt_lds_min32(int32_t *p, int32_t v) { if (v < *p) *p = v; }
WT_DEV void wt_lds_add64(unsigned long long *p, unsigned long long v) { *p += v; }
WT_DEV unsigned long long wt_glb_add64(unsigned long long *p, unsigned long long v) {
    unsigned long long o = *p; *p += v; return o;
}
WT_DEV void wt_glb_or64(unsigned long long *p, unsigned long long v) { *p |= v; }
WT_DEV unsigned long long wt_status_load(unsigned long long *p) { return *p; }
WT_DEV void wt_status_store(unsigned long long *p, unsigned long long v) { *p = v; }
WT_DEV void wt_backoff() {}
#else
WT_DEV int wt_popc64(uint64_t x) { return __popcll(x); }
WT_DEV int wt_ctz64(uint64_t x) { return __ffsll((unsigned long long) x) - 1; }
WT_DEV void wt_lds_or64(uint64_t *p, uint64_t v) { atomicOr((unsigned long long *) p, (unsigned long long) v); }
WT_DEV void wt_lds_min32(int32_t *p, int32_t v) { atomicMin(p, v); }
WT_DEV void wt_lds_add64(unsigned long long *p, unsigned long long v) { atomicAdd(p, v); }
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

WT_DEV double wt_inc_beta(double a, double b, double x) {
    if (x <= 0) return 0;
    if (x >= 1) return 1;
    double lnfront = lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log1p(-x);
    if (x < (a + 1) / (a + b + 2)) return exp(lnfront) * wt_betacf(a, b, x) / a;
    return 1 - exp(lnfront) * wt_betacf(b, a, 1 - x) / b;
}

WT_DEV double wt_tdist_Q(double t, double nu) {
    if (wt_isnan(t) || wt_isnan(nu) || nu <= 0) return wt_nan();
    if (isinf(t)) return t > 0 ? 0.0 : 1.0;
    double x = nu / (nu + t * t);
    double tail = 0.5 * wt_inc_beta(nu / 2, 0.5, x);
    return t >= 0 ? tail : 1 - tail;
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
    sh->nbits = P.W;
    sh->row = k + ch;                 // one extra boundary row per chromosome
    sh->next_bp = 0x7fffffff;
    sh->n_emit = 0;
    sh->bp_sum = 0;
    sh->n_intervals = 0;
    sh->goffset = 0;
}

// ---------------------------------------------------------------------------
// Phase 1: clear the bitmaps
// ---------------------------------------------------------------------------
WT_DEV void wt_phase_zero(const WtParams &P, WtCtx &c, int tid, int nt) {
    const int nS = P.n_tracks * P.spitch;
    for (int x = tid; x < nS; x += nt) c.S[x] = 0;
    for (int x = tid; x < P.n_words; x += nt) { c.U[x] = 0; c.E[x] = 0; }
}

// ---------------------------------------------------------------------------
// Phase 2: stream the window's intervals, build S_i and U.
// One 64-lane group per track (coalesced reads of start[]/finish[]).
// Interval classes relative to the window [w0,w1):
//   f == w0           : true breakpoint at w0, covers nothing here
//   s >= w1           : first interval beyond the window: candidate for next_bp
//   otherwise         : covers part of the window; clipped start bit in S_i;
//                       true start/finish bits in U; f >= w1 -> candidate next_bp
// ---------------------------------------------------------------------------
WT_DEV void wt_phase_load(const WtParams &P, WtCtx &c, int tid, int nt) {
    const WtShared *sh = c.sh;
    const int N = P.n_tracks;
    const int32_t w0 = sh->w0, w1 = sh->w1;
    const int group = tid >> 6, lane = tid & 63, ngroups = nt >> 6;
    const uint32_t *row0 = P.widx + (size_t) sh->row * N;
    const uint32_t *row1 = row0 + N;
    for (int i = group; i < N; i += ngroups) {
        const long long seg = (long long) sh->chrom * N + i;
        const long long off = P.seg_off[seg];
        const long long n = P.seg_off[seg + 1] - off;
        long long lo = row0[i], hi = row1[i];
        if (hi >= n) hi = n - 1;
        uint64_t *Si = c.S + (size_t) i * P.spitch;
        if (lane == 0 && hi >= lo) wt_lds_add64(&c.sh->n_intervals, (unsigned long long) (hi - lo + 1));
        for (long long jr = lo + lane; jr <= hi; jr += 64) {
            const int32_t s = P.start[off + jr];
            const int32_t f = P.finish[off + jr];
            if (jr == lo) c.gbase[i] = off + lo - 1 + (f == w0 ? 1 : 0);
            if (f == w0) { wt_lds_or64(&c.U[0], 1ull); continue; }
            if (s >= w1) { wt_lds_min32(&c.sh->next_bp, s); continue; }
            const int cs = s > w0 ? s - w0 : 0;
            wt_lds_or64(&Si[cs >> 6], 1ull << (cs & 63));
            if (s >= w0) wt_lds_or64(&c.U[cs >> 6], 1ull << (cs & 63));
            if (f < w1) {
                const int cf = f - w0;
                wt_lds_or64(&c.U[cf >> 6], 1ull << (cf & 63));
            } else {
                wt_lds_min32(&c.sh->next_bp, f);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Phase 3: per-track exclusive popcount prefix over the words of S_i
// ---------------------------------------------------------------------------
WT_DEV void wt_phase_count(const WtParams &P, WtCtx &c, int tid, int nt) {
    for (int i = tid; i < P.n_tracks; i += nt) {
        const uint64_t *Si = c.S + (size_t) i * P.spitch;
        uint16_t *ci = c.cnt + (size_t) i * P.cpitch;
        unsigned run = 0;
        for (int w = 0; w < P.n_words; w++) {
            ci[w] = (uint16_t) run;
            run += (unsigned) wt_popc64(Si[w]);
        }
    }
}

// Does an interval of track i cover window position p?  If so return its value.
template <class ValT>
WT_DEV bool wt_fetch(const WtParams &P, const WtCtx &c, int i, int word, uint64_t mask, int32_t p_abs, double &v) {
    const uint64_t sw = c.S[(size_t) i * P.spitch + word];
    const unsigned r = (unsigned) c.cnt[(size_t) i * P.cpitch + word] + (unsigned) wt_popc64(sw & mask);
    if (r == 0) return false;
    const long long g = c.gbase[i] + r;
    if (p_abs >= P.finish[g]) return false;
    v = (double) ((const ValT *) P.value)[g];
    return true;
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

// ---------------------------------------------------------------------------
// Per-run reducers.  One lane evaluates one run start p (window-relative).
// Tracks are visited in index order in f64 exactly like the reference loops.
// Returns the reducer value; n0/n1 receive the in-play counts of set 0 / set 1
// (one-sample ops: everything is "set 0").
// `col`/`colstride`: this lane's scratch column in LDS (median, MWU only).
// ---------------------------------------------------------------------------
template <int OP, class ValT, class ScrT>
WT_DEV double wt_eval_position(const WtParams &P, const WtCtx &c, int p, int &n0, int &n1,
                               char *scratch, int lane_col, int colstride) {
    const int N = P.n_tracks;
    const int word = p >> 6;
    const uint64_t mask = wt_mask_incl(p & 63);
    const int32_t p_abs = c.sh->w0 + p;
    const double *dflt = P.defaults;
    n0 = 0; n1 = 0;

    if (OP == WT_OP_SUM || OP == WT_OP_PRODUCT || OP == WT_OP_MEAN) {
        // reducers.c:259-292, 313-346, 367-402
        double acc = (OP == WT_OP_PRODUCT) ? 1.0 : 0.0;
        bool nan = false;
        for (int i = 0; i < N; i++) {
            double x;
            const bool cov = wt_fetch<ValT>(P, c, i, word, mask, p_abs, x);
            if (!cov) x = dflt[i];
            n0 += cov;
            if (wt_isnan(x)) nan = true;
            if (OP == WT_OP_PRODUCT) acc *= x; else acc += x;
        }
        if (nan) return wt_nan();
        if (OP == WT_OP_MEAN) acc /= N;
        return acc;
    }
    if (OP == WT_OP_MIN || OP == WT_OP_MAX) {
        // reducers.c:125-168, 192-235: seed is 0 (not the default) when track 0 is absent
        double best;
        bool nan = false;
        {
            double x;
            const bool cov = wt_fetch<ValT>(P, c, 0, word, mask, p_abs, x);
            best = cov ? x : 0.0;
            n0 += cov;
            if (wt_isnan(best)) nan = true;
        }
        for (int i = 1; i < N; i++) {
            double x;
            const bool cov = wt_fetch<ValT>(P, c, i, word, mask, p_abs, x);
            if (!cov) x = dflt[i];
            n0 += cov;
            if (wt_isnan(x)) nan = true;
            if (OP == WT_OP_MAX) { if (best < x) best = x; } else { if (best > x) best = x; }
        }
        return nan ? wt_nan() : best;
    }
    if (OP == WT_OP_VAR || OP == WT_OP_STDDEV || OP == WT_OP_ENTROPY || OP == WT_OP_CV) {
        // reducers.c:428-479 (var), 511-563 (stddev; entropy installs the same pop, :665),
        // 672-725 (CV).  Pass 1 rounds every value through `float`.
        double mean = 0;
        bool nan = false;
        for (int i = 0; i < N; i++) {
            double x;
            const bool cov = wt_fetch<ValT>(P, c, i, word, mask, p_abs, x);
            if (!cov) x = dflt[i];
            n0 += cov;
            const float fx = (float) x;
            if (wt_isnanf(fx)) nan = true;
            mean += (double) fx;
        }
        if (nan || wt_isnan(mean)) return wt_nan();
        if (OP == WT_OP_VAR && N < 2) return wt_nan();
        mean /= N;
        if (OP == WT_OP_CV && mean == 0) return wt_nan();
        double acc = 0;
        for (int i = 0; i < N; i++) {
            double x;
            const bool cov = wt_fetch<ValT>(P, c, i, word, mask, p_abs, x);
            if (OP == WT_OP_VAR) {
                if (!cov) continue;          // :470-475 ignores absent tracks
            } else if (!cov) x = dflt[i];
            const double diff = mean - x;
            acc += diff * diff;
        }
        acc /= N;
        if (OP == WT_OP_VAR) return acc;
        acc = sqrt(acc);
        if (OP == WT_OP_CV) acc /= mean;
        return acc;
    }
    if (OP == WT_OP_TTEST) {
        // setComparisons.c:60-117: sums over in-play tracks, counts over all tracks
        const int na = P.n_set0, nb = N - P.n_set0;
        double s1 = 0, q1 = 0, s2 = 0, q2 = 0;
        for (int i = 0; i < na; i++) {
            double x;
            if (wt_fetch<ValT>(P, c, i, word, mask, p_abs, x)) { n0++; s1 += x; q1 += x * x; }
        }
        for (int i = na; i < N; i++) {
            double x;
            if (wt_fetch<ValT>(P, c, i, word, mask, p_abs, x)) { n1++; s2 += x; q2 += x * x; }
        }
        const double m1 = s1 / na, m2 = s2 / nb;
        const double msq1 = q1 / na, msq2 = q2 / nb;
        const double var1 = msq1 - m1 * m1, var2 = msq2 - m2 * m2;
        if (var1 + var2 == 0) return wt_nan();
        double t = (m1 - m2) / sqrt(var1 / na + var2 / nb);
        if (t < 0) t = -t;
        const double den = var1 / na + var2 / nb;
        const double c1 = (double) ((long long) na * na * (na - 1));
        const double c2 = (double) ((long long) nb * nb * (nb - 1));
        const double nu = den * den / ((var1 * var1) / c1 + (var2 * var2) / c2);
        return 2 * wt_tdist_Q(t, nu);
    }
    if (OP == WT_OP_MEDIAN) {
        // reducers.c:780-813: upper median of the default-substituted values.
        // Selection by bitwise binary search over order-preserving keys kept in
        // this lane's LDS column (no divergence, no writes after the gather).
        typedef typename std::conditional<sizeof(ScrT) == 4, uint32_t, uint64_t>::type KeyT;
        KeyT *col = (KeyT *) scratch + lane_col;
        bool nan = false;
        for (int i = 0; i < N; i++) {
            double x;
            const bool cov = wt_fetch<ValT>(P, c, i, word, mask, p_abs, x);
            if (!cov) x = dflt[i];
            n0 += cov;
            if (wt_isnan(x)) nan = true;
            if (sizeof(ScrT) == 4) col[(size_t) i * colstride] = (KeyT) wt_key32((float) x);
            else col[(size_t) i * colstride] = (KeyT) wt_key64(x);
        }
        if (nan) return wt_nan();
        const int kth = N / 2;       // 0-based rank of vals[N/2]
        // largest key K such that count(keys < K) <= kth  ==  the kth smallest key
        KeyT K = 0;
        for (int b = (int) sizeof(KeyT) * 8 - 1; b >= 0; b--) {
            const KeyT trial = K | ((KeyT) 1 << b);
            int below = 0;
            for (int i = 0; i < N; i++) below += (col[(size_t) i * colstride] < trial);
            if (below <= kth) K = trial;
        }
        return (sizeof(ScrT) == 4) ? (double) wt_unkey32((uint32_t) K) : wt_unkey64((uint64_t) K);
    }
    if (OP == WT_OP_MWU) {
        // setComparisons.c:293-366.  The lane's LDS column holds (value,set)
        // pairs, insertion-sorted stably by value (set-0 entries are inserted
        // first, so inside a tie group they precede set-1 entries exactly as
        // after glibc's stable merge-sort qsort).  Then the reference's scan.
        const int na = P.n_set0, nb = N - P.n_set0;
        ScrT *val = (ScrT *) scratch + lane_col;                   // [N][colstride]
        uint8_t *set = (uint8_t *) ((ScrT *) scratch + (size_t) N * colstride) + lane_col;
        bool nan = false;
        for (int i = 0; i < N; i++) {
            double x;
            const bool cov = wt_fetch<ValT>(P, c, i, word, mask, p_abs, x);
            if (!cov) x = dflt[i];
            if (i < na) n0 += cov; else n1 += cov;
            if (wt_isnan(x)) nan = true;
            // stable insertion: shift strictly greater elements up
            int j = i;
            while (j > 0 && (double) val[(size_t) (j - 1) * colstride] > x) {
                val[(size_t) j * colstride] = val[(size_t) (j - 1) * colstride];
                set[(size_t) j * colstride] = set[(size_t) (j - 1) * colstride];
                j--;
            }
            val[(size_t) j * colstride] = (ScrT) x;
            set[(size_t) j * colstride] = (uint8_t) (i >= na);
        }
        if (nan) return wt_nan();
        const double mu = (double) (na * nb / 2);                               // :386 int division
        const double sigma = sqrt((double) (na * nb * (na + nb + 1) / 12));     // :387 int division
        double U1 = 0;
        int prev = 0, ties = 0, prevTies = 0;
        for (int idx = 0; idx < N && prev < na; idx++) {
            if (!set[(size_t) idx * colstride]) {
                const ScrT x = val[(size_t) idx * colstride];
                U1 += idx - prev;
                if (ties) {
                    for (int j = idx + 1; j < N && val[(size_t) j * colstride] == x && set[(size_t) j * colstride]; j++)
                        prevTies++;
                    U1 -= prevTies / 2.0;
                    U1 += (ties - prevTies) / 2.0;
                    if (prevTies == ties) prevTies = ties = 0;
                } else {
                    for (int j = idx + 1; j < N && val[(size_t) j * colstride] == x; j++)
                        if (set[(size_t) j * colstride]) ties++;
                    if (ties) U1 += ties / 2.0;
                }
                prev++;
            }
        }
        if (U1 > mu) return 2 * erf((mu - U1) / sigma);
        return 2 * erf((U1 - mu) / sigma);
    }
    if (OP == WT_OP_MULTIPLEX) {
        for (int i = 0; i < N; i++) {
            double x;
            n0 += wt_fetch<ValT>(P, c, i, word, mask, p_abs, x);
        }
        return 0.0;
    }
    return wt_nan();
}

// First true breakpoint after window position p (absolute coordinate).
WT_DEV int32_t wt_next_breakpoint(const WtParams &P, const WtCtx &c, int p) {
    int w = p >> 6;
    uint64_t bits = c.U[w] & ~wt_mask_incl(p & 63);
    while (!bits) {
        if (++w >= P.n_words) return c.sh->next_bp;
        bits = c.U[w];
    }
    return c.sh->w0 + w * 64 + wt_ctz64(bits);
}

// ---------------------------------------------------------------------------
// Phase 4: evaluate every breakpoint owned by the window
// ---------------------------------------------------------------------------
template <int OP, class ValT, class ScrT>
WT_DEV void wt_phase_eval(const WtParams &P, WtCtx &c, WtLane &L, int tid, int nt) {
    const bool two = (OP == WT_OP_TTEST || OP == WT_OP_MWU);
    const int N = P.n_tracks;
    const int na = two ? P.n_set0 : N, nb = N - na;
    L.emit = 0;
    unsigned long long bp = 0;
    int n_emit = 0;
#pragma unroll
    for (int it = 0; it < WT_MAX_ITERS; it++) {
        const int p = it * nt + tid;
        if (p >= P.W) continue;
        if (!((c.U[p >> 6] >> (p & 63)) & 1ull)) continue;
        int n0, n1;
        const double r = wt_eval_position<OP, ValT, ScrT>(P, c, p, n0, n1, c.scratch, tid, nt);
        bool emit;
        if (two) {
            // setComparisons.c:48-54 / 282-288: both Multiplexers in play
            const bool a = (P.flags & WT_STRICT_SET0) ? (n0 == na) : (n0 > 0);
            const bool b = (P.flags & WT_STRICT_SET1) ? (n1 == nb) : (n1 > 0);
            emit = a && b;
        } else {
            // multiplexer.c:120,125
            emit = (P.flags & WT_STRICT_SET0) ? (n0 == N) : (n0 > 0);
        }
        if (!emit) continue;
        const int32_t fin = wt_next_breakpoint(P, c, p);
        L.res[it] = r;
        L.fin[it] = fin;
        L.emit |= 1u << it;
        wt_lds_or64(&c.E[p >> 6], 1ull << (p & 63));
        bp += (unsigned long long) (fin - (c.sh->w0 + p));
        n_emit++;
    }
    if (n_emit) wt_lds_add64(&c.sh->bp_sum, bp);
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
// Phase 6 (one lane): decoupled look-back for the global run offset.
// status[k] = AGG|count once window k knows its own count,
//             PFX|inclusive_prefix once it also knows everything before it.
// Windows are handed out in order by the ticket, so every predecessor has
// started; the spin is bounded and reports WT_ERR_LOOKBACK instead of hanging.
// ---------------------------------------------------------------------------
WT_DEV void wt_phase_lookback(const WtParams &P, WtCtx &c, long long k) {
    WtShared *sh = c.sh;
    const unsigned long long mine = c.epfx[P.n_words];
    sh->n_emit = (int32_t) mine;
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
    wt_status_store(&P.status[k], WT_FLAG_PFX | (excl + mine));
    sh->goffset = (long long) excl;
    // chromosome run offsets + totals
    const int ch = sh->chrom;
    if (k == P.c_first_win[ch]) P.chrom_run_off[ch] = (long long) excl;
    if (k == P.n_windows - 1) {
        P.chrom_run_off[P.n_chrom] = (long long) (excl + mine);
        P.counters[WT_CTR_RUNS] = excl + mine;
    }
    if (sh->bp_sum) wt_glb_add64(&P.counters[WT_CTR_BP], sh->bp_sum);
    if (sh->n_intervals) wt_glb_add64(&P.counters[WT_CTR_INTERVALS], sh->n_intervals);
    if ((long long) (excl + mine) > P.capacity) wt_glb_or64(&P.counters[WT_CTR_ERROR], WT_ERR_CAPACITY);
}

// ---------------------------------------------------------------------------
// Phase 7: write the emitted runs at their global positions (coalesced: lanes
// hold consecutive positions)
// ---------------------------------------------------------------------------
template <int OP, class ValT>
WT_DEV void wt_phase_write(const WtParams &P, WtCtx &c, const WtLane &L, int tid, int nt) {
    const long long goff = c.sh->goffset;
    const int32_t w0 = c.sh->w0;
#pragma unroll
    for (int it = 0; it < WT_MAX_ITERS; it++) {
        if (!((L.emit >> it) & 1u)) continue;
        const int p = it * nt + tid;
        const int w = p >> 6, b = p & 63;
        const uint64_t below = b ? wt_mask_incl(b - 1) : 0ull;
        const long long idx = goff + c.epfx[w] + wt_popc64(c.E[w] & below);
        if (idx >= P.capacity) continue;
        P.o_start[idx] = w0 + p;
        P.o_finish[idx] = L.fin[it];
        if (OP == WT_OP_MULTIPLEX) {
            const int N = P.n_tracks;
            const uint64_t mask = wt_mask_incl(b);
            for (int i = 0; i < N; i++) {
                double x;
                const bool cov = wt_fetch<ValT>(P, c, i, w, mask, w0 + p, x);
                P.o_tile[idx * N + i] = cov ? x : P.defaults[i];
                P.o_inplay[idx * N + i] = (uint8_t) cov;
            }
        } else {
            P.o_value[idx] = L.res[it];
        }
    }
}

// ---------------------------------------------------------------------------
// Window index: one lane per input interval; interval jr of (chrom, track)
// claims every window boundary b with finish[jr-1] < b <= finish[jr]:
//   widx[row(b)][track] = jr  == first interval with finish >= b.
// Rows past the last interval get n (= none).  Empty (chrom,track) segments
// rely on the rows having been zeroed (0 == n).
// ---------------------------------------------------------------------------
WT_DEV void wt_index_interval(const WtParams &P, long long g) {
    const int N = P.n_tracks;
    // segment of g: last seg with seg_off[seg] <= g
    long long lo = 0, hi = (long long) P.n_chrom * N;     // invariant: seg_off[lo] <= g < seg_off[hi]
    while (hi - lo > 1) {
        const long long mid = (lo + hi) >> 1;
        if (P.seg_off[mid] <= g) lo = mid; else hi = mid;
    }
    const long long seg = lo;
    const int ch = (int) (seg / N), i = (int) (seg % N);
    const long long jr = g - P.seg_off[seg];
    const long long n = P.seg_off[seg + 1] - P.seg_off[seg];
    const long long cb = P.cbase[ch];
    const long long nw = P.c_nwin[ch];
    const long long rowbase = P.c_first_win[ch] + ch;
    const long long f = P.finish[g];
    long long m_lo = 0;
    if (jr > 0) m_lo = (P.finish[g - 1] - cb) / P.W + 1;     // finish[g-1] > cbase always
    long long m_hi = (f - cb) / P.W;
    if (m_hi > nw) m_hi = nw;
    for (long long m = m_lo; m <= m_hi; m++) P.widx[(size_t) (rowbase + m) * N + i] = (uint32_t) jr;
    if (jr == n - 1)
        for (long long m = m_hi + 1; m <= nw; m++) P.widx[(size_t) (rowbase + m) * N + i] = (uint32_t) n;
}

#endif  // WT_CORE_H_
