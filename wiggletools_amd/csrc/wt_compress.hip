// wt_compress.hip -- device-side run compression: the reference's
// CompressionWiggleIterator (reference src/unaryOps.c:235-253), which the default writer puts in
// front of every non-bedGraph output (src/wigWriter.c:263-267), applied to a run list that is
// still in HBM so that only the merged runs cross PCIe.
//
// Reference rule (sequential): a run joins the current group iff same chromosome, its start ==
// the previous run's finish, and (both NaN or |value - value of the group's FIRST run| < 1e-6).
// The comparison is against the group LEADER, so the rule is not a function of neighbours only.
// Parallel formulation, exact for every input:
//   1. classify each run against its predecessor (one lane per run, bitmaps by wave ballot):
//        SURE LEADER   not contiguous / other chromosome / NaN-ness differs / |dv| >= 2e-6+eps
//                      (then |v - leader| >= 1e-6 whatever the leader is, because the
//                       predecessor is within 1e-6 of its leader)
//        SURE MEMBER   contiguous and value identical to the predecessor's (it shares the
//                      predecessor's fate, and the predecessor is in the current group)
//        UNCERTAIN     contiguous, 0 < |dv| < 2e-6+eps: depends on the leader's value
//   2. resolve: one lane per sure leader walks forward over the bitmaps (64 runs per step) and
//      applies the reference rule to the uncertain runs it meets, promoting some to leaders.
//      Uncertain runs are rare in real tracks, so the walk is almost always a bitmap skim.
//   3. rank leaders (two-level popcount scan) and emit: start/value of the leader, finish of
//      the last run before the next leader.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/wiggletools_amd.h"

namespace {

#define WC_BLOCK 256
#define WC_WORDS_PER_BLOCK 2048      // bitmap words one block ranks / emits

__device__ __forceinline__ bool wc_isnan(double x) { return x != x; }

// 1. classification -> LEAD / UNC bitmaps (bit r of word r/64)
// (n_ptr != NULL: the run count is read on the device -- the streaming pipeline compresses a batch
//  before its host has learnt how many runs the reduce kernel emitted; grids are then sized by capacity)
__global__ void __launch_bounds__(WC_BLOCK) wc_classify(const int32_t *start, const int32_t *finish, const double *value,
                                                        long long n, const unsigned long long *n_ptr,
                                                        unsigned long long *lead, unsigned long long *unc,
                                                        unsigned long long *first_sure) {
    if (n_ptr) n = (long long) *n_ptr < n ? (long long) *n_ptr : n;
    const long long r = (long long) blockIdx.x * WC_BLOCK + threadIdx.x;
    bool is_lead = false, is_unc = false;
    if (r < n) {
        if (r == 0) {
            is_lead = true;
        } else {
            const double v = value[r], pv = value[r - 1];
            const bool contiguous = start[r] == finish[r - 1];
            const bool vn = wc_isnan(v), pn = wc_isnan(pv);
            if (!contiguous || vn != pn) is_lead = true;
            else if (vn) is_lead = false;                    // both NaN: merges (unaryOps.c:248)
            else {
                const double d = fabs(v - pv);
                if (d >= 2.000001e-6) is_lead = true;
                else if (d != 0.0) is_unc = true;
            }
        }
    }
    const unsigned long long lb = __ballot(is_lead), ub = __ballot(is_unc);
    if ((threadIdx.x & 63) == 0 && r < n) { lead[r >> 6] = lb; unc[r >> 6] = ub; }
    if (first_sure) {       // first run (other than run 0) that leads a group whatever came before it
        const unsigned long long sure = r == 0 ? (lb & ~1ull) : lb;
        if ((threadIdx.x & 63) == 0 && r < n && sure) atomicMin(first_sure, (unsigned long long) (r + (__ffsll(sure) - 1)));
    }
}

// Open start (pipeline batches): the list continues a previous batch, whose last group may extend
// into this one -- whether the runs BEFORE the first sure leader join that group depends on its
// leader's value, which this batch does not know (and a compressed group hides its members'
// values, so merging them here could change what the consumer's own CompressionWiggleIterator
// decides).  They are passed through one by one: every run before the first sure leader leads.
__global__ void __launch_bounds__(WC_BLOCK) wc_open_start(unsigned long long *lead, const unsigned long long *first_sure,
                                                          long long n, const unsigned long long *n_ptr) {
    if (n_ptr) n = (long long) *n_ptr < n ? (long long) *n_ptr : n;
    long long f = (long long) (*first_sure < (unsigned long long) n ? *first_sure : (unsigned long long) n);
    const long long w = (long long) blockIdx.x * WC_BLOCK + threadIdx.x;
    if (w * 64 >= f) return;
    const long long top = f - w * 64;
    lead[w] |= top >= 64 ? ~0ull : ((1ull << top) - 1ull);
}

// chromosome starts are leaders whatever the coordinates say (strcmp(chrom) test, unaryOps.c:248)
__global__ void wc_chrom_starts(const int64_t *chrom_run_off, int n_chrom, long long n, unsigned long long *lead,
                                unsigned long long *unc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chrom) return;
    const long long r = chrom_run_off[c];
    if (r < n && r < chrom_run_off[c + 1]) {
        atomicOr(&lead[r >> 6], 1ull << (r & 63));
        atomicAnd(&unc[r >> 6], ~(1ull << (r & 63)));
    }
}

// next set bit of (a|b) at or after position p (p < n), or n
__device__ long long wc_next(const unsigned long long *a, const unsigned long long *b, long long p, long long n) {
    long long w = p >> 6;
    const long long nw = (n + 63) >> 6;
    unsigned long long bits = (a[w] | (b ? b[w] : 0ull)) & (~0ull << (p & 63));
    while (!bits) {
        if (++w >= nw) return n;
        bits = a[w] | (b ? b[w] : 0ull);
    }
    const long long q = (w << 6) + (__ffsll(bits) - 1);
    return q < n ? q : n;
}

// 2. resolve uncertain runs: one lane per sure leader
__global__ void __launch_bounds__(WC_BLOCK) wc_resolve(const double *value, long long n, const unsigned long long *n_ptr,
                                                       const unsigned long long *lead,
                                                       const unsigned long long *unc, unsigned long long *promoted) {
    if (n_ptr) n = (long long) *n_ptr < n ? (long long) *n_ptr : n;
    const long long r = (long long) blockIdx.x * WC_BLOCK + threadIdx.x;
    if (r >= n || !((lead[r >> 6] >> (r & 63)) & 1ull)) return;
    double leader = value[r];
    long long p = r + 1;
    while (p < n) {
        const long long q = wc_next(lead, unc, p, n);
        if (q >= n || ((lead[q >> 6] >> (q & 63)) & 1ull)) break;      // next sure leader: its own lane takes over
        // q is uncertain (contiguous, both values non-NaN): the reference test against the leader
        const double v = value[q];
        if (!(fabs(v - leader) < 0.000001)) {
            atomicOr(&promoted[q >> 6], 1ull << (q & 63));
            leader = v;
        }
        p = q + 1;
    }
}

// 3a. final leader bitmap + per-block leader counts
__global__ void __launch_bounds__(WC_BLOCK) wc_count(unsigned long long *lead, const unsigned long long *promoted,
                                                     long long n_words, const unsigned long long *n_ptr,
                                                     unsigned long long *block_count) {
    if (n_ptr) { const long long nw = ((long long) *n_ptr + 63) >> 6; n_words = nw < n_words ? nw : n_words; }
    __shared__ unsigned int red[WC_BLOCK];
    const long long w0 = (long long) blockIdx.x * WC_WORDS_PER_BLOCK;
    unsigned int c = 0;
    for (int k = threadIdx.x; k < WC_WORDS_PER_BLOCK; k += WC_BLOCK) {
        const long long w = w0 + k;
        if (w < n_words) {
            const unsigned long long f = lead[w] | promoted[w];
            lead[w] = f;
            c += (unsigned) __popcll(f);
        }
    }
    red[threadIdx.x] = c;
    __syncthreads();
    for (int s = WC_BLOCK / 2; s > 0; s >>= 1) {
        if ((int) threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_count[blockIdx.x] = red[0];
}

// 3b. exclusive scan of the block counts (one block; the list is short: n / 131072 entries)
__global__ void wc_scan_blocks(unsigned long long *block_count, long long n_blocks, unsigned long long *total) {
    // (blocks beyond a device-side run count hold 0: wc_count writes every block)
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long run = 0;
        for (long long b = 0; b < n_blocks; b++) { const unsigned long long c = block_count[b]; block_count[b] = run; run += c; }
        *total = run;
    }
}

// 3c. emit merged runs; also the per-chromosome offsets of the compressed list
__global__ void __launch_bounds__(WC_BLOCK) wc_emit(const int32_t *start, const int32_t *finish, const double *value, long long n,
                                                    const unsigned long long *n_ptr,
                                                    const unsigned long long *lead, long long n_words,
                                                    const unsigned long long *block_off, long long capacity,
                                                    int32_t *o_start, int32_t *o_finish, double *o_value) {
    if (n_ptr) {
        n = (long long) *n_ptr < n ? (long long) *n_ptr : n;
        const long long nw = (n + 63) >> 6;
        n_words = nw < n_words ? nw : n_words;
    }
    __shared__ unsigned int pfx[WC_WORDS_PER_BLOCK];
    const long long w0 = (long long) blockIdx.x * WC_WORDS_PER_BLOCK;
    for (int k = threadIdx.x; k < WC_WORDS_PER_BLOCK; k += WC_BLOCK) {
        const long long w = w0 + k;
        pfx[k] = (w < n_words) ? (unsigned) __popcll(lead[w]) : 0u;
    }
    __syncthreads();
    if (threadIdx.x == 0) {            // 2048-entry exclusive scan; tiny next to the emission below
        unsigned int run = 0;
        for (int k = 0; k < WC_WORDS_PER_BLOCK; k++) { const unsigned int c = pfx[k]; pfx[k] = run; run += c; }
    }
    __syncthreads();
    const unsigned long long base = block_off[blockIdx.x];
    // one lane per run of the block's words
    for (long long k = threadIdx.x; k < (long long) WC_WORDS_PER_BLOCK * 64; k += WC_BLOCK) {
        const long long r = (w0 << 6) + k;
        if (r >= n) break;
        const unsigned long long word = lead[r >> 6];
        if (!((word >> (r & 63)) & 1ull)) continue;
        const unsigned long long below = (r & 63) ? (word & ((1ull << (r & 63)) - 1ull)) : 0ull;
        const long long o = (long long) (base + pfx[(r >> 6) - w0] + (unsigned) __popcll(below));
        if (o >= capacity) continue;
        const long long nx = (r + 1 < n) ? wc_next(lead, nullptr, r + 1, n) : n;
        o_start[o] = start[r];
        o_finish[o] = finish[nx - 1];
        o_value[o] = value[r];
    }
}

// rank of run r among leaders = number of leaders before r (for the chromosome offsets)
__global__ void wc_chrom_offsets(const int64_t *chrom_run_off, int n_chrom, long long n, const unsigned long long *lead,
                                 const unsigned long long *block_off, unsigned long long total, int64_t *o_chrom_run_off) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_chrom) return;
    const long long r = chrom_run_off[c];
    if (r >= n) { o_chrom_run_off[c] = (int64_t) total; return; }
    const long long w = r >> 6, b = w / WC_WORDS_PER_BLOCK;
    unsigned long long cnt = block_off[b];
    for (long long x = b * WC_WORDS_PER_BLOCK; x < w; x++) cnt += (unsigned) __popcll(lead[x]);
    if (r & 63) cnt += (unsigned) __popcll(lead[w] & ((1ull << (r & 63)) - 1ull));
    o_chrom_run_off[c] = (int64_t) cnt;
}

}  // namespace

extern "C" const char *wtamd_last_error(void);

namespace {
thread_local std::string g_err;
}

#define WC_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "wiggletools_amd: %s: %s\n", #expr, hipGetErrorString(e_));   \
            return WTAMD_ERR_HIP;                                                         \
        }                                                                                 \
    } while (0)

extern "C" int wtamd_runs_compress(const wtamd_runs *in, int64_t n_runs, int32_t n_chrom, wtamd_runs *out,
                                   int64_t *n_out, void *stream) {
    if (!in || !out || !n_out || n_runs < 0 || !in->chrom_run_off) return WTAMD_ERR_ARG;
    hipStream_t s = (hipStream_t) stream;
    if (n_runs == 0) {
        if (out->chrom_run_off) WC_HIP(hipMemsetAsync(out->chrom_run_off, 0, sizeof(int64_t) * (n_chrom + 1), s));
        *n_out = 0;
        WC_HIP(hipStreamSynchronize(s));
        return WTAMD_OK;
    }
    const long long n = n_runs;
    const long long n_words = (n + 63) >> 6;
    const long long n_blocks = (n_words + WC_WORDS_PER_BLOCK - 1) / WC_WORDS_PER_BLOCK;
    unsigned long long *d_lead = nullptr, *d_unc = nullptr, *d_prom = nullptr, *d_blk = nullptr, *d_total = nullptr;
    WC_HIP(hipMalloc(&d_lead, sizeof(unsigned long long) * n_words));
    WC_HIP(hipMalloc(&d_unc, sizeof(unsigned long long) * n_words));
    WC_HIP(hipMalloc(&d_prom, sizeof(unsigned long long) * n_words));
    WC_HIP(hipMalloc(&d_blk, sizeof(unsigned long long) * (n_blocks + 1)));
    d_total = d_blk + n_blocks;
    WC_HIP(hipMemsetAsync(d_prom, 0, sizeof(unsigned long long) * n_words, s));
    const unsigned grid_runs = (unsigned) ((n + WC_BLOCK - 1) / WC_BLOCK);
    hipLaunchKernelGGL(wc_classify, dim3(grid_runs), dim3(WC_BLOCK), 0, s, in->start, in->finish, in->value, n, (const unsigned long long *) nullptr, d_lead, d_unc,
                       (unsigned long long *) nullptr);
    hipLaunchKernelGGL(wc_chrom_starts, dim3((unsigned) ((n_chrom + 63) / 64)), dim3(64), 0, s, in->chrom_run_off, (int) n_chrom, n,
                       d_lead, d_unc);
    hipLaunchKernelGGL(wc_resolve, dim3(grid_runs), dim3(WC_BLOCK), 0, s, in->value, n, (const unsigned long long *) nullptr, d_lead, d_unc, d_prom);
    hipLaunchKernelGGL(wc_count, dim3((unsigned) n_blocks), dim3(WC_BLOCK), 0, s, d_lead, d_prom, n_words, (const unsigned long long *) nullptr, d_blk);
    hipLaunchKernelGGL(wc_scan_blocks, dim3(1), dim3(64), 0, s, d_blk, n_blocks, d_total);
    hipLaunchKernelGGL(wc_emit, dim3((unsigned) n_blocks), dim3(WC_BLOCK), 0, s, in->start, in->finish, in->value, n,
                       (const unsigned long long *) nullptr, d_lead, n_words,
                       d_blk, (long long) out->capacity, out->start, out->finish, out->value);
    unsigned long long total = 0;
    WC_HIP(hipGetLastError());
    WC_HIP(hipMemcpyAsync(&total, d_total, sizeof(total), hipMemcpyDeviceToHost, s));
    WC_HIP(hipStreamSynchronize(s));
    if (out->chrom_run_off) {
        hipLaunchKernelGGL(wc_chrom_offsets, dim3((unsigned) ((n_chrom + 1 + 63) / 64)), dim3(64), 0, s, in->chrom_run_off, (int) n_chrom,
                           n, d_lead, d_blk, total, out->chrom_run_off);
        WC_HIP(hipGetLastError());
        WC_HIP(hipStreamSynchronize(s));
    }
    (void) hipFree(d_lead); (void) hipFree(d_unc); (void) hipFree(d_prom); (void) hipFree(d_blk);
    *n_out = (int64_t) total;
    return (int64_t) total > out->capacity ? WTAMD_ERR_CAPACITY : WTAMD_OK;
}


// ---- in-stream flavour for the pipeline (wt_pipe.h): one chromosome, the run count still on the
// device (*d_n, at most `capacity`), scratch provided by the caller (WC scratch words: see
// wt_compress_scratch_words), nothing waits.  *d_n_out receives the number of merged runs.
long long wt_compress_scratch_words(long long capacity) {
    const long long n_words = (capacity + 63) >> 6;
    const long long n_blocks = (n_words + WC_WORDS_PER_BLOCK - 1) / WC_WORDS_PER_BLOCK;
    return 3 * n_words + n_blocks + 4;
}

int wt_compress_async(const int32_t *start, const int32_t *finish, const double *value, const unsigned long long *d_n,
                      long long capacity, unsigned long long *scratch, int32_t *o_start, int32_t *o_finish, double *o_value,
                      unsigned long long *d_n_out, hipStream_t s) {
    if (capacity <= 0) return WTAMD_OK;
    const long long n_words = (capacity + 63) >> 6;
    const long long n_blocks = (n_words + WC_WORDS_PER_BLOCK - 1) / WC_WORDS_PER_BLOCK;
    unsigned long long *d_lead = scratch, *d_unc = scratch + n_words, *d_prom = scratch + 2 * n_words, *d_blk = scratch + 3 * n_words;
    unsigned long long *d_first = d_blk + n_blocks + 2;
    WC_HIP(hipMemsetAsync(scratch, 0, sizeof(unsigned long long) * (size_t) (3 * n_words + n_blocks + 1), s));
    WC_HIP(hipMemsetAsync(d_first, 0xff, sizeof(unsigned long long), s));
    const unsigned grid_runs = (unsigned) ((capacity + WC_BLOCK - 1) / WC_BLOCK);
    hipLaunchKernelGGL(wc_classify, dim3(grid_runs), dim3(WC_BLOCK), 0, s, start, finish, value, capacity, d_n, d_lead, d_unc, d_first);
    hipLaunchKernelGGL(wc_open_start, dim3((unsigned) ((n_words + WC_BLOCK - 1) / WC_BLOCK)), dim3(WC_BLOCK), 0, s, d_lead, d_first,
                       capacity, d_n);
    hipLaunchKernelGGL(wc_resolve, dim3(grid_runs), dim3(WC_BLOCK), 0, s, value, capacity, d_n, d_lead, d_unc, d_prom);
    hipLaunchKernelGGL(wc_count, dim3((unsigned) n_blocks), dim3(WC_BLOCK), 0, s, d_lead, d_prom, n_words, d_n, d_blk);
    hipLaunchKernelGGL(wc_scan_blocks, dim3(1), dim3(64), 0, s, d_blk, n_blocks, d_n_out);
    hipLaunchKernelGGL(wc_emit, dim3((unsigned) n_blocks), dim3(WC_BLOCK), 0, s, start, finish, value, capacity, d_n, d_lead, n_words,
                       d_blk, capacity, o_start, o_finish, o_value);
    WC_HIP(hipGetLastError());
    return WTAMD_OK;
}
