// wt_synth.hip -- synthetic workload generator of SURVEY 8d, on device (BENCH / TEST PLUMBING, not
// part of the hot path): N tracks of sorted, non-overlapping runs over a list of chromosomes,
// run length ~ 1 + Geometric(1/l), value = k/8 with k ~ U{0..levels-1} (exact in f32 and f64, ties
// on purpose), a run is dropped (gap) with probability gap_prob.
//
// Counter-based, as the survey asks ("so host and device can regenerate any tile without storing
// 100 x 3 Gb"): position x of (chromosome c, track t) is a breakpoint iff a hash of
// (seed, c, t, x) falls below 2^32 / l -- a Bernoulli trial per bp, hence geometric run lengths --
// and the run starting at a breakpoint takes its value and its gap flag from a second hash of the
// same key.  Any (c, t, range) can therefore be regenerated independently, by any GPU, in any
// order: a whole 3.1 Gbp x 100-track genome never has to be resident, bench.py generates one work
// item at a time right before it is processed.
//
// Two passes over the positions, 4096 per block (16 consecutive positions per lane): count the
// kept runs per block; [exclusive scan by the caller]; write every run at its rank.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/wiggletools_amd.h"
#include "wt_devscope.h"

int wt_fail_ext(int code, const std::string &msg);     // wt_engine.hip

#define WS_BLOCK 256
#define WS_PER_LANE 16
#define WS_TILE (WS_BLOCK * WS_PER_LANE)
#define WS_MAX_CHROM 64

namespace {

struct WsArgs {
    unsigned long long seed;
    int32_t n_chrom, n_tracks;
    uint32_t bp_thresh;         // breakpoint iff low 32 hash bits < bp_thresh (0: every position, l = 1)
    uint32_t gap_thresh;        // run dropped iff 32 bits of the second hash < gap_thresh
    uint32_t levels;
    int32_t chrom_len[WS_MAX_CHROM];
    long long chrom_block_off[WS_MAX_CHROM + 1];    // first block of chromosome c; its blocks are [track][tile]
};

__device__ __forceinline__ unsigned long long ws_mix(unsigned long long z) {     // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct WsKey {
    unsigned long long base;
    uint32_t bp_thresh;
    __device__ __forceinline__ unsigned long long h(int32_t x) const { return ws_mix(base + 0x9E3779B97F4A7C15ull * (unsigned long long) (x + 1)); }
    __device__ __forceinline__ bool is_bp(int32_t x) const { return x == 0 || bp_thresh == 0 || (uint32_t) h(x) < bp_thresh; }
};

__device__ __forceinline__ void ws_locate(const WsArgs &A, long long block, int &c, int &t, int32_t &x0) {
    c = 0;
    while (c + 1 < A.n_chrom && A.chrom_block_off[c + 1] <= block) c++;
    const long long tiles = (A.chrom_len[c] + WS_TILE - 1) / WS_TILE;
    const long long r = block - A.chrom_block_off[c];
    t = (int) (r / tiles);
    x0 = (int32_t) ((r % tiles) * WS_TILE);
}

__device__ __forceinline__ WsKey ws_key(const WsArgs &A, int c, int t) {
    WsKey k;
    k.base = ws_mix(A.seed ^ ((unsigned long long) c << 40) ^ ((unsigned long long) t << 8) ^ 0x5bd1e995ull);
    k.bp_thresh = A.bp_thresh;
    return k;
}

// kept-run test + value of the run starting at breakpoint x
__device__ __forceinline__ bool ws_run(const WsArgs &A, const WsKey &k, int32_t x, float &v) {
    const unsigned long long g = ws_mix(k.h(x) ^ 0xD6E8FEB86659FD93ull);
    v = (float) ((uint32_t) (g >> 40) % A.levels) * 0.125f;
    return (uint32_t) g >= A.gap_thresh;
}

// (grid-stride over the blocks: HIP caps a launch at 2^32 threads -- 500 tracks x chromosome 1 are 30 M blocks)
__global__ void __launch_bounds__(WS_BLOCK) ws_count_kernel(const WsArgs A, long long *counts, long long n_blocks) {
  for (long long blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    int c, t;
    int32_t x0;
    ws_locate(A, blk, c, t, x0);
    const WsKey k = ws_key(A, c, t);
    const int32_t len = A.chrom_len[c];
    int n = 0;
    const int32_t xa = x0 + (int32_t) threadIdx.x * WS_PER_LANE;
#pragma unroll 4
    for (int q = 0; q < WS_PER_LANE; q++) {
        const int32_t x = xa + q;
        float v;
        if (x < len && k.is_bp(x) && ws_run(A, k, x, v)) n++;
    }
    __shared__ int wsum[WS_BLOCK / 64];
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < WS_BLOCK / 64; w++) tot += wsum[w];
        counts[blk] = tot;
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(WS_BLOCK) ws_fill_kernel(const WsArgs A, const long long *block_off, int32_t *start,
                                                            int32_t *finish, float *value, long long n_blocks) {
  for (long long blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    int c, t;
    int32_t x0;
    ws_locate(A, blk, c, t, x0);
    const WsKey k = ws_key(A, c, t);
    const int32_t len = A.chrom_len[c];
    const int32_t xa = x0 + (int32_t) threadIdx.x * WS_PER_LANE;
    // pass 1: the lane's kept runs
    int n = 0;
    uint32_t bpmask = 0, keepmask = 0;
#pragma unroll 4
    for (int q = 0; q < WS_PER_LANE; q++) {
        const int32_t x = xa + q;
        float v;
        if (x < len && k.is_bp(x)) {
            bpmask |= 1u << q;
            if (ws_run(A, k, x, v)) { keepmask |= 1u << q; n++; }
        }
    }
    // exclusive rank of the lane inside the block
    __shared__ int wsum[WS_BLOCK / 64];
    int incl = n;
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int before = incl - n;
    for (int w = 0; w < (int) (threadIdx.x >> 6); w++) before += wsum[w];
    long long g = block_off[blk] + before;
    // pass 2: a run ends at the next breakpoint (inside the lane's 16 positions, or scanned for)
    while (keepmask) {
        const int q = __ffs(keepmask) - 1;
        keepmask &= keepmask - 1;
        const int32_t x = xa + q;
        const uint32_t later = bpmask & ~((2u << q) - 1u);
        int32_t f;
        if (later) {
            f = xa + (__ffs(later) - 1);
        } else {
            f = xa + WS_PER_LANE;
            while (f < len && !k.is_bp(f)) f++;
            if (f > len) f = len;
        }
        float v;
        (void) ws_run(A, k, x, v);
        start[g] = x + 1;           // 1-based inclusive (wiggleIterator.h:23)
        finish[g] = f + 1;          // exclusive
        value[g] = v;
        g++;
    }
    __syncthreads();
  }
}

int ws_args(unsigned long long seed, int n_chrom, const int32_t *chrom_len, int n_tracks, double mean_run, double gap_prob,
            int levels, WsArgs &A, long long &n_blocks) {
    if (n_chrom <= 0 || n_chrom > WS_MAX_CHROM || !chrom_len || n_tracks <= 0 || mean_run < 1.0 || levels <= 0 || gap_prob < 0 || gap_prob >= 1)
        return wt_fail_ext(WTAMD_ERR_ARG, "wtamd_synth: bad arguments");
    A.seed = seed; A.n_chrom = n_chrom; A.n_tracks = n_tracks;
    A.bp_thresh = mean_run <= 1.0 ? 0u : (uint32_t) (4294967296.0 / mean_run);
    A.gap_thresh = (uint32_t) (4294967296.0 * gap_prob);
    A.levels = (uint32_t) levels;
    long long off = 0;
    for (int c = 0; c < n_chrom; c++) {
        if (chrom_len[c] <= 0) return wt_fail_ext(WTAMD_ERR_ARG, "wtamd_synth: chromosome length must be positive");
        A.chrom_len[c] = chrom_len[c];
        A.chrom_block_off[c] = off;
        off += (long long) ((chrom_len[c] + WS_TILE - 1) / WS_TILE) * n_tracks;
    }
    A.chrom_block_off[n_chrom] = off;
    n_blocks = off;
    return WTAMD_OK;
}

}  // namespace

extern "C" {

// Number of count blocks (entries of `counts` / `block_off`), and for every (chrom, track) segment
// the index of its first block (seg_first_block: n_chrom * n_tracks + 1 entries, host).
int wtamd_synth_plan(int n_chrom, const int32_t *chrom_len, int n_tracks, int64_t *n_blocks, int64_t *seg_first_block) {
    WsArgs A;
    long long nb = 0;
    const int rc = ws_args(0, n_chrom, chrom_len, n_tracks, 16.0, 0.0, 1, A, nb);
    if (rc != WTAMD_OK) return rc;
    if (n_blocks) *n_blocks = nb;
    if (seg_first_block) {
        for (int c = 0; c < n_chrom; c++) {
            const long long tiles = (chrom_len[c] + WS_TILE - 1) / WS_TILE;
            for (int t = 0; t < n_tracks; t++) seg_first_block[(int64_t) c * n_tracks + t] = A.chrom_block_off[c] + tiles * t;
        }
        seg_first_block[(int64_t) n_chrom * n_tracks] = nb;
    }
    return WTAMD_OK;
}

// counts: DEVICE, n_blocks int64.
int wtamd_synth_count(uint64_t seed, int n_chrom, const int32_t *chrom_len, int n_tracks, double mean_run, double gap_prob,
                      int levels, int64_t *counts, void *stream) {
    WsArgs A;
    long long nb = 0;
    const int rc = ws_args(seed, n_chrom, chrom_len, n_tracks, mean_run, gap_prob, levels, A, nb);
    if (rc != WTAMD_OK) return rc;
    hipLaunchKernelGGL(ws_count_kernel, dim3((unsigned) (nb < (1ll << 22) ? nb : (1ll << 22))), dim3(WS_BLOCK), 0, (hipStream_t) stream, A,
                       (long long *) counts, nb);
    if (hipGetLastError() != hipSuccess) return wt_fail_ext(WTAMD_ERR_HIP, "wtamd_synth_count: launch failed");
    return WTAMD_OK;
}

// block_off: DEVICE, exclusive prefix sum of `counts`; start / finish / value: DEVICE, sum(counts) entries.
int wtamd_synth_fill(uint64_t seed, int n_chrom, const int32_t *chrom_len, int n_tracks, double mean_run, double gap_prob,
                     int levels, const int64_t *block_off, int32_t *start, int32_t *finish, float *value, void *stream) {
    WsArgs A;
    long long nb = 0;
    const int rc = ws_args(seed, n_chrom, chrom_len, n_tracks, mean_run, gap_prob, levels, A, nb);
    if (rc != WTAMD_OK) return rc;
    hipLaunchKernelGGL(ws_fill_kernel, dim3((unsigned) (nb < (1ll << 22) ? nb : (1ll << 22))), dim3(WS_BLOCK), 0, (hipStream_t) stream, A,
                       (const long long *) block_off, start, finish, value, nb);
    if (hipGetLastError() != hipSuccess) return wt_fail_ext(WTAMD_ERR_HIP, "wtamd_synth_fill: launch failed");
    return WTAMD_OK;
}

}  // extern "C"
