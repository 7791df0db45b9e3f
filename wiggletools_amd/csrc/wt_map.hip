// wt_map.hip -- the reference's `map`-able unary operators applied to whole run lists on device
// (SURVEY 8f row 3): scale, offset, ln, log, exp, pow, abs (reference src/unaryOps.c:650-949).
//
// The reference wraps every track in one lazy operator iterator (one indirect call per interval,
// commandParser.c:115-211).  Here the operator is an elementwise pass over the track set's value
// array before it is multiplexed: a run list has one value per input run, so mapping the runs is
// cheaper than fusing the operator into the reducer's per-(track, position) loop would be.
// ln / log DROP runs whose value is <= 0 (unaryOps.c:764-765), gt / gte / lt / lte the runs failing the
// comparison (:390-399) -- that changes which tracks are in play downstream, so those also compact the run lists (flag -> block counts -> scan ->
// scatter) and rewrite the segment offsets.
// Values become f64 (the reference's operators compute in double); default values are transformed
// on the host by wtamd_map_default, including the `float` truncation several ctors apply
// (SURVEY Q14).  Bound: HBM, 12 B read + 16 B written per run.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/wiggletools_amd.h"
#include "wt_devscope.h"
#include "wt_mapop.h"

#define WM_BLOCK 256
#define WM_ITEMS 16
#define WM_TILE (WM_BLOCK * WM_ITEMS)

extern "C" const char *wtamd_last_error(void);
int wt_fail_ext(int code, const std::string &msg);     // wt_engine.hip

namespace {

template <class ValT>
__global__ void __launch_bounds__(WM_BLOCK) wm_map_kernel(int op, double param, double lg, const ValT *in, long long n,
                                                           double *out, unsigned long long *block_keep) {
    __shared__ unsigned int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const long long base = (long long) blockIdx.x * WM_TILE;
    unsigned kept = 0;
#pragma unroll 4
    for (int q = 0; q < WM_ITEMS; q++) {
        const long long g = base + threadIdx.x + (long long) q * WM_BLOCK;
        if (g < n) {
            bool keep;
            const double r = wm_apply(op, param, lg, (double) in[g], keep);
            // a dropped run keeps a NaN payload nobody reads; the flag is re-derived from the input
            out[g] = r;
            kept += keep ? 1u : 0u;
        }
    }
    if (block_keep) {
        atomicAdd(&cnt, kept);
        __syncthreads();
        if (threadIdx.x == 0) block_keep[blockIdx.x] = cnt;
    }
}

// exclusive scan of the block counts (one wave; the tile is 4096 runs, so even 2.4e9 runs are
// only 5.8e5 entries)
__global__ void wm_scan_blocks(unsigned long long *block_keep, long long n_blocks, unsigned long long *total) {
    const int lane = threadIdx.x;
    unsigned long long carry = 0;
    for (long long b0 = 0; b0 < n_blocks; b0 += 64) {
        const long long b = b0 + lane;
        unsigned long long v = b < n_blocks ? block_keep[b] : 0ull, incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = (unsigned long long) __shfl_up((long long) incl, d);
            if (lane >= d) incl += o;
        }
        if (b < n_blocks) block_keep[b] = carry + incl - v;
        carry += (unsigned long long) __shfl((long long) incl, 63);
    }
    if (lane == 0) *total = carry;
}

template <class ValT>
__device__ inline bool wm_kept(const ValT *in, long long g, int op, double param) {
    bool keep;
    (void) wm_apply(op, param, 1.0, (double) in[g], keep);      // the flag does not depend on the log base
    return keep;
}

// stable scatter of the kept runs of one tile (order inside the tile: by global index)
template <class ValT>
__global__ void __launch_bounds__(WM_BLOCK) wm_compact_kernel(int op, double param, const ValT *in, const int32_t *start,
                                                               const int32_t *finish, const double *mapped, long long n,
                                                               const unsigned long long *block_off, int32_t *o_start,
                                                               int32_t *o_finish, double *o_value) {
    __shared__ unsigned int wave_tot[WM_BLOCK / 64];
    const long long base = (long long) blockIdx.x * WM_TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long off = block_off[blockIdx.x];
    // every thread owns WM_ITEMS CONSECUTIVE runs so that the output order is the input order
    const long long g0 = base + (long long) threadIdx.x * WM_ITEMS;
    unsigned mask = 0, mine = 0;
#pragma unroll
    for (int q = 0; q < WM_ITEMS; q++)
        if (g0 + q < n && wm_kept(in, g0 + q, op, param)) { mask |= 1u << q; mine++; }
    unsigned incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = (unsigned) __shfl_up((int) incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned before = incl - mine;
    for (int w = 0; w < wave; w++) before += wave_tot[w];
    unsigned long long o = off + before;
#pragma unroll
    for (int q = 0; q < WM_ITEMS; q++)
        if ((mask >> q) & 1u) {
            o_start[o] = start[g0 + q];
            o_finish[o] = finish[g0 + q];
            o_value[o] = mapped[g0 + q];
            o++;
        }
}

// new segment offsets: number of kept runs before every old segment boundary
template <class ValT>
__global__ void wm_seg_offsets(int op, double param, const ValT *in, const int64_t *seg_off, long long n_seg, long long n,
                               const unsigned long long *block_off, unsigned long long total, int64_t *o_seg_off) {
    const long long s = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_seg) return;
    const long long p = seg_off[s];
    if (p >= n) { o_seg_off[s] = (int64_t) total; return; }
    const long long b = p / WM_TILE;
    unsigned long long c = block_off[b];
    for (long long g = b * WM_TILE; g < p; g++) c += wm_kept(in, g, op, param) ? 1ull : 0ull;
    o_seg_off[s] = (int64_t) c;
}


// ---------------------------------------------------------------------------
// Per-track operator chains inside the streaming pipeline (wtamd_pipe_set_map): the same operators,
// a chain of up to WTAMD_MAP_CHAIN_MAX of them per track (`sum ln a scale 2 b` wraps tracks
// differently; `map` wraps all of them alike, commandParser.c:115-211), applied to one batch
// between its arrival in HBM and the window index.  Nothing returns to the host: run counts, block
// offsets and the new segment offsets stay on device, the kernels after it read the compacted
// lists through the rewritten seg_off[].
// ---------------------------------------------------------------------------
struct WmChain { int32_t n_ops; int32_t op[WTAMD_MAP_CHAIN_MAX]; double param[WTAMD_MAP_CHAIN_MAX]; double lg[WTAMD_MAP_CHAIN_MAX]; };

// OutT = float: every operator of every chain is float32-exact on float32 input (abs, the comparisons, scale by +-1):
// the batch stays float32 and with it on the exact difference-array kernels (wt_map_chain_f32_exact).
template <class ValT, class OutT>
__global__ void __launch_bounds__(WM_BLOCK) wm_chain_kernel(const WmChain *chains, const int64_t *seg, int n_tracks,
                                                             const ValT *in, long long n, OutT *out, uint8_t *keep_flag,
                                                             unsigned long long *block_keep) {
    __shared__ unsigned int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const long long base = (long long) blockIdx.x * WM_TILE;
    unsigned kept = 0;
    int t = -1;
    long long t_end = -1;
    for (int q = 0; q < WM_ITEMS; q++) {
        const long long g = base + threadIdx.x + (long long) q * WM_BLOCK;
        if (g >= n) break;
        if (g >= t_end) {       // track of run g: last t with seg[t] <= g
            int lo = 0, hi = n_tracks;
            while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (seg[m] <= g) lo = m; else hi = m; }
            t = lo; t_end = seg[t + 1];
        }
        const WmChain &c = chains[t];
        double v = (double) in[g];
        bool keep = true;
        for (int k = 0; k < c.n_ops && keep; k++) v = wm_apply(c.op[k], c.param[k], c.lg[k], v, keep);
        out[g] = (OutT) v;
        if (keep_flag) keep_flag[g] = keep ? 1 : 0;
        kept += keep ? 1u : 0u;
    }
    if (block_keep) {
        atomicAdd(&cnt, kept);
        __syncthreads();
        if (threadIdx.x == 0) block_keep[blockIdx.x] = cnt;
    }
}

template <class OutT>
__global__ void __launch_bounds__(WM_BLOCK) wm_compact_flag_kernel(const uint8_t *keep_flag, const int32_t *start, const int32_t *finish,
                                                                    const OutT *mapped, long long n,
                                                                    const unsigned long long *block_off, int32_t *o_start,
                                                                    int32_t *o_finish, OutT *o_value) {
    __shared__ unsigned int wave_tot[WM_BLOCK / 64];
    const long long base = (long long) blockIdx.x * WM_TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long off = block_off[blockIdx.x];
    const long long g0 = base + (long long) threadIdx.x * WM_ITEMS;
    unsigned mask = 0, mine = 0;
#pragma unroll
    for (int q = 0; q < WM_ITEMS; q++)
        if (g0 + q < n && keep_flag[g0 + q]) { mask |= 1u << q; mine++; }
    unsigned incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = (unsigned) __shfl_up((int) incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned before = incl - mine;
    for (int w = 0; w < wave; w++) before += wave_tot[w];
    unsigned long long o = off + before;
#pragma unroll
    for (int q = 0; q < WM_ITEMS; q++)
        if ((mask >> q) & 1u) {
            o_start[o] = start[g0 + q];
            o_finish[o] = finish[g0 + q];
            o_value[o] = mapped[g0 + q];
            o++;
        }
}

__global__ void wm_seg_offsets_flag(const uint8_t *keep_flag, const int64_t *seg_off, long long n_seg, long long n,
                                    const unsigned long long *block_off, const unsigned long long *total, int64_t *o_seg_off) {
    const long long s = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_seg) return;
    const long long p = seg_off[s];
    if (p >= n) { o_seg_off[s] = (int64_t) *total; return; }
    const long long b = p / WM_TILE;
    unsigned long long c = block_off[b];
    for (long long g = b * WM_TILE; g < p; g++) c += keep_flag[g] ? 1ull : 0ull;
    o_seg_off[s] = (int64_t) c;
}

}  // namespace

// Scratch of wt_map_chain_async in 8-byte words for batches of up to `capacity` runs: the mapped
// values before compaction, the block counts (+ total), the keep flags.
long long wt_map_scratch_words(long long capacity) {
    const long long nb = (capacity + WM_TILE - 1) / WM_TILE;
    return capacity + (nb + 2) + (capacity + 7) / 8;
}

bool wt_map_op_drops(int op) { return op == WTAMD_MAP_LN || op == WTAMD_MAP_LOG || op >= WTAMD_MAP_GT; }

// Operators whose result on a float32 value is a float32 whatever the value: |v|, the comparisons (1 or dropped),
// scale by +-1 (the `diff` idiom: Sum over [a, scale -1 b], commandParser.c:589-600).
bool wt_map_op_f32_exact(int op, double param) {
    if (op == WTAMD_MAP_ABS || (op >= WTAMD_MAP_GT && op <= WTAMD_MAP_LTE)) return true;
    if (op == WTAMD_MAP_SCALE) return param == 1.0 || param == -1.0;
    if (op == WTAMD_MAP_OFFSET) return param == 0.0;
    return false;
}

// d_chains: n_tracks device WmChain records (wt_map_upload_chains).  `drops`: some chain holds an
// operator that drops runs -- then o_start / o_finish / o_value receive the compacted lists and
// d_seg_out the new offsets; otherwise only o_value is written (the coordinates and d_seg_in stay
// what the kernels downstream read).  Everything is enqueued on `stream`, nothing waits.
int wt_map_chain_async(const void *d_chains, int n_tracks, bool drops, const int64_t *d_seg_in, long long n, const int32_t *start,
                       const int32_t *finish, const void *value, bool value_is_f64, unsigned long long *scratch,
                       int32_t *o_start, int32_t *o_finish, double *o_value, int64_t *d_seg_out, hipStream_t stream, bool out_f32) {
    if (n <= 0) {
        if (drops) return hipMemsetAsync(d_seg_out, 0, sizeof(int64_t) * ((size_t) n_tracks + 1), stream) == hipSuccess ? WTAMD_OK : WTAMD_ERR_HIP;
        return WTAMD_OK;
    }
    const long long nb = (n + WM_TILE - 1) / WM_TILE;
    double *d_mapped = drops ? (double *) scratch : o_value;
    unsigned long long *d_blk = scratch + n;
    uint8_t *d_flag = (uint8_t *) (d_blk + nb + 2);
    const WmChain *ch = (const WmChain *) d_chains;
    if (out_f32 && value_is_f64) return WTAMD_ERR_ARG;      // (float32 output is for float32 input only)
    if (value_is_f64)
        hipLaunchKernelGGL((wm_chain_kernel<double, double>), dim3((unsigned) nb), dim3(WM_BLOCK), 0, stream, ch, d_seg_in, n_tracks,
                           (const double *) value, n, d_mapped, drops ? d_flag : nullptr, drops ? d_blk : nullptr);
    else if (out_f32)
        hipLaunchKernelGGL((wm_chain_kernel<float, float>), dim3((unsigned) nb), dim3(WM_BLOCK), 0, stream, ch, d_seg_in, n_tracks,
                           (const float *) value, n, (float *) d_mapped, drops ? d_flag : nullptr, drops ? d_blk : nullptr);
    else
        hipLaunchKernelGGL((wm_chain_kernel<float, double>), dim3((unsigned) nb), dim3(WM_BLOCK), 0, stream, ch, d_seg_in, n_tracks,
                           (const float *) value, n, d_mapped, drops ? d_flag : nullptr, drops ? d_blk : nullptr);
    if (drops) {
        hipLaunchKernelGGL(wm_scan_blocks, dim3(1), dim3(64), 0, stream, d_blk, nb, d_blk + nb);
        if (out_f32)
            hipLaunchKernelGGL(wm_compact_flag_kernel<float>, dim3((unsigned) nb), dim3(WM_BLOCK), 0, stream, d_flag, start, finish,
                               (const float *) d_mapped, n, d_blk, o_start, o_finish, (float *) o_value);
        else
            hipLaunchKernelGGL(wm_compact_flag_kernel<double>, dim3((unsigned) nb), dim3(WM_BLOCK), 0, stream, d_flag, start, finish, d_mapped, n,
                           d_blk, o_start, o_finish, o_value);
        hipLaunchKernelGGL(wm_seg_offsets_flag, dim3((unsigned) ((n_tracks + 1 + 255) / 256)), dim3(256), 0, stream, d_flag, d_seg_in,
                           (long long) n_tracks, n, d_blk, d_blk + nb, d_seg_out);
    }
    return hipGetLastError() == hipSuccess ? WTAMD_OK : WTAMD_ERR_HIP;
}

// Host chains -> device records (log of the base / radix precomputed as wtamd_runs_map does).
int wt_map_upload_chains(const wtamd_map_chain *chains, int n_tracks, void **d_out, bool *drops, bool *f32_exact) {
    std::vector<WmChain> h((size_t) n_tracks);
    *drops = false;
    if (f32_exact) {
        *f32_exact = !getenv("WTAMD_MAP_F64");
        for (int t = 0; t < n_tracks && *f32_exact; t++)
            for (int k = 0; k < chains[t].n_ops && k < WTAMD_MAP_CHAIN_MAX; k++)
                if (!wt_map_op_f32_exact(chains[t].op[k], chains[t].param[k])) *f32_exact = false;
    }
    for (int t = 0; t < n_tracks; t++) {
        const wtamd_map_chain &c = chains[t];
        if (c.n_ops < 0 || c.n_ops > WTAMD_MAP_CHAIN_MAX) return wt_fail_ext(WTAMD_ERR_ARG, "wtamd_pipe_set_map: chain length");
        h[(size_t) t].n_ops = c.n_ops;
        for (int k = 0; k < WTAMD_MAP_CHAIN_MAX; k++) {
            const int op = k < c.n_ops ? c.op[k] : WTAMD_MAP_COUNT_;
            if (k < c.n_ops && (op < 0 || op >= WTAMD_MAP_COUNT_)) return wt_fail_ext(WTAMD_ERR_ARG, "wtamd_pipe_set_map: unknown operator");
            if (k < c.n_ops && (op == WTAMD_MAP_LOG || op == WTAMD_MAP_EXPB) && !(c.param[k] > 0))
                return wt_fail_ext(WTAMD_ERR_ARG, "wtamd_pipe_set_map: base / radix must be positive");
            h[(size_t) t].op[k] = op;
            h[(size_t) t].param[k] = k < c.n_ops ? c.param[k] : 0;
            h[(size_t) t].lg[k] = (op == WTAMD_MAP_LOG || op == WTAMD_MAP_EXPB) ? log(c.param[k]) : 1.0;
            if (k < c.n_ops && wt_map_op_drops(op)) *drops = true;
        }
    }
    void *d = nullptr;
    if (hipMalloc(&d, sizeof(WmChain) * (size_t) n_tracks) != hipSuccess) return wt_fail_ext(WTAMD_ERR_HIP, "hipMalloc(map chains)");
    if (hipMemcpy(d, h.data(), sizeof(WmChain) * (size_t) n_tracks, hipMemcpyHostToDevice) != hipSuccess) {
        (void) hipFree(d);
        return wt_fail_ext(WTAMD_ERR_HIP, "hipMemcpy(map chains)");
    }
    *d_out = d;
    return WTAMD_OK;
}

extern "C" {

#define WM_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return wt_fail_ext(WTAMD_ERR_HIP, std::string(#call ": ") + hipGetErrorString(e_)); } while (0)

int wtamd_runs_map(int map_op, double param, int64_t n_seg, const int64_t *seg_off, const int32_t *start,
                   const int32_t *finish, const void *value, int value_is_f64, int32_t *o_start, int32_t *o_finish,
                   double *o_value, int64_t *o_seg_off, void *stream) {
    if (map_op < 0 || map_op >= WTAMD_MAP_COUNT_ || n_seg < 0 || !seg_off || !o_value || !o_seg_off)
        return wt_fail_ext(WTAMD_ERR_ARG, "wtamd_runs_map: bad argument");
    if ((map_op == WTAMD_MAP_LOG || map_op == WTAMD_MAP_EXPB) && !(param > 0))
        return wt_fail_ext(WTAMD_ERR_ARG, "wtamd_runs_map: base / radix must be positive");
    hipStream_t s = (hipStream_t) stream;
    const long long n = seg_off[n_seg];
    const bool drops = map_op == WTAMD_MAP_LN || map_op == WTAMD_MAP_LOG || map_op >= WTAMD_MAP_GT;
    const double lg = (map_op == WTAMD_MAP_LOG || map_op == WTAMD_MAP_EXPB) ? log(param) : 1.0;
    if (n == 0) {
        for (int64_t q = 0; q <= n_seg; q++) o_seg_off[q] = 0;
        return WTAMD_OK;
    }
    const long long n_blocks = (n + WM_TILE - 1) / WM_TILE;
    unsigned long long *d_blk = nullptr;
    double *d_mapped = o_value;
    int64_t *d_seg = nullptr, *d_oseg = nullptr;
    WtDevScope scope;
    if (drops) {
        WM_HIP(scope.alloc(&d_blk, sizeof(unsigned long long) * (n_blocks + 1)));
        WM_HIP(scope.alloc(&d_mapped, sizeof(double) * n));
        WM_HIP(scope.alloc(&d_seg, sizeof(int64_t) * (n_seg + 1) * 2));
        d_oseg = d_seg + (n_seg + 1);
        WM_HIP(hipMemcpyAsync(d_seg, seg_off, sizeof(int64_t) * (n_seg + 1), hipMemcpyHostToDevice, s));
    }
    if (value_is_f64)
        hipLaunchKernelGGL(wm_map_kernel<double>, dim3((unsigned) n_blocks), dim3(WM_BLOCK), 0, s, map_op, param, lg,
                           (const double *) value, n, d_mapped, d_blk);
    else
        hipLaunchKernelGGL(wm_map_kernel<float>, dim3((unsigned) n_blocks), dim3(WM_BLOCK), 0, s, map_op, param, lg,
                           (const float *) value, n, d_mapped, d_blk);
    WM_HIP(hipGetLastError());
    if (!drops) {
        // coordinates are unchanged: copy them only if the caller asked for separate arrays
        if (o_start && o_start != start) WM_HIP(hipMemcpyAsync(o_start, start, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, s));
        if (o_finish && o_finish != finish) WM_HIP(hipMemcpyAsync(o_finish, finish, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, s));
        WM_HIP(hipStreamSynchronize(s));
        for (int64_t q = 0; q <= n_seg; q++) o_seg_off[q] = seg_off[q];
        return WTAMD_OK;
    }
    if (!o_start || !o_finish) return wt_fail_ext(WTAMD_ERR_ARG, "wtamd_runs_map: operators that drop runs need output coordinate arrays");
    hipLaunchKernelGGL(wm_scan_blocks, dim3(1), dim3(64), 0, s, d_blk, n_blocks, d_blk + n_blocks);
    unsigned long long total = 0;
    WM_HIP(hipMemcpyAsync(&total, d_blk + n_blocks, sizeof total, hipMemcpyDeviceToHost, s));
    WM_HIP(hipStreamSynchronize(s));
    const unsigned seg_grid = (unsigned) ((n_seg + 1 + 255) / 256);
    if (value_is_f64) {
        hipLaunchKernelGGL(wm_compact_kernel<double>, dim3((unsigned) n_blocks), dim3(WM_BLOCK), 0, s, map_op, param,
                           (const double *) value, start, finish, d_mapped, n, d_blk, o_start, o_finish, o_value);
        hipLaunchKernelGGL(wm_seg_offsets<double>, dim3(seg_grid), dim3(256), 0, s, map_op, param, (const double *) value, d_seg,
                           (long long) n_seg, n, d_blk, total, d_oseg);
    } else {
        hipLaunchKernelGGL(wm_compact_kernel<float>, dim3((unsigned) n_blocks), dim3(WM_BLOCK), 0, s, map_op, param,
                           (const float *) value, start, finish, d_mapped, n, d_blk, o_start, o_finish, o_value);
        hipLaunchKernelGGL(wm_seg_offsets<float>, dim3(seg_grid), dim3(256), 0, s, map_op, param, (const float *) value, d_seg,
                           (long long) n_seg, n, d_blk, total, d_oseg);
    }
    WM_HIP(hipGetLastError());
    WM_HIP(hipMemcpyAsync(o_seg_off, d_oseg, sizeof(int64_t) * (n_seg + 1), hipMemcpyDeviceToHost, s));
    WM_HIP(hipStreamSynchronize(s));
    return WTAMD_OK;
}

}  // extern "C"
