// wt_bwdev.hip -- BigWig sections decoded ON THE DEVICE: compressed file bytes in, run lists
// (start, finish, float value) of a pipeline batch out.  Replaces, for the file leg of the engine,
// what the reference gets from libBigWig's host-side zlib inflate + interval loop
// (src/bigWiggleReader.c:36-83) and what csrc/wt_bigwig.cpp does on host threads: with 100 files the
// host inflate bounded the whole path at 1.6e7 bp/s, and the compressed bytes are 2.5 x
// smaller than the run lists on the PCIe link.
//
// Kernels, all on the pipeline's decode stream, nothing returns to the host in between:
//   wt_bw_copy_kernel     pinned host staging (file bytes as read) -> HBM, 16 B per lane
//   wt_bw_inflate_kernel  ONE LANE PER SECTION: 64 independent zlib streams per wavefront
//                         (csrc/wt_inflate.h: limit-based canonical Huffman decoding, one symbol and its first
//                         8 bytes per branch-free step, loads landing at round boundaries; 288 B symbol table +
//                         32 B LZ77 ring of LDS per lane: 20 KB per wavefront, EIGHT wavefronts per CU), plain
//                         bytes to a strided scratch buffer.  Bound: dependent-instruction latency / VALU issue
//                         (a serial bit stream per lane); throughput comes from sections in flight.
//   wt_bw_count_kernel    one wavefront per section: Adler-32 against the stream's trailer, header, per-item piece
//                         counts (1-based shift, 10 000-bp boxing, clip window: csrc/wt_bwdev_core.h), order /
//                         extent checks
//   wt_bw_scan_kernel     one workgroup: exclusive scan over the sections -> piece offsets, the
//                         batch's device-side seg_off[], total and error word (pinned host status)
//   wt_bw_scatter_kernel  one wavefront per section: pieces to the slot's SoA arrays (coalesced)
// HBM traffic: 4.8 B (compressed) + 2 x 12 B (plain, written and read twice) + 12 B per interval --
// a few % of the reduce kernels' traffic; the inflate kernel is the one that costs time.
#include <hip/hip_runtime.h>
#include <algorithm>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/wiggletools_amd.h"
#include "wt_inflate.h"
#include "wt_bwdev_core.h"

int wt_fail_ext(int code, const std::string &msg);     // wt_engine.hip

namespace {

typedef unsigned int wt_u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) wt_bw_copy_kernel(const wt_u32x4 *src, wt_u32x4 *dst, long long n16) {
    const long long stride = (long long) gridDim.x * 256 * 4;
    for (long long i = (long long) blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride) {
        wt_u32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) if (i + 256 * q < n16) v[q] = __builtin_nontemporal_load(src + i + 256 * q);
#pragma unroll
        for (int q = 0; q < 4; q++) if (i + 256 * q < n16) dst[i + 256 * q] = v[q];
    }
}

#define WT_BW_INF_LANES 64

// RING = 8: 320 B of LDS per lane, eight wavefronts per CU (two per SIMD); RING = 64: 544 B, four per CU, matches up to
// 252 bytes back served from LDS (WTAMD_INFLATE_RING=64; kept for comparison).
template <int RING>
__global__ void __launch_bounds__(WT_BW_INF_LANES) wt_bw_inflate_kernel(const WtBwSection *secs, const WtBwTrack *tracks, int n_sec,
                                                                         const uint8_t *comp, uint8_t *plain, uint32_t plain_stride,
                                                                         int32_t *plain_len, uint32_t *trailer_at) {
    __shared__ uint32_t s_perm[(WT_INF_PERM / 4) * WT_BW_INF_LANES];
    __shared__ uint32_t s_ring[RING * WT_BW_INF_LANES];
    const int lane = threadIdx.x;
    const int i = blockIdx.x * WT_BW_INF_LANES + lane;
    if (i >= n_sec) return;
    const WtBwSection sc = secs[i];
    const bool compressed = tracks[sc.track].compressed != 0;
    WtInfMem m;
    m.perm = (WT_AS_LDS uint8_t *) (s_perm + lane);
    m.ring = (WT_AS_LDS uint32_t *) (s_ring + lane); m.stride = WT_BW_INF_LANES;
    WtInflateT<RING> z;
    wt_inf_begin(z, comp + sc.comp_off, sc.comp_size, plain + (size_t) i * plain_stride, plain_stride, false);
    if (!compressed) {          // an uncompressed file: the section bytes are copied (a stored block in disguise)
        z.st = sc.comp_size ? WT_INF_ST_STORED : WT_INF_ST_DONE;
        z.stored_rem = sc.comp_size;
        z.last = true;
        if (sc.comp_size > plain_stride) wt_inf_fail(z, WT_INF_ERR_SPACE);
    }
    plain_len[i] = (int32_t) wt_inf_run(z, m);
    // where the stream's final block ended: the Adler-32 trailer sits THERE, not at the end of the index leaf -- a leaf may
    // carry padding behind its stream (round 4 read the trailer at comp_off + comp_size - 4 and rejected such files: the
    // advisor's finding).  Kept in counts[], which the count kernel reads before it writes the section's count there.
    trailer_at[i] = wt_inf_end_byte(z);
}

int wt_bw_inflate_ring() {
    static const int ring = (getenv("WTAMD_INFLATE_RING") && atoi(getenv("WTAMD_INFLATE_RING")) == 64) ? 64 : 8;
    return ring;
}

__device__ __forceinline__ uint32_t wt_wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wt_wave_scan(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

__global__ void __launch_bounds__(64) wt_bw_count_kernel(const WtBwSection *secs, const WtBwTrack *tracks, int n_sec, const uint8_t *comp,
                                                          const uint8_t *plain, uint32_t plain_stride, const int32_t *plain_len,
                                                          uint32_t *counts, uint32_t *err) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n_sec) return;
    const WtBwSection sc = secs[i];
    const WtBwTrack tk = tracks[sc.track];
    const int32_t len = plain_len[i];
    uint32_t bad = 0, total = 0;
    if (len < 0) {
        bad = WT_BW_ERR_INFLATE;
    } else {
        const uint8_t *p = plain + (size_t) i * plain_stride;
        if (tk.compressed && sc.comp_size >= 6) {
            // Adler-32 of what was inflated against the stream's trailer (RFC 1950; libBigWig's uncompress() checks it,
            // so a damaged payload that still parses must not pass here either): A = 1 + sum d_i, B = sum of the running A.
            // Every lane takes a slice; B = n + sum over slices of (their own B + bytes behind the slice x their A).
            // Lanes take the dwords of the section INTERLEAVED (one coalesced 256-byte read per wavefront instruction):
            // A = 1 + sum d_i,  B = n + sum (n - i) d_i = n + n * sum d_i - sum i * d_i.
            const uint32_t n = (uint32_t) len;
            unsigned long long A = 0, C = 0;            // sum d_i, sum i * d_i of this lane's bytes
            for (uint32_t q = 4u * (uint32_t) lane; q < n; q += 256u) {
                const uint32_t w = *(const uint32_t *) (p + q);
                uint32_t a = 0, c = 0;
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t d = q + k < n ? (w >> (8u * k)) & 255u : 0u;
                    a += d; c += k * d;
                }
                A += a;
                C += (unsigned long long) q * a + c;
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { A += __shfl_xor(A, o, 64); C += __shfl_xor(C, o, 64); }
            const unsigned long long B = (unsigned long long) n * A - C;
            const uint32_t ad = (uint32_t) (((B + n) % 65521ull) << 16) | (uint32_t) ((A + 1ull) % 65521ull);
            const uint32_t at = counts[i];              // left there by the inflate kernel: the end of the final block
            if (at + 4u > sc.comp_size) {
                bad |= WT_BW_ERR_INFLATE;               // no room for a trailer: truncated
            } else {
                const uint8_t *t = comp + sc.comp_off + at;
                const uint32_t want = ((uint32_t) t[0] << 24) | ((uint32_t) t[1] << 16) | ((uint32_t) t[2] << 8) | (uint32_t) t[3];
                if (ad != want) bad |= WT_BW_ERR_INFLATE;
            }
        }
        WtBwHdr h;
        if (!wt_bw_parse_hdr(p, (uint32_t) len, h)) {
            bad |= WT_BW_ERR_SECTION;
        } else if (h.chrom_id == tk.chrom_id) {     // (else: a block of another chromosome sharing the index leaf)
            for (uint32_t k0 = 0; k0 < h.count; k0 += 64) {
                const uint32_t k = k0 + lane;
                if (k < h.count) {
                    uint32_t s0, e0, vb, item_bad = 0;
                    wt_bw_item(p, h, k, s0, e0, vb);
                    if (s0 < sc.leaf_start || e0 > sc.leaf_end || e0 <= s0) item_bad |= WT_BW_ERR_EXTENT;
                    if (e0 >= (uint32_t) WTAMD_MAX_COORD) item_bad |= WT_BW_ERR_COORD;
                    if (k > 0) {
                        uint32_t ps, pe, pv;
                        wt_bw_item(p, h, k - 1, ps, pe, pv);
                        if (s0 < pe) item_bad |= WT_BW_ERR_EXTENT;  // unsorted or overlapping items
                    }
                    // (a malformed item is not boxed: the batch is rejected anyway, and a bogus span would be walked
                    // stretch by stretch -- up to 430 000 iterations of one lane)
                    if (!item_bad) total += wt_bw_pieces(s0, e0, tk, [](int32_t, int32_t) {});
                    bad |= item_bad;
                }
            }
        }
    }
    total = wt_wave_sum(total);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) bad |= __shfl_xor(bad, o, 64);
    if (lane == 0) {
        counts[i] = total;
        if (bad) atomicOr(err, bad);
    }
}

// status[0] = error bits, status[1] = pieces of the batch (pinned host memory)
__global__ void __launch_bounds__(1024) wt_bw_scan_kernel(const WtBwTrack *tracks, int n_tracks, int n_sec, const uint32_t *counts,
                                                           long long *offsets, int64_t *seg_off, long long capacity,
                                                           uint32_t *err, unsigned long long *status) {
    __shared__ long long s_part[1024];
    __shared__ long long s_carry;
    const int t = threadIdx.x;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n_sec; base += 1024 * 8) {
        long long v[8], sum = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int i = base + t * 8 + q;
            v[q] = i < n_sec ? (long long) counts[i] : 0;
            sum += v[q];
        }
        s_part[t] = sum;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const long long x = t >= o ? s_part[t - o] : 0;
            __syncthreads();
            s_part[t] += x;
            __syncthreads();
        }
        long long run = s_carry + s_part[t] - sum;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int i = base + t * 8 + q;
            if (i < n_sec) offsets[i] = run;
            run += v[q];
        }
        __syncthreads();
        if (t == 1023) s_carry += s_part[1023];
        __syncthreads();
    }
    const long long total = s_carry;
    if (t == 0) offsets[n_sec] = total;
    __syncthreads();
    // any error: the batch is handed on EMPTY (the kernels downstream must not read what was not written)
    const bool failed = *err != 0 || total > capacity;
    for (int k = t; k <= n_tracks; k += 1024) {
        long long o = total;
        if (failed) o = 0;
        else if (k < n_tracks) {
            // the first section at or after this track's slice (empty slices take the next track's offset)
            const int fs = tracks[k].first_section;
            o = fs < n_sec ? offsets[fs] : total;
        }
        seg_off[k] = o;
    }
    if (t == 0) {
        uint32_t e = *err;
        if (total > capacity) e |= WT_BW_ERR_CAPACITY;
        *err = e;
        status[0] = e;
        status[1] = (unsigned long long) total;
    }
}

__global__ void __launch_bounds__(64) wt_bw_scatter_kernel(const WtBwSection *secs, const WtBwTrack *tracks, int n_sec,
                                                            const uint8_t *plain, uint32_t plain_stride, const int32_t *plain_len,
                                                            const long long *offsets, long long capacity, const uint32_t *err,
                                                            int32_t *o_start, int32_t *o_finish, float *o_value) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n_sec) return;
    if (*err) return;           // a failed batch is empty (see the scan kernel)
    const int32_t len = plain_len[i];
    if (len < 0) return;
    const WtBwSection sc = secs[i];
    const WtBwTrack tk = tracks[sc.track];
    const uint8_t *p = plain + (size_t) i * plain_stride;
    WtBwHdr h;
    if (!wt_bw_parse_hdr(p, (uint32_t) len, h) || h.chrom_id != tk.chrom_id) return;
    long long at = offsets[i];
    const long long end = offsets[i + 1];
    for (uint32_t k0 = 0; k0 < h.count; k0 += 64) {
        const uint32_t k = k0 + lane;
        uint32_t s0 = 0, e0 = 0, vb = 0, n = 0;
        if (k < h.count) {
            wt_bw_item(p, h, k, s0, e0, vb);
            n = wt_bw_pieces(s0, e0, tk, [](int32_t, int32_t) {});
        }
        const uint32_t incl = wt_wave_scan(n, lane);
        long long w = at + (long long) (incl - n);
        if (n) {
            const float v = __uint_as_float(vb);
            wt_bw_pieces(s0, e0, tk, [&](int32_t a, int32_t b) {
                if (w < end && w < capacity) { o_start[w] = a; o_finish[w] = b; o_value[w] = v; }
                w++;
            });
        }
        at += (long long) __shfl(incl, 63, 64);
    }
}

}  // namespace

// sections resident per launch of the inflate kernel: CUs x wavefronts per CU x 64 lanes
long long wt_bw_fill_sections(int num_cu) {
    const void *kern = wt_bw_inflate_ring() == 64 ? (const void *) wt_bw_inflate_kernel<64> : (const void *) wt_bw_inflate_kernel<8>;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, WT_BW_INF_LANES, 0) != hipSuccess || per_cu < 1) {
        (void) hipGetLastError();
        per_cu = 0;
    }
    // The runtime's occupancy calculator is not the hardware's dispatcher: the same code object gets 8 workgroups per CU from
    // the HIP runtime PyTorch bundles and 4 from ROCm 7.2's own (round 5: a process linked against /opt/rocm sized its batches
    // for half the lanes -- 98 batches of 55 296 sections where the other runtime cut 59 of 100 000 -- and inflated at 5.8
    // instead of 8.1 sections / us).  What the workgroups of this kernel need is known: static LDS and registers
    // (hipFuncGetAttributes) against the CU's 160 KB of LDS and 512 registers per SIMD lane; the larger of the two answers
    // is used -- a batch twice the resident lanes costs a second round of the launch, half a batch wastes half the GPU.
    hipFuncAttributes fa{};
    int by_need = 0;
    if (hipFuncGetAttributes(&fa, kern) == hipSuccess) {
        int dev = 0;
        (void) hipGetDevice(&dev);
        int lds_cu = 0;
        if (hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || lds_cu <= 0) { (void) hipGetLastError(); lds_cu = 160 * 1024; }
        const int by_lds = fa.sharedSizeBytes > 0 ? lds_cu / (int) fa.sharedSizeBytes : 32;
        const int regs = fa.numRegs > 0 ? (fa.numRegs + 7) & ~7 : 128;
        int waves_simd = 512 / regs;
        if (waves_simd > 8) waves_simd = 8;
        if (waves_simd < 1) waves_simd = 1;
        by_need = std::min(by_lds, 4 * waves_simd);      // (one wavefront per workgroup, four SIMDs per CU)
        if (by_need > 32) by_need = 32;
    } else {
        (void) hipGetLastError();
    }
    if (const char *e = getenv("WTAMD_BW_WAVES_PER_CU")) { const int v = atoi(e); if (v >= 1 && v <= 32) { per_cu = v; by_need = 0; } }
    static const bool trace = getenv("WTAMD_TRACE") != nullptr;
    if (trace) fprintf(stderr, "[bwdev] inflate kernel: %d workgroups per CU by the runtime's calculator, %d by LDS / registers (%d B, %d regs)\n", per_cu, by_need,
                       (int) fa.sharedSizeBytes, (int) fa.numRegs);
    if (by_need > per_cu) per_cu = by_need;
    if (per_cu < 1) per_cu = 3;
    return (long long) num_cu * per_cu * WT_BW_INF_LANES;
}

// scratch a batch of `n_sec` sections with `plain_stride` bytes each needs, in bytes
long long wt_bw_scratch_bytes(long long n_sec, long long plain_stride) {
    const long long n = n_sec > 0 ? n_sec : 1;
    // plain | plain_len (i32) | counts (u32) | offsets (i64, n + 1) | err (u32, padded)
    return n * plain_stride + n * 4 + n * 4 + (n + 1) * 8 + 64 + 256;
}

// Enqueues copy (on s_copy) -> [event] -> inflate, count, scan, scatter (on s_dec).
//   h_bytes / d_bytes: the batch's staging (tables + file bytes), pinned host and device twin, n_bytes (padded to 16)
//   d_comp: where the file bytes start inside d_bytes;  d_secs / d_tracks: the tables inside d_bytes
//   scratch: wt_bw_scratch_bytes() of device memory
//   d_seg_off: n_tracks + 1 offsets written on device;  h_status: 2 words of pinned host memory
int wt_bw_decode_async(const void *h_bytes, void *d_bytes, long long n_bytes, const void *d_comp, const void *d_secs, const void *d_tracks, int n_tracks,
                       long long n_sec, long long plain_stride, void *scratch, long long capacity, int32_t *o_start, int32_t *o_finish,
                       float *o_value, int64_t *d_seg_off, unsigned long long *h_status, int copy_blocks, hipStream_t s_copy,
                       hipEvent_t e_copied, hipStream_t s_dec) {
    if (n_sec < 0 || n_sec > 0x7FFFFFFFll - 64 || plain_stride <= 0 || (plain_stride & 15) || plain_stride > 0x7FFFFFFFll)
        return wt_fail_ext(WTAMD_ERR_ARG, "wt_bw_decode_async: bad section count / stride");
    const long long n = n_sec > 0 ? n_sec : 1;
    uint8_t *plain = (uint8_t *) scratch;
    int32_t *plain_len = (int32_t *) (plain + n * plain_stride);
    uint32_t *counts = (uint32_t *) (plain_len + n);
    long long *offsets = (long long *) (((uintptr_t) (counts + n) + 7) & ~(uintptr_t) 7);
    uint32_t *err = (uint32_t *) (offsets + n + 1);
#define WT_BW_HIP(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return wt_fail_ext(WTAMD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
    const long long n16 = (n_bytes + 15) / 16;
    // the copy engine by default (one contiguous range: no per-range cost), WTAMD_BW_COPY=kernel: the copy kernel
    static const bool sdma = !(getenv("WTAMD_BW_COPY") && !strcmp(getenv("WTAMD_BW_COPY"), "kernel"));
    if (n16 > 0 && sdma) {
        // the copy engine instead of a kernel: nothing of the transfer runs on the CUs the inflate kernel occupies
        WT_BW_HIP(hipMemcpyAsync(d_bytes, h_bytes, (size_t) n16 * 16, hipMemcpyHostToDevice, s_copy));
    } else if (n16 > 0) {
        long long grid = copy_blocks > 0 ? copy_blocks : 64;
        const long long need = (n16 + 1023) / 1024;
        if (grid > need) grid = need;
        hipLaunchKernelGGL(wt_bw_copy_kernel, dim3((unsigned) grid), dim3(256), 0, s_copy, (const wt_u32x4 *) h_bytes, (wt_u32x4 *) d_bytes, n16);
        WT_BW_HIP(hipGetLastError());
    }
    WT_BW_HIP(hipEventRecord(e_copied, s_copy));
    WT_BW_HIP(hipStreamWaitEvent(s_dec, e_copied, 0));
    WT_BW_HIP(hipMemsetAsync(err, 0, sizeof(uint32_t), s_dec));
    const WtBwSection *secs = (const WtBwSection *) d_secs;
    const WtBwTrack *tracks = (const WtBwTrack *) d_tracks;
    if (n_sec > 0) {
        const unsigned g = (unsigned) ((n_sec + WT_BW_INF_LANES - 1) / WT_BW_INF_LANES);
        if (wt_bw_inflate_ring() == 64)
            hipLaunchKernelGGL(wt_bw_inflate_kernel<64>, dim3(g), dim3(WT_BW_INF_LANES), 0, s_dec, secs, tracks, (int) n_sec,
                               (const uint8_t *) d_comp, plain, (uint32_t) plain_stride, plain_len, counts);
        else
            hipLaunchKernelGGL(wt_bw_inflate_kernel<8>, dim3(g), dim3(WT_BW_INF_LANES), 0, s_dec, secs, tracks, (int) n_sec,
                               (const uint8_t *) d_comp, plain, (uint32_t) plain_stride, plain_len, counts);
        WT_BW_HIP(hipGetLastError());
        hipLaunchKernelGGL(wt_bw_count_kernel, dim3((unsigned) n_sec), dim3(64), 0, s_dec, secs, tracks, (int) n_sec, (const uint8_t *) d_comp, plain,
                           (uint32_t) plain_stride, plain_len, counts, err);
        WT_BW_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(wt_bw_scan_kernel, dim3(1), dim3(1024), 0, s_dec, tracks, n_tracks, (int) n_sec, counts, offsets, d_seg_off,
                       capacity, err, h_status);
    WT_BW_HIP(hipGetLastError());
    if (n_sec > 0) {
        hipLaunchKernelGGL(wt_bw_scatter_kernel, dim3((unsigned) n_sec), dim3(64), 0, s_dec, secs, tracks, (int) n_sec, plain,
                           (uint32_t) plain_stride, plain_len, offsets, capacity, err, o_start, o_finish, o_value);
        WT_BW_HIP(hipGetLastError());
    }
#undef WT_BW_HIP
    return WTAMD_OK;
}
