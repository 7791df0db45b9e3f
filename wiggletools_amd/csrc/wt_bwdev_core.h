// wt_bwdev_core.h -- BigWig section -> run-list pieces, the per-item arithmetic shared by the device
// kernels (csrc/wt_bwdev.hip) and their CPU emulation (tests/emu/wt_bw_emu.cpp).
//
// A section (the plain bytes of one zlib stream of a BigWig file, or the raw bytes of an
// uncompressed file) is a 24-byte header and `count` items of type 1 bedGraph (start, end, value),
// 2 variableStep (start, value; end = start + span) or 3 fixedStep (value; start = chromStart +
// k * step).  What the reference's reader does with the intervals libBigWig hands it
// (src/bigWiggleReader.c):
//   * 0-based half-open -> 1-based start, exclusive finish                          :39-40
//   * a whole-chromosome read queries 10 000-bp stretches [1 + 10000 k, 1 + 10000 (k+1)) for
//     1 + 10000 k < chromosome length and boxes every interval into the stretch   :42-44, :73-83
//     (an interval crossing a stretch edge arrives as several pieces; nothing at or beyond the
//     last stretch start that is >= length)
//   * after seek() ONE region query, boxed into [start, finish) only               :91-92, :125-145
// Here: pieces of item k of a section = its stretches (box) or the item itself, each clipped to the
// track's [clip_lo, clip_hi) (the seek window, or "from the iterator's current element on") and
// dropped when empty.
#ifndef WT_BWDEV_CORE_H_
#define WT_BWDEV_CORE_H_

#include <stdint.h>
#include <string.h>

#include "../../include/wiggletools_amd.h"

#ifndef WT_HD
#if defined(__HIPCC__)
#define WT_HD __host__ __device__ __forceinline__
#else
#define WT_HD inline
#endif
#endif

#define WT_BW_STRETCH 10000

// One section / one track of a batch: the public table types (include/wiggletools_amd.h), read by the kernels.
typedef wtamd_bw_section WtBwSection;
typedef wtamd_bw_track WtBwTrack;

// Error bits a batch reports (device counter / emulator return)
#define WT_BW_ERR_INFLATE 1u    // a zlib stream did not inflate
#define WT_BW_ERR_SECTION 2u    // malformed section (size vs item count, unknown type)
#define WT_BW_ERR_EXTENT 4u     // an item lies outside its index leaf's extents, or items out of order
#define WT_BW_ERR_COORD 8u      // a coordinate beyond the engine's maximum
#define WT_BW_ERR_CAPACITY 16u  // more pieces than the host's bound

struct WtBwHdr {
    uint32_t chrom_id, start, end, step, span;
    uint32_t type, count;
};

WT_HD uint32_t wt_bw_u32(const uint8_t *p) {
    return (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16) | ((uint32_t) p[3] << 24);
}

// False: malformed.
WT_HD bool wt_bw_parse_hdr(const uint8_t *p, uint32_t len, WtBwHdr &h) {
    if (len < 24) return false;
    h.chrom_id = wt_bw_u32(p); h.start = wt_bw_u32(p + 4); h.end = wt_bw_u32(p + 8);
    h.step = wt_bw_u32(p + 12); h.span = wt_bw_u32(p + 16);
    h.type = p[20];
    h.count = (uint32_t) p[22] | ((uint32_t) p[23] << 8);
    if (h.type < 1 || h.type > 3) return false;
    const uint32_t item = h.type == 1 ? 12u : h.type == 2 ? 8u : 4u;
    return 24u + item * h.count <= len;
}

// Item k: 0-based half-open [s0, e0) and its value bits.  `p` = section bytes (4-byte aligned).
WT_HD void wt_bw_item(const uint8_t *p, const WtBwHdr &h, uint32_t k, uint32_t &s0, uint32_t &e0, uint32_t &vbits) {
    const uint32_t *q = (const uint32_t *) (p + 24);
    if (h.type == 1) { s0 = q[3 * k]; e0 = q[3 * k + 1]; vbits = q[3 * k + 2]; }
    else if (h.type == 2) { s0 = q[2 * k]; e0 = s0 + h.span; vbits = q[2 * k + 1]; }
    else { s0 = h.start + k * h.step; e0 = s0 + h.span; vbits = q[k]; }
}

// Pieces of the 0-based item [s0, e0): calls emit(start, finish) (1-based start, exclusive finish) for
// each, in order; returns their number.  emit may be a counting no-op.
template <class F>
WT_HD uint32_t wt_bw_pieces(uint32_t s0, uint32_t e0, const WtBwTrack &t, F emit) {
    const int64_t s = (int64_t) s0 + 1, f = (int64_t) e0 + 1;      // bigWiggleReader.c:39-40
    const int64_t lo = t.clip_lo, hi = t.clip_hi;
    uint32_t n = 0;
    if (!t.box) {
        const int64_t a = s > lo ? s : lo, b = f < hi ? f : hi;
        if (a < b) { emit((int32_t) a, (int32_t) b); n = 1; }
        return n;
    }
    const int64_t length = t.chrom_len;
    if (f <= lo || s >= hi) return 0;
    // stretches k with 1 + 10000 k < min(length, f), from the one holding s -- or, cheaper, from the one
    // holding clip_lo when that lies beyond it (earlier stretches end at or before clip_lo)
    int64_t k = (s - 1) / WT_BW_STRETCH;
    if (lo > s) { const int64_t kl = (lo - 1) / WT_BW_STRETCH; if (kl > k) k = kl; }
    for (;; k++) {
        const int64_t a = 1 + k * WT_BW_STRETCH, b = a + WT_BW_STRETCH;
        if (a >= length || a >= f || a >= hi) break;
        int64_t bs = s > a ? s : a, bf = f < b ? f : b;             // :42-44
        if (bs < lo) bs = lo;
        if (bf > hi) bf = hi;
        if (bs < bf) { emit((int32_t) bs, (int32_t) bf); n++; }
    }
    return n;
}

// Upper bound of the pieces a section can produce, from what the host knows without inflating it.
static inline int64_t wt_bw_section_bound(uint32_t plain_bytes, uint32_t leaf_start, uint32_t leaf_end, int box) {
    int64_t items = plain_bytes > 24 ? (int64_t) (plain_bytes - 24) / 4 : 0;      // fixedStep items are the smallest
    if (items > 65535) items = 65535;
    if (box) items += ((int64_t) leaf_end - (int64_t) leaf_start) / WT_BW_STRETCH + 2;
    return items;
}

#endif  // WT_BWDEV_CORE_H_
