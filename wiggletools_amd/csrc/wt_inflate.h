// wt_inflate.h -- DEFLATE / zlib (RFC 1950, 1951) decoder written for ONE GPU LANE PER STREAM.
//
// Why: a BigWig file is a list of independent zlib streams ("sections" of <= a few thousand items,
// reference src/bigWiggleReader.c:52-83 reads them through libBigWig, which inflates every one on the
// host).  The file leg of the engine was bound by exactly that host inflate.  Sections are
// independent, so here every lane of a wavefront inflates its own section: 64 streams per wavefront,
// tens of thousands in flight per GPU, no cross-lane communication at all.
//
// What makes a serial bit-stream decoder fit a lane:
//   * no look-up tables sized 2^bits: canonical Huffman decoding by LIMITS.  For a code with
//     count[k] symbols of length k the left-aligned 15-bit window w of the stream has length
//         len = 1 + #{ k in 1..14 : w >= lim[k] },   lim[k] = (first[k] + count[k]) << (15 - k)
//     and the symbol is perm[adj[len] + (w >> (15 - len))].  The 2 x 15 limits live in REGISTERS
//     (statically indexed arrays), `adj` (16 entries) and `perm` (the symbols sorted by code) in
//     LDS -- 2 dependent LDS reads per symbol, 704 bytes of LDS per lane.
//   * the code lengths of a dynamic block are parked in the unused top 4 bits of perm[] while
//     the table is built in place; the 19-symbol code-length code lives in two 64-bit registers.
//   * LZ77 history: the last 64 output bytes of every lane are kept in an LDS ring; matches with a
//     distance <= 64 (the bulk of them in 12-byte record data) never touch global memory, longer
//     ones read the lane's own earlier output back.  Output is gathered to dwords in a register.
//   * one state machine step per loop iteration, a match copies at most WT_INF_COPY bytes per
//     iteration, so a lane inside a 258-byte match does not stall the 63 others.
//
// The same code compiles for the host (tests/emu, tests/test_inflate.py: checked against zlib's
// own output on stored / fixed / dynamic streams of every compression level) -- `stride` is 1 there
// and the "LDS" arrays are plain memory.
//
// Memory layout: element i of a lane's array lies at base[i * stride] (+ lane), stride = lanes of
// the workgroup: lanes that walk their tables in lock step (table construction) hit consecutive
// addresses, random accesses are at worst 2-way bank conflicted (u16).
#ifndef WT_INFLATE_H_
#define WT_INFLATE_H_

#include <stdint.h>

#ifndef WT_HD
#if defined(__HIPCC__)
#define WT_HD __host__ __device__ __forceinline__
#else
#define WT_HD inline
#endif
#endif

#define WT_INF_PERM 320         // [0, 288) literal / length symbols, [288, 320) distance symbols
#define WT_INF_DBASE 288
#define WT_INF_AUX 32           // [0, 16) literal / length adj (or counters), [16, 32) distance
#define WT_INF_RING 64
#define WT_INF_COPY 4           // match bytes copied per state machine step

// bytes of "LDS" one lane needs
#define WT_INF_LANE_BYTES (WT_INF_PERM * 2 + WT_INF_AUX * 2 + WT_INF_RING)

enum {
    WT_INF_OK = 0,
    WT_INF_ERR_HEADER = 1,      // not a zlib / deflate stream (CMF / FLG), preset dictionary
    WT_INF_ERR_BLOCK = 2,       // reserved block type, stored LEN / NLEN mismatch
    WT_INF_ERR_CODE = 3,        // over-subscribed or unusable Huffman code, bad repeat
    WT_INF_ERR_SYMBOL = 4,      // a bit pattern no code word matches, length / distance symbol out of range
    WT_INF_ERR_DIST = 5,        // distance beyond the start of the output
    WT_INF_ERR_SPACE = 6,       // output does not fit the capacity
    WT_INF_ERR_INPUT = 7        // the stream ends before the final block does
};

enum { WT_INF_ST_ZHDR = 0, WT_INF_ST_BLOCK = 1, WT_INF_ST_SYM = 2, WT_INF_ST_STORED = 3, WT_INF_ST_DONE = 4, WT_INF_ST_ERR = 5 };

struct WtInfMem {
    uint16_t *perm;     // WT_INF_PERM entries
    uint16_t *aux;      // WT_INF_AUX entries
    uint8_t *ring;      // WT_INF_RING bytes
    int stride;         // elements between consecutive entries of this lane
};

struct WtInflate {
    // input: 32-bit words at 4-byte aligned addresses
    const uint32_t *in_w;       // aligned base
    uint32_t in_next;           // next word to fetch
    uint32_t in_words;          // words that may be read
    uint32_t nw;                // prefetched word in_w[in_next - 1 + ...] (see wt_inf_fetch)
    uint64_t bb;                // bit buffer, LSB first
    int32_t bc;                 // valid bits in bb
    int64_t bits_left;          // bits of the stream not yet consumed (underflow = truncated input)
    // output
    uint8_t *out;               // 4-byte aligned
    uint32_t out_pos, out_cap;
    uint32_t acc;               // bytes of the current output dword
    // state
    int32_t st, err;
    uint32_t copy_rem, copy_dist;
    uint32_t stored_rem;
    bool last, raw;             // last block seen; raw deflate (no zlib wrapper)
    uint32_t llim[16], dlim[16];    // [1..15] used
};

WT_HD uint32_t wt_inf_bitrev15(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x) >> 17;
#else
    uint32_t r = 0;
    for (int i = 0; i < 15; i++) r |= ((x >> i) & 1u) << (14 - i);
    return r;
#endif
}

WT_HD uint32_t wt_inf_load(const WtInflate &z, uint32_t i) { return i < z.in_words ? z.in_w[i] : 0u; }

// Starts a stream of `n_bytes` at `src` (any alignment; up to 3 bytes before and 3 after it, inside
// the same allocation, are read and ignored) writing at most `cap` bytes to `dst` (4-byte aligned, cap
// rounded up to a multiple of 4 must be writable).
WT_HD void wt_inf_begin(WtInflate &z, const uint8_t *src, uint32_t n_bytes, uint8_t *dst, uint32_t cap, bool raw_deflate) {
    const uintptr_t a = (uintptr_t) src;
    const uint32_t mis = (uint32_t) (a & 3u);
    z.in_w = (const uint32_t *) (a - mis);
    z.in_words = (mis + n_bytes + 3u) >> 2;
    z.bits_left = (int64_t) n_bytes * 8;
    z.bb = 0; z.bc = 0;
    z.in_next = 0;
    if (z.in_words > 0) {
        z.bb = (uint64_t) (z.in_w[0] >> (8 * mis));
        z.bc = 32 - 8 * (int32_t) mis;
        z.in_next = 1;
    }
    z.nw = wt_inf_load(z, z.in_next);
    z.out = dst; z.out_pos = 0; z.out_cap = cap; z.acc = 0;
    z.st = raw_deflate ? WT_INF_ST_BLOCK : WT_INF_ST_ZHDR;
    z.err = WT_INF_OK;
    z.copy_rem = z.copy_dist = 0; z.stored_rem = 0;
    z.last = false; z.raw = raw_deflate;
#pragma unroll
    for (int k = 0; k < 16; k++) { z.llim[k] = 0; z.dlim[k] = 0; }
}

// After this bc >= 33 (or the input is exhausted: zero bits follow, bits_left catches over-reads).
WT_HD void wt_inf_refill(WtInflate &z) {
    if (z.bc <= 32) {
        z.bb |= (uint64_t) z.nw << z.bc;
        z.bc += 32;
        z.in_next++;
        z.nw = wt_inf_load(z, z.in_next);      // consumed at the next refill: its latency hides behind the decode
    }
}

WT_HD uint32_t wt_inf_bits(WtInflate &z, int n) {      // n <= 32, after a refill guaranteeing enough bits
    const uint32_t v = (uint32_t) (z.bb & ((1ull << n) - 1ull));
    z.bb >>= n; z.bc -= n; z.bits_left -= n;
    return v;
}

WT_HD void wt_inf_fail(WtInflate &z, int code) { z.st = WT_INF_ST_ERR; z.err = code; z.copy_rem = 0; }

// One Huffman symbol.  lim[1..15] registers, adj / perm in lane memory.  Returns -1 on a pattern no
// code word matches.
WT_HD int wt_inf_decode(WtInflate &z, const uint32_t (&lim)[16], const uint16_t *adj, const uint16_t *perm, int stride) {
    const uint32_t w = wt_inf_bitrev15((uint32_t) z.bb & 0x7FFFu);
    int len = 1;
#pragma unroll
    for (int k = 1; k <= 14; k++) len += (w >= lim[k]) ? 1 : 0;
    if (w >= lim[15]) return -1;
    const int32_t a = (int32_t) (int16_t) adj[len * stride];
    const int32_t idx = a + (int32_t) (w >> (15 - len));
    z.bb >>= len; z.bc -= len; z.bits_left -= len;
    return (int) (perm[idx * stride] & 0x1FFu);
}

// Canonical table of the `n` symbols whose lengths sit in the top 4 bits of perm[0..n): fills lim[],
// adj[] (16 entries) and the low 9 bits of perm[].  False: over-subscribed.
WT_HD bool wt_inf_build(uint32_t (&lim)[16], uint16_t *adj, uint16_t *perm, int n, int stride) {
#pragma unroll
    for (int k = 0; k < 16; k++) adj[k * stride] = 0;
    for (int s = 0; s < n; s++) {
        const int l = perm[s * stride] >> 12;
        adj[l * stride] = (uint16_t) (adj[l * stride] + 1);
    }
    uint32_t code = 0, off = 0;
    bool ok = true;
    lim[0] = 0;
#pragma unroll
    for (int k = 1; k <= 15; k++) {
        const uint32_t c = adj[k * stride];
        if (code + c > (1u << k)) ok = false;
        lim[k] = (code + c) << (15 - k);
        adj[k * stride] = (uint16_t) off;           // insertion cursor of length k
        off += c;
        code = (code + c) << 1;
    }
    if (!ok) return false;
    for (int s = 0; s < n; s++) {
        const int l = perm[s * stride] >> 12;
        if (l) {
            const uint32_t j = adj[l * stride];
            adj[l * stride] = (uint16_t) (j + 1);
            perm[j * stride] = (uint16_t) ((perm[j * stride] & 0xF000u) | (uint32_t) s);
        }
    }
    // cursors -> adj[k] = (symbols shorter than k) - first code of length k
    uint32_t prev_off = 0;
#pragma unroll
    for (int k = 1; k <= 15; k++) {
        const uint32_t next = adj[k * stride];
        const uint32_t first = (k == 1) ? 0u : (lim[k - 1] >> (15 - k));
        adj[k * stride] = (uint16_t) (int16_t) ((int32_t) prev_off - (int32_t) first);
        prev_off = next;
    }
    return true;
}

WT_HD void wt_inf_emit(WtInflate &z, const WtInfMem &m, uint32_t b) {
    m.ring[(z.out_pos & (WT_INF_RING - 1)) * m.stride] = (uint8_t) b;
    z.acc |= b << (8 * (z.out_pos & 3u));
    z.out_pos++;
    if ((z.out_pos & 3u) == 0) {
        *(uint32_t *) (z.out + z.out_pos - 4) = z.acc;
        z.acc = 0;
    }
}

// Block header (RFC 1951 3.2.3 - 3.2.7): sets up the tables of a fixed / dynamic block or the byte
// count of a stored one.
WT_HD void wt_inf_block(WtInflate &z, const WtInfMem &m) {
    const int S = m.stride;
    wt_inf_refill(z);
    z.last = wt_inf_bits(z, 1) != 0;
    const uint32_t type = wt_inf_bits(z, 2);
    if (type == 0) {
        wt_inf_bits(z, z.bc & 7);                   // to the byte boundary (bc counts from it)
        wt_inf_refill(z);
        const uint32_t len = wt_inf_bits(z, 16), nlen = wt_inf_bits(z, 16);
        if ((len ^ nlen) != 0xFFFFu) { wt_inf_fail(z, WT_INF_ERR_BLOCK); return; }
        z.stored_rem = len;
        z.st = len ? WT_INF_ST_STORED : (z.last ? WT_INF_ST_DONE : WT_INF_ST_BLOCK);
        return;
    }
    if (type == 3) { wt_inf_fail(z, WT_INF_ERR_BLOCK); return; }
    int hlit = 288, hdist = 30;
    for (int s = 0; s < WT_INF_PERM; s++) m.perm[s * S] = 0;
    if (type == 1) {
        for (int s = 0; s < 288; s++) m.perm[s * S] = (uint16_t) ((s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8) << 12);
        for (int s = 0; s < 30; s++) m.perm[(WT_INF_DBASE + s) * S] = (uint16_t) (5 << 12);
    } else {
        hlit = (int) wt_inf_bits(z, 5) + 257;
        hdist = (int) wt_inf_bits(z, 5) + 1;
        const int hclen = (int) wt_inf_bits(z, 4) + 4;
        if (hlit > 286 || hdist > 30) { wt_inf_fail(z, WT_INF_ERR_CODE); return; }
        // the code-length code: 19 lengths of 3 bits, packed 3 bits per symbol
        uint64_t cl = 0;
        wt_inf_refill(z);
#pragma unroll
        for (int i = 0; i < 19; i++) {
            const int order = i == 0 ? 16 : i == 1 ? 17 : i == 2 ? 18 : i == 3 ? 0 : (i & 1) ? (8 - ((i - 3) >> 1)) : (7 + ((i - 2) >> 1));
            if (i == 10) wt_inf_refill(z);
            if (i < hclen) cl |= (uint64_t) wt_inf_bits(z, 3) << (3 * order);
        }
        uint64_t cnt = 0;                           // 8-bit counters per length
#pragma unroll
        for (int s = 0; s < 19; s++) cnt += 1ull << (8 * (int) ((cl >> (3 * s)) & 7u));
        uint32_t clim[8];
        int32_t cadj[8];
        uint64_t cur = 0;                           // insertion cursors, 8 bits per length
        {
            uint32_t code = 0, off = 0;
            bool ok = true;
            clim[0] = 0; cadj[0] = 0;
#pragma unroll
            for (int k = 1; k <= 7; k++) {
                const uint32_t c = (uint32_t) (cnt >> (8 * k)) & 255u;
                if (code + c > (1u << k)) ok = false;
                clim[k] = (code + c) << (7 - k);
                cadj[k] = (int32_t) off - (int32_t) code;
                cur |= (uint64_t) off << (8 * k);
                off += c;
                code = (code + c) << 1;
            }
            if (!ok) { wt_inf_fail(z, WT_INF_ERR_CODE); return; }
        }
        // its symbols sorted by code: aux[0..18] (the block's own adj[] is built afterwards)
#pragma unroll
        for (int s = 0; s < 19; s++) {
            const int l = (int) ((cl >> (3 * s)) & 7u);
            if (l) {
                const uint32_t j = (uint32_t) (cur >> (8 * l)) & 255u;
                cur += 1ull << (8 * l);
                m.aux[j * S] = (uint16_t) s;
            }
        }
        const int total = hlit + hdist;
        int i = 0;
        uint32_t prev = 0;
        while (i < total) {
            wt_inf_refill(z);
            const uint32_t w7 = wt_inf_bitrev15((uint32_t) z.bb & 0x7Fu) >> 8;     // 7-bit window, first bit on top
            int len = 1;
#pragma unroll
            for (int k = 1; k <= 6; k++) len += (w7 >= clim[k]) ? 1 : 0;
            if (w7 >= clim[7]) { wt_inf_fail(z, WT_INF_ERR_SYMBOL); return; }
            int32_t a = 0;
#pragma unroll
            for (int k = 1; k <= 7; k++) a = (len == k) ? cadj[k] : a;
            const uint32_t sym = m.aux[(a + (int32_t) (w7 >> (7 - len))) * S];
            z.bb >>= len; z.bc -= len; z.bits_left -= len;
            uint32_t rep = 1, val = sym;
            if (sym == 16) {
                if (i == 0) { wt_inf_fail(z, WT_INF_ERR_CODE); return; }
                rep = 3 + wt_inf_bits(z, 2); val = prev;
            } else if (sym == 17) {
                rep = 3 + wt_inf_bits(z, 3); val = 0;
            } else if (sym == 18) {
                rep = 11 + wt_inf_bits(z, 7); val = 0;
            }
            if (i + (int) rep > total) { wt_inf_fail(z, WT_INF_ERR_CODE); return; }
            for (uint32_t r = 0; r < rep; r++, i++) {
                const int pos = i < hlit ? i : WT_INF_DBASE + (i - hlit);
                m.perm[pos * S] = (uint16_t) (val << 12);
            }
            prev = val;
        }
        if ((m.perm[256 * S] >> 12) == 0) { wt_inf_fail(z, WT_INF_ERR_CODE); return; }    // no end-of-block code
    }
    if (!wt_inf_build(z.llim, m.aux, m.perm, 288, S) ||
        !wt_inf_build(z.dlim, m.aux + 16 * S, m.perm + WT_INF_DBASE * S, 32, S)) { wt_inf_fail(z, WT_INF_ERR_CODE); return; }
    z.st = WT_INF_ST_SYM;
}

// One step of the state machine.  Returns false once the lane has nothing more to do.
WT_HD bool wt_inf_step(WtInflate &z, const WtInfMem &m) {
    if (z.st >= WT_INF_ST_DONE) return false;
    const int S = m.stride;
    if (z.copy_rem) {
#pragma unroll
        for (int q = 0; q < WT_INF_COPY; q++) {
            if (z.copy_rem) {
                uint32_t b;
                if (z.copy_dist <= WT_INF_RING) b = m.ring[((z.out_pos - z.copy_dist) & (WT_INF_RING - 1)) * S];
                else b = z.out[z.out_pos - z.copy_dist];        // flushed long ago (>= 64 bytes back)
                wt_inf_emit(z, m, b);
                z.copy_rem--;
            }
        }
        if (z.copy_rem) return true;
    }
    if (z.bits_left < 0) { wt_inf_fail(z, WT_INF_ERR_INPUT); return false; }
    if (z.st == WT_INF_ST_SYM) {
        wt_inf_refill(z);
        const int sym = wt_inf_decode(z, z.llim, m.aux, m.perm, S);
        if (sym < 0) { wt_inf_fail(z, WT_INF_ERR_SYMBOL); return false; }
        if (sym < 256) {
            if (z.out_pos >= z.out_cap) { wt_inf_fail(z, WT_INF_ERR_SPACE); return false; }
            wt_inf_emit(z, m, (uint32_t) sym);
        } else if (sym == 256) {
            z.st = z.last ? WT_INF_ST_DONE : WT_INF_ST_BLOCK;
        } else {
            const uint32_t ls = (uint32_t) sym - 257u;
            if (ls > 28u) { wt_inf_fail(z, WT_INF_ERR_SYMBOL); return false; }
            uint32_t len;
            if (ls < 8u) len = ls + 3u;
            else if (ls == 28u) len = 258u;
            else {
                const uint32_t e = (ls - 4u) >> 2;
                len = ((4u + (ls & 3u)) << e) + 3u + wt_inf_bits(z, (int) e);
            }
            wt_inf_refill(z);
            const int ds = wt_inf_decode(z, z.dlim, m.aux + 16 * S, m.perm + WT_INF_DBASE * S, S);
            if (ds < 0 || ds > 29) { wt_inf_fail(z, WT_INF_ERR_SYMBOL); return false; }
            uint32_t dist;
            if (ds < 4) dist = (uint32_t) ds + 1u;
            else {
                const uint32_t e = ((uint32_t) ds >> 1) - 1u;
                dist = ((2u + ((uint32_t) ds & 1u)) << e) + 1u + wt_inf_bits(z, (int) e);
            }
            if (dist > z.out_pos) { wt_inf_fail(z, WT_INF_ERR_DIST); return false; }
            if (z.out_pos + len > z.out_cap) { wt_inf_fail(z, WT_INF_ERR_SPACE); return false; }
            z.copy_rem = len; z.copy_dist = dist;
        }
        return true;
    }
    if (z.st == WT_INF_ST_STORED) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (z.stored_rem) {
                if (z.out_pos >= z.out_cap) { wt_inf_fail(z, WT_INF_ERR_SPACE); return false; }
                if (q == 0 || q == 2) wt_inf_refill(z);
                wt_inf_emit(z, m, wt_inf_bits(z, 8));
                z.stored_rem--;
            }
        }
        if (!z.stored_rem) z.st = z.last ? WT_INF_ST_DONE : WT_INF_ST_BLOCK;
        return true;
    }
    if (z.st == WT_INF_ST_BLOCK) {
        wt_inf_block(z, m);
        return z.st < WT_INF_ST_DONE || z.st == WT_INF_ST_DONE;
    }
    // WT_INF_ST_ZHDR: RFC 1950 -- CMF, FLG
    wt_inf_refill(z);
    const uint32_t cmf = wt_inf_bits(z, 8), flg = wt_inf_bits(z, 8);
    if ((cmf & 15u) != 8u || (cmf >> 4) > 7u || ((cmf << 8) | flg) % 31u != 0u || (flg & 32u)) { wt_inf_fail(z, WT_INF_ERR_HEADER); return false; }
    z.st = WT_INF_ST_BLOCK;
    return true;
}

// Flushes the last partial dword; returns the number of bytes produced or -(error code).
WT_HD int64_t wt_inf_finish(WtInflate &z) {
    if (z.st == WT_INF_ST_DONE && z.bits_left < 0) { z.st = WT_INF_ST_ERR; z.err = WT_INF_ERR_INPUT; }
    if (z.st != WT_INF_ST_DONE) return -(int64_t) (z.err ? z.err : WT_INF_ERR_INPUT);
    if (z.out_pos & 3u) *(uint32_t *) (z.out + (z.out_pos & ~3u)) = z.acc;
    return (int64_t) z.out_pos;
}

#endif  // WT_INFLATE_H_
