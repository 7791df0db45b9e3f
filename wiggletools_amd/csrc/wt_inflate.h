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
//     and 2 x 15 adj values live in REGISTERS
//     (statically indexed arrays), `perm` (the 286 literal / length symbols sorted by code) in LDS --
//     ONE dependent LDS read per symbol -- the 30 distance symbols in 5 registers: 640 bytes of LDS per lane,
//     so that FOUR wavefronts (one per SIMD) share a CU's 160 KB.
//   * the code lengths of a dynamic block are parked in the unused top 4 bits of perm[] while
//     the table is built in place, the per-length counters borrow the ring's 16 slots (whose history waits
//     in registers); the 19-symbol code-length code lives in 64-bit registers.
//   * LZ77 history: the last 16 output DWORDS of every lane are kept in an LDS ring; a match with a
//     distance <= 60 (the bulk of them in 12-byte record data) reads three of them -- one LDS round
//     trip for up to 8 bytes -- and never touches global memory; longer ones read the lane's own
//     earlier output back.  Output is gathered to dwords in a register and stored whole.
//   * the input is prefetched 16 bytes (~16 symbols) ahead into registers: the loop never waits for
//     a load it has just issued;
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

#if defined(__HIP_DEVICE_COMPILE__)
// Explicit address spaces: with generic pointers hipcc folded "ring byte or far byte" into ONE flat_load per
// byte followed by s_waitcnt vmcnt(0) lgkmcnt(0) -- every copied byte waited for every outstanding store.
#define WT_AS_GLOBAL __attribute__((address_space(1)))
#define WT_AS_LDS __attribute__((address_space(3)))
#else
#define WT_AS_GLOBAL
#define WT_AS_LDS
#endif

// hipcc rewrites a chain `a = c_k ? t[k + 1] : a` over a statically indexed table into ONE variable-index load
// t[len] -- and a variable index keeps the whole decoder state in scratch memory.  An empty asm on the running
// value hides the chain from that rewrite (no instruction is emitted).
#if defined(__HIP_DEVICE_COMPILE__)
#define WT_INF_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define WT_INF_OPAQUE(x) (void) 0
#endif

#define WT_INF_PERM 288         // literal / length symbols sorted by code (the 30 distance symbols live in registers)
#define WT_INF_RING 16          // dwords of LZ77 history per lane
#define WT_INF_RING_DIST 56     // matches up to this distance are served from the ring
#define WT_INF_COPY 8           // match bytes copied per state machine step

// bytes of "LDS" one lane needs: 640 -> 40 KB per wavefront, four wavefronts per CU (one per SIMD)
#define WT_INF_LANE_BYTES (WT_INF_PERM * 2 + WT_INF_RING * 4)

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
    WT_AS_LDS uint16_t *perm;   // WT_INF_PERM entries
    WT_AS_LDS uint32_t *ring;   // WT_INF_RING dwords (borrowed as 16 counters while a block's tables are built)
    int stride;                 // elements between consecutive entries of this lane
};

struct WtInfQuad { uint32_t x, y, z, w; };

struct WtInflate {
    // input: 16-byte chunks at 16-byte aligned addresses; `cur` is being consumed, `nxt` is in flight
    const WT_AS_GLOBAL uint32_t *in_w;  // aligned base
    uint32_t in_chunk;          // next chunk to fetch
    uint32_t in_chunks;         // chunks that may be read
    WtInfQuad cur, nxt;
    uint32_t qi;                // words of `cur` already taken (the queue is shifted, cur.x is the next one)
    bool nxt_empty;             // `nxt` was moved into `cur` and its successor has not arrived yet
    uint32_t words;             // words taken so far
    uint32_t head_bits;         // bits of the first word that precede the stream
    uint32_t n_bytes;
    uint64_t bb;                // bit buffer, LSB first
    int32_t bc;                 // valid bits in bb
    // output
    WT_AS_GLOBAL uint32_t *out; // 4-byte aligned
    uint32_t out_pos, out_cap;
    uint32_t acc;               // bytes of the current (partial) output dword
    // state
    int32_t st, err;
    uint32_t copy_rem, copy_dist;
    uint32_t stored_rem;
    bool last, raw;             // last block seen; raw deflate (no zlib wrapper)
    uint32_t llim[16], dlim[16];    // [1..15] used
    int32_t ladj[16], dadj[16];
    uint32_t dperm[5];          // the distance symbols sorted by code, 5 bits each, 6 per word
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

// Chunk i of the input (past the end: the last chunk again -- a lane that gets there has over-read and fails,
// see wt_inf_overread).  UNCONDITIONAL on purpose: a load under a branch is merged with the "no load" value by a
// register copy, and the copy waits for the load right where it was issued.
WT_HD WtInfQuad wt_inf_load(const WtInflate &z, uint32_t i) {
    const uint32_t last = z.in_chunks ? z.in_chunks - 1u : 0u;
    const WT_AS_GLOBAL uint32_t *p = z.in_w + 4 * (size_t) (i < last ? i : last);
    WtInfQuad q;
    q.x = p[0]; q.y = p[1]; q.z = p[2]; q.w = p[3];
    return q;
}

WT_HD void wt_inf_fail(WtInflate &z, int code) { z.st = WT_INF_ST_ERR; z.err = code; z.copy_rem = 0; }

// Bits consumed beyond the end of the stream?  (Checked when a chunk is fetched -- a stream that keeps decoding
// garbage past its end cannot run forever -- and at the end.)
WT_HD bool wt_inf_overread(const WtInflate &z) {
    const int64_t consumed = (int64_t) z.words * 32 - (int64_t) z.head_bits - (int64_t) z.bc;
    return consumed > (int64_t) z.n_bytes * 8;
}

// (The queue is SHIFTED, never indexed: one variable index into the state struct and hipcc keeps the whole
// struct in scratch memory.)  `nxt` is refilled by wt_inf_step's prefetch one step after it was moved into `cur`;
// the blocking load here serves the block-header code, which reads many words inside one step.
// HOT = true (the symbol loop): `nxt` is known to be there (a step takes at most two words, the prefetch lands one
// step after `nxt` was taken) -- no load, hence no wait, on this path.
template <bool HOT>
WT_HD uint32_t wt_inf_word(WtInflate &z) {
    const uint32_t w = z.cur.x;
    z.cur.x = z.cur.y; z.cur.y = z.cur.z; z.cur.z = z.cur.w;
    z.qi++;
    z.words++;
    if (z.qi == 4) {
        if (!HOT && z.nxt_empty) { z.nxt = wt_inf_load(z, z.in_chunk); z.in_chunk++; }
        z.cur = z.nxt;
        z.qi = 0;
        z.nxt_empty = true;
    }
    return w;
}

// Starts a stream of `n_bytes` at `src` (any alignment; the 16-byte aligned chunks around it must be readable
// inside the same allocation -- up to 15 bytes before and 31 after) writing at most `cap` bytes to `dst` (4-byte
// aligned, cap rounded up to a multiple of 4 must be writable).
WT_HD void wt_inf_begin(WtInflate &z, const uint8_t *src, uint32_t n_bytes, uint8_t *dst, uint32_t cap, bool raw_deflate) {
    const uintptr_t a = (uintptr_t) src;
    const uint32_t mis = (uint32_t) (a & 15u);
    z.in_w = (const WT_AS_GLOBAL uint32_t *) (a - mis);
    z.in_chunks = (mis + n_bytes + 15u) >> 4;
    z.n_bytes = n_bytes;
    z.cur = wt_inf_load(z, 0);
    z.nxt = wt_inf_load(z, 1);
    z.in_chunk = 2;
    z.qi = 0;
    z.nxt_empty = false;
    z.words = 0;
    for (uint32_t k = 0; k < (mis >> 2); k++) (void) wt_inf_word<false>(z);       // whole words before the stream are skipped
    z.words = 0;
    const uint32_t w0 = wt_inf_word<false>(z);
    z.head_bits = 8u * (mis & 3u);
    z.bb = (uint64_t) (w0 >> z.head_bits);
    z.bc = 32 - (int32_t) z.head_bits;
    z.out = (WT_AS_GLOBAL uint32_t *) dst; z.out_pos = 0; z.out_cap = cap; z.acc = 0;
    z.st = raw_deflate ? WT_INF_ST_BLOCK : WT_INF_ST_ZHDR;
    z.err = WT_INF_OK;
    z.copy_rem = z.copy_dist = 0; z.stored_rem = 0;
    z.last = false; z.raw = raw_deflate;
#pragma unroll
    for (int k = 0; k < 16; k++) { z.llim[k] = 0; z.dlim[k] = 0; z.ladj[k] = 0; z.dadj[k] = 0; }
#pragma unroll
    for (int k = 0; k < 5; k++) z.dperm[k] = 0;
}

// After this bc >= 33 (past the end of the input the last chunk repeats; wt_inf_overread catches it).
template <bool HOT = false>
WT_HD void wt_inf_refill(WtInflate &z) {
    if (z.bc <= 32) {
        z.bb |= (uint64_t) wt_inf_word<HOT>(z) << z.bc;
        z.bc += 32;
    }
}

WT_HD uint32_t wt_inf_bits(WtInflate &z, int n) {      // n <= 32, after a refill guaranteeing enough bits
    const uint32_t v = (uint32_t) (z.bb & ((1ull << n) - 1ull));
    z.bb >>= n; z.bc -= n;
    return v;
}

// Length and sorted-symbol index of the code word on top of the stream: lim[1..15] / adj[1..15] registers.
// Returns false on a pattern no code word matches.
WT_HD bool wt_inf_code(WtInflate &z, const uint32_t (&lim)[16], const int32_t (&adj)[16], int32_t &idx) {
    const uint32_t w = wt_inf_bitrev15((uint32_t) z.bb & 0x7FFFu);
    int len = 1;
    int32_t a = adj[1];
#pragma unroll
    for (int k = 1; k <= 14; k++) {
        const bool ge = w >= lim[k];
        len += ge ? 1 : 0;
        a = ge ? adj[k + 1] : a;
        WT_INF_OPAQUE(a);
    }
    idx = a + (int32_t) (w >> (15 - len));
    z.bb >>= len; z.bc -= len;
    return w < lim[15];
}

// lim[] / adj[] of a canonical code from the 16 per-length counts in cnt[] (lane memory); leaves the insertion
// cursor of every length in cnt[].  False: over-subscribed.
WT_HD bool wt_inf_limits(uint32_t (&lim)[16], int32_t (&adj)[16], WT_AS_LDS uint32_t *cnt, int stride) {
    uint32_t code = 0, off = 0;
    bool ok = true;
    lim[0] = 0; adj[0] = 0;
#pragma unroll
    for (int k = 1; k <= 15; k++) {
        const uint32_t c = cnt[k * stride];
        if (code + c > (1u << k)) ok = false;
        lim[k] = (code + c) << (15 - k);
        adj[k] = (int32_t) off - (int32_t) code;    // (symbols shorter than k) - first code of length k
        cnt[k * stride] = off;                      // insertion cursor of length k
        off += c;
        code = (code + c) << 1;
    }
    return ok;
}

// Appends the low n (1..8) bytes of `bytes` to the output: full dwords go to global memory and to the ring,
// the partial one stays in `acc` (it reaches the ring when a match is about to read it).
WT_HD void wt_inf_put(WtInflate &z, const WtInfMem &m, uint64_t bytes, uint32_t n) {
    if (n < 8) bytes &= (1ull << (8 * n)) - 1ull;
    const uint32_t sh = (z.out_pos & 3u) * 8u;
    const uint32_t d = z.out_pos >> 2;
    const uint32_t e0 = z.acc | (uint32_t) (bytes << sh);
    const uint64_t t = bytes >> (32u - sh);
    const uint32_t e1 = (uint32_t) t, e2 = (uint32_t) (t >> 32);
    const uint32_t full = ((z.out_pos & 3u) + n) >> 2;
    uint32_t acc = e0;
    if (full >= 1) {
        z.out[d] = e0;
        m.ring[(d & (WT_INF_RING - 1)) * m.stride] = e0;
        acc = e1;
        if (full >= 2) {
            z.out[d + 1] = e1;
            m.ring[((d + 1) & (WT_INF_RING - 1)) * m.stride] = e1;
            acc = e2;
        }
    }
    z.acc = acc;
    z.out_pos += n;
}

// One literal: the common case of wt_inf_put.
WT_HD void wt_inf_put1(WtInflate &z, const WtInfMem &m, uint32_t b) {
    const uint32_t acc = z.acc | (b << ((z.out_pos & 3u) * 8u));
    z.out_pos++;
    z.acc = acc;
    if ((z.out_pos & 3u) == 0) {
        const uint32_t d = (z.out_pos >> 2) - 1u;
        z.out[d] = acc;
        m.ring[(d & (WT_INF_RING - 1)) * m.stride] = acc;
        z.acc = 0;
    }
}

// Block header (RFC 1951 3.2.3 - 3.2.7): sets up the tables of a fixed / dynamic block or the byte count of a
// stored one.  Executed once or twice per stream: compactness matters more than speed here.
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
    // the 16 ring slots become the per-length counters / cursors of the table construction: the history they hold
    // (live when this is not the stream's first block) waits in registers
    uint32_t saved[WT_INF_RING];
#pragma unroll
    for (int k = 0; k < WT_INF_RING; k++) saved[k] = m.ring[k * S];
    WT_AS_LDS uint32_t *cnt = m.ring;
    uint64_t dl0 = 0, dl1 = 0;                      // lengths of the distance symbols, 4 bits each (16 per word)
    int hlit = 288, hdist = 30;
    for (int s = 0; s < WT_INF_PERM; s++) m.perm[s * S] = 0;
    if (type == 1) {
        for (int s = 0; s < 288; s++) m.perm[s * S] = (uint16_t) ((s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8) << 12);
        dl0 = 0x5555555555555555ull; dl1 = 0x0055555555555555ull;       // 30 codes of 5 bits
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
        uint64_t cc = 0;                            // 8-bit counters per length
#pragma unroll
        for (int s = 0; s < 19; s++) cc += 1ull << (8 * (int) ((cl >> (3 * s)) & 7u));
        uint32_t clim[8];
        int32_t cadj[8];
        uint64_t cur = 0;                           // insertion cursors, 8 bits per length
        {
            uint32_t code = 0, off = 0;
            bool ok = true;
            clim[0] = 0; cadj[0] = 0;
#pragma unroll
            for (int k = 1; k <= 7; k++) {
                const uint32_t c = (uint32_t) (cc >> (8 * k)) & 255u;
                if (code + c > (1u << k)) ok = false;
                clim[k] = (code + c) << (7 - k);
                cadj[k] = (int32_t) off - (int32_t) code;
                cur |= (uint64_t) off << (8 * k);
                off += c;
                code = (code + c) << 1;
            }
            if (!ok) { wt_inf_fail(z, WT_INF_ERR_CODE); return; }
        }
        // its symbols sorted by code, 5 bits each: 12 in cp0, the rest in cp1
        uint64_t cp0 = 0, cp1 = 0;
#pragma unroll
        for (int s = 0; s < 19; s++) {
            const int l = (int) ((cl >> (3 * s)) & 7u);
            if (l) {
                const uint32_t j = (uint32_t) (cur >> (8 * l)) & 255u;
                cur += 1ull << (8 * l);
                if (j < 12u) cp0 |= (uint64_t) s << (5u * j); else cp1 |= (uint64_t) s << (5u * (j - 12u));
            }
        }
        const int total = hlit + hdist;
        int i = 0;
        uint32_t prev = 0;
        while (i < total) {
            wt_inf_refill(z);
            const uint32_t w7 = wt_inf_bitrev15((uint32_t) z.bb & 0x7Fu) >> 8;     // 7-bit window, first bit on top
            int len = 1;
            int32_t a = cadj[1];
#pragma unroll
            for (int k = 1; k <= 6; k++) {
                const bool ge = w7 >= clim[k];
                len += ge ? 1 : 0;
                a = ge ? cadj[k + 1] : a;
                WT_INF_OPAQUE(a);
            }
            if (w7 >= clim[7]) { wt_inf_fail(z, WT_INF_ERR_SYMBOL); return; }
            const uint32_t j = (uint32_t) (a + (int32_t) (w7 >> (7 - len)));
            const uint32_t sym = (uint32_t) ((j < 12u ? cp0 >> (5u * j) : cp1 >> (5u * (j - 12u))) & 31u);
            z.bb >>= len; z.bc -= len;
            uint32_t rep = 1, val = sym;
            if (sym == 16) {
                if (i == 0) { wt_inf_fail(z, WT_INF_ERR_CODE); return; }
                rep = 3 + wt_inf_bits(z, 2); val = prev;
            } else if (sym == 17) {
                rep = 3 + wt_inf_bits(z, 3); val = 0;
            } else if (sym == 18) {
                rep = 11 + wt_inf_bits(z, 7); val = 0;
            } else if (sym > 18) {
                wt_inf_fail(z, WT_INF_ERR_SYMBOL); return;
            }
            if (i + (int) rep > total) { wt_inf_fail(z, WT_INF_ERR_CODE); return; }
            for (uint32_t r = 0; r < rep; r++, i++) {
                if (i < hlit) m.perm[i * S] = (uint16_t) (val << 12);
                else {
                    const uint32_t q = (uint32_t) (i - hlit);
                    if (q < 16u) dl0 |= (uint64_t) val << (4u * q); else dl1 |= (uint64_t) val << (4u * (q - 16u));
                }
            }
            prev = val;
        }
        if ((m.perm[256 * S] >> 12) == 0) { wt_inf_fail(z, WT_INF_ERR_CODE); return; }    // no end-of-block code
    }
    // literal / length table: counts -> limits -> symbols sorted by code, in place (lengths in the top 4 bits)
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 16; k++) cnt[k * S] = 0;
    for (int s = 0; s < WT_INF_PERM; s++) {
        const int l = m.perm[s * S] >> 12;
        cnt[l * S] = cnt[l * S] + 1u;
    }
    ok = wt_inf_limits(z.llim, z.ladj, cnt, S) && ok;
    for (int s = 0; s < WT_INF_PERM; s++) {
        const int l = m.perm[s * S] >> 12;
        if (l) {
            const uint32_t j = cnt[l * S];
            cnt[l * S] = j + 1u;
            if (j < (uint32_t) WT_INF_PERM) m.perm[j * S] = (uint16_t) ((m.perm[j * S] & 0xF000u) | (uint32_t) s);
        }
    }
    // distance table: 30 symbols, sorted into 5 registers
#pragma unroll
    for (int k = 0; k < 16; k++) cnt[k * S] = 0;
    for (int s = 0; s < 30; s++) {
        const uint32_t l = (uint32_t) ((s < 16 ? dl0 >> (4 * s) : dl1 >> (4 * (s - 16))) & 15u);
        cnt[l * S] = cnt[l * S] + 1u;
    }
    ok = wt_inf_limits(z.dlim, z.dadj, cnt, S) && ok;
    uint32_t dp[5] = {0u, 0u, 0u, 0u, 0u};
    for (int s = 0; s < 30; s++) {
        const uint32_t l = (uint32_t) ((s < 16 ? dl0 >> (4 * s) : dl1 >> (4 * (s - 16))) & 15u);
        if (l) {
            const uint32_t j = cnt[l * S];
            cnt[l * S] = j + 1u;
            const uint32_t word = j / 6u, sh = 5u * (j - 6u * word);
#pragma unroll
            for (int q = 0; q < 5; q++) dp[q] |= (word == (uint32_t) q) ? ((uint32_t) s << sh) : 0u;
        }
    }
#pragma unroll
    for (int q = 0; q < 5; q++) z.dperm[q] = dp[q];
#pragma unroll
    for (int k = 0; k < WT_INF_RING; k++) m.ring[k * S] = saved[k];
    if (!ok) { wt_inf_fail(z, WT_INF_ERR_CODE); return; }
    z.st = WT_INF_ST_SYM;
}

WT_HD bool wt_inf_step_body(WtInflate &z, const WtInfMem &m);
WT_HD bool wt_inf_step_cold(WtInflate &z, const WtInfMem &m);

// One step of the state machine.  Returns false once the lane has nothing more to do.
// The input prefetch brackets the step: the 16-byte load of the chunk after next is ISSUED before the step's
// work and its registers are only touched after it -- a load whose result is merged into the loop-carried state
// right where it was issued costs a full memory round trip per chunk (hipcc copies it at the end of the
// branch; seen in the ISA as s_waitcnt vmcnt(0) three instructions after the load).
WT_HD bool wt_inf_step(WtInflate &z, const WtInfMem &m) {
    if (z.st >= WT_INF_ST_DONE) return false;
    const bool need = z.nxt_empty;
    const uint32_t chunk0 = z.in_chunk;
    WtInfQuad tmp;                  // (only read under `need`)
    if (need) {
        if (wt_inf_overread(z)) { wt_inf_fail(z, WT_INF_ERR_INPUT); return false; }
        tmp = wt_inf_load(z, chunk0);
    }
    const bool more = wt_inf_step_body(z, m);
    if (need && z.nxt_empty && z.in_chunk == chunk0) { z.nxt = tmp; z.in_chunk = chunk0 + 1; z.nxt_empty = false; }
    return more;
}

WT_HD bool wt_inf_step_body(WtInflate &z, const WtInfMem &m) {
    const int S = m.stride;
    if (z.copy_rem) {
        // up to 8 bytes of the match: three dwords around the source, one LDS round trip
        const uint32_t src = z.out_pos - z.copy_dist;
        const uint32_t d0 = src >> 2;
        uint32_t r0, r1, r2;
        if (z.copy_dist <= WT_INF_RING_DIST) {
            m.ring[((z.out_pos >> 2) & (WT_INF_RING - 1)) * S] = z.acc;        // the partial dword may be part of the source
            r0 = m.ring[(d0 & (WT_INF_RING - 1)) * S];
            r1 = m.ring[((d0 + 1) & (WT_INF_RING - 1)) * S];
            r2 = m.ring[((d0 + 2) & (WT_INF_RING - 1)) * S];
        } else {                                    // flushed long ago (full dwords are stored at once)
            r0 = z.out[d0]; r1 = z.out[d0 + 1]; r2 = z.out[d0 + 2];
        }
        const uint32_t sh = (src & 3u) * 8u;
        uint64_t w = ((uint64_t) r0 | ((uint64_t) r1 << 32)) >> sh;
        if (sh) w |= (uint64_t) r2 << (64u - sh);
        if (z.copy_dist < 8u) {                     // overlapping copy: the first `dist` bytes repeat
            const uint32_t s8 = 8u * z.copy_dist;
            w &= (1ull << s8) - 1ull;
            w |= w << s8;
            if (2u * s8 < 64u) w |= w << (2u * s8);
            if (4u * s8 < 64u) w |= w << (4u * s8);
        }
        const uint32_t n = z.copy_rem < (uint32_t) WT_INF_COPY ? z.copy_rem : (uint32_t) WT_INF_COPY;
        wt_inf_put(z, m, w, n);
        z.copy_rem -= n;
        if (z.copy_rem) return true;
    }
    if (z.st == WT_INF_ST_SYM) {
        wt_inf_refill<true>(z);
        int32_t idx;
        if (!wt_inf_code(z, z.llim, z.ladj, idx)) { wt_inf_fail(z, WT_INF_ERR_SYMBOL); return false; }
        const uint32_t sym = m.perm[idx * S] & 0x1FFu;
        if (sym < 256u) {
            if (z.out_pos >= z.out_cap) { wt_inf_fail(z, WT_INF_ERR_SPACE); return false; }
            wt_inf_put1(z, m, sym);
        } else if (sym == 256u) {
            z.st = z.last ? WT_INF_ST_DONE : WT_INF_ST_BLOCK;
        } else {
            const uint32_t ls = sym - 257u;
            if (ls > 28u) { wt_inf_fail(z, WT_INF_ERR_SYMBOL); return false; }
            uint32_t len;
            if (ls < 8u) len = ls + 3u;
            else if (ls == 28u) len = 258u;
            else {
                const uint32_t e = (ls - 4u) >> 2;
                len = ((4u + (ls & 3u)) << e) + 3u + wt_inf_bits(z, (int) e);
            }
            wt_inf_refill<true>(z);
            int32_t di;
            if (!wt_inf_code(z, z.dlim, z.dadj, di) || (uint32_t) di > 29u) { wt_inf_fail(z, WT_INF_ERR_SYMBOL); return false; }
            const uint32_t word = (uint32_t) di / 6u, dsh = 5u * ((uint32_t) di - 6u * word);
            uint32_t dw = z.dperm[0];
            dw = word == 1u ? z.dperm[1] : dw; WT_INF_OPAQUE(dw);
            dw = word == 2u ? z.dperm[2] : dw; WT_INF_OPAQUE(dw);
            dw = word == 3u ? z.dperm[3] : dw; WT_INF_OPAQUE(dw);
            dw = word == 4u ? z.dperm[4] : dw; WT_INF_OPAQUE(dw);
            const uint32_t ds = (dw >> dsh) & 31u;
            if (ds > 29u) { wt_inf_fail(z, WT_INF_ERR_SYMBOL); return false; }
            uint32_t dist;
            if (ds < 4u) dist = ds + 1u;
            else {
                const uint32_t e = (ds >> 1) - 1u;
                dist = ((2u + (ds & 1u)) << e) + 1u + wt_inf_bits(z, (int) e);
            }
            if (dist > z.out_pos) { wt_inf_fail(z, WT_INF_ERR_DIST); return false; }
            if (z.out_pos + len > z.out_cap) { wt_inf_fail(z, WT_INF_ERR_SPACE); return false; }
            z.copy_rem = len; z.copy_dist = dist;
        }
        return true;
    }
    // every other state: rare, and it leaves `nxt` filled -- the symbol loop relies on it (wt_inf_word<true>)
    const bool more = wt_inf_step_cold(z, m);
    if (z.nxt_empty) { z.nxt = wt_inf_load(z, z.in_chunk); z.in_chunk++; z.nxt_empty = false; }
    return more;
}

WT_HD bool wt_inf_step_cold(WtInflate &z, const WtInfMem &m) {
    if (z.st == WT_INF_ST_STORED) {
        wt_inf_refill(z);
        const uint32_t n = z.stored_rem < 4u ? z.stored_rem : 4u;
        if (z.out_pos + n > z.out_cap) { wt_inf_fail(z, WT_INF_ERR_SPACE); return false; }
        wt_inf_put(z, m, (uint64_t) wt_inf_bits(z, 8 * (int) n), n);
        z.stored_rem -= n;
        if (!z.stored_rem) z.st = z.last ? WT_INF_ST_DONE : WT_INF_ST_BLOCK;
        return true;
    }
    if (z.st == WT_INF_ST_BLOCK) {
        wt_inf_block(z, m);
        return z.st != WT_INF_ST_ERR;
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
    if (z.st == WT_INF_ST_DONE && wt_inf_overread(z)) { z.st = WT_INF_ST_ERR; z.err = WT_INF_ERR_INPUT; }
    if (z.st != WT_INF_ST_DONE) return -(int64_t) (z.err ? z.err : WT_INF_ERR_INPUT);
    if (z.out_pos & 3u) z.out[z.out_pos >> 2] = z.acc;
    return (int64_t) z.out_pos;
}

#endif  // WT_INFLATE_H_
