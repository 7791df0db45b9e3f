// wt_inflate.h -- DEFLATE / zlib (RFC 1950, 1951) decoder written for ONE GPU LANE PER STREAM.
//
// Why: a BigWig file is a list of independent zlib streams ("sections" of <= a few thousand items,
// reference src/bigWiggleReader.c:52-83 reads them through libBigWig, which inflates every one on the
// host).  The file leg of the engine was bound by exactly that host inflate.  Sections are
// independent, so here every lane of a wavefront inflates its own section: 64 streams per wavefront,
// tens of thousands in flight per GPU, no cross-lane communication at all.
//
// Round 4 rewrite.  The round-3 decoder took 11.1 ms per 63 500 sections with ONE wavefront per
// SIMD, 42 % of its cycles waiting: a step was "one symbol OR 8 bytes of a match" (7 500 steps per section, each
// paying for the literal, the match and the copy path because 64 lanes are never in the same state), the input
// prefetch and every match beyond the LDS ring were loads whose results the same step consumed -- one global
// round trip (s_waitcnt vmcnt(0), which on gfx9 also waits for the step's own stores) in most steps.  Now:
//   * ONE STEP = ONE SYMBOL AND ITS FIRST 8 BYTES: decode a literal / length code; a length goes on to its
//     distance and the first <= 8 bytes of the copy in the same step; literal byte, match bytes and stored bytes
//     leave through ONE put.  5 400 steps per section instead of 7 500, and one copy of the code that is shared.
//   * NO LOAD IS CONSUMED INSIDE A STEP.  Steps run in ROUNDS of WT_INF_ROUND; global loads (the next 16 input
//     bytes, the source of a match beyond the ring) are issued whenever needed and LAND at the next round
//     boundary (wt_inf_land) -- the only place that waits for memory.  A lane whose match source has not landed
//     idles for the rest of its round (2 steps on average, ~7 % of the matches); a lane whose input queue is
//     about to run dry idles likewise (32 buffered bytes: never, for streams that are not adversarial).
//   * 320 BYTES OF LDS PER LANE instead of 640 -- TWO wavefronts per SIMD: the sorted symbol table holds the LOW
//     BYTE of a symbol only (288 B) -- inside one code length the sorted symbols ascend, so "literal or length
//     code" is index >= threshold[length], and the threshold rides in the register that already holds the
//     length's index adjustment -- and the ring is 8 dwords.  The table is built without a second array: the
//     code lengths of a dynamic block are DECODED TWICE (count, rewind the bit stream, place).
//   * conflict-free LDS: a lane's bytes live in its own bank (dword-interleaved layout).
//
// Decoding itself is table-free canonical Huffman by LIMITS: for a code with count[k] symbols of length k the
// left-aligned 15-bit window w of the stream has length
//         len = 1 + #{ k in 1..14 : w >= lim[k] },   lim[k] = (first[k] + count[k]) << (15 - k)
// and the symbol is perm[adj[len] + (w >> (15 - len))].  2 x 15 limits, 2 x 15 adjustments in REGISTERS
// (statically indexed), the 30 distance symbols in 5 registers.
//
// The same code compiles for the host (tests/emu, tests/test_bwdev.py: checked against zlib's own output on
// stored / fixed / dynamic streams of every compression level) -- `stride` is 1 there and the "LDS" arrays are
// plain memory.
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

#define WT_INF_PERM 288         // literal / length symbols sorted by code, low byte only
#ifndef WT_INF_RING
#define WT_INF_RING 8           // dwords of LZ77 history per lane (power of two, >= 8: its first 8 dwords double as
#endif                          // the 16 per-length counters while a block's tables are built)
#define WT_INF_RING_DIST(R) (4 * (R) - 4)  // matches up to this distance are served from a ring of R dwords
#define WT_INF_COPY 8           // match bytes copied per step
#define WT_INF_FAR 16           // bytes of a match beyond the ring fetched by one load (5 dwords)
#ifndef WT_INF_ROUND
#define WT_INF_ROUND 4          // steps between two landings
#endif

// bytes of "LDS" one lane needs with a ring of R dwords: 320 for R = 8 -> 20 KB per wavefront, EIGHT per CU
#define WT_INF_LANE_BYTES(R) (WT_INF_PERM + 4 * (R))

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

// states: SYM and STORED produce output inside wt_inf_step; ZHDR and BLOCK are worked off at a round boundary
enum { WT_INF_ST_SYM = 0, WT_INF_ST_STORED = 1, WT_INF_ST_ZHDR = 2, WT_INF_ST_BLOCK = 3, WT_INF_ST_DONE = 4, WT_INF_ST_ERR = 5 };

// A lane's two arrays.  Element layout (S = stride = lanes of the workgroup): dword d of a lane lies S dwords
// after its dword d - 1, i.e. every lane keeps to its own LDS bank whatever it indexes.
struct WtInfMem {
    WT_AS_LDS uint8_t *perm;    // the lane's dword 0 of the symbol table (byte j at ((j >> 2) * S) * 4 + (j & 3))
    WT_AS_LDS uint32_t *ring;   // the lane's dword 0 of the ring (dword k at k * S)
    int stride;
};

WT_HD uint32_t wt_inf_perm_get(const WtInfMem &m, uint32_t j) { return m.perm[(((j >> 2) * (uint32_t) m.stride) << 2) + (j & 3u)]; }
WT_HD void wt_inf_perm_set(const WtInfMem &m, uint32_t j, uint32_t v) { m.perm[(((j >> 2) * (uint32_t) m.stride) << 2) + (j & 3u)] = (uint8_t) v; }
// the 16 per-length counters / cursors of the table construction: 16-bit halves of the ring's first 8 dwords
WT_HD uint32_t wt_inf_cnt_get(const WtInfMem &m, uint32_t k) {
    return ((WT_AS_LDS uint16_t *) m.ring)[(((k >> 1) * (uint32_t) m.stride) << 1) + (k & 1u)];
}
WT_HD void wt_inf_cnt_set(const WtInfMem &m, uint32_t k, uint32_t v) {
    ((WT_AS_LDS uint16_t *) m.ring)[(((k >> 1) * (uint32_t) m.stride) << 1) + (k & 1u)] = (uint16_t) v;
}

struct WtInfQuad { uint32_t x, y, z, w; };

// The landing registers of a match source beyond the ring.  On the device the load is INLINE ASSEMBLY with its
// destination tied to these registers ("+v"): written as plain C++ the compiler gives the load fresh registers and
// copies them into the loop-carried ones right behind it -- s_waitcnt vmcnt(1) inside the step, i.e. the global
// round trip per step this design exists to avoid (86 % of the steps see a lane with such a match).  The compiler does
// not know about that load; wt_inf_land() waits for it explicitly.
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint32_t wt_inf_u32x4 __attribute__((ext_vector_type(4)));
struct WtInfFar { wt_inf_u32x4 q; uint32_t t; };
#else
struct WtInfFar { uint32_t q[4]; uint32_t t; };
#endif

template <int RING>
struct WtInflateT {
    // input: 16-byte chunks at 16-byte aligned addresses.  `cur` is being consumed (qn words left, shifted so that
    // cur.x is the next one), `q1` follows it, `pend` is in flight and lands at the next round boundary.
    const WT_AS_GLOBAL uint32_t *in_w;  // aligned base
    uint32_t mis;               // bytes between the aligned base and the stream
    uint32_t in_chunk;          // next chunk to fetch
    uint32_t in_chunks;         // chunks that may be read
    WtInfQuad cur, q1, pend;
    uint32_t qn;
    bool q1_valid, pend_valid;
    uint32_t wpos;              // words taken so far, counted from the aligned base
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
    // a match beyond the ring: WT_INF_FAR bytes of its source, loaded into fpend, landed into fq
    uint32_t fq[5];
    WtInfFar fpend;
    uint32_t far_have;          // bytes of fq not yet copied
    uint32_t far_len;           // bytes the load in flight will deliver
    bool far_pending;
    // tables: literal / length code and distance code
    uint32_t llim[16], dlim[16];    // [1..15] used
    uint32_t lpk[16];           // (index adjustment << 16) | first sorted index of the length that holds a symbol >= 256
    int32_t dadj[16];
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

// Chunk i of the input (callers keep i < in_chunks).
template <int RING>
WT_HD WtInfQuad wt_inf_load(const WtInflateT<RING> &z, uint32_t i) {
    const WT_AS_GLOBAL uint32_t *p = z.in_w + 4 * (size_t) i;
    WtInfQuad q;
    q.x = p[0]; q.y = p[1]; q.z = p[2]; q.w = p[3];
    return q;
}

template <int RING>
// (st is written LAST: two branches that end in stores of the same constant to different fields -- copy_rem = 0 here,
// st = WT_INF_ST_SYM = 0 at the end of the block header -- are merged by hipcc into one store through a selected
// pointer, and a selected pointer keeps the fields in scratch memory: a scratch reload at the top of the step loop and
// with it an s_waitcnt vmcnt(0) per step.)
WT_HD void wt_inf_fail(WtInflateT<RING> &z, int code) { z.copy_rem = 0; z.err = code; z.st = WT_INF_ST_ERR; }

// bits of the stream consumed so far
template <int RING>
WT_HD int64_t wt_inf_bitpos(const WtInflateT<RING> &z) { return (int64_t) z.wpos * 32 - (int64_t) z.bc - 8 * (int64_t) z.mis; }

template <int RING>
WT_HD bool wt_inf_overread(const WtInflateT<RING> &z) { return wt_inf_bitpos(z) > (int64_t) z.n_bytes * 8; }

// The next word of the queue.  (The queue is SHIFTED, never indexed: one variable index into the state struct and
// hipcc keeps the whole struct in scratch memory.)  An empty queue yields zero words: the caller has made sure that
// nothing more can arrive (end of the input; wt_inf_overread catches a stream that decodes beyond it).
template <int RING>
WT_HD uint32_t wt_inf_word(WtInflateT<RING> &z) {
    const uint32_t w = z.cur.x;
    z.cur.x = z.cur.y; z.cur.y = z.cur.z; z.cur.z = z.cur.w; z.cur.w = 0;
    z.wpos++;
    if (z.qn) z.qn--;
    if (z.qn == 0 && z.q1_valid) { z.cur = z.q1; z.qn = 4; z.q1_valid = false; }
    return w;
}

// The blocking flavour for the block-header code, which reads many words at one go: fills the queue on the spot.
template <int RING>
WT_HD uint32_t wt_inf_word_cold(WtInflateT<RING> &z) {
    if (z.qn == 0) {
        if (z.pend_valid) { z.cur = z.pend; z.qn = 4; z.pend_valid = false; }
        else if (z.in_chunk < z.in_chunks) { z.cur = wt_inf_load(z, z.in_chunk); z.in_chunk++; z.qn = 4; }
    }
    return wt_inf_word(z);
}

// Positions the reader at bit `pos` of the stream (blocking; whatever was in flight is dropped).
template <int RING>
WT_HD void wt_inf_seek_bits(WtInflateT<RING> &z, uint32_t pos) {
    const uint32_t byte = z.mis + (pos >> 3), word = byte >> 2, ch = word >> 2;
    const WtInfQuad zero = {0u, 0u, 0u, 0u};
    z.cur = ch < z.in_chunks ? wt_inf_load(z, ch) : zero;
    z.q1_valid = ch + 1u < z.in_chunks;
    z.q1 = z.q1_valid ? wt_inf_load(z, ch + 1u) : zero;
    z.in_chunk = ch + 2u;
    z.pend_valid = false;
    z.qn = 4;
    z.wpos = word & ~3u;
    for (uint32_t k = 0; k < (word & 3u); k++) (void) wt_inf_word(z);     // whole words before the position are skipped
    const uint32_t w0 = wt_inf_word(z);
    const uint32_t head = 8u * (byte & 3u) + (pos & 7u);
    z.bb = (uint64_t) (w0 >> head);
    z.bc = 32 - (int32_t) head;
}

// Starts a stream of `n_bytes` at `src` (any alignment; the 16-byte aligned chunks around it must be readable
// inside the same allocation -- up to 15 bytes before and 15 after) writing at most `cap` bytes to `dst` (4-byte
// aligned, cap rounded up to a multiple of 4 must be writable).
template <int RING>
WT_HD void wt_inf_begin(WtInflateT<RING> &z, const uint8_t *src, uint32_t n_bytes, uint8_t *dst, uint32_t cap, bool raw_deflate) {
    const uintptr_t a = (uintptr_t) src;
    z.mis = (uint32_t) (a & 15u);
    z.in_w = (const WT_AS_GLOBAL uint32_t *) (a - z.mis);
    z.in_chunks = (z.mis + n_bytes + 15u) >> 4;
    z.n_bytes = n_bytes;
    z.pend.x = z.pend.y = z.pend.z = z.pend.w = 0;
    wt_inf_seek_bits(z, 0);
    z.out = (WT_AS_GLOBAL uint32_t *) dst; z.out_pos = 0; z.out_cap = cap; z.acc = 0;
    z.st = raw_deflate ? WT_INF_ST_BLOCK : WT_INF_ST_ZHDR;
    z.err = WT_INF_OK;
    z.copy_rem = z.copy_dist = 0; z.stored_rem = 0;
    z.last = false; z.raw = raw_deflate;
    z.far_have = z.far_len = 0; z.far_pending = false;
#pragma unroll
    for (int k = 0; k < 5; k++) { z.fq[k] = 0; z.dperm[k] = 0; }
#pragma unroll
    for (int k = 0; k < 4; k++) z.fpend.q[k] = 0;
    z.fpend.t = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { z.llim[k] = 0; z.dlim[k] = 0; z.lpk[k] = 0; z.dadj[k] = 0; }
}

// After this bc >= 33 (COLD: fills the queue on the spot; otherwise the caller has checked that two words wait).
template <bool COLD, int RING>
WT_HD void wt_inf_refill(WtInflateT<RING> &z) {
    if (z.bc <= 32) {
        const uint32_t w = COLD ? wt_inf_word_cold(z) : wt_inf_word(z);
        z.bb |= (uint64_t) w << z.bc;
        z.bc += 32;
    }
}

template <int RING>
WT_HD uint32_t wt_inf_bits(WtInflateT<RING> &z, int n) {      // n <= 32, after a refill guaranteeing enough bits
    const uint32_t v = (uint32_t) (z.bb & ((1ull << n) - 1ull));
    z.bb >>= n; z.bc -= n;
    return v;
}

// Length and table word of the code word on top of the stream: lim[1..15] / tab[1..15] registers; idx_base = the
// position of the code word among the code words of its length.  Returns false on a pattern no code word matches.
template <int RING, class T>
WT_HD bool wt_inf_code(WtInflateT<RING> &z, const uint32_t (&lim)[16], const T (&tab)[16], T &t, uint32_t &within) {
    const uint32_t w = wt_inf_bitrev15((uint32_t) z.bb & 0x7FFFu);
    int len = 1;
    T a = tab[1];
#pragma unroll
    for (int k = 1; k <= 14; k++) {
        const bool ge = w >= lim[k];
        len += ge ? 1 : 0;
        a = ge ? tab[k + 1] : a;
        WT_INF_OPAQUE(a);
    }
    t = a;
    within = w >> (15 - len);
    z.bb >>= len; z.bc -= len;
    return w < lim[15];
}

// lim[] of a canonical code from the 16 per-length counts (lane memory); calls tab(k, first code of length k,
// symbols shorter than k) for k = 1..15 and leaves the insertion cursor of every length in the counters.
// False: over-subscribed.
template <class F>
WT_HD bool wt_inf_limits(uint32_t (&lim)[16], const WtInfMem &m, F tab) {
    uint32_t code = 0, off = 0;
    bool ok = true;
    lim[0] = 0;
#pragma unroll
    for (int k = 1; k <= 15; k++) {
        const uint32_t c = wt_inf_cnt_get(m, (uint32_t) k);
        if (code + c > (1u << k)) ok = false;
        lim[k] = (code + c) << (15 - k);
        tab(k, code, off);
        wt_inf_cnt_set(m, (uint32_t) k, off);       // insertion cursor of length k
        off += c;
        code = (code + c) << 1;
    }
    return ok;
}

// Appends the low n (1..8) bytes of `bytes` to the output: full dwords go to global memory and to the ring,
// the partial one stays in `acc` (it reaches the ring when a match is about to read it).
template <int RING>
WT_HD void wt_inf_put(WtInflateT<RING> &z, const WtInfMem &m, uint64_t bytes, uint32_t n) {
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
        m.ring[(d & (RING - 1)) * m.stride] = e0;
        acc = e1;
        if (full >= 2) {
            z.out[d + 1] = e1;
            m.ring[((d + 1) & (RING - 1)) * m.stride] = e1;
            acc = e2;
        }
    }
    z.acc = acc;
    z.out_pos += n;
}

// The code lengths of a dynamic block's two alphabets (RFC 1951 3.2.7), run-length coded with the code-length
// code: calls f(i, length) for i = 0 .. total - 1.  Run twice per block (count, then place).
struct WtInfClen {
    uint32_t clim[8];
    int32_t cadj[8];
    uint64_t cp0, cp1;          // the code-length symbols sorted by code, 5 bits each: 12 in cp0, the rest in cp1
};

template <int RING, class F>
WT_HD bool wt_inf_lengths(WtInflateT<RING> &z, const WtInfClen &c, int total, F f) {
    int i = 0;
    uint32_t prev = 0;
    while (i < total) {
        wt_inf_refill<true>(z);
        const uint32_t w7 = wt_inf_bitrev15((uint32_t) z.bb & 0x7Fu) >> 8;     // 7-bit window, first bit on top
        int len = 1;
        int32_t a = c.cadj[1];
#pragma unroll
        for (int k = 1; k <= 6; k++) {
            const bool ge = w7 >= c.clim[k];
            len += ge ? 1 : 0;
            a = ge ? c.cadj[k + 1] : a;
            WT_INF_OPAQUE(a);
        }
        if (w7 >= c.clim[7]) { wt_inf_fail(z, WT_INF_ERR_SYMBOL); return false; }
        const uint32_t j = (uint32_t) (a + (int32_t) (w7 >> (7 - len)));
        const uint32_t sym = (uint32_t) ((j < 12u ? c.cp0 >> (5u * j) : c.cp1 >> (5u * (j - 12u))) & 31u);
        z.bb >>= len; z.bc -= len;
        uint32_t rep = 1, val = sym;
        if (sym == 16) {
            if (i == 0) { wt_inf_fail(z, WT_INF_ERR_CODE); return false; }
            rep = 3 + wt_inf_bits(z, 2); val = prev;
        } else if (sym == 17) {
            rep = 3 + wt_inf_bits(z, 3); val = 0;
        } else if (sym == 18) {
            rep = 11 + wt_inf_bits(z, 7); val = 0;
        } else if (sym > 18) {
            wt_inf_fail(z, WT_INF_ERR_SYMBOL); return false;
        }
        if (i + (int) rep > total) { wt_inf_fail(z, WT_INF_ERR_CODE); return false; }
        for (uint32_t r = 0; r < rep; r++, i++) f(i, val);
        prev = val;
    }
    return true;
}

WT_HD uint32_t wt_inf_fixed_length(int s) { return s < 144 ? 8u : s < 256 ? 9u : s < 280 ? 7u : 8u; }

// Block header (RFC 1951 3.2.3 - 3.2.7): sets up the tables of a fixed / dynamic block or the byte count of a
// stored one.  Executed once or twice per stream: compactness matters more than speed here.
template <int RING>
WT_HD void wt_inf_block(WtInflateT<RING> &z, const WtInfMem &m) {
    const int S = m.stride;
    wt_inf_refill<true>(z);
    z.last = wt_inf_bits(z, 1) != 0;
    const uint32_t type = wt_inf_bits(z, 2);
    if (type == 0) {
        wt_inf_bits(z, z.bc & 7);                   // to the byte boundary (bc counts from it)
        wt_inf_refill<true>(z);
        const uint32_t len = wt_inf_bits(z, 16), nlen = wt_inf_bits(z, 16);
        if ((len ^ nlen) != 0xFFFFu) { wt_inf_fail(z, WT_INF_ERR_BLOCK); return; }
        z.stored_rem = len;
        z.st = len ? WT_INF_ST_STORED : (z.last ? WT_INF_ST_DONE : WT_INF_ST_BLOCK);
        return;
    }
    if (type == 3) { wt_inf_fail(z, WT_INF_ERR_BLOCK); return; }
    // the ring's first 8 dwords become the 16 per-length counters / cursors of the table construction: the history
    // they hold (live when this is not the stream's first block) waits in registers
    uint32_t saved[8];
#pragma unroll
    for (int k = 0; k < 8; k++) saved[k] = m.ring[k * S];
    uint64_t dl0 = 0, dl1 = 0;                      // lengths of the distance symbols, 4 bits each (16 per word)
    uint32_t lits[16];                              // literals (symbols < 256) per length: the counters when symbol 256 comes up
    int hlit = 288, hdist = 30;
    bool ok = true;
    WtInfClen c;
    uint32_t mark = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { wt_inf_cnt_set(m, (uint32_t) k, 0); lits[k] = 0; }
    auto count = [&](int i, uint32_t l) {
        if (i == 256) {
#pragma unroll
            for (int k = 0; k < 16; k++) lits[k] = wt_inf_cnt_get(m, (uint32_t) k);
            if (l == 0) ok = false;                 // no end-of-block code
        }
        if (i < hlit) wt_inf_cnt_set(m, l, wt_inf_cnt_get(m, l) + 1u);
        else {
            const uint32_t q = (uint32_t) (i - hlit);
            if (q < 16u) dl0 |= (uint64_t) l << (4u * q); else dl1 |= (uint64_t) l << (4u * (q - 16u));
        }
    };
    auto place = [&](int i, uint32_t l) {
        if (l && i < hlit) {
            const uint32_t j = wt_inf_cnt_get(m, l);
            wt_inf_cnt_set(m, l, j + 1u);
            if (j < (uint32_t) WT_INF_PERM) wt_inf_perm_set(m, j, (uint32_t) i);
        }
    };
    if (type == 1) {
        for (int s = 0; s < 288; s++) count(s, wt_inf_fixed_length(s));
        dl0 = 0x5555555555555555ull; dl1 = 0x0055555555555555ull;       // 30 codes of 5 bits
    } else {
        hlit = (int) wt_inf_bits(z, 5) + 257;
        hdist = (int) wt_inf_bits(z, 5) + 1;
        const int hclen = (int) wt_inf_bits(z, 4) + 4;
        if (hlit > 286 || hdist > 30) { wt_inf_fail(z, WT_INF_ERR_CODE); return; }
        // the code-length code: 19 lengths of 3 bits, packed 3 bits per symbol
        uint64_t cl = 0;
        wt_inf_refill<true>(z);
#pragma unroll
        for (int i = 0; i < 19; i++) {
            const int order = i == 0 ? 16 : i == 1 ? 17 : i == 2 ? 18 : i == 3 ? 0 : (i & 1) ? (8 - ((i - 3) >> 1)) : (7 + ((i - 2) >> 1));
            if (i == 10) wt_inf_refill<true>(z);
            if (i < hclen) cl |= (uint64_t) wt_inf_bits(z, 3) << (3 * order);
        }
        uint64_t cc = 0;                            // 8-bit counters per length
#pragma unroll
        for (int s = 0; s < 19; s++) cc += 1ull << (8 * (int) ((cl >> (3 * s)) & 7u));
        uint64_t cur = 0;                           // insertion cursors, 8 bits per length
        {
            uint32_t code = 0, off = 0;
            bool cok = true;
            c.clim[0] = 0; c.cadj[0] = 0;
#pragma unroll
            for (int k = 1; k <= 7; k++) {
                const uint32_t n = (uint32_t) (cc >> (8 * k)) & 255u;
                if (code + n > (1u << k)) cok = false;
                c.clim[k] = (code + n) << (7 - k);
                c.cadj[k] = (int32_t) off - (int32_t) code;
                cur |= (uint64_t) off << (8 * k);
                off += n;
                code = (code + n) << 1;
            }
            if (!cok) { wt_inf_fail(z, WT_INF_ERR_CODE); return; }
        }
        c.cp0 = 0; c.cp1 = 0;
#pragma unroll
        for (int s = 0; s < 19; s++) {
            const int l = (int) ((cl >> (3 * s)) & 7u);
            if (l) {
                const uint32_t j = (uint32_t) (cur >> (8 * l)) & 255u;
                cur += 1ull << (8 * l);
                if (j < 12u) c.cp0 |= (uint64_t) s << (5u * j); else c.cp1 |= (uint64_t) s << (5u * (j - 12u));
            }
        }
        mark = (uint32_t) wt_inf_bitpos(z);
        if (!wt_inf_lengths(z, c, hlit + hdist, count)) return;
    }
    // literal / length code: limits, and per length (index adjustment << 16) | first index holding a symbol >= 256
    uint32_t lpk[16];
    lpk[0] = 0;
    ok = wt_inf_limits(z.llim, m, [&](int k, uint32_t code, uint32_t off) {
        lpk[k] = ((uint32_t) ((int32_t) off - (int32_t) code) << 16) | ((off + lits[k]) & 0xFFFFu);
    }) && ok;
#pragma unroll
    for (int k = 0; k < 16; k++) z.lpk[k] = lpk[k];
    // ... its symbols sorted by code (low bytes): the lengths once more, this time to place them
    if (type == 1) {
        for (int s = 0; s < 288; s++) place(s, wt_inf_fixed_length(s));
    } else {
        // (the whole sequence again: a repeat code may run from one alphabet into the other)
        wt_inf_seek_bits(z, mark);
        if (!wt_inf_lengths(z, c, hlit + hdist, place)) return;
    }
    // distance table: 30 symbols, sorted into 5 registers
#pragma unroll
    for (int k = 0; k < 16; k++) wt_inf_cnt_set(m, (uint32_t) k, 0);
    for (int s = 0; s < 30; s++) {
        const uint32_t l = (uint32_t) ((s < 16 ? dl0 >> (4 * s) : dl1 >> (4 * (s - 16))) & 15u);
        wt_inf_cnt_set(m, l, wt_inf_cnt_get(m, l) + 1u);
    }
    int32_t dadj[16];
    dadj[0] = 0;
    ok = wt_inf_limits(z.dlim, m, [&](int k, uint32_t code, uint32_t off) { dadj[k] = (int32_t) off - (int32_t) code; }) && ok;
#pragma unroll
    for (int k = 0; k < 16; k++) z.dadj[k] = dadj[k];
    uint32_t dp[5] = {0u, 0u, 0u, 0u, 0u};
    for (int s = 0; s < 30; s++) {
        const uint32_t l = (uint32_t) ((s < 16 ? dl0 >> (4 * s) : dl1 >> (4 * (s - 16))) & 15u);
        if (l) {
            const uint32_t j = wt_inf_cnt_get(m, l);
            wt_inf_cnt_set(m, l, j + 1u);
            const uint32_t word = j / 6u, sh = 5u * (j - 6u * word);
#pragma unroll
            for (int q = 0; q < 5; q++) dp[q] |= (word == (uint32_t) q) ? ((uint32_t) s << sh) : 0u;
        }
    }
#pragma unroll
    for (int q = 0; q < 5; q++) z.dperm[q] = dp[q];
#pragma unroll
    for (int k = 0; k < 8; k++) m.ring[k * S] = saved[k];
    if (!ok) { wt_inf_fail(z, WT_INF_ERR_CODE); return; }
    z.st = WT_INF_ST_SYM;
}

// ---- round boundary: everything that waits for memory, and everything rare
// Returns false once the lane has nothing more to do.
template <int RING>
WT_HD bool wt_inf_land(WtInflateT<RING> &z, const WtInfMem &m) {
    if (z.st >= WT_INF_ST_DONE) return false;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the round's one wait for memory: `pend` and `fpend` have landed
#endif
    if (z.pend_valid) {
        if (z.qn == 0) { z.cur = z.pend; z.qn = 4; }
        else { z.q1 = z.pend; z.q1_valid = true; }
        z.pend_valid = false;
    }
    if (z.far_pending) {
#pragma unroll
        for (int k = 0; k < 4; k++) z.fq[k] = z.fpend.q[k];
        z.fq[4] = z.fpend.t;
        z.far_have = z.far_len;
        z.far_pending = false;
    }
    if (wt_inf_overread(z)) { wt_inf_fail(z, WT_INF_ERR_INPUT); return false; }
    if (z.st == WT_INF_ST_ZHDR) {                   // RFC 1950 -- CMF, FLG
        wt_inf_refill<true>(z);
        const uint32_t cmf = wt_inf_bits(z, 8), flg = wt_inf_bits(z, 8);
        if ((cmf & 15u) != 8u || (cmf >> 4) > 7u || ((cmf << 8) | flg) % 31u != 0u || (flg & 32u)) { wt_inf_fail(z, WT_INF_ERR_HEADER); return false; }
        z.st = WT_INF_ST_BLOCK;
    }
    if (z.st == WT_INF_ST_BLOCK) {
        wt_inf_block(z, m);
        if (z.st >= WT_INF_ST_DONE) return false;
    }
    if (!z.q1_valid && !z.pend_valid && z.in_chunk < z.in_chunks) {
        z.pend = wt_inf_load(z, z.in_chunk);
        z.in_chunk++;
        z.pend_valid = true;
    }
    return true;
}

// ---- one step: at most one symbol, at most WT_INF_COPY bytes.  No load issued here is consumed here.
template <int RING>
WT_HD void wt_inf_step(WtInflateT<RING> &z, const WtInfMem &m) {
    const int S = m.stride;
    uint64_t bytes = 0;
    uint32_t n = 0;
    // two words cover the worst symbol (15 + 5 + 15 + 13 bits after a refill to >= 33)
    const bool fed = z.qn + (z.q1_valid ? 4u : 0u) >= 2u || (!z.pend_valid && z.in_chunk >= z.in_chunks);
    if (z.copy_rem == 0 && fed && z.st <= WT_INF_ST_STORED) {
        wt_inf_refill<false>(z);
        if (z.st == WT_INF_ST_SYM) {
            uint32_t pk, within;
            const bool hit = wt_inf_code(z, z.llim, z.lpk, pk, within);
            const uint32_t idx = (uint32_t) (((int32_t) pk >> 16) + (int32_t) within);
            const uint32_t s8 = hit ? wt_inf_perm_get(m, idx < (uint32_t) WT_INF_PERM ? idx : 0u) : 0u;
            if (!hit) {
                wt_inf_fail(z, WT_INF_ERR_SYMBOL);
            } else if (idx < (pk & 0xFFFFu)) {      // a literal
                if (z.out_pos >= z.out_cap) wt_inf_fail(z, WT_INF_ERR_SPACE);
                else { bytes = s8; n = 1; }
            } else if (s8 == 0u) {                  // 256: end of block
                z.st = z.last ? WT_INF_ST_DONE : WT_INF_ST_BLOCK;
            } else {
                const uint32_t ls = s8 - 1u;        // length symbol 257 + ls
                uint32_t len;
                if (ls < 8u) len = ls + 3u;
                else if (ls == 28u) len = 258u;
                else {
                    const uint32_t e = (ls - 4u) >> 2;
                    len = ((4u + (ls & 3u)) << e) + 3u + wt_inf_bits(z, (int) (e < 6u ? e : 0u));
                }
                wt_inf_refill<false>(z);
                int32_t da;
                uint32_t dwithin;
                const bool dhit = wt_inf_code(z, z.dlim, z.dadj, da, dwithin);
                const uint32_t di = (uint32_t) (da + (int32_t) dwithin);
                const uint32_t word = di / 6u, dsh = 5u * (di - 6u * word);
                uint32_t dw = z.dperm[0];
                dw = word == 1u ? z.dperm[1] : dw; WT_INF_OPAQUE(dw);
                dw = word == 2u ? z.dperm[2] : dw; WT_INF_OPAQUE(dw);
                dw = word == 3u ? z.dperm[3] : dw; WT_INF_OPAQUE(dw);
                dw = word == 4u ? z.dperm[4] : dw; WT_INF_OPAQUE(dw);
                const uint32_t ds = (dw >> dsh) & 31u;
                uint32_t dist = ds + 1u;
                if (ds >= 4u) {
                    const uint32_t e = (ds >> 1) - 1u;
                    dist = ((2u + (ds & 1u)) << e) + 1u + wt_inf_bits(z, (int) (e < 14u ? e : 0u));
                }
                if (ls > 28u || !dhit || di > 29u || ds > 29u) wt_inf_fail(z, WT_INF_ERR_SYMBOL);
                else if (dist > z.out_pos) wt_inf_fail(z, WT_INF_ERR_DIST);
                else if (z.out_pos + len > z.out_cap) wt_inf_fail(z, WT_INF_ERR_SPACE);
                else { z.copy_rem = len; z.copy_dist = dist; }
            }
        } else {                                    // WT_INF_ST_STORED: up to 4 bytes (bc is a multiple of 8 here)
            const uint32_t k = z.stored_rem < 4u ? z.stored_rem : 4u;
            if (z.out_pos + k > z.out_cap) wt_inf_fail(z, WT_INF_ERR_SPACE);
            else {
                bytes = (uint64_t) wt_inf_bits(z, 8 * (int) k);
                n = k;
                z.stored_rem -= k;
                if (!z.stored_rem) z.st = z.last ? WT_INF_ST_DONE : WT_INF_ST_BLOCK;
            }
        }
    }
    if (z.copy_rem) {
        const uint32_t src = z.out_pos - z.copy_dist;
        const uint32_t sh = (src & 3u) * 8u;
        if (z.copy_dist <= (uint32_t) WT_INF_RING_DIST(RING)) {
            // up to 8 bytes of the match: three dwords around the source, one LDS round trip
            const uint32_t d0 = src >> 2;
            m.ring[((z.out_pos >> 2) & (RING - 1)) * S] = z.acc;          // the partial dword may be part of the source
            const uint32_t r0 = m.ring[(d0 & (RING - 1)) * S];
            const uint32_t r1 = m.ring[((d0 + 1) & (RING - 1)) * S];
            const uint32_t r2 = m.ring[((d0 + 2) & (RING - 1)) * S];
            uint64_t w = ((uint64_t) r0 | ((uint64_t) r1 << 32)) >> sh;
            if (sh) w |= (uint64_t) r2 << (64u - sh);
            if (z.copy_dist < 8u) {                 // overlapping copy: the first `dist` bytes repeat
                const uint32_t s8 = 8u * z.copy_dist;
                w &= (1ull << s8) - 1ull;
                w |= w << s8;
                if (2u * s8 < 64u) w |= w << (2u * s8);
                if (4u * s8 < 64u) w |= w << (4u * s8);
            }
            n = z.copy_rem < (uint32_t) WT_INF_COPY ? z.copy_rem : (uint32_t) WT_INF_COPY;
            bytes = w;
            z.copy_rem -= n;
        } else if (z.far_have) {                    // landed: the next <= 8 bytes wait in fq (never overlapping: dist > FAR)
            uint64_t w = ((uint64_t) z.fq[0] | ((uint64_t) z.fq[1] << 32)) >> sh;
            if (sh) w |= (uint64_t) z.fq[2] << (64u - sh);
            n = z.far_have < (uint32_t) WT_INF_COPY ? z.far_have : (uint32_t) WT_INF_COPY;
            bytes = w;
            z.far_have -= n;
            z.copy_rem -= n;
            z.fq[0] = z.fq[2]; z.fq[1] = z.fq[3]; z.fq[2] = z.fq[4];
        } else if (!z.far_pending) {
            // beyond the ring: the lane's own output is read back (full dwords were stored as they filled up, the
            // partial one is flushed now); the bytes land at the next round boundary and the lane idles until then
            const uint32_t d0 = src >> 2;
            if (z.out_pos & 3u) z.out[z.out_pos >> 2] = z.acc;
#if defined(__HIP_DEVICE_COMPILE__)
            const WT_AS_GLOBAL uint32_t *sp = z.out + d0;
            asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dword %1, %2, off offset:16"
                         : "+v"(z.fpend.q), "+v"(z.fpend.t) : "v"(sp) : "memory");
#else
            for (int k = 0; k < 4; k++) z.fpend.q[k] = z.out[d0 + k];
            z.fpend.t = z.out[d0 + 4];
#endif
            z.far_len = z.copy_rem < (uint32_t) WT_INF_FAR ? z.copy_rem : (uint32_t) WT_INF_FAR;
            z.far_pending = true;
        }
    }
    if (n) wt_inf_put(z, m, bytes, n);
}

// Flushes the last partial dword; returns the number of bytes produced or -(error code).
template <int RING>
WT_HD int64_t wt_inf_finish(WtInflateT<RING> &z) {
    if (z.st == WT_INF_ST_DONE && wt_inf_overread(z)) { z.st = WT_INF_ST_ERR; z.err = WT_INF_ERR_INPUT; }
    if (z.st != WT_INF_ST_DONE) return -(int64_t) (z.err ? z.err : WT_INF_ERR_INPUT);
    if (z.out_pos & 3u) z.out[z.out_pos >> 2] = z.acc;
    return (int64_t) z.out_pos;
}

// The whole stream: rounds of WT_INF_ROUND steps between landings.
template <int RING>
WT_HD int64_t wt_inf_run(WtInflateT<RING> &z, const WtInfMem &m) {
    while (wt_inf_land(z, m)) {
#pragma unroll 1
        for (int r = 0; r < WT_INF_ROUND; r++) wt_inf_step(z, m);
    }
    return wt_inf_finish(z);
}

#endif  // WT_INFLATE_H_
