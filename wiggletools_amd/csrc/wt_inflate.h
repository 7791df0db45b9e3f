// wt_inflate.h -- DEFLATE / zlib (RFC 1950, 1951) decoder written for ONE GPU LANE PER STREAM.
//
// Why: a BigWig file is a list of independent zlib streams ("sections" of <= a few thousand items,
// reference src/bigWiggleReader.c:52-83 reads them through libBigWig, which inflates every one on the
// host).  The file leg of the engine was bound by exactly that host inflate (DESIGN A.5).  Sections are
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
// Round 6: A LITERAL LEADS (WT_INF_LEAD).  A step decodes the symbol at the bit position and, as if that one were a literal, the symbol
// behind it; when the first IS a literal the step emits it and carries on with the second as "the symbol" (a literal, a match and its
// first 7 bytes, the end of the block).  3 770 steps per 1024-item section instead of 5 400 (measured on the bench's sections through
// the host build: wtemu_inflate_ring), each a quarter longer (a third chain of compares behind the first two): 11.1 -> 9.5 ms per
// batch of 100 000 sections on MI355X, and 9.1 ms with rounds of 2 steps instead of 4 (tools/experiments/r6_inf_prof.sh).
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
// the chains of compares: lengths below K1 are compared for by every wavefront, K1 .. K2 - 1 and K2 .. 14 behind wave-uniform tests on the
// codes' longest words (literal / length code, distance code)
#ifndef WT_INF_LK1
#define WT_INF_LK1 12
#define WT_INF_LK2 13
#endif
#ifndef WT_INF_DK1
#define WT_INF_DK1 9
#define WT_INF_DK2 11
#endif
#ifndef WT_INF_LEAD
#define WT_INF_LEAD 1           // a literal in front of a symbol is emitted by the same step (round 6)
#endif
#ifndef WT_INF_ROUND
#define WT_INF_ROUND 2          // steps between two landings (round 6, with two symbols per step: 2 -> 9.12 ms, 3 -> 9.18, 4 -> 9.42, 6 -> 11.0 per batch)
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
    // input: 16-byte chunks ("quads") at 16-byte aligned addresses.  The stream is read POSITIONALLY: `bp` is the bit of
    // the current quad `cur` the next symbol starts at, `q1` is the quad behind it, `pend` is in flight and lands at the
    // next round boundary.  A step peeks 64 bits at bp (wt_inf_peek: selects + two v_alignbit, no state) and adds what
    // it consumed to bp; a bit BUFFER refilled word by word (round-4 first version) cost two conditional refills with
    // their queue shifts per step, ~50 VALU instructions and six branches.
    const WT_AS_GLOBAL uint32_t *in_w;  // aligned base
    uint32_t mis;               // bytes between the aligned base and the stream
    uint32_t in_chunk;          // next quad to fetch
    uint32_t in_chunks;         // quads that may be read
    WtInfQuad cur, q1, pend;
    bool q1_valid, pend_valid;
    uint32_t bp;                // 0 .. 127 (+ what the last step consumed: normalised at the start of the next)
    uint32_t qbase;             // quads before `cur`
    uint32_t n_bytes;
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
    uint32_t lmax, dmax;        // longest code word of the two codes
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
WT_HD int64_t wt_inf_bitpos(const WtInflateT<RING> &z) { return (int64_t) z.qbase * 128 + (int64_t) z.bp - 8 * (int64_t) z.mis; }

// byte offset, from the start of the stream, of what follows the final block (a zlib stream's Adler-32 trailer)
template <int RING>
WT_HD uint32_t wt_inf_end_byte(const WtInflateT<RING> &z) { const int64_t b = wt_inf_bitpos(z); return b > 0 ? (uint32_t) ((b + 7) >> 3) : 0u; }

template <int RING>
WT_HD bool wt_inf_overread(const WtInflateT<RING> &z) { return wt_inf_bitpos(z) > (int64_t) z.n_bytes * 8; }

WT_HD uint32_t wt_inf_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {        // low 32 bits of (hi:lo) >> sh, sh < 32
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t) ((((uint64_t) hi << 32) | (uint64_t) lo) >> (sh & 31u));
#endif
}

// 64 bits of the stream from bit bp (< 128) of the current quad on; the bits beyond `cur` come from q1 (whatever it
// holds when it is not valid: a stream never CONSUMES bits it does not have, wt_inf_overread sees to the rest).
template <int RING>
WT_HD uint64_t wt_inf_peek(const WtInflateT<RING> &z) {
    const uint32_t i = z.bp >> 5, sh = z.bp & 31u;
    const bool i1 = i == 1u, i2 = i == 2u, i3 = i >= 3u;
    uint32_t w0 = z.cur.x, w1 = z.cur.y, w2 = z.cur.z;
    w0 = i1 ? z.cur.y : w0; w1 = i1 ? z.cur.z : w1; w2 = i1 ? z.cur.w : w2;
    w0 = i2 ? z.cur.z : w0; w1 = i2 ? z.cur.w : w1; w2 = i2 ? z.q1.x : w2;
    w0 = i3 ? z.cur.w : w0; w1 = i3 ? z.q1.x : w1; w2 = i3 ? z.q1.y : w2;
    return (uint64_t) wt_inf_alignbit(w1, w0, sh) | ((uint64_t) wt_inf_alignbit(w2, w1, sh) << 32);
}

// Positions the reader at bit `pos` of the stream (blocking; whatever was in flight is dropped).
template <int RING>
WT_HD void wt_inf_seek_bits(WtInflateT<RING> &z, uint32_t pos) {
    const uint32_t t = 8u * z.mis + pos, qi = t >> 7;
    const WtInfQuad zero = {0u, 0u, 0u, 0u};
    z.cur = qi < z.in_chunks ? wt_inf_load(z, qi) : zero;
    z.q1_valid = qi + 1u < z.in_chunks;
    z.q1 = z.q1_valid ? wt_inf_load(z, qi + 1u) : zero;
    z.in_chunk = qi + 2u;
    z.pend_valid = false;
    z.bp = t & 127u;
    z.qbase = qi;
}

// The blocking flavour of the reader, for the block-header code (which reads many words at one go): makes bp < 128
// and q1 valid (while the input lasts), fetching on the spot.
template <int RING>
WT_HD void wt_inf_cold_norm(WtInflateT<RING> &z) {
    const WtInfQuad zero = {0u, 0u, 0u, 0u};
    while (z.bp >= 128u) {
        if (z.q1_valid) z.cur = z.q1;
        else if (z.pend_valid) { z.cur = z.pend; z.pend_valid = false; }
        else if (z.in_chunk < z.in_chunks) { z.cur = wt_inf_load(z, z.in_chunk); z.in_chunk++; }
        else z.cur = zero;
        z.q1_valid = false;
        z.q1 = zero;
        z.bp -= 128u;
        z.qbase++;
    }
    if (!z.q1_valid) {
        if (z.pend_valid) { z.q1 = z.pend; z.pend_valid = false; z.q1_valid = true; }
        else if (z.in_chunk < z.in_chunks) { z.q1 = wt_inf_load(z, z.in_chunk); z.in_chunk++; z.q1_valid = true; }
    }
}

// the next n (<= 32) bits, consumed (cold paths only)
template <int RING>
WT_HD uint32_t wt_inf_bits(WtInflateT<RING> &z, int n) {
    wt_inf_cold_norm(z);
    const uint32_t v = (uint32_t) (wt_inf_peek(z) & ((1ull << n) - 1ull));
    z.bp += (uint32_t) n;
    return v;
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
    z.lmax = z.dmax = 1;
#pragma unroll
    for (int k = 0; k < 5; k++) { z.fq[k] = 0; z.dperm[k] = 0; }
#pragma unroll
    for (int k = 0; k < 4; k++) z.fpend.q[k] = 0;
    z.fpend.t = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { z.llim[k] = 0; z.dlim[k] = 0; z.lpk[k] = 0; z.dadj[k] = 0; }
}

// true if the condition holds for ANY lane of the wavefront (wave-uniform: branches on it are scalar branches)
WT_HD bool wt_inf_any(bool c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(c) != 0ull;
#else
    return c;
#endif
}

// Length of the code word that starts the left-aligned 15-bit window w, and the table word of that length:
// lim[1..15] / tab[1..15] registers.  (w >= lim[15]: no code word matches; the caller checks.)
// A code whose longest word has M bits never needs the comparisons k >= M (lim[k] = lim[15] there): `maxlen` is the
// lane's M, and the tail of the chain sits behind wave-uniform branches (K1 < K2: first comparisons of the two tails).
template <int K1, int K2, class T>
WT_HD int wt_inf_code(uint32_t w, const uint32_t (&lim)[16], const T (&tab)[16], uint32_t maxlen, T &t) {
    int len = 1;
    T a = tab[1];
#define WT_INF_CODE_STEP(k)                                                                       \
    {                                                                                             \
        const bool ge = w >= lim[k];                                                              \
        len += ge ? 1 : 0;                                                                        \
        a = ge ? tab[(k) + 1] : a;                                                                \
        WT_INF_OPAQUE(a);                                                                         \
    }
#pragma unroll
    for (int k = 1; k < K1; k++) WT_INF_CODE_STEP(k)
    if (wt_inf_any(maxlen > (uint32_t) K1)) {
#pragma unroll
        for (int k = K1; k < K2; k++) WT_INF_CODE_STEP(k)
        if (wt_inf_any(maxlen > (uint32_t) K2)) {
#pragma unroll
            for (int k = K2; k <= 14; k++) WT_INF_CODE_STEP(k)
        }
    }
#undef WT_INF_CODE_STEP
    t = a;
    return len;
}

// lim[] of a canonical code from the 16 per-length counts (lane memory); calls tab(k, first code of length k,
// symbols shorter than k) for k = 1..15 and leaves the insertion cursor of every length in the counters.
// False: over-subscribed.
template <class F>
WT_HD bool wt_inf_limits(uint32_t (&lim)[16], const WtInfMem &m, uint32_t &maxlen, F tab) {
    uint32_t code = 0, off = 0;
    bool ok = true;
    lim[0] = 0;
    maxlen = 1;
#pragma unroll
    for (int k = 1; k <= 15; k++) {
        const uint32_t c = wt_inf_cnt_get(m, (uint32_t) k);
        if (c) maxlen = (uint32_t) k;
        if (code + c > (1u << k)) ok = false;
        lim[k] = (code + c) << (15 - k);
        tab(k, code, off);
        wt_inf_cnt_set(m, (uint32_t) k, off);       // insertion cursor of length k
        off += c;
        code = (code + c) << 1;
    }
    return ok;
}

// Appends the low n (0..8) bytes of `bytes` to the output: full dwords go to global memory and to the ring, the partial
// one stays in `acc` (it reaches the ring when a match is about to read it).  Branch-free but for the two stores.
template <int RING>
WT_HD void wt_inf_put(WtInflateT<RING> &z, const WtInfMem &m, uint64_t bytes, uint32_t n) {
    bytes &= n < 8u ? (1ull << (8u * n)) - 1ull : ~0ull;
    const uint32_t sh = (z.out_pos & 3u) * 8u;
    const uint32_t d = z.out_pos >> 2;
    const uint32_t e0 = z.acc | (uint32_t) (bytes << sh);
    const uint64_t t = bytes >> (32u - sh);
    const uint32_t e1 = (uint32_t) t, e2 = (uint32_t) (t >> 32);
    const uint32_t full = ((z.out_pos & 3u) + n) >> 2;
    if (full >= 1u) {
        z.out[d] = e0;
        m.ring[(d & (RING - 1)) * m.stride] = e0;
    }
    if (full >= 2u) {
        z.out[d + 1] = e1;
        m.ring[((d + 1) & (RING - 1)) * m.stride] = e1;
    }
    uint32_t acc = e0;
    acc = full >= 1u ? e1 : acc;
    acc = full >= 2u ? e2 : acc;
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
        wt_inf_cold_norm(z);
        const uint32_t w7 = wt_inf_bitrev15((uint32_t) wt_inf_peek(z) & 0x7Fu) >> 8;     // 7-bit window, first bit on top
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
        z.bp += (uint32_t) len;
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
    z.last = wt_inf_bits(z, 1) != 0;
    const uint32_t type = wt_inf_bits(z, 2);
    if (type == 0) {
        z.bp += (8u - (z.bp & 7u)) & 7u;           // to the byte boundary (quads are byte aligned)
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
#pragma unroll
        for (int i = 0; i < 19; i++) {
            const int order = i == 0 ? 16 : i == 1 ? 17 : i == 2 ? 18 : i == 3 ? 0 : (i & 1) ? (8 - ((i - 3) >> 1)) : (7 + ((i - 2) >> 1));
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
    ok = wt_inf_limits(z.llim, m, z.lmax, [&](int k, uint32_t code, uint32_t off) {
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
    ok = wt_inf_limits(z.dlim, m, z.dmax, [&](int k, uint32_t code, uint32_t off) { dadj[k] = (int32_t) off - (int32_t) code; }) && ok;
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
    if (z.pend_valid) { z.q1 = z.pend; z.q1_valid = true; z.pend_valid = false; }      // (only issued while q1 is empty)
    if (z.far_pending) {
#pragma unroll
        for (int k = 0; k < 4; k++) z.fq[k] = z.fpend.q[k];
        z.fq[4] = z.fpend.t;
        z.far_have = z.far_len;
        z.far_pending = false;
    }
    if (wt_inf_overread(z)) { wt_inf_fail(z, WT_INF_ERR_INPUT); return false; }
    if (z.st == WT_INF_ST_ZHDR) {                   // RFC 1950 -- CMF, FLG
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
// Written WITHOUT BRANCHES around arithmetic: 64 lanes are never in the same state, so a wavefront pays for every
// path of every step anyway, and the structurised form of "if literal / if length / if refill ..." cost one SALU
// instruction for every three VALU ones plus 36 branches per step (SQ counters of the first round-4 version: 3.1e9
// VALU, 1.0e9 SALU, 0.32e9 branch instructions per launch; VALUBusy 53 % with two wavefronts per SIMD).  Every lane
// decodes a literal / length code AND a distance code from its 64-bit window, reads the ring, and selects; the only
// branches left guard memory operations and the rare paths (stored bytes, a match source beyond the ring).
template <int RING>
WT_HD void wt_inf_step(WtInflateT<RING> &z, const WtInfMem &m) {
    const int S = m.stride;
    // ---- input window
    const bool no_more = !z.pend_valid && z.in_chunk >= z.in_chunks;
    const bool adv = z.bp >= 128u && (z.q1_valid || no_more);     // on to the next quad
    z.cur.x = adv ? z.q1.x : z.cur.x; z.cur.y = adv ? z.q1.y : z.cur.y; z.cur.z = adv ? z.q1.z : z.cur.z; z.cur.w = adv ? z.q1.w : z.cur.w;
    z.q1_valid = z.q1_valid && !adv;
    z.bp -= adv ? 128u : 0u;
    z.qbase += adv ? 1u : 0u;
    // a symbol takes up to 48 bits (15 + 5 + 15 + 13): q1 must be there once bp is beyond 64 (or the input be over)
    const bool fed = z.bp < 64u || (z.bp < 128u && (z.q1_valid || no_more));
    const uint64_t win0 = wt_inf_peek(z);
    const bool dec = z.copy_rem == 0u && fed && z.st == WT_INF_ST_SYM;
    // ---- literal / length code
#if WT_INF_LEAD
    // Round 6: A LITERAL LEADS.  The symbol at the bit position (A) is decoded, and so is the one behind it as if A were a literal (B:
    // the same tables, a second chain of compares and a second table byte); when A is a literal the step emits it AND goes on with B
    // as "the symbol" -- another literal, a match with its first 7 bytes, the end of the block.  In 12-byte record data two symbols in
    // five are literals in front of a match or of another literal: 5 400 steps per section become ~3 100, each a fifth longer.
    // (63 bits of the 64-bit window at most: 15 + 48.)  The leading literal is part of the partial dword the match's source may read
    // (acc1) and leaves through the same put as the bytes behind it.
    const uint32_t wA = wt_inf_bitrev15((uint32_t) win0 & 0x7FFFu);
    uint32_t pkA;
    const int lenA = wt_inf_code<WT_INF_LK1, WT_INF_LK2>(wA, z.llim, z.lpk, z.lmax, pkA);
    const uint32_t idxA = (uint32_t) (((int32_t) pkA >> 16) + (int32_t) (wA >> (15 - lenA)));
    const uint32_t s8A = wt_inf_perm_get(m, idxA < (uint32_t) WT_INF_PERM ? idxA : 0u);
    const uint64_t winB = win0 >> lenA;
    const uint32_t wB = wt_inf_bitrev15((uint32_t) winB & 0x7FFFu);
    uint32_t pkB;
    const int lenB = wt_inf_code<WT_INF_LK1, WT_INF_LK2>(wB, z.llim, z.lpk, z.lmax, pkB);
    const uint32_t idxB = (uint32_t) (((int32_t) pkB >> 16) + (int32_t) (wB >> (15 - lenB)));
    const uint32_t s8B = wt_inf_perm_get(m, idxB < (uint32_t) WT_INF_PERM ? idxB : 0u);
    const bool lead = dec && idxA < (pkA & 0xFFFFu) && wA < z.llim[15] && z.out_pos < z.out_cap;
    const uint64_t win = lead ? winB : win0;
    const uint32_t w1 = lead ? wB : wA;
    const uint32_t pk = lead ? pkB : pkA;
    const int len1 = lead ? lenB : lenA;
    const uint32_t idx = lead ? idxB : idxA;
    const uint32_t s8 = lead ? s8B : s8A;
    const uint32_t nlead = lead ? 1u : 0u;
    const uint32_t out_pos1 = z.out_pos + nlead;                                // where the symbol's own bytes go
    const uint32_t acc1 = lead ? z.acc | (s8A << ((z.out_pos & 3u) * 8u)) : z.acc;      // the partial dword with the leading literal in it
#else
    const uint64_t win = win0;
    const uint32_t w1 = wt_inf_bitrev15((uint32_t) win & 0x7FFFu);
    uint32_t pk;
    const int len1 = wt_inf_code<WT_INF_LK1, WT_INF_LK2>(w1, z.llim, z.lpk, z.lmax, pk);
    const uint32_t idx = (uint32_t) (((int32_t) pk >> 16) + (int32_t) (w1 >> (15 - len1)));
    const uint32_t s8 = wt_inf_perm_get(m, idx < (uint32_t) WT_INF_PERM ? idx : 0u);
    const bool lead = false;
    const uint32_t nlead = 0u, lenA = 0u, s8A = 0u;
    const uint32_t out_pos1 = z.out_pos;
    const uint32_t acc1 = z.acc;
    (void) lead; (void) lenA; (void) s8A;
#endif
    const bool is_lit = idx < (pk & 0xFFFFu);
    const uint64_t a1 = win >> len1;
    // ---- distance code, decoded BEFORE the symbol is back from LDS: right behind the length code, as if the length
    // had no extra bits (lengths 3 .. 10: all but one match in two thousand of 12-byte records; the others take the
    // branch below) -- the table read and this chain overlap instead of following each other
    uint32_t w2 = wt_inf_bitrev15((uint32_t) a1 & 0x7FFFu);
    int32_t da;
    int len2 = wt_inf_code<WT_INF_DK1, WT_INF_DK2>(w2, z.dlim, z.dadj, z.dmax, da);
    // ---- the symbol read as a length symbol 257 + ls
    const uint32_t ls = s8 - 1u;
    const bool l_ext = ls >= 8u && ls < 28u;
    const uint32_t e = l_ext ? (ls - 4u) >> 2 : 0u;
    uint32_t lbase = ((4u + (ls & 3u)) << e) + 3u;
    lbase = ls < 8u ? ls + 3u : lbase;
    lbase = ls == 28u ? 258u : lbase;
    const uint32_t len = lbase + ((uint32_t) a1 & ((1u << e) - 1u));
    uint64_t a2 = a1;
    if (dec && !is_lit && l_ext) {                  // a length with extra bits: the distance code starts behind them
        a2 = a1 >> e;
        w2 = wt_inf_bitrev15((uint32_t) a2 & 0x7FFFu);
        len2 = wt_inf_code<WT_INF_DK1, WT_INF_DK2>(w2, z.dlim, z.dadj, z.dmax, da);
    }
    const uint32_t di = (uint32_t) (da + (int32_t) (w2 >> (15 - len2)));
    const uint32_t word = di / 6u, dsh = 5u * (di - 6u * word);
    uint32_t dw = z.dperm[0];
    dw = word == 1u ? z.dperm[1] : dw; WT_INF_OPAQUE(dw);
    dw = word == 2u ? z.dperm[2] : dw; WT_INF_OPAQUE(dw);
    dw = word == 3u ? z.dperm[3] : dw; WT_INF_OPAQUE(dw);
    dw = word == 4u ? z.dperm[4] : dw; WT_INF_OPAQUE(dw);
    const uint32_t ds = (dw >> dsh) & 31u;
    const bool d_ext = ds >= 4u && ds < 30u;
    const uint32_t de = d_ext ? (ds >> 1) - 1u : 0u;
    uint32_t dbase = ((2u + (ds & 1u)) << de) + 1u;
    dbase = ds < 4u ? ds + 1u : dbase;
    const uint64_t a3 = a2 >> len2;
    const uint32_t dist = dbase + ((uint32_t) a3 & ((1u << de) - 1u));
    // ---- what the symbol is, and whether it is acceptable (booleans: lane masks, combined on the scalar unit)
    const bool is_eob = !is_lit && s8 == 0u;
    const bool is_m = !is_lit && s8 != 0u;
    const bool bad_sym = w1 >= z.llim[15] || (is_m && (ls > 28u || w2 >= z.dlim[15] || di > 29u || ds > 29u));
    const bool bad_dist = is_m && dist > out_pos1;
    const bool bad_space = is_lit ? out_pos1 >= z.out_cap : (is_m && out_pos1 + len > z.out_cap);
    const bool ok = dec && !(bad_sym || bad_dist || bad_space), fail = dec && (bad_sym || bad_dist || bad_space);
    z.bp += ok ? (is_m ? (uint32_t) len1 + e + (uint32_t) len2 + de : (uint32_t) len1) + (lead ? (uint32_t) lenA : 0u) : 0u;
    if (fail) wt_inf_fail(z, bad_sym ? WT_INF_ERR_SYMBOL : bad_dist ? WT_INF_ERR_DIST : WT_INF_ERR_SPACE);      // (rare)
    z.st = (ok && is_eob) ? (z.last ? WT_INF_ST_DONE : WT_INF_ST_BLOCK) : z.st;
    z.copy_rem = (ok && is_m) ? len : z.copy_rem;
    z.copy_dist = (ok && is_m) ? dist : z.copy_dist;
    uint64_t bytes = (uint64_t) s8;
    uint32_t n = (ok && is_lit) ? 1u : 0u;
    // ---- stored bytes (rare: a wavefront usually skips this): up to 4 (the stream is byte aligned here)
    if (z.st == WT_INF_ST_STORED && fed) {
        const uint32_t k = z.stored_rem < 4u ? z.stored_rem : 4u;
        if (z.out_pos + k > z.out_cap) wt_inf_fail(z, WT_INF_ERR_SPACE);
        else {
            bytes = win0;
            n = k;
            z.bp += 8u * k;
            z.stored_rem -= k;
            if (!z.stored_rem) z.st = z.last ? WT_INF_ST_DONE : WT_INF_ST_BLOCK;
        }
    }
    // ---- match bytes: from the ring (every lane reads it, the lanes inside a match near enough use it) ...
    const bool cp = z.copy_rem != 0u;
    const bool near = z.copy_dist <= (uint32_t) WT_INF_RING_DIST(RING);
    const uint32_t src = out_pos1 - z.copy_dist;
    const uint32_t sh = (src & 3u) * 8u, d0 = src >> 2;
    m.ring[((z.out_pos >> 2) & (RING - 1)) * S] = acc1;                   // the partial dword may be part of the source
    const uint32_t r0 = m.ring[(d0 & (RING - 1)) * S];
    const uint32_t r1 = m.ring[((d0 + 1) & (RING - 1)) * S];
    const uint32_t r2 = m.ring[((d0 + 2) & (RING - 1)) * S];
    uint64_t w = (uint64_t) wt_inf_alignbit(r1, r0, sh) | ((uint64_t) wt_inf_alignbit(r2, r1, sh) << 32);
    const uint32_t room = (uint32_t) WT_INF_COPY - nlead;                    // (a leading literal takes one of the put's 8 bytes)
    const uint32_t nc = z.copy_rem < room ? z.copy_rem : room;
    if (wt_inf_any(cp && z.copy_dist < nc)) {      // overlapping copy (run-length-like matches: rare in record data --
        const uint32_t dd = z.copy_dist, s8b = 8u * (dd & 7u);     // a wavefront usually skips this): the first `dist` bytes repeat
        const bool o8 = dd < 8u, o4 = dd < 4u, o2 = dd < 2u;
        w &= o8 ? (1ull << s8b) - 1ull : ~0ull;
        w |= o8 ? w << s8b : 0ull;
        w |= o4 ? w << ((2u * s8b) & 63u) : 0ull;
        w |= o2 ? w << ((4u * s8b) & 63u) : 0ull;
    }
    // ... or from the bytes a load beyond the ring has landed (never overlapping: dist > WT_INF_FAR)
    const bool far_take = cp && !near && z.far_have != 0u;
    const uint64_t wf = (uint64_t) wt_inf_alignbit(z.fq[1], z.fq[0], sh) | ((uint64_t) wt_inf_alignbit(z.fq[2], z.fq[1], sh) << 32);
    const uint32_t nf = z.far_have < (uint32_t) WT_INF_COPY ? z.far_have : (uint32_t) WT_INF_COPY;
    const bool use_ring = cp && near;
    const uint32_t took = use_ring ? nc : (far_take ? nf : 0u);
    bytes = use_ring ? w : (far_take ? wf : bytes);
    n = (use_ring || far_take) ? took : n;
    z.copy_rem -= took;
    z.far_have -= far_take ? nf : 0u;
    z.fq[0] = far_take ? z.fq[2] : z.fq[0]; z.fq[1] = far_take ? z.fq[3] : z.fq[1]; z.fq[2] = far_take ? z.fq[4] : z.fq[2];
    if (cp && !near && !far_take && !z.far_pending) {
        // beyond the ring: the lane's own output is read back (full dwords were stored as they filled up, the
        // partial one is flushed now); the bytes land at the next round boundary and the lane idles until then
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
#if WT_INF_LEAD
    bytes = lead ? (uint64_t) s8A | (bytes << 8) : bytes;
    n += nlead;
#endif
    wt_inf_put(z, m, bytes, n);
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
