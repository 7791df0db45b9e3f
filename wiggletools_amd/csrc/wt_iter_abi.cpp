// wt_iter_abi.cpp -- DROP-IN LAYER: the reference's own C API for the hot path
// (reference src/wiggletools.h:80-103, src/multiplexer.h:38-41, src/multiSet.h:32-33),
// implemented on top of the bulk GPU engine (wtamd_*).  Host-side C++ above the
// C ABI, mirroring the reference's names, argument meaning and error behaviour
// (message to stdout/stderr, then exit(1)).
//
// How a lazy pull API is fed to a bulk engine:
//   * a Drainer pops the N child iterators (each from ONE thread, as the
//     reference requires) into a host SoA batch: one chromosome, run starts in
//     [lo, hi).  An interval that crosses `hi` is carried into the next batch;
//     one interval beyond `hi` per track is included as a sentinel so the last
//     run of the batch gets its true finish (no seam artefacts, coordinates stay
//     bit-exact).  Batches grow geometrically (2 Kbp -> ~4 M intervals).
//   * popMultiplexer() walks the run tile the GPU materialised for the batch
//     and keeps every field of struct multiplexer_st coherent (other reference
//     translation units read them: mWigWriter.c:182-197, statistics.c:432-442).
//   * a reducer constructor (MeanReduction, ...) takes the Multiplexer over: the
//     first (tiny, already drained) batch is pushed back and from then on whole
//     batches go through the FUSED multiplex+reduce kernel; the Multiplexer's
//     per-run fields are then no longer maintained (allowed when the
//     reducer owns the multiplexer, which is how commandParser.c builds them).
//   * TTestReduction / MWUReduction take over both Multiplexers of the Multiset
//     and run the two-sample kernels over the joint track list.
//
// There is no CPU evaluation path here: every run, aligned tile or reduced
// value comes from the HIP kernels.  If no GPU is present the first engine call
// fails and the process exits(1) with the engine's message.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <vector>

#include "../../include/wiggletools_amd.h"

#define WT_WEAK __attribute__((weak))

namespace {

[[noreturn]] void die(const char *what) {
    fprintf(stderr, "wiggletools_amd: %s: %s\n", what, wtamd_last_error());
    exit(1);
}

struct Ivl {
    char *chrom;
    int32_t start, finish;
    double value;
};

// One child iterator plus intervals that were popped from it but pushed back.
struct TrackSource {
    WiggleIterator *it = nullptr;
    std::deque<Ivl> pending;

    bool empty() const { return pending.empty() && it->done; }
    Ivl head() const {
        if (!pending.empty()) return pending.front();
        Ivl v = { it->chrom, it->start, it->finish, it->value };
        return v;
    }
    void advance() {
        if (!pending.empty()) pending.pop_front();
        else pop(it);
    }
};

const int64_t kTargetIntervals = 4 << 20;   // intervals per steady-state batch
const int64_t kFirstSpan = 2048;            // bp of the priming batch

struct Drainer {
    std::vector<TrackSource> src;
    std::vector<double> defaults;
    // current batch (one chromosome)
    char *chrom = nullptr;
    int32_t lo = 0, hi = 0;
    std::vector<int64_t> seg_off;
    std::vector<int32_t> start, finish;
    std::vector<double> value;
    std::vector<float> value32;
    std::vector<char> popped;       // per interval: 0 sentinel (still in the source), 1 consumed,
                                    // 2 consumed but already requeued because it crosses `hi`
    bool all_f32 = true;
    bool have = false;
    // continuation state
    bool continuing = false;        // next batch continues `chrom` at `hi`
    int64_t span = kFirstSpan;

    int n_tracks() const { return (int) src.size(); }

    // Pushes the current batch back so that another consumer can start over from it.
    void rewind() {
        if (!have) return;
        for (int i = n_tracks() - 1; i >= 0; i--) {
            for (int64_t g = seg_off[i + 1] - 1; g >= seg_off[i]; g--) {
                if (popped[g] != 1) continue;
                Ivl v = { chrom, start[g], finish[g], value[g] };
                src[i].pending.push_front(v);
            }
        }
        have = false;
        continuing = false;
        span = kFirstSpan;
    }

    void reset() {      // after seek: forget everything that was buffered
        for (auto &s : src) s.pending.clear();
        have = false;
        continuing = false;
        span = kFirstSpan;
    }

    bool next_batch() {
        const int N = n_tracks();
        have = false;
        // chromosome and range start
        if (continuing) {
            lo = hi;
        } else {
            chrom = nullptr;
            for (int i = 0; i < N; i++) {
                if (src[i].empty()) continue;
                char *c = src[i].head().chrom;
                if (!chrom || strcmp(c, chrom) < 0) chrom = c;     // multiplexer.c:56
            }
            if (!chrom) return false;
            int64_t m = INT32_MAX;
            for (int i = 0; i < N; i++) {
                if (src[i].empty()) continue;
                Ivl h = src[i].head();
                if (strcmp(h.chrom, chrom) == 0 && h.start < m) m = h.start;
            }
            lo = (int32_t) m;
        }
        const int64_t hi64 = (int64_t) lo + span;
        hi = hi64 >= INT32_MAX ? INT32_MAX : (int32_t) hi64;

        seg_off.assign(1, 0);
        start.clear(); finish.clear(); value.clear(); popped.clear();
        all_f32 = true;
        bool more = false;
        for (int i = 0; i < N; i++) {
            TrackSource &s = src[i];
            while (!s.empty()) {
                Ivl h = s.head();
                if (h.chrom != chrom && strcmp(h.chrom, chrom) != 0) break;
                const bool inside = h.start < hi;
                start.push_back(h.start); finish.push_back(h.finish); value.push_back(h.value);
                popped.push_back(inside ? 1 : 0);
                if (all_f32 && !(std::isnan(h.value) || (double) (float) h.value == h.value)) all_f32 = false;
                if (!inside) { more = true; break; }          // sentinel: stays in the source
                s.advance();
                if (h.finish > hi) {                          // crosses the cut: needed again next time
                    popped.back() = 2;
                    s.pending.push_front(h);
                    more = true;
                    // make sure it is not drained twice in this batch
                    break;
                }
            }
            seg_off.push_back((int64_t) start.size());
        }
        // a carried interval sits at the front of `pending` with start < hi: next batch must not
        // stop at it as a "sentinel" -- it is taken because its start < new hi as well.
        continuing = more;
        have = true;
        // grow / shrink towards the interval budget
        const int64_t n = (int64_t) start.size();
        if (n < kTargetIntervals / 2 && span < ((int64_t) 1 << 31)) span *= 2;
        else if (n > kTargetIntervals * 2 && span > kFirstSpan) span /= 2;
        if (all_f32) {
            value32.resize(value.size());
            for (size_t k = 0; k < value.size(); k++) value32[k] = (float) value[k];
        }
        return true;
    }

    wtamd_trackset *upload() {
        wtamd_tracks t;
        memset(&t, 0, sizeof(t));
        t.n_chrom = 1;
        t.n_tracks = n_tracks();
        t.seg_off = seg_off.data();
        t.start = start.data();
        t.finish = finish.data();
        t.value = all_f32 ? (const void *) value32.data() : (const void *) value.data();
        t.value_is_f64 = all_f32 ? 0 : 1;
        t.defaults = defaults.data();
        t.range_lo = &lo;
        t.range_hi = &hi;
        wtamd_trackset *ts = nullptr;
        if (wtamd_trackset_create_host(&t, &ts) != WTAMD_OK) die("wtamd_trackset_create_host");
        return ts;
    }
};

// ---------------------------------------------------------------------------
// Multiplexer
// ---------------------------------------------------------------------------
struct MuxState {
    Drainer dr;
    std::vector<int32_t> rs, rf;
    std::vector<double> rcount;     // per-run inplay_count
    std::vector<double> tile;
    std::vector<uint8_t> ip;
    int64_t n = 0, cur = 0;
    bool materialised = false;
    bool taken_over = false;        // a reducer owns the drainer now
};

MuxState *mux_state(Multiplexer *m) { return (MuxState *) m->data; }

void mux_materialise(Multiplexer *m) {
    MuxState *S = mux_state(m);
    wtamd_trackset *ts = S->dr.upload();
    int64_t cap = wtamd_trackset_max_runs(ts);
    if (cap < 1) cap = 1;
    const int N = m->count;
    S->rs.resize(cap); S->rf.resize(cap); S->rcount.resize(cap);
    S->tile.resize((size_t) cap * N); S->ip.resize((size_t) cap * N);
    wtamd_runs r;
    memset(&r, 0, sizeof(r));
    r.capacity = cap; r.start = S->rs.data(); r.finish = S->rf.data(); r.value = S->rcount.data();
    int64_t n = 0;
    if (wtamd_multiplex_host(ts, m->strict ? WTAMD_STRICT_SET0 : 0, &r, S->tile.data(), S->ip.data(), &n) != WTAMD_OK)
        die("wtamd_multiplex_host");
    wtamd_trackset_destroy(ts);
    S->n = n; S->cur = 0; S->materialised = true;
}

void mux_pop(Multiplexer *m) {
    MuxState *S = mux_state(m);
    if (S->taken_over) { m->done = 1; return; }
    for (;;) {
        if (!S->dr.have) {
            if (!S->dr.next_batch()) { m->done = 1; return; }
            S->materialised = false;
        }
        if (!S->materialised) mux_materialise(m);
        if (S->cur < S->n) {
            const int N = m->count;
            const int64_t r = S->cur++;
            m->chrom = S->dr.chrom;
            m->start = S->rs[r];
            m->finish = S->rf[r];
            for (int i = 0; i < N; i++) {
                m->values[i] = S->tile[(size_t) r * N + i];
                m->inplay[i] = (wt_bool) S->ip[(size_t) r * N + i];
            }
            m->inplay_count = (int) S->rcount[r];
            return;
        }
        S->dr.have = false;     // batch exhausted
    }
}

void mux_seek(Multiplexer *m, const char *chrom, int start, int finish) {
    MuxState *S = mux_state(m);
    m->done = 0;
    for (int i = 0; i < m->count; i++) seek(m->iters[i], chrom, start, finish);   // multiplexer.c:133-134
    S->dr.reset();
    S->materialised = false;
    S->taken_over = false;
    m->inplay_count = 0;
    popMultiplexer(m);
}

// ---------------------------------------------------------------------------
// Reducers (one- and two-sample): iterate over the fused kernel's run list
// ---------------------------------------------------------------------------
struct RedState {
    Drainer dr;
    int op = 0;
    uint32_t flags = 0;
    int n_set0 = 0;
    std::vector<int32_t> rs, rf;
    std::vector<double> rv;
    int64_t n = 0, cur = 0;
    Multiplexer *multi = nullptr;       // one-sample
    Multiset *multiset = nullptr;       // two-sample
};

struct RedData {        // wi->data: must be free()-able like the reference's (wiggleIterator.c:52-55)
    RedState *state;
};

RedState *red_state(WiggleIterator *wi) { return ((RedData *) wi->data)->state; }

void red_pop(WiggleIterator *wi) {
    if (wi->done) return;
    RedState *R = red_state(wi);
    for (;;) {
        if (R->cur < R->n) {
            const int64_t r = R->cur++;
            wi->chrom = R->dr.chrom;
            wi->start = R->rs[r];
            wi->finish = R->rf[r];
            wi->value = R->rv[r];
            return;
        }
        if (!R->dr.next_batch()) {
            wi->done = 1;
            if (R->multi) R->multi->done = 1;
            if (R->multiset) R->multiset->done = 1;
            return;
        }
        wtamd_trackset *ts = R->dr.upload();
        int64_t cap = wtamd_trackset_max_runs(ts);
        if (cap < 1) cap = 1;
        R->rs.resize(cap); R->rf.resize(cap); R->rv.resize(cap);
        wtamd_runs runs;
        memset(&runs, 0, sizeof(runs));
        runs.capacity = cap; runs.start = R->rs.data(); runs.finish = R->rf.data(); runs.value = R->rv.data();
        wtamd_reduce_desc d = { R->op, R->flags, R->n_set0, 0 };
        int64_t n = 0;
        if (wtamd_reduce_host(ts, &d, &runs, &n) != WTAMD_OK) die("wtamd_reduce_host");
        wtamd_trackset_destroy(ts);
        R->n = n; R->cur = 0;
    }
}

void red_take_over(RedState *R, Multiplexer *m) {
    MuxState *S = mux_state(m);
    S->dr.rewind();
    for (auto &s : S->dr.src) R->dr.src.push_back(std::move(s));
    for (double d : S->dr.defaults) R->dr.defaults.push_back(d);
    S->dr.src.clear();
    S->taken_over = true;
}

void red_seek(WiggleIterator *wi, const char *chrom, int start, int finish) {
    // reference WiggleReducerSeek (reducers.c:25-29) / SetComparisonSeek (setComparisons.c:25-29):
    // seek the children, then pop once.
    RedState *R = red_state(wi);
    for (auto &s : R->dr.src) seek(s.it, chrom, start, finish);
    R->dr.reset();
    R->n = R->cur = 0;
    if (R->multi) R->multi->done = 0;
    if (R->multiset) R->multiset->done = 0;
    wi->done = 0;
    pop(wi);
}

WiggleIterator *make_reducer(Multiplexer *m, int op) {
    RedState *R = new RedState();
    R->op = op;
    R->flags = m->strict ? WTAMD_STRICT_SET0 : 0;
    R->multi = m;
    red_take_over(R, m);
    RedData *d = (RedData *) calloc(1, sizeof(RedData));
    d->state = R;
    const double dflt = wtamd_reducer_default(op, m->count, m->default_values);
    return newWiggleIterator(d, &red_pop, &red_seek, dflt, 0);
}

WiggleIterator *make_set_reducer(Multiset *ms, int op) {
    RedState *R = new RedState();
    R->op = op;
    R->multiset = ms;
    R->n_set0 = ms->multis[0]->count;
    R->flags = (ms->multis[0]->strict ? WTAMD_STRICT_SET0 : 0) | (ms->multis[1]->strict ? WTAMD_STRICT_SET1 : 0);
    red_take_over(R, ms->multis[0]);
    red_take_over(R, ms->multis[1]);
    RedData *d = (RedData *) calloc(1, sizeof(RedData));
    d->state = R;
    return newWiggleIterator(d, &red_pop, &red_seek, NAN, 0);     // setComparisons.c:130,389
}

// ---------------------------------------------------------------------------
// Select / FillIn: host iterators over popMultiplexer (reference reducers.c:41-119)
// ---------------------------------------------------------------------------
struct SelData { Multiplexer *multi; int index; wt_bool trim; };

void sel_seek(WiggleIterator *wi, const char *chrom, int start, int finish) {
    SelData *d = (SelData *) wi->data;
    seekMultiplexer(d->multi, chrom, start, finish);
    pop(wi);
}

void select_pop(WiggleIterator *wi) {
    if (wi->done) return;
    SelData *d = (SelData *) wi->data;
    Multiplexer *m = d->multi;
    if (m->done) { wi->done = 1; return; }
    while (m->inplay[d->index]) {              // reducers.c:52-58 (sic: skips runs where the track IS in play)
        popMultiplexer(m);
        if (m->done) { wi->done = 1; return; }
    }
    wi->value = m->values[d->index];
    wi->chrom = m->chrom; wi->start = m->start; wi->finish = m->finish;
    popMultiplexer(m);
}

void fillin_pop(WiggleIterator *wi) {
    if (wi->done) return;
    SelData *d = (SelData *) wi->data;
    Multiplexer *m = d->multi;
    if (m->done) { wi->done = 1; return; }
    if (d->trim) {
        while (!m->inplay[0]) {
            popMultiplexer(m);
            if (m->done) { wi->done = 1; return; }
        }
    }
    wi->chrom = m->chrom; wi->start = m->start; wi->finish = m->finish;
    wi->value = m->inplay[1] ? m->values[1] : m->default_values[1];
    popMultiplexer(m);
}

// ---------------------------------------------------------------------------
// Multiset stepping (K-way alignment of already aligned Multiplexer run streams;
// K is 2 in practice).  Linear scans instead of the reference's heaps
// (multiSet.c:21-101), same run sequence.
// ---------------------------------------------------------------------------
void multiset_step(Multiset *s) {
    const int K = s->count;
    // close (multiSet.c:21-31)
    for (int k = 0; k < K; k++) {
        Multiplexer *m = s->multis[k];
        if (s->inplay[k] && m->finish == s->finish) {
            popMultiplexer(m);
            s->inplay[k] = 0;
            s->inplay_count--;
        }
    }
    // anything waiting on this chromosome?
    bool waiting = false;
    if (s->chrom)
        for (int k = 0; k < K; k++) {
            Multiplexer *m = s->multis[k];
            if (!s->inplay[k] && !m->done && strcmp(m->chrom, s->chrom) == 0) waiting = true;
        }
    if (!s->inplay_count && !waiting) {
        // queue up the next chromosome (multiSet.c:33-58)
        s->chrom = nullptr;
        for (int k = 0; k < K; k++) {
            Multiplexer *m = s->multis[k];
            if (!m->done && (!s->chrom || strcmp(m->chrom, s->chrom) < 0)) s->chrom = m->chrom;
        }
        if (!s->chrom) { s->done = 1; return; }
    }
    int min_start = INT32_MAX;
    for (int k = 0; k < K; k++) {
        Multiplexer *m = s->multis[k];
        if (!s->inplay[k] && !m->done && strcmp(m->chrom, s->chrom) == 0 && m->start < min_start) min_start = m->start;
    }
    s->start = s->inplay_count ? s->finish : min_start;          // multiSet.c:93-96
    for (int k = 0; k < K; k++) {                                // admit, multiSet.c:60-68
        Multiplexer *m = s->multis[k];
        if (!s->inplay[k] && !m->done && strcmp(m->chrom, s->chrom) == 0 && m->start == s->start) {
            s->inplay[k] = 1;
            s->inplay_count++;
        }
    }
    int fin = INT32_MAX;                                         // multiSet.c:70-78
    for (int k = 0; k < K; k++) {
        Multiplexer *m = s->multis[k];
        if (s->inplay[k]) { if (m->finish < fin) fin = m->finish; }
        else if (!m->done && strcmp(m->chrom, s->chrom) == 0 && m->start < fin) fin = m->start;
    }
    s->finish = fin;
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------
// Iterator core (weak: the reference's own wiggleIterator.o / unaryOps.o win when
// this library is linked into the reference build)
// ---------------------------------------------------------------------------
WT_WEAK WiggleIterator *newWiggleIterator(void *data, void (*popFunction)(WiggleIterator *),
                                          void (*seekFunction)(WiggleIterator *, const char *, int, int),
                                          double default_value, wt_bool overlapping) {
    WiggleIterator *wi = (WiggleIterator *) calloc(1, sizeof(WiggleIterator));
    wi->data = data;
    wi->pop = popFunction;
    wi->seek = seekFunction;
    wi->value = 1;                  // value-less bed regions count 1 (wiggleIterator.c:26)
    wi->overlaps = overlapping;
    wi->default_value = default_value;
    pop(wi);                        // a fresh iterator already holds its first element (:32)
    return wi;
}

WT_WEAK void pop(WiggleIterator *wi) {
    if (!wi->done) wi->pop(wi);
}

WT_WEAK void runWiggleIterator(WiggleIterator *wi) {
    while (!wi->done) wi->pop(wi);
}

WT_WEAK void seek(WiggleIterator *wi, const char *chrom, int start, int finish) {
    wi->done = 0;
    wi->seek(wi, chrom, start, finish);
}

WT_WEAK void destroyWiggleIterator(WiggleIterator *wi) {
    free(wi->data);
    free(wi);
}

// union of overlapping regions (reference unaryOps.c:60-96), needed because the
// Multiplexer's children must be non-overlapping (multiplexer.c:163)
struct WtUnionData { WiggleIterator *iter; };

static void wt_union_pop(WiggleIterator *wi) {
    WiggleIterator *it = ((WtUnionData *) wi->data)->iter;
    if (it->done) { wi->done = 1; return; }
    int count = 0;
    while (!it->done) {
        if (!count) {
            wi->chrom = it->chrom; wi->start = it->start; wi->finish = it->finish; wi->value = it->value;
        } else if (wi->chrom == it->chrom && wi->finish > it->start) {      // pointer identity, like :76
            if (it->finish > wi->finish) wi->finish = it->finish;
        } else {
            break;
        }
        count++;
        pop(it);
    }
    wi->done = (count == 0);
}

static void wt_union_seek(WiggleIterator *wi, const char *chrom, int start, int finish) {
    WiggleIterator *it = ((WtUnionData *) wi->data)->iter;
    seek(it, chrom, start, finish);
    wi->done = 0;
    pop(wi);
}

WT_WEAK WiggleIterator *UnionWiggleIterator(WiggleIterator *i) {
    WtUnionData *d = (WtUnionData *) calloc(1, sizeof(WtUnionData));
    d->iter = i;
    return newWiggleIterator(d, &wt_union_pop, &wt_union_seek, i->default_value, 0);
}

WT_WEAK WiggleIterator *NonOverlappingWiggleIterator(WiggleIterator *i) {
    return i->overlaps ? UnionWiggleIterator(i) : i;
}

// ---------------------------------------------------------------------------
// Multiplexer
// ---------------------------------------------------------------------------
void popMultiplexer(Multiplexer *m) {
    if (!m->done) m->pop(m);
}

void runMultiplexer(Multiplexer *m) {
    while (!m->done) m->pop(m);
}

void seekMultiplexer(Multiplexer *m, const char *chrom, int start, int finish) {
    m->done = 0;
    m->seek(m, chrom, start, finish);
}

Multiplexer *newCoreMultiplexer(void *data, int count, void (*popFn)(Multiplexer *),
                                void (*seekFn)(Multiplexer *, const char *, int, int)) {
    Multiplexer *m = (Multiplexer *) calloc(1, sizeof(Multiplexer));
    m->count = count;
    m->values = (double *) calloc((size_t) count, sizeof(double));
    m->default_values = (double *) calloc((size_t) count, sizeof(double));
    m->inplay = (wt_bool *) calloc((size_t) count, sizeof(wt_bool));
    m->pop = popFn;
    m->seek = seekFn;
    m->data = data;
    return m;       // starts / finishes stay NULL: this engine has no heaps
}

Multiplexer *newMultiplexer(WiggleIterator **iters, int count, wt_bool strict) {
    MuxState *S = new MuxState();
    Multiplexer *m = newCoreMultiplexer(S, count, &mux_pop, &mux_seek);
    m->strict = strict;
    m->iters = (WiggleIterator **) calloc((size_t) count, sizeof(WiggleIterator *));
    S->dr.src.resize((size_t) count);
    for (int i = 0; i < count; i++) {
        m->iters[i] = NonOverlappingWiggleIterator(iters[i]);       // multiplexer.c:163
        m->default_values[i] = m->iters[i]->default_value;
        m->values[i] = m->iters[i]->default_value;
        S->dr.src[i].it = m->iters[i];
        S->dr.defaults.push_back(m->iters[i]->default_value);
    }
    popMultiplexer(m);                                              // primed like multiplexer.c:167
    return m;
}

// ---------------------------------------------------------------------------
// Multiset
// ---------------------------------------------------------------------------
void popMultiset(Multiset *s) {
    if (!s->done) multiset_step(s);
}

void seekMultiset(Multiset *s, const char *chrom, int start, int finish) {
    s->done = 0;
    for (int k = 0; k < s->count; k++) seekMultiplexer(s->multis[k], chrom, start, finish);
    for (int k = 0; k < s->count; k++) s->inplay[k] = 0;
    s->inplay_count = 0;
    s->chrom = nullptr;
    popMultiset(s);
}

Multiset *newMultiset(Multiplexer **multis, int count) {
    Multiset *s = (Multiset *) calloc(1, sizeof(Multiset));
    s->count = count;
    s->multis = multis;                                             // keeps the caller's array (multiSet.c:118)
    s->inplay = (wt_bool *) calloc((size_t) count, sizeof(wt_bool));
    s->values = (double **) calloc((size_t) count, sizeof(double *));
    for (int k = 0; k < count; k++) s->values[k] = multis[k]->values;
    popMultiset(s);
    return s;
}

// ---------------------------------------------------------------------------
// Reducers
// ---------------------------------------------------------------------------
WiggleIterator *SumReduction(Multiplexer *m) { return make_reducer(m, WTAMD_OP_SUM); }
WiggleIterator *ProductReduction(Multiplexer *m) { return make_reducer(m, WTAMD_OP_PRODUCT); }
WiggleIterator *MeanReduction(Multiplexer *m) { return make_reducer(m, WTAMD_OP_MEAN); }
WiggleIterator *VarianceReduction(Multiplexer *m) { return make_reducer(m, WTAMD_OP_VAR); }
WiggleIterator *StdDevReduction(Multiplexer *m) { return make_reducer(m, WTAMD_OP_STDDEV); }
WiggleIterator *EntropyReduction(Multiplexer *m) { return make_reducer(m, WTAMD_OP_ENTROPY); }
WiggleIterator *CVReduction(Multiplexer *m) { return make_reducer(m, WTAMD_OP_CV); }
WiggleIterator *MedianReduction(Multiplexer *m) { return make_reducer(m, WTAMD_OP_MEDIAN); }
WiggleIterator *MinReduction(Multiplexer *m) { return make_reducer(m, WTAMD_OP_MIN); }
WiggleIterator *MaxReduction(Multiplexer *m) { return make_reducer(m, WTAMD_OP_MAX); }

WiggleIterator *SelectReduction(Multiplexer *m, int index) {
    SelData *d = (SelData *) calloc(1, sizeof(SelData));
    d->multi = m; d->index = index;
    return newWiggleIterator(d, &select_pop, &sel_seek, m->default_values[index], 0);
}

WiggleIterator *FillInReduction(Multiplexer *m, wt_bool trim) {
    if (m->count != 2) {
        printf("The fill in operator can only work on 2 iterators! Got %i\n", m->count);    // reducers.c:110-113
        exit(1);
    }
    SelData *d = (SelData *) calloc(1, sizeof(SelData));
    d->multi = m; d->trim = trim;
    return newWiggleIterator(d, &fillin_pop, &sel_seek, m->default_values[1], 0);
}

WiggleIterator *TTestReduction(Multiset *s) {
    if (s->count != 2 || s->multis[0]->count < 3 || s->multis[1]->count < 3) {
        puts("The t-test function only works for two sets with enough elements to compute variance");   // setComparisons.c:125-128
        exit(1);
    }
    return make_set_reducer(s, WTAMD_OP_TTEST);
}

WiggleIterator *MWUReduction(Multiset *s) {
    if (s->count != 2 || s->multis[0]->count == 0 || s->multis[1]->count == 0) {
        puts("The Mann-Whitney U function only works for two non-empty sets");                          // setComparisons.c:374-377
        exit(1);
    }
    return make_set_reducer(s, WTAMD_OP_MWU);
}

}  // extern "C"
