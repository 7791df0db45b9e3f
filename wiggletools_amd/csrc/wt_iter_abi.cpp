// wt_iter_abi.cpp -- DROP-IN LAYER: the reference's own C API for the hot path
// (reference src/wiggletools.h:80-103, src/multiplexer.h:38-41, src/multiSet.h:32-33),
// implemented on top of the streaming pipeline of the bulk GPU engine (wtamd_pipe_*).
// Host-side C++ above the C ABI, mirroring the reference's names, argument meaning and
// error behaviour (message to stdout/stderr, then exit(1)).
//
// How a lazy pull API is fed to a bulk engine:
//   * a Drainer pops the N child iterators (each from ONE thread, as the reference
//     requires) straight into the PINNED staging arrays of a pipeline slot: one
//     chromosome, run starts in [lo, hi).  An interval that reaches `hi` (finish >= hi)
//     stays the current element of its source and is seen again by the next batch -- so a
//     track finishing exactly at the cut still marks the breakpoint there; one interval
//     beyond `hi` per track is included as a sentinel so the last run of the batch gets its
//     true finish (no seam artefacts, coordinates stay bit-exact).  Values are staged as
//     float32 until a value that is not float32-exact shows up (then float64, for good).
//   * a Feeder keeps the pipeline `depth` batches deep: while the consumer walks the runs
//     of batch k (pinned output), batch k+1 is on the GPU and the Drainer has already
//     filled and shipped it -- the overlap the reference gets from its producer threads
//     (bufferedReader.c:41-55,99-109), here across host, PCIe and GPU.
//   * popMultiplexer() walks the run tile the GPU materialised for the batch and keeps
//     every field of struct multiplexer_st coherent (other reference translation units
//     read them: mWigWriter.c:182-197, statistics.c:432-442).  The tile batches are sized
//     by runs x tracks, so memory stays bounded whatever the track count.
//   * a reducer constructor (MeanReduction, ...) takes the Multiplexer over: what the
//     Multiplexer had drained is pushed back and from then on whole batches go through
//     the FUSED multiplex+reduce kernel; the Multiplexer's per-run fields are then no
//     longer maintained (allowed when the reducer owns the multiplexer, which
//     is how commandParser.c builds them).
//   * TTestReduction / MWUReduction take over both Multiplexers of the Multiset and run
//     the two-sample kernels over the joint track list.
//
// There is no CPU evaluation path here: every run, aligned tile or reduced value comes
// from the HIP kernels.  If no GPU is present the first engine call fails and the
// process exits(1) with the engine's message.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/wiggletools_amd.h"
#include "wt_mapop.h"
#include "wt_bigwig_int.h"
#include <unistd.h>

#define WT_WEAK __attribute__((weak))

extern "C" WiggleIterator *NonOverlappingWiggleIterator(WiggleIterator *);     // unaryOps.c:98-103 (weak definition below)

namespace {

[[noreturn]] void die(const char *what) {
    fprintf(stderr, "wiggletools_amd: %s: %s\n", what, wtamd_last_error());
    exit(1);
}

double now_ms() {
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
const bool g_trace = getenv("WTAMD_TRACE") != nullptr;    // host-side timeline of the Feeder on stderr

int64_t env_i64(const char *name, int64_t dflt) {
    const char *e = getenv(name);
    if (!e || !*e) return dflt;
    const long long v = atoll(e);
    return v > 0 ? (int64_t) v : dflt;
}

struct Ivl {
    const char *chrom;      // interned
    int32_t start, finish;
    double value;
};

// Chromosome names are interned once per distinct name: child readers only promise a stable
// `char *` while they stay on a chromosome (bedReader.c:62-68 allocates a fresh string per
// chromosome, others may reuse a buffer), whereas the pointers this layer hands out are printed
// later by writer threads (wigWriter.c:175) and compared by identity (unaryOps.c:76).
struct Interner {
    std::vector<char *> names;
    const char *get(const char *raw) {
        for (char *n : names)
            if (strcmp(n, raw) == 0) return n;
        names.push_back(strdup(raw));
        return names.back();
    }
};

// Bulk side door of a child iterator built by this library (wtamd_ArrayReader, ...): first member of
// its `data`; recognised by its pop function (wt_bulk_pop).  peek() exposes the upcoming intervals
// of the current chromosome as SoA arrays, the first being the iterator's current element;
// advance() consumes k of them and refreshes the iterator's visible fields.
struct BulkSource {
    int64_t (*peek)(BulkSource *, const int32_t **start, const int32_t **finish, const float **value);
    void (*advance)(BulkSource *, WiggleIterator *, int64_t k);
    // true: the arrays peek() points into never move or change (wtamd_ArrayReader) -- they may be read
    // by the device later, where they lie; false: valid until the next advance() only (copied at once)
    bool stable;
};

void wt_bulk_pop(WiggleIterator *wi) {
    BulkSource *b = (BulkSource *) wi->data;
    b->advance(b, wi, 1);
}

}  // namespace

// the reference's bufferedReader.c, replaced: every reader built on it becomes a bulk source (csrc/wt_bufreader.h)
#include "wt_bufreader.h"

// The layer in the order its pieces build on each other (each header opens the anonymous namespace itself):
#include "wt_abi_common.h"
#include "wt_abi_feeder.h"
#include "wt_abi_reduce.h"
#include "wt_abi_readers.h"
#include "wt_abi_bwdev.h"
#include "wt_abi_ops.h"
#include "wt_abi_integrators.h"


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
    if (g_trace) fprintf(stderr, "[multiplexer] new %.3f\n", now_ms());
    MuxState *S = new MuxState();
    Multiplexer *m = newCoreMultiplexer(S, count, &mux_pop, &mux_seek);
    m->strict = strict;
    m->iters = (WiggleIterator **) calloc((size_t) count, sizeof(WiggleIterator *));
    S->fd.src.resize((size_t) count);
    for (int i = 0; i < count; i++) {
        m->iters[i] = NonOverlappingWiggleIterator(iters[i]);       // multiplexer.c:163
        m->default_values[i] = m->iters[i]->default_value;
        m->values[i] = m->iters[i]->default_value;
        // operator iterators of this library are peeled off: the raw child is drained, the chain runs on device
        WiggleIterator *raw = wt_unwrap_maps(m->iters[i], S->fd.src[i].chain);
        S->fd.src[i].it = raw;
        for (int k = 0; k < S->fd.src[i].chain.n_ops; k++) {
            const int op = S->fd.src[i].chain.op[k];
            if (op == WTAMD_MAP_LN || op == WTAMD_MAP_LOG || op >= WTAMD_MAP_GT) S->fd.src[i].drops = true;
        }
        if (raw->pop == &wt_bulk_pop) S->fd.src[i].bulk = (BulkSource *) raw->data;
        else S->fd.src[i].bulk = wt_bufreader_bulk(raw);        // a reader on this library's bufferedReader (csrc/wt_bufreader.h)
        S->fd.defaults.push_back(m->iters[i]->default_value);
    }
    popMultiplexer(m);                                              // primed like multiplexer.c:167
    if (g_trace) fprintf(stderr, "[multiplexer] primed %.3f\n", now_ms());
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

WiggleIterator *FTestReduction(Multiset *s) {
    // The reference's F-test (setComparisons.c:152-243) is broken -- its inner loops advance
    // `index` instead of `index2` (:183,199), undefined behaviour -- and is out of
    // scope here.  Exported so that commandParser.c:642-644 links; fails the reference's way.
    (void) s;
    puts("The F-test is not available in the wiggletools_amd engine (the reference implementation is undefined behaviour, setComparisons.c:183,199)");
    exit(1);
}

WiggleIterator *MWUReduction(Multiset *s) {
    if (s->count != 2 || s->multis[0]->count == 0 || s->multis[1]->count == 0) {
        puts("The Mann-Whitney U function only works for two non-empty sets");                          // setComparisons.c:374-377
        exit(1);
    }
    return make_set_reducer(s, WTAMD_OP_MWU);
}

// ---------------------------------------------------------------------------
// Bulk doors (include/wiggletools_amd.h)
// ---------------------------------------------------------------------------
WiggleIterator *wtamd_ArrayReader(int n_chrom, const char *const *chrom_names, const int64_t *seg_off,
                                  const int32_t *start, const int32_t *finish, const float *value,
                                  double default_value) {
    ArrReader *a = new (calloc(1, sizeof(ArrReader))) ArrReader();     // free()-able, like every iterator's data
    a->hdr.peek = &arr_peek;
    a->hdr.advance = &arr_advance;
    a->hdr.stable = true;
    a->n_chrom = n_chrom;
    a->names = (char **) calloc((size_t) (n_chrom > 0 ? n_chrom : 1), sizeof(char *));
    a->seg_off = (int64_t *) calloc((size_t) n_chrom + 1, sizeof(int64_t));
    for (int c = 0; c < n_chrom; c++) a->names[c] = strdup(chrom_names[c]);
    for (int c = 0; c <= n_chrom; c++) a->seg_off[c] = seg_off[c];
    a->start = start; a->finish = finish; a->value = value;
    a->c = -1; a->j = 0; a->end = 0;           // settle() moves to the first non-empty chromosome
    WiggleIterator *wi = (WiggleIterator *) calloc(1, sizeof(WiggleIterator));
    wi->data = a;
    wi->pop = &wt_bulk_pop;
    wi->seek = &arr_seek;
    wi->value = 1;
    wi->default_value = default_value;
    a->settle(wi);                             // a fresh iterator already holds its first element (wiggleIterator.c:32)
    return wi;
}

// The same arrays behind a reader written the way the reference writes its binary-file readers (bigWiggleReader.c:52-151,
// bamReader.c, bigBedReader.c): a producer thread pushes one interval at a time into the buffered reader, the iterator's
// pop is BufferedReaderPop.  It exists to exercise and to time the buffered reader's bulk door (csrc/wt_bufreader.h) with
// a producer that costs nothing but the protocol.
namespace {
struct BufArrReader {
    int n_chrom;
    char **names;
    int64_t *seg_off;
    const int32_t *start, *finish;
    const float *value;
    BufferedReaderData *buf;
    int only;                       // after seek(): this chromosome only, clipped to [win_start, win_finish)
    int32_t win_start, win_finish;
};

void *bufarr_produce(void *arg) {
    BufArrReader *a = (BufArrReader *) arg;
    for (int c = 0; c < a->n_chrom; c++) {
        if (a->only >= 0 && c != a->only) continue;
        for (int64_t j = a->seg_off[c]; j < a->seg_off[c + 1]; j++) {
            int32_t s = a->start[j], f = a->finish[j];
            if (a->only >= 0) {
                if (f <= a->win_start) continue;
                if (s >= a->win_finish) break;
                if (s < a->win_start) s = a->win_start;
                if (f > a->win_finish) f = a->win_finish;
            }
            if (pushValuesToBuffer(a->buf, a->names[c], s, f, (double) a->value[j])) return nullptr;
        }
    }
    endBufferedSignal(a->buf);
    return nullptr;
}

void bufarr_pop(WiggleIterator *wi) {
    BufArrReader *a = (BufArrReader *) wi->data;
    BufferedReaderPop(wi, a->buf);
}

void bufarr_seek(WiggleIterator *wi, const char *chrom, int start, int finish) {
    BufArrReader *a = (BufArrReader *) wi->data;
    killBufferedReader(a->buf);
    free(a->buf);
    a->buf = nullptr;
    a->only = a->n_chrom;           // unknown chromosome: nothing
    for (int c = 0; c < a->n_chrom; c++)
        if (strcmp(a->names[c], chrom) == 0) a->only = c;
    a->win_start = start; a->win_finish = finish;
    launchBufferedReader(&bufarr_produce, a, &a->buf);
    wi->done = 0;
    bufarr_pop(wi);
}
}  // namespace

WiggleIterator *wtamd_BufferedArrayReader(int n_chrom, const char *const *chrom_names, const int64_t *seg_off,
                                          const int32_t *start, const int32_t *finish, const float *value,
                                          double default_value) {
    BufArrReader *a = (BufArrReader *) calloc(1, sizeof(BufArrReader));
    a->n_chrom = n_chrom;
    a->names = (char **) calloc((size_t) (n_chrom > 0 ? n_chrom : 1), sizeof(char *));
    a->seg_off = (int64_t *) calloc((size_t) n_chrom + 1, sizeof(int64_t));
    for (int c = 0; c < n_chrom; c++) a->names[c] = strdup(chrom_names[c]);
    for (int c = 0; c <= n_chrom; c++) a->seg_off[c] = seg_off[c];
    a->start = start; a->finish = finish; a->value = value;
    a->only = -1;
    WiggleIterator *wi = (WiggleIterator *) calloc(1, sizeof(WiggleIterator));
    wi->data = a;
    wi->pop = &bufarr_pop;
    wi->seek = &bufarr_seek;
    wi->value = 1;
    wi->default_value = default_value;
    launchBufferedReader(&bufarr_produce, a, &a->buf);
    bufarr_pop(wi);
    return wi;
}

int64_t wtamd_iterator_next_block(WiggleIterator *wi, const char **chrom, const int32_t **start,
                                  const int32_t **finish, const double **value) {
    if (!wi || wi->pop != &red_pop) return -1;          // reducers of this library only
    RedState *R = red_state(wi);
    if (R->block_done) red_pop(wi);                     // the previous block emptied its batch: fetch the next
    if (wi->done) return 0;
    Feeder &F = R->fd;
    const int64_t first = R->cur - 1;                   // the iterator's current element
    const int64_t n = F.res.n_runs - first;
    if (chrom) *chrom = F.res_chrom;
    *start = F.res.start + first;
    *finish = F.res.finish + first;
    *value = F.res.value + first;
    R->cur = F.res.n_runs;
    R->block_done = true;
    return n;
}

WiggleIterator *wtamd_MapIterator(WiggleIterator *child, int map_op, double param) {
    if (map_op < 0 || map_op >= WTAMD_MAP_COUNT_) { puts("wtamd_MapIterator: unknown operator"); exit(1); }
    if ((map_op == WTAMD_MAP_LOG || map_op == WTAMD_MAP_EXPB) && !(param > 0)) { puts("wtamd_MapIterator: base / radix must be positive"); exit(1); }
    int depth = 1;
    for (WiggleIterator *w = child; w->pop == &map_pop; w = ((MapIter *) w->data)->child) depth++;
    if (depth > WTAMD_MAP_CHAIN_MAX) { puts("wtamd_MapIterator: operator chain too deep"); exit(1); }
    MapIter *m = (MapIter *) calloc(1, sizeof(MapIter));        // free()-able, like every iterator's data
    m->child = child; m->op = map_op; m->param = param;
    m->lg = (map_op == WTAMD_MAP_LOG || map_op == WTAMD_MAP_EXPB) ? log(param) : 1.0;
    WiggleIterator *wi = (WiggleIterator *) calloc(1, sizeof(WiggleIterator));
    wi->data = m;
    wi->pop = &map_pop;
    wi->seek = &map_seek;
    wi->value = 1;
    wi->overlaps = child->overlaps;
    wi->default_value = wtamd_map_default(map_op, param, child->default_value);
    map_settle(wi);
    return wi;
}

WiggleIterator *wtamd_BigWiggleReader(const char *path, int box) {
    wtamd_bw *bw = nullptr;
    static const bool trace_open = getenv("WTAMD_TRACE_OPEN") != nullptr;
    const double t_open0 = trace_open ? now_ms() : 0;
    if (wtamd_bw_open(path, &bw) != WTAMD_OK) exit(1);     // message printed (bigWiggleReader.c:116-118)
    const double t_open1 = trace_open ? now_ms() : 0;
    BwReader *r = new BwReader();
    BwHandle *h = (BwHandle *) calloc(1, sizeof(BwHandle));
    h->hdr.peek = &bw_peek;
    h->hdr.advance = &bw_advance;
    h->hdr.stable = false;
    h->r = r;
    r->bw = bw;
    r->box = box;
    r->p_box = box;
    for (int i = 0; i < wtamd_bw_n_chrom(bw); i++) r->names.push_back(wtamd_bw_chrom_name(bw, i));
    std::sort(r->names.begin(), r->names.end(), [](const std::string &a, const std::string &b) { return strcmp(a.c_str(), b.c_str()) < 0; });
    for (const std::string &n : r->names) r->cnames.push_back(strdup(n.c_str()));
    WiggleIterator *wi = (WiggleIterator *) calloc(1, sizeof(WiggleIterator));
    wi->data = h;
    wi->pop = &wt_bulk_pop;
    wi->seek = &bw_seek;
    wi->value = 1;
    wi->default_value = 0;                      // bigWiggleReader.c:150
    // priming (wiggleIterator.c:32): the first data block is decoded here, on the caller's thread -- 100 files used
    // to cost 100 thread starts and hand-shakes before the first run could be computed
    r->cur = 0;
    r->p_blocks = 1;
    bw_decode(r, r->buf[0]);
    if (r->failed) { fprintf(stderr, "wiggletools_amd: BigWig decode failed\n"); exit(1); }
    r->j = 0; r->end = r->buf[0].n;
    if (r->buf[0].chrom < 0) r->done = true;
    bw_settle(r, wi);
    if (trace_open) fprintf(stderr, "[reader] open %.3f ms, names + priming block %.3f ms\n", t_open1 - t_open0, now_ms() - t_open1);
    return wi;
}

// Releases what the reference's destroyWiggleIterator cannot know about: the file, the two decode buffers and the
// producer thread (joined).  The iterator itself stays the caller's (free() / destroyWiggleIterator as usual); it
// must not be popped, sought or handed to a reducer afterwards.  Readers that are never closed keep an idle thread
// and a file descriptor each for the life of the process, like the reference's (bigWiggleReader.c never joins).
int wtamd_BigWiggleReader_close(WiggleIterator *wi) {
    if (!wi || wi->pop != &wt_bulk_pop || !wi->data) return WTAMD_ERR_ARG;
    BwHandle *h = (BwHandle *) wi->data;
    if (h->hdr.peek != &bw_peek || !h->r) return WTAMD_ERR_ARG;
    BwReader *r = h->r;
    if (r->started) {
        { std::unique_lock<std::mutex> lk(r->mu); r->cv.wait(lk, [&] { return r->want_buf < 0; }); r->quit = true; }
        r->cv.notify_all();
        if (r->th.joinable()) r->th.join();
    }
    wtamd_bw_close(r->bw);
    bw_free(r->buf[0]); bw_free(r->buf[1]);
    // (the chromosome names stay: consumers may still hold the pointers they were handed)
    delete r;
    h->r = nullptr;
    wi->done = true;
    return WTAMD_OK;
}

WiggleIterator *wtamd_AUCIntegrator(WiggleIterator *wi) { return make_integrator(wi, nullptr, 0); }
WiggleIterator *wtamd_MeanIntegrator(WiggleIterator *wi) { return make_integrator(wi, nullptr, 1); }
WiggleIterator *wtamd_PearsonIntegrator(Multiplexer *multi) {
    if (multi->count != 2) { puts("wtamd_PearsonIntegrator: the Multiplexer must hold exactly two tracks"); exit(1); }
    return make_integrator(nullptr, multi, 2);
}

int wtamd_BigWiggleReaders(int n, const char *const *paths, int box, WiggleIterator **out) {
    if (n < 0 || (n > 0 && (!paths || !out))) return WTAMD_ERR_ARG;
    if (g_trace) fprintf(stderr, "[readers] open %d files %.3f\n", n, now_ms());
    // files opened side by side are about to feed a reducer: the HIP runtime starts up on a helper thread meanwhile
    if (n >= 2 && !(getenv("WTAMD_BW_DEVICE") && atoi(getenv("WTAMD_BW_DEVICE")) == 0)) wtamd_warmup_async();
    const int T = std::max(1, std::min({n, wt_usable_cores(), 16}));
    std::vector<std::thread> th;
    for (int w = 0; w < T; w++)
        th.emplace_back([=] { for (int i = w; i < n; i += T) out[i] = wtamd_BigWiggleReader(paths[i], box); });
    for (auto &t : th) t.join();
    if (g_trace) fprintf(stderr, "[readers] opened %.3f\n", now_ms());
    return WTAMD_OK;
}

int wtamd_iterator_compress_output(WiggleIterator *wi, int on) {
    if (!wi || wi->pop != &red_pop) return WTAMD_ERR_ARG;
    RedState *R = red_state(wi);
    R->fd.compress_on = on != 0;
    if (!R->fd.pipe) return R->fd.opened_once ? WTAMD_OK : WTAMD_ERR_ARG;
    int rc = WTAMD_OK;
    for (wtamd_pipe *q : R->fd.pipes) { const int r1 = wtamd_pipe_set_compress(q, on); if (r1 != WTAMD_OK) rc = r1; }
    return rc;
}

int wtamd_iterator_pipe_stats(WiggleIterator *wi, wtamd_pipe_stats *out) {
    if (!wi || !out || wi->pop != &red_pop) return WTAMD_ERR_ARG;
    RedState *R = red_state(wi);
    if (!R->fd.pipe) {
        if (!R->fd.opened_once) return WTAMD_ERR_ARG;
        *out = R->fd.last_stats;
        return WTAMD_OK;
    }
    R->fd.sum_stats(out);
    return WTAMD_OK;
}

int64_t wtamd_drain(WiggleIterator *wi, int64_t *covered_bp, double *value_sum) {
    int64_t n = 0, bp = 0;
    double acc = 0;
    while (!wi->done) {
        n++;
        bp += wi->finish - wi->start;
        if (wi->value == wi->value) acc += wi->value;
        wi->pop(wi);
    }
    if (covered_bp) *covered_bp = bp;
    if (value_sum) *value_sum = acc;
    return n;
}

}  // extern "C"
