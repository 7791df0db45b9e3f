/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Driver for the *compiled reference* (oracle/_ref/libwiggletools_ref.so, built
 * by oracle/Makefile straight from /root/reference/src without modification).
 * It feeds the reference's own newMultiplexer / ...Reduction / newMultiset with
 * array-backed WiggleIterators and records what the reference emits, so that
 *   (a) oracle/wt_oracle.c can be validated against the real thing, and
 *   (b) bench.py can time the real reference as cpu_baseline.kind="reference".
 *
 * The same harness also drives OUR drop-in library (it exports the same C API),
 * which is how the GPU parity tests of the drop-in layer read like calls into
 * the reference.
 *
 * The reference library is opened with dlopen(RTLD_LAZY): reference unaryOps.c
 * refers to BigWiggleReader/BamReader/... whose sources need libBigWig/htslib
 * (absent here); those symbols stay unresolved and are never called.
 *
 * Everything in this file is our own code; the only thing taken from the
 * reference is its ABI (struct layouts in include/wiggletools_amd.h).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/wiggletools_amd.h"

typedef struct {
    int32_t n_chrom, n_tracks;
    const int64_t *seg_off;
    const int32_t *start, *finish;
    const double *value;
    const double *defaults;
} wto_tracks;

/* function pointers into the reference library */
static void *g_lib;
static WiggleIterator *(*r_newWiggleIterator)(void *, void (*)(WiggleIterator *),
                                              void (*)(WiggleIterator *, const char *, int, int), double, wt_bool);
static Multiplexer *(*r_newMultiplexer)(WiggleIterator **, int, wt_bool);
static Multiset *(*r_newMultiset)(Multiplexer **, int);
static void (*r_popMultiplexer)(Multiplexer *);
static void (*r_popMultiset)(Multiset *);
static void (*r_seekMultiset)(Multiset *, const char *, int, int);
static WiggleIterator *(*r_TeeWiggleIterator)(WiggleIterator *, FILE *, wt_bool, wt_bool);
static Multiplexer *(*r_TeeMultiplexer)(Multiplexer *, FILE *, wt_bool, wt_bool);
static WiggleIterator *(*r_ArrayReader)(int, const char *const *, const int64_t *, const int32_t *, const int32_t *,
                                        const float *, double);
static WiggleIterator *(*r_BufferedArrayReader)(int, const char *const *, const int64_t *, const int32_t *, const int32_t *,
                                        const float *, double);
static int64_t (*r_next_block)(WiggleIterator *, const char **, const int32_t **, const int32_t **, const double **);
static int (*r_compress_output)(WiggleIterator *, int);
static WiggleIterator *(*r_MapIterator)(WiggleIterator *, int, double);     /* wtamd_MapIterator (tested library only) */
static int g_compress_mode;    /* 1: reducers handed to the writers are asked to merge their runs on the device first */
void ref_set_compress_mode(int on) { g_compress_mode = on; }
static void (*r_pop)(WiggleIterator *);
static void (*r_seek)(WiggleIterator *, const char *, int, int);
/* src/bufferedReader.h of the library under test (the reference's bufferedReader.o, or this repository's drop-in for it) */
typedef struct bufferedReaderData_st BufData;
static void (*r_launchBufferedReader)(void *(*)(void *), void *, BufData **);
static wt_bool (*r_pushValuesToBuffer)(BufData *, const char *, int, int, double);
static void (*r_endBufferedSignal)(BufData *);
static void (*r_killBufferedReader)(BufData *);
static void (*r_BufferedReaderPop)(WiggleIterator *, BufData *);
static WiggleIterator *(*r_SmartReader)(char *, wt_bool);
static WiggleIterator *(*r_AUCIntegrator)(WiggleIterator *);
static WiggleIterator *(*r_PearsonIntegrator)(Multiplexer *);
/* the tested library's fused integrator doors (wtamd_AUCIntegrator, ...) and its pipeline counters */
static WiggleIterator *(*r_door_auc)(WiggleIterator *);
static WiggleIterator *(*r_door_mean)(WiggleIterator *);
static WiggleIterator *(*r_door_pearson)(Multiplexer *);
static int (*r_pipe_stats)(WiggleIterator *, void *);
static WiggleIterator *(*r_CompressionWiggleIterator)(WiggleIterator *);
static WiggleIterator *(*r_ScaleWiggleIterator)(WiggleIterator *, double);
static WiggleIterator *(*r_ShiftWiggleIterator)(WiggleIterator *, double);
static WiggleIterator *(*r_NaturalLogWiggleIterator)(WiggleIterator *);
static WiggleIterator *(*r_LogWiggleIterator)(WiggleIterator *, double);
static WiggleIterator *(*r_NaturalExpWiggleIterator)(WiggleIterator *);
static WiggleIterator *(*r_ExpWiggleIterator)(WiggleIterator *, double);
static WiggleIterator *(*r_PowerWiggleIterator)(WiggleIterator *, double);
static WiggleIterator *(*r_AbsWiggleIterator)(WiggleIterator *);
static WiggleIterator *(*r_HighPassFilterWiggleIterator)(WiggleIterator *, double, wt_bool);
static WiggleIterator *(*r_reduction[10])(Multiplexer *);
static WiggleIterator *(*r_set_reduction[2])(Multiset *);      /* TTestReduction, MWUReduction (optional) */

static const char *k_red_names[10] = {
    "SumReduction", "ProductReduction", "MeanReduction", "VarianceReduction", "StdDevReduction",
    "EntropyReduction", "CVReduction", "MinReduction", "MaxReduction", "MedianReduction"
};

int ref_open(const char *path) {
    if (g_lib) return 0;
    g_lib = dlopen(path, RTLD_LAZY | RTLD_LOCAL);
    if (!g_lib) { fprintf(stderr, "ref_open: %s\n", dlerror()); return -1; }
#define BIND(var, name) do { *(void **) (&var) = dlsym(g_lib, name); \
        if (!var) { fprintf(stderr, "ref_open: missing %s\n", name); return -2; } } while (0)
    BIND(r_newWiggleIterator, "newWiggleIterator");
    BIND(r_newMultiplexer, "newMultiplexer");
    BIND(r_newMultiset, "newMultiset");
    BIND(r_popMultiplexer, "popMultiplexer");
    BIND(r_popMultiset, "popMultiset");
    BIND(r_pop, "pop");
    BIND(r_seek, "seek");
    for (int i = 0; i < 10; i++) BIND(r_reduction[i], k_red_names[i]);
#undef BIND
    /* optional symbols: the compiled reference lacks the set comparisons (GSL), the
     * drop-in library lacks readers / integrators it does not replace */
#define OPT(var, name) *(void **) (&var) = dlsym(g_lib, name)
    OPT(r_seekMultiset, "seekMultiset");
    OPT(r_launchBufferedReader, "launchBufferedReader");
    OPT(r_pushValuesToBuffer, "pushValuesToBuffer");
    OPT(r_endBufferedSignal, "endBufferedSignal");
    OPT(r_killBufferedReader, "killBufferedReader");
    OPT(r_BufferedReaderPop, "BufferedReaderPop");
    OPT(r_TeeWiggleIterator, "TeeWiggleIterator");
    OPT(r_TeeMultiplexer, "TeeMultiplexer");
    OPT(r_ArrayReader, "wtamd_ArrayReader");
    OPT(r_BufferedArrayReader, "wtamd_BufferedArrayReader");
    OPT(r_next_block, "wtamd_iterator_next_block");
    OPT(r_compress_output, "wtamd_iterator_compress_output");
    OPT(r_MapIterator, "wtamd_MapIterator");
    OPT(r_SmartReader, "SmartReader");
    OPT(r_AUCIntegrator, "AUCIntegrator");
    OPT(r_PearsonIntegrator, "PearsonIntegrator");
    OPT(r_door_auc, "wtamd_AUCIntegrator");
    OPT(r_door_mean, "wtamd_MeanIntegrator");
    OPT(r_door_pearson, "wtamd_PearsonIntegrator");
    OPT(r_pipe_stats, "wtamd_iterator_pipe_stats");
    OPT(r_CompressionWiggleIterator, "CompressionWiggleIterator");
    OPT(r_ScaleWiggleIterator, "ScaleWiggleIterator");
    OPT(r_ShiftWiggleIterator, "ShiftWiggleIterator");
    OPT(r_NaturalLogWiggleIterator, "NaturalLogWiggleIterator");
    OPT(r_LogWiggleIterator, "LogWiggleIterator");
    OPT(r_NaturalExpWiggleIterator, "NaturalExpWiggleIterator");
    OPT(r_ExpWiggleIterator, "ExpWiggleIterator");
    OPT(r_PowerWiggleIterator, "PowerWiggleIterator");
    OPT(r_AbsWiggleIterator, "AbsWiggleIterator");
    OPT(r_HighPassFilterWiggleIterator, "HighPassFilterWiggleIterator");
    OPT(r_set_reduction[0], "TTestReduction");
    OPT(r_set_reduction[1], "MWUReduction");
#undef OPT
    return 0;
}

/* ------------------------------------------------------------------ */
/* Array-backed child iterator                                         */
/* ------------------------------------------------------------------ */
typedef struct {
    const wto_tracks *t;
    char **names;      /* stable char* per chromosome (reference compares pointers, unaryOps.c:76) */
    int track;
    int c;             /* current chromosome */
    int64_t j;         /* next interval index inside (c, track) */
    /* optional seek window */
    int have_win; int win_c; int win_start, win_finish;
    /* held: no data until the first seek -- what the reference's readers do under the CLI's `seek`
     * (holdFire, commandParser.c:615-624, bufferedReader.c:164-166) */
    int hold;
    /* child mode 3: the iterator hands out ONE name buffer and rewrites it per chromosome (allowed: readers only promise a
     * stable char* while they stay on a chromosome; the reference's multiplexer compares by strcmp, multiplexer.c:56) */
    char namebuf[64];
} arr_iter;

static int g_hold;     /* children made from now on are held until their first seek */
static int g_child_mode;   /* 1: children are the tested library's own bulk-capable wtamd_ArrayReader; 2: even tracks only;
                            * 3: plain children that reuse one name buffer across chromosomes */
static int g_block_mode;   /* 1: reducer output is taken through wtamd_iterator_next_block */

static void arr_pop(WiggleIterator *wi) {
    arr_iter *a = (arr_iter *) wi->data;
    const wto_tracks *t = a->t;
    if (a->hold && !a->have_win) { wi->done = 1; return; }
    for (;;) {
        if (a->c >= t->n_chrom) { wi->done = 1; return; }
        int64_t seg = (int64_t) a->c * t->n_tracks + a->track;
        int64_t lo = t->seg_off[seg], hi = t->seg_off[seg + 1];
        if (a->j < lo) a->j = lo;
        if (a->j >= hi) { a->c++; a->j = -1; if (a->have_win) { wi->done = 1; return; } continue; }
        int32_t s = t->start[a->j], f = t->finish[a->j];
        if (a->have_win) {
            if (f <= a->win_start) { a->j++; continue; }
            if (s >= a->win_finish) { wi->done = 1; return; }
            if (s < a->win_start) s = a->win_start;
            if (f > a->win_finish) f = a->win_finish;
        }
        if (g_child_mode == 3) { strncpy(a->namebuf, a->names[a->c], sizeof a->namebuf - 1); wi->chrom = a->namebuf; }
        else wi->chrom = a->names[a->c];
        wi->start = s; wi->finish = f;
        wi->value = t->value[a->j];
        a->j++;
        return;
    }
}

static void arr_seek(WiggleIterator *wi, const char *chrom, int start, int finish) {
    arr_iter *a = (arr_iter *) wi->data;
    a->have_win = 1; a->win_start = start; a->win_finish = finish;
    a->c = a->t->n_chrom; a->j = -1;
    for (int c = 0; c < a->t->n_chrom; c++)
        if (!strcmp(a->names[c], chrom)) { a->c = c; break; }
    wi->done = 0;
    arr_pop(wi);
}

void ref_set_modes(int child_mode, int block_mode) { g_child_mode = child_mode; g_block_mode = block_mode; }

/* One track of the [chrom][track] layout as contiguous float32 SoA arrays for wtamd_ArrayReader. */
static WiggleIterator *make_array_child(const wto_tracks *t, char **names, int track) {
    int64_t n = 0;
    for (int c = 0; c < t->n_chrom; c++) n += t->seg_off[(int64_t) c * t->n_tracks + track + 1] - t->seg_off[(int64_t) c * t->n_tracks + track];
    int64_t *so = (int64_t *) calloc((size_t) t->n_chrom + 1, sizeof(int64_t));
    int32_t *s = (int32_t *) malloc(sizeof(int32_t) * (size_t) (n + 1)), *f = (int32_t *) malloc(sizeof(int32_t) * (size_t) (n + 1));
    float *v = (float *) malloc(sizeof(float) * (size_t) (n + 1));
    int64_t k = 0;
    for (int c = 0; c < t->n_chrom; c++) {
        int64_t lo = t->seg_off[(int64_t) c * t->n_tracks + track], hi = t->seg_off[(int64_t) c * t->n_tracks + track + 1];
        so[c] = k;
        for (int64_t g = lo; g < hi; g++, k++) { s[k] = t->start[g]; f[k] = t->finish[g]; v[k] = (float) t->value[g]; }
    }
    so[t->n_chrom] = k;
    if (g_child_mode == 5) return r_BufferedArrayReader(t->n_chrom, (const char *const *) names, so, s, f, v, t->defaults[track]);
    return r_ArrayReader(t->n_chrom, (const char *const *) names, so, s, f, v, t->defaults[track]);
}

/* `map <op>`: every child wrapped in one operator iterator -- the reference's own constructors
 * (unaryOps.c; commandParser.c:115-211 builds lt / lte as scale -1, gt -x) or, in the tested library,
 * wtamd_MapIterator (whose chains then run on the device inside the pipeline). */
static int g_map_op = -1;
static double g_map_param;
void ref_set_map(int map_op, double param) { g_map_op = map_op; g_map_param = param; }

static WiggleIterator *wrap_map(WiggleIterator *c, int map_op, double param) {
    if (r_MapIterator) return r_MapIterator(c, map_op, param);
    switch (map_op) {
    case 0: return r_ScaleWiggleIterator(c, param);
    case 1: return r_ShiftWiggleIterator(c, param);
    case 2: return r_NaturalLogWiggleIterator(c);
    case 3: return r_LogWiggleIterator(c, param);
    case 4: return r_NaturalExpWiggleIterator(c);
    case 5: return r_ExpWiggleIterator(c, param);
    case 6: return r_PowerWiggleIterator(c, param);
    case 7: return r_AbsWiggleIterator(c);
    case 8: return r_HighPassFilterWiggleIterator(c, param, 0);
    case 9: return r_HighPassFilterWiggleIterator(c, param, 1);
    case 10: return r_HighPassFilterWiggleIterator(r_ScaleWiggleIterator(c, -1), -param, 0);
    case 11: return r_HighPassFilterWiggleIterator(r_ScaleWiggleIterator(c, -1), -param, 1);
    default: return c;
    }
}

/* ---- child mode 4: a stand-in for the reference's binary-file readers, written against src/bufferedReader.h the way
 * bigWiggleReader.c is (:85-150): a reader THREAD pushes the track's intervals into the library's block buffer
 * (pushValuesToBuffer), the iterator's pop is BufferedReaderPop, seek kills + frees the buffer, relaunches the thread on
 * the region and skips / clips like BigWiggleReaderSeek (:125-145).  (The real readers need libBigWig / htslib.) */
typedef struct {
    const wto_tracks *t;
    char **names;
    int track;
    BufData *buf;
    const char *chrom;      /* region after seek (NULL: everything) */
    int start, stop;
} buf_reader;

static void *buf_reader_thread(void *ptr) {
    buf_reader *d = (buf_reader *) ptr;
    const wto_tracks *t = d->t;
    for (int c = 0; c < t->n_chrom; c++) {
        if (d->chrom && strcmp(d->names[c], d->chrom)) continue;
        const int64_t seg = (int64_t) c * t->n_tracks + d->track;
        for (int64_t j = t->seg_off[seg]; j < t->seg_off[seg + 1]; j++) {
            int s = t->start[j], f = t->finish[j];
            if (d->chrom) {         /* a region query hands back the overlapping intervals, boxed into it (:42-44) */
                if (f <= d->start || s >= d->stop) continue;
                if (s < d->start) s = d->start;
                if (f > d->stop) f = d->stop;
            }
            if (r_pushValuesToBuffer(d->buf, d->names[c], s, f, t->value[j])) return NULL;      /* killed */
        }
    }
    r_endBufferedSignal(d->buf);
    return NULL;
}

static void buf_reader_pop(WiggleIterator *wi) {
    buf_reader *d = (buf_reader *) wi->data;
    r_BufferedReaderPop(wi, d->buf);
}

static void buf_reader_seek(WiggleIterator *wi, const char *chrom, int start, int finish) {
    buf_reader *d = (buf_reader *) wi->data;
    if (d->buf) {
        r_killBufferedReader(d->buf);
        free(d->buf);
        d->buf = NULL;
    }
    d->chrom = chrom; d->start = start; d->stop = finish;
    r_launchBufferedReader(&buf_reader_thread, d, &d->buf);
    wi->done = 0;
    buf_reader_pop(wi);
    while (!wi->done && (strcmp(wi->chrom, chrom) < 0 || (strcmp(chrom, wi->chrom) == 0 && wi->finish <= start))) buf_reader_pop(wi);
    if (!wi->done && strcmp(chrom, wi->chrom) == 0 && wi->start < start) wi->start = start;
}

static WiggleIterator *make_buffered_child(const wto_tracks *t, char **names, int track) {
    buf_reader *d = (buf_reader *) calloc(1, sizeof(buf_reader));
    d->t = t; d->names = names; d->track = track;
    if (!g_hold) r_launchBufferedReader(&buf_reader_thread, d, &d->buf);       /* (held: launched by the first seek) */
    return r_newWiggleIterator(d, buf_reader_pop, buf_reader_seek, t->defaults[track], 0);
}

static WiggleIterator *make_plain_child(const wto_tracks *t, char **names, int track);
static WiggleIterator *make_child(const wto_tracks *t, char **names, int track) {
    WiggleIterator *c = make_plain_child(t, names, track);
    return g_map_op >= 0 ? wrap_map(c, g_map_op, g_map_param) : c;
}

static WiggleIterator *make_plain_child(const wto_tracks *t, char **names, int track) {
    if (r_ArrayReader && (g_child_mode == 1 || (g_child_mode == 2 && (track & 1) == 0))) return make_array_child(t, names, track);
    if (g_child_mode == 4 && r_launchBufferedReader) return make_buffered_child(t, names, track);
    if (g_child_mode == 5 && r_BufferedArrayReader) return make_array_child(t, names, track);     /* the library's own reader on its buffered reader */
    arr_iter *a = (arr_iter *) calloc(1, sizeof(arr_iter));
    a->t = t; a->names = names; a->track = track; a->c = 0; a->j = -1;
    a->hold = g_hold;
    return r_newWiggleIterator(a, arr_pop, arr_seek, t->defaults[track], 0);
}

static char **make_names(int n_chrom) {
    char **names = (char **) calloc((size_t) n_chrom, sizeof(char *));
    for (int c = 0; c < n_chrom; c++) {
        names[c] = (char *) malloc(16);
        snprintf(names[c], 16, "c%05d", c);    /* strcmp order == index order */
    }
    return names;
}

static int name_to_index(const char *name) { return atoi(name + 1); }

static Multiplexer *make_multiplexer(const wto_tracks *t, char **names, int lo, int hi, int strict) {
    int n = hi - lo;
    WiggleIterator **iters = (WiggleIterator **) calloc((size_t) n, sizeof(WiggleIterator *));
    for (int i = 0; i < n; i++) iters[i] = make_child(t, names, lo + i);
    Multiplexer *m = r_newMultiplexer(iters, n, (wt_bool) (strict != 0));
    free(iters);   /* newMultiplexer copies the array (multiplexer.c:160-166) */
    return m;
}

/* The reference's writers over the tested library's reducer / Multiplexer: `write` / `write_bg`
 * (TeeWiggleIterator, wigWriter.c:261-276: CompressionWiggleIterator in front unless bedGraph, a
 * writer thread printing 10 000-entry blocks) and `mwrite` / `mwrite_bg` (TeeMultiplexer,
 * mWigWriter.c:286-327, which reads struct multiplexer_st fields directly, :182-197).  Returns the
 * number of pops, < 0 when the library lacks the writers. */
int64_t ref_write_reduce(const wto_tracks *t, int op, unsigned flags, const char *path, int bedgraph) {
    if (!g_lib || op < 0 || op > 9 || !r_TeeWiggleIterator) return -2;
    FILE *f = fopen(path, "w");
    if (!f) return -3;
    char **names = make_names(t->n_chrom);
    Multiplexer *m = make_multiplexer(t, names, 0, t->n_tracks, flags & 1u);
    WiggleIterator *red = r_reduction[op](m);
    if (g_compress_mode && r_compress_output && !bedgraph && r_compress_output(red, 1) != 0) return -7;
    WiggleIterator *w = r_TeeWiggleIterator(red, f, (wt_bool) bedgraph, 0);
    int64_t n = 0;
    while (!w->done) { n++; r_pop(w); }
    fclose(f);
    return n;
}

int64_t ref_mwrite(const wto_tracks *t, unsigned flags, const char *path, int bedgraph) {
    if (!g_lib || !r_TeeMultiplexer) return -2;
    FILE *f = fopen(path, "w");
    if (!f) return -3;
    char **names = make_names(t->n_chrom);
    Multiplexer *m = r_TeeMultiplexer(make_multiplexer(t, names, 0, t->n_tracks, flags & 1u), f, (wt_bool) bedgraph, 0);
    int64_t n = 0;
    while (!m->done) { n++; r_popMultiplexer(m); }
    fclose(f);
    return n;
}

/* Consumer bulk door: a few plain pops first (the two protocols mix), then whole blocks. */
static int64_t take_blocks(WiggleIterator *r, int64_t cap, int32_t *o_chrom, int32_t *o_start, int32_t *o_finish, double *o_value) {
    int64_t n = 0;
    for (int k = 0; k < 3 && !r->done; k++) {
        if (n >= cap) return -1;
        o_chrom[n] = name_to_index(r->chrom); o_start[n] = r->start; o_finish[n] = r->finish; o_value[n] = r->value;
        n++;
        r_pop(r);
    }
    for (;;) {
        const char *c; const int32_t *s, *f; const double *v;
        int64_t m = r_next_block(r, &c, &s, &f, &v);
        if (m < 0) return -5;
        if (m == 0) break;
        if (n + m > cap) return -1;
        for (int64_t q = 0; q < m; q++) { o_chrom[n] = name_to_index(c); o_start[n] = s[q]; o_finish[n] = f[q]; o_value[n] = v[q]; n++; }
    }
    return r->done ? n : -6;
}

/* Runs the reference reducer `op` (0..9) and records every run it emits. */
int64_t ref_reduce(const wto_tracks *t, int op, unsigned flags, int64_t cap,
                   int32_t *o_chrom, int32_t *o_start, int32_t *o_finish, double *o_value) {
    if (!g_lib || op < 0 || op > 9) return -2;
    char **names = make_names(t->n_chrom);
    Multiplexer *m = make_multiplexer(t, names, 0, t->n_tracks, flags & 1u);
    WiggleIterator *r = r_reduction[op](m);
    if (g_block_mode && r_next_block) return take_blocks(r, cap, o_chrom, o_start, o_finish, o_value);
    int64_t n = 0;
    while (!r->done) {
        if (n >= cap) return -1;
        o_chrom[n] = name_to_index(r->chrom);
        o_start[n] = r->start; o_finish[n] = r->finish; o_value[n] = r->value;
        n++;
        r_pop(r);
    }
    return n;   /* everything is leaked by design, like the reference */
}

/* Same after seek(chrom_index, start, finish) on the reducer (reducers.c:25-29). */
int64_t ref_reduce_seek(const wto_tracks *t, int op, unsigned flags, int chrom, int start, int finish,
                        int64_t cap, int32_t *o_chrom, int32_t *o_start, int32_t *o_finish, double *o_value) {
    if (!g_lib || op < 0 || op > 9) return -2;
    char **names = make_names(t->n_chrom);
    Multiplexer *m = make_multiplexer(t, names, 0, t->n_tracks, flags & 1u);
    WiggleIterator *r = r_reduction[op](m);
    r_seek(r, names[chrom], start, finish);
    int64_t n = 0;
    while (!r->done) {
        if (n >= cap) return -1;
        o_chrom[n] = name_to_index(r->chrom);
        o_start[n] = r->start; o_finish[n] = r->finish; o_value[n] = r->value;
        n++;
        r_pop(r);
    }
    return n;
}

/* The CLI's `seek chr s f <reducer>` (commandParser.c:615-624): children are HELD (no data) while
 * the Multiplexer(s) and the reducer are constructed, then seek() is called on the reducer
 * (reducers.c:25-29 / setComparisons.c:25-29).  op 0..9 one-sample, 10/11 two-sample (n_set0). */
int64_t ref_reduce_seek_held(const wto_tracks *t, int op, int n_set0, unsigned flags, int chrom, int start, int finish,
                             int64_t cap, int32_t *o_chrom, int32_t *o_start, int32_t *o_finish, double *o_value) {
    if (!g_lib || op < 0 || op > 11) return -2;
    if (op >= 10 && !r_set_reduction[op - 10]) return -2;
    char **names = make_names(t->n_chrom);
    WiggleIterator *r;
    g_hold = 1;
    if (op < 10) {
        Multiplexer *m = make_multiplexer(t, names, 0, t->n_tracks, flags & 1u);
        r = r_reduction[op](m);
    } else {
        Multiplexer **ms = (Multiplexer **) calloc(2, sizeof(Multiplexer *));
        ms[0] = make_multiplexer(t, names, 0, n_set0, flags & 1u);
        ms[1] = make_multiplexer(t, names, n_set0, t->n_tracks, flags & 2u);
        r = r_set_reduction[op - 10](r_newMultiset(ms, 2));
    }
    g_hold = 0;
    if (!r->done) return -4;            /* held children: nothing before the seek */
    r_seek(r, names[chrom], start, finish);
    int64_t n = 0;
    while (!r->done) {
        if (n >= cap) return -1;
        o_chrom[n] = name_to_index(r->chrom);
        o_start[n] = r->start; o_finish[n] = r->finish; o_value[n] = r->value;
        n++;
        r_pop(r);
    }
    return n;
}

/* seekMultiset (multiSet.c:103-113) over two Multiplexers of held children; records the runs
 * where both sets are in play, like ref_multiset. */
int64_t ref_multiset_seek_held(const wto_tracks *t, int n_set0, unsigned flags, int chrom, int start, int finish,
                               int64_t cap, int32_t *o_chrom, int32_t *o_start, int32_t *o_finish,
                               double *o_tile, uint8_t *o_inplay) {
    if (!g_lib || !r_seekMultiset) return -2;
    char **names = make_names(t->n_chrom);
    Multiplexer **ms = (Multiplexer **) calloc(2, sizeof(Multiplexer *));
    g_hold = 1;
    ms[0] = make_multiplexer(t, names, 0, n_set0, flags & 1u);
    ms[1] = make_multiplexer(t, names, n_set0, t->n_tracks, flags & 2u);
    Multiset *S = r_newMultiset(ms, 2);
    g_hold = 0;
    if (!S->done) return -4;
    r_seekMultiset(S, names[chrom], start, finish);
    int N = t->n_tracks;
    int64_t n = 0;
    while (!S->done) {
        if (S->inplay[0] && S->inplay[1]) {
            if (n >= cap) return -1;
            o_chrom[n] = name_to_index(S->chrom);
            o_start[n] = S->start; o_finish[n] = S->finish;
            for (int k = 0; k < 2; k++) {
                int base = k ? n_set0 : 0;
                for (int i = 0; i < ms[k]->count; i++) {
                    o_tile[n * N + base + i] = S->values[k][i];
                    o_inplay[n * N + base + i] = (uint8_t) ms[k]->inplay[i];
                }
            }
            n++;
        }
        r_popMultiset(S);
    }
    return n;
}

/* Steps the reference Multiplexer and dumps values[]/inplay[] per run. */
int64_t ref_multiplex(const wto_tracks *t, unsigned flags, int64_t cap,
                      int32_t *o_chrom, int32_t *o_start, int32_t *o_finish,
                      double *o_tile, uint8_t *o_inplay) {
    if (!g_lib) return -2;
    char **names = make_names(t->n_chrom);
    Multiplexer *m = make_multiplexer(t, names, 0, t->n_tracks, flags & 1u);
    int N = t->n_tracks;
    int64_t n = 0;
    while (!m->done) {
        if (n >= cap) return -1;
        o_chrom[n] = name_to_index(m->chrom);
        o_start[n] = m->start; o_finish[n] = m->finish;
        for (int i = 0; i < N; i++) {
            o_tile[n * N + i] = m->values[i];
            o_inplay[n * N + i] = (uint8_t) m->inplay[i];
        }
        n++;
        r_popMultiplexer(m);
    }
    return n;
}

/* Steps the reference Multiset over two Multiplexers (tracks [0,n_set0) and the
 * rest); records the runs where both are in play (setComparisons.c:48-54) with
 * the per-track values/inplay the two-sample reducers would read. */
int64_t ref_multiset(const wto_tracks *t, int n_set0, unsigned flags, int64_t cap,
                     int32_t *o_chrom, int32_t *o_start, int32_t *o_finish,
                     double *o_tile, uint8_t *o_inplay) {
    if (!g_lib) return -2;
    char **names = make_names(t->n_chrom);
    Multiplexer **ms = (Multiplexer **) calloc(2, sizeof(Multiplexer *));
    ms[0] = make_multiplexer(t, names, 0, n_set0, flags & 1u);
    ms[1] = make_multiplexer(t, names, n_set0, t->n_tracks, flags & 2u);
    Multiset *S = r_newMultiset(ms, 2);
    int N = t->n_tracks;
    int64_t n = 0;
    while (!S->done) {
        if (S->inplay[0] && S->inplay[1]) {
            if (n >= cap) return -1;
            o_chrom[n] = name_to_index(S->chrom);
            o_start[n] = S->start; o_finish[n] = S->finish;
            for (int k = 0; k < 2; k++) {
                int base = k ? n_set0 : 0;
                for (int i = 0; i < ms[k]->count; i++) {
                    o_tile[n * N + base + i] = S->values[k][i];
                    o_inplay[n * N + base + i] = (uint8_t) ms[k]->inplay[i];
                }
            }
            n++;
        }
        r_popMultiset(S);
    }
    return n;
}

/* Two-sample reducer (op 10 = ttest, 11 = MWU) through newMultiset; only for libraries
 * that export TTestReduction / MWUReduction. */
int64_t ref_reduce2(const wto_tracks *t, int op, int n_set0, unsigned flags, int64_t cap,
                    int32_t *o_chrom, int32_t *o_start, int32_t *o_finish, double *o_value) {
    if (!g_lib || op < 10 || op > 11 || !r_set_reduction[op - 10]) return -2;
    char **names = make_names(t->n_chrom);
    Multiplexer **ms = (Multiplexer **) calloc(2, sizeof(Multiplexer *));
    ms[0] = make_multiplexer(t, names, 0, n_set0, flags & 1u);
    ms[1] = make_multiplexer(t, names, n_set0, t->n_tracks, flags & 2u);
    Multiset *S = r_newMultiset(ms, 2);
    WiggleIterator *r = r_set_reduction[op - 10](S);
    int64_t n = 0;
    while (!r->done) {
        if (n >= cap) return -1;
        o_chrom[n] = name_to_index(r->chrom);
        o_start[n] = r->start; o_finish[n] = r->finish; o_value[n] = r->value;
        n++;
        r_pop(r);
    }
    return n;
}

/* Default value the reference reducer ctor computes. */
double ref_reducer_default(int op, int n, const double *defaults) {
    if (!g_lib || op < 0 || op > 9) return NAN;
    int64_t *off = (int64_t *) calloc((size_t) n + 1, sizeof(int64_t));
    wto_tracks t = { 1, n, off, NULL, NULL, NULL, defaults };
    char **names = make_names(1);
    Multiplexer *m = make_multiplexer(&t, names, 0, n, 0);
    WiggleIterator *r = r_reduction[op](m);
    return r->default_value;
}

/* AUC of the reference reducer output (statistics.c:103-120). */
double ref_auc_of_reduce(const wto_tracks *t, int op, unsigned flags) {
    if (!g_lib || op < 0 || op > 9 || !r_AUCIntegrator) return NAN;
    char **names = make_names(t->n_chrom);
    Multiplexer *m = make_multiplexer(t, names, 0, t->n_tracks, flags & 1u);
    WiggleIterator *a = r_AUCIntegrator(r_reduction[op](m));
    while (!a->done) r_pop(a);
    return *(double *) a->data;
}

/* Pearson of tracks 0 and 1 (statistics.c:414-465). */
/* The tested library's integrator doors: kind 0 AUC, 1 mean over reducer `op`; 2 Pearson over the 2-track
 * Multiplexer.  info[0] = pops the integrator needed, info[1] = bytes the pipeline shipped device -> host,
 * info[2] = runs the reducer computed (kinds 0 / 1; -1 when the library has no counters). */
double ref_door_integrate(const wto_tracks *t, int op, unsigned flags, int kind, int64_t *info) {
    info[0] = info[1] = info[2] = -1;
    if (!g_lib) return NAN;
    char **names = make_names(t->n_chrom);
    WiggleIterator *a = NULL, *r = NULL;
    if (kind == 2) {
        if (t->n_tracks != 2 || !r_door_pearson) return NAN;
        a = r_door_pearson(make_multiplexer(t, names, 0, 2, 0));
    } else {
        if (op < 0 || op > 9 || !(kind ? r_door_mean : r_door_auc)) return NAN;
        r = r_reduction[op](make_multiplexer(t, names, 0, t->n_tracks, flags & 1u));
        a = (kind ? r_door_mean : r_door_auc)(r);
    }
    int64_t pops = 0;
    while (!a->done) { r_pop(a); pops++; }
    info[0] = pops;
    if (r && r_pipe_stats) {
        int64_t st[32];
        memset(st, 0, sizeof st);
        if (r_pipe_stats(r, st) == 0) { info[1] = st[5]; info[2] = st[2]; }    /* wtamd_pipe_stats: d2h_bytes, runs */
    }
    return *(double *) a->data;
}

/* The same doors driven the way `apply` drives an integrator (apply.c: one seek per region, pop to the end, read the
 * value): `pre_pops` pops first (a seek MID-STREAM: the source is primed and has batches under way), then for every
 * region (chrom index, start, finish) seek + pop until done.  out[0] = the value before the first seek, out[1 + k] =
 * the value after region k (the sums go on across seeks: statistics.c:38-43,84-88,406-410).  Returns 0, < 0 on error. */
int ref_door_integrate_seek(const wto_tracks *t, int op, unsigned flags, int kind, int pre_pops, int n_regions,
                            const int32_t *regions, double *out) {
    if (!g_lib) return -2;
    char **names = make_names(t->n_chrom);
    WiggleIterator *a = NULL;
    if (kind == 2) {
        if (t->n_tracks != 2 || !r_door_pearson) return -3;
        a = r_door_pearson(make_multiplexer(t, names, 0, 2, 0));
    } else {
        if (op < 0 || op > 9 || !(kind ? r_door_mean : r_door_auc)) return -3;
        a = (kind ? r_door_mean : r_door_auc)(r_reduction[op](make_multiplexer(t, names, 0, t->n_tracks, flags & 1u)));
    }
    for (int k = 0; k < pre_pops && !a->done; k++) r_pop(a);
    out[0] = *(double *) a->data;
    for (int k = 0; k < n_regions; k++) {
        r_seek(a, names[regions[3 * k]], regions[3 * k + 1], regions[3 * k + 2]);
        while (!a->done) r_pop(a);
        out[1 + k] = *(double *) a->data;
    }
    return 0;
}

double ref_pearson(const wto_tracks *t) {
    if (!g_lib || t->n_tracks != 2 || !r_PearsonIntegrator) return NAN;
    char **names = make_names(t->n_chrom);
    Multiplexer *m = make_multiplexer(t, names, 0, 2, 0);
    WiggleIterator *p = r_PearsonIntegrator(m);
    while (!p->done) r_pop(p);
    return *(double *) p->data;
}

/* `map`-able unary operators (unaryOps.c:650-949) over ONE track: drains the reference's operator
 * iterator wrapped around the array-backed child.  map_op: 0 scale, 1 offset, 2 ln, 3 log base
 * param, 4 exp (natural), 5 exp radix param, 6 pow, 7 abs, 8 gt, 9 gte, 10 lt, 11 lte (built as
 * commandParser.c:180-199 builds them).  *o_default receives the operator
 * iterator's default_value.  Returns the number of intervals or < 0. */
int64_t ref_map(const wto_tracks *t, int track, int map_op, double param, int64_t cap,
                int32_t *o_chrom, int32_t *o_start, int32_t *o_finish, double *o_value, double *o_default) {
    if (!g_lib || !r_ScaleWiggleIterator) return -2;
    char **names = make_names(t->n_chrom);
    WiggleIterator *c = make_plain_child(t, names, track), *w = NULL;
    switch (map_op) {
    case 0: w = r_ScaleWiggleIterator(c, param); break;
    case 1: w = r_ShiftWiggleIterator(c, param); break;
    case 2: w = r_NaturalLogWiggleIterator(c); break;
    case 3: w = r_LogWiggleIterator(c, param); break;
    case 4: w = r_NaturalExpWiggleIterator(c); break;
    case 5: w = r_ExpWiggleIterator(c, param); break;
    case 6: w = r_PowerWiggleIterator(c, param); break;
    case 7: w = r_AbsWiggleIterator(c); break;
    case 8: w = r_HighPassFilterWiggleIterator(c, param, 0); break;
    case 9: w = r_HighPassFilterWiggleIterator(c, param, 1); break;
    case 10: w = r_HighPassFilterWiggleIterator(r_ScaleWiggleIterator(c, -1), -param, 0); break;
    case 11: w = r_HighPassFilterWiggleIterator(r_ScaleWiggleIterator(c, -1), -param, 1); break;
    default: return -3;
    }
    if (o_default) *o_default = w->default_value;
    int64_t n = 0;
    while (!w->done) {
        if (n >= cap) return -1;
        o_chrom[n] = name_to_index(w->chrom);
        o_start[n] = w->start; o_finish[n] = w->finish; o_value[n] = w->value;
        n++;
        r_pop(w);
    }
    return n;
}

/* Compression (unaryOps.c:235-263) of the reference reducer output. */
int64_t ref_reduce_compressed(const wto_tracks *t, int op, unsigned flags, int64_t cap,
                              int32_t *o_chrom, int32_t *o_start, int32_t *o_finish, double *o_value) {
    if (!g_lib || op < 0 || op > 9 || !r_CompressionWiggleIterator) return -2;
    char **names = make_names(t->n_chrom);
    Multiplexer *m = make_multiplexer(t, names, 0, t->n_tracks, flags & 1u);
    WiggleIterator *r = r_CompressionWiggleIterator(r_reduction[op](m));
    int64_t n = 0;
    while (!r->done) {
        if (n >= cap) return -1;
        o_chrom[n] = name_to_index(r->chrom);
        o_start[n] = r->start; o_finish[n] = r->finish; o_value[n] = r->value;
        n++;
        r_pop(r);
    }
    return n;
}

/* Runs `op` over text files through the reference's own readers
 * (SmartReader, unaryOps.c:1168-1204: .wig/.bg/.bed only here).
 * Chromosome names are returned as one '\n'-joined buffer of unique names in
 * order of first appearance; o_chrom indexes into it. */
int64_t ref_reduce_files(int n_files, char **paths, int op, unsigned flags, int64_t cap,
                         int32_t *o_chrom, int32_t *o_start, int32_t *o_finish, double *o_value,
                         char *names_buf, int names_cap) {
    if (!g_lib || op < 0 || op > 9 || !r_SmartReader) return -2;
    WiggleIterator **iters = (WiggleIterator **) calloc((size_t) n_files, sizeof(WiggleIterator *));
    for (int i = 0; i < n_files; i++) iters[i] = r_SmartReader(paths[i], 0);
    Multiplexer *m = r_newMultiplexer(iters, n_files, (wt_bool) (flags & 1u));
    WiggleIterator *r = r_reduction[op](m);
    int64_t n = 0;
    int n_names = 0;
    char *last = NULL;
    names_buf[0] = 0;
    while (!r->done) {
        if (n >= cap) return -1;
        if (!last || strcmp(last, r->chrom)) {
            if ((int) (strlen(names_buf) + strlen(r->chrom) + 2) > names_cap) return -3;
            if (n_names) strcat(names_buf, "\n");
            strcat(names_buf, r->chrom);
            n_names++;
            last = r->chrom;
        }
        o_chrom[n] = n_names - 1;
        o_start[n] = r->start; o_finish[n] = r->finish; o_value[n] = r->value;
        n++;
        r_pop(r);
    }
    return n;
}

/* Wall-clock of the reference path, sink = none (the reference's `do`):
 * returns seconds; *o_runs / *o_bp receive runs emitted and bp covered. */
double ref_time_reduce(const wto_tracks *t, int op, unsigned flags, int64_t *o_runs, int64_t *o_bp) {
    if (!g_lib || op < 0 || op > 9) return -1;
    char **names = make_names(t->n_chrom);
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    Multiplexer *m = make_multiplexer(t, names, 0, t->n_tracks, flags & 1u);
    WiggleIterator *r = r_reduction[op](m);
    int64_t n = 0, bp = 0;
    volatile double sinkv = 0;
    while (!r->done) {
        n++; bp += r->finish - r->start; sinkv += r->value;
        r->pop(r);
    }
    clock_gettime(CLOCK_MONOTONIC, &b);
    *o_runs = n; *o_bp = bp;
    return (double) (b.tv_sec - a.tv_sec) + 1e-9 * (double) (b.tv_nsec - a.tv_nsec);
}
