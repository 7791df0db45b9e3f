/*
 * wiggletools_amd.h -- C ABI of the MI355X-native multiplexer / reducer engine.
 *
 * Two layers are declared here:
 *
 *  (1) DROP-IN LAYER.  The symbols the reference's own callers bind
 *      (reference src/commandParser.c:500-569,635-651 call them; they are
 *      declared in reference src/wiggletools.h:80-103 and
 *      src/multiplexer.h:38-41, src/multiSet.h:32-33).  Same names, same C
 *      signatures, same struct layouts (reference src/wiggleIterator.h:21-35,
 *      src/multiplexer.h:21-36, src/multiSet.h:20-30), so that
 *      libwiggletools_amd.so can be linked in place of multiplexer.o /
 *      multiSet.o / reducers.o / setComparisons.o.  See INTEGRATION.md.
 *
 *  (2) BULK LAYER (wtamd_*).  Plain pointers + sizes.  This is what the
 *      drop-in layer itself calls once it has drained the child iterators
 *      into SoA batches, and what a bulk reader (BigWig section decoder)
 *      or bench.py binds directly when the tracks are already resident
 *      in HBM.
 *
 * No torch / C++ types appear in any signature.
 */
#ifndef WIGGLETOOLS_AMD_H_
#define WIGGLETOOLS_AMD_H_

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* (1) DROP-IN LAYER                                                   */
/* ------------------------------------------------------------------ */

/* The reference spells its booleans `#define bool char`
 * (reference src/wiggletools.h:18-22).  We cannot #define bool in a header
 * that C++ also includes, so the ABI type is named explicitly. */
typedef char wt_bool;

typedef struct wiggleIterator_st WiggleIterator;
typedef struct multiplexer_st Multiplexer;
typedef struct multiset_st Multiset;

/* Layout == reference src/wiggleIterator.h:21-35 (88 bytes on LP64). */
struct wiggleIterator_st {
    char *chrom;
    int start;              /* 1-based, inclusive  */
    int finish;             /* exclusive           */
    double value;
    void *valuePtr;
    wt_bool done;
    int strand;
    void *data;
    void (*pop)(WiggleIterator *);
    void (*seek)(WiggleIterator *, const char *, int, int);
    wt_bool overlaps;
    double default_value;
    WiggleIterator *append;
};

/* Layout == reference src/multiplexer.h:21-36.  `starts`/`finishes` are the
 * reference's FibHeap pointers; this engine has no heaps and keeps them NULL. */
struct multiplexer_st {
    char *chrom;
    int start;
    int finish;
    double *values;
    double *default_values;
    int count, inplay_count;
    wt_bool *inplay;
    WiggleIterator **iters;
    wt_bool done;
    wt_bool strict;
    void (*pop)(Multiplexer *);
    void (*seek)(Multiplexer *, const char *, int, int);
    void *starts, *finishes;
    void *data;
};

/* Layout == reference src/multiSet.h:20-30. */
struct multiset_st {
    char *chrom;
    int start;
    int finish;
    double **values;
    int count, inplay_count;
    wt_bool *inplay;
    Multiplexer **multis;
    wt_bool done;
    void *starts, *finishes;
};

/* Iterator core -- replaces reference src/wiggleIterator.c:20-70.
 * (Provided so the library is self-contained; when linked into the reference
 * build the reference's wiggleIterator.o may be kept instead -- the semantics
 * are identical: ctor primes the first element, pop() guards on done.) */
WiggleIterator *newWiggleIterator(void *data, void (*pop)(WiggleIterator *),
                                  void (*seek)(WiggleIterator *, const char *, int, int),
                                  double default_value, wt_bool overlapping);
void pop(WiggleIterator *);
void seek(WiggleIterator *, const char *, int, int);
void runWiggleIterator(WiggleIterator *);
void destroyWiggleIterator(WiggleIterator *);

/* Multiplexer -- replaces reference src/multiplexer.c:22-169. */
Multiplexer *newMultiplexer(WiggleIterator **iters, int count, wt_bool strict);
Multiplexer *newCoreMultiplexer(void *data, int count, void (*pop)(Multiplexer *),
                                void (*seek)(Multiplexer *, const char *, int, int));
void popMultiplexer(Multiplexer *);
void seekMultiplexer(Multiplexer *, const char *chrom, int start, int finish);
void runMultiplexer(Multiplexer *);

/* Multiset -- replaces reference src/multiSet.c:80-128. */
Multiset *newMultiset(Multiplexer **multis, int count);
void popMultiset(Multiset *);
void seekMultiset(Multiset *, const char *chrom, int start, int finish);

/* Reducers -- replace reference src/reducers.c (ctor lines in brackets). */
WiggleIterator *SumReduction(Multiplexer *);      /* reducers.c:294-307 */
WiggleIterator *ProductReduction(Multiplexer *);  /* reducers.c:348-361 */
WiggleIterator *MeanReduction(Multiplexer *);     /* reducers.c:404-422 */
WiggleIterator *VarianceReduction(Multiplexer *); /* reducers.c:481-505 */
WiggleIterator *StdDevReduction(Multiplexer *);   /* reducers.c:565-590 */
WiggleIterator *EntropyReduction(Multiplexer *);  /* reducers.c:640-666 */
WiggleIterator *CVReduction(Multiplexer *);       /* reducers.c:727-751 */
WiggleIterator *MedianReduction(Multiplexer *);   /* reducers.c:815-834 */
WiggleIterator *MinReduction(Multiplexer *);      /* reducers.c:237-253 */
WiggleIterator *MaxReduction(Multiplexer *);      /* reducers.c:170-186 */
WiggleIterator *SelectReduction(Multiplexer *, int);      /* reducers.c:67-72, host */
WiggleIterator *FillInReduction(Multiplexer *, wt_bool);  /* reducers.c:108-119, host */

/* Two-sample tests -- replace reference src/setComparisons.c. */
WiggleIterator *TTestReduction(Multiset *);  /* setComparisons.c:123-131 */
WiggleIterator *MWUReduction(Multiset *);    /* setComparisons.c:372-390 */
/* setComparisons.c:232-243.  The reference's F-test pop is broken (its inner loops advance
 * `index` instead of `index2`, :183,199 -- undefined behaviour) and is out of scope:
 * exported so that the link change of INTEGRATION.md links (commandParser.c:642-644 calls it);
 * prints a message and exit(1)s, the reference's own error convention. */
WiggleIterator *FTestReduction(Multiset *);

/* Behavioural difference at this boundary (documented, loud): the reference accepts any C `int`
 * coordinate; this engine refuses run finishes above WTAMD_MAX_COORD = 2^31 - 65537 (message +
 * exit(1) from the drop-in layer, WTAMD_ERR_ARG from the bulk layer) because its 32-bit window
 * arithmetic keeps a margin below INT32_MAX.  No genome assembly comes near it. */

/* ------------------------------------------------------------------ */
/* (2) BULK LAYER                                                      */
/* ------------------------------------------------------------------ */

/* Reducer selector.  One code per reference ...ReductionPop. */
enum wtamd_op {
    WTAMD_OP_SUM = 0,      /* SumReductionPop       reducers.c:259-292 */
    WTAMD_OP_PRODUCT = 1,  /* ProductReductionPop   reducers.c:313-346 */
    WTAMD_OP_MEAN = 2,     /* MeanReductionPop      reducers.c:367-402 */
    WTAMD_OP_VAR = 3,      /* VarianceReductionPop  reducers.c:428-479 */
    WTAMD_OP_STDDEV = 4,   /* StdDevReductionPop    reducers.c:511-563 */
    WTAMD_OP_ENTROPY = 5,  /* == STDDEV pop (reducers.c:665 installs StdDevReductionPop) */
    WTAMD_OP_CV = 6,       /* CVReductionPop        reducers.c:672-725 */
    WTAMD_OP_MIN = 7,      /* MinReductionPop       reducers.c:192-235 */
    WTAMD_OP_MAX = 8,      /* MaxReductionPop       reducers.c:125-168 */
    WTAMD_OP_MEDIAN = 9,   /* MedianReductionPop    reducers.c:780-813 */
    WTAMD_OP_TTEST = 10,   /* TTestReductionPop     setComparisons.c:35-121 */
    WTAMD_OP_MWU = 11,     /* MWUReductionPop       setComparisons.c:269-370 */
    WTAMD_OP_COUNT_ = 12
};

/* Flags for wtamd_reduce_desc.flags */
#define WTAMD_STRICT_SET0 1u   /* Multiplexer `strict` (set 0, or the only set) */
#define WTAMD_STRICT_SET1 2u   /* `strict` of the second Multiplexer (two-sample ops) */

/* Status codes (every wtamd_* function returning int). */
#define WTAMD_OK 0
#define WTAMD_ERR_ARG 1        /* bad argument                         */
#define WTAMD_ERR_HIP 2        /* HIP runtime error (see wtamd_last_error) */
#define WTAMD_ERR_CAPACITY 3   /* output buffers too small             */
#define WTAMD_ERR_NODEVICE 4   /* no gfx950 device visible             */
#define WTAMD_ERR_INTERNAL 5   /* kernel reported an internal fault    */

typedef struct wtamd_trackset wtamd_trackset; /* opaque: N tracks resident in HBM */

/*
 * Track storage ("run lists"): for chromosome c (0..n_chrom-1, already in
 * strcmp order, cf. reference multiplexer.c:56) and track i (0..n_tracks-1)
 * the intervals of that track on that chromosome are the slice
 *     [seg_off[c*n_tracks+i], seg_off[c*n_tracks+i+1])
 * of the three parallel arrays start[] (1-based inclusive), finish[]
 * (exclusive) and value[].  Within a slice intervals are sorted and
 * non-overlapping (the Multiplexer's precondition, multiplexer.c:163).
 * value[] is float32 (BigWig payload type) or float64.
 */
typedef struct {
    int32_t n_chrom;
    int32_t n_tracks;
    const int64_t *seg_off;   /* HOST pointer, n_chrom*n_tracks+1 entries */
    const int32_t *start;     /* host or device, see create function */
    const int32_t *finish;
    const void *value;        /* float* or double* */
    int32_t value_is_f64;     /* 0: float32, 1: float64 */
    const double *defaults;   /* HOST pointer, n_tracks default_values */
    /* Optional (NULL = whole chromosome): per chromosome, only runs whose START lies in
     * [range_lo[c], range_hi[c]) are produced.  This is how one chromosome is cut into
     * batches (drop-in layer) or shards (multi-GPU): a run spanning a cut belongs to the
     * piece that holds its start and keeps its true finish, so pieces concatenate to the
     * unsharded output.  INT32_MAX as range_hi = "to the end".  HOST pointers, n_chrom entries. */
    const int32_t *range_lo;
    const int32_t *range_hi;
} wtamd_tracks;

typedef struct {
    int32_t op;        /* enum wtamd_op */
    uint32_t flags;    /* WTAMD_STRICT_* */
    int32_t n_set0;    /* two-sample ops: tracks [0,n_set0) = set 0, rest = set 1; else 0 */
    int32_t reserved;
} wtamd_reduce_desc;

/* Output of one reduction: R runs in (chrom, start) order. All DEVICE pointers
 * unless the _host entry point is used. */
typedef struct {
    int64_t capacity;        /* in: entries available in the arrays below */
    int32_t *start;          /* out: run start  (1-based inclusive)  */
    int32_t *finish;         /* out: run finish (exclusive)          */
    double *value;           /* out: reducer value                   */
    int64_t *chrom_run_off;  /* out: n_chrom+1 entries, runs of chrom c = [off[c],off[c+1]) */
} wtamd_runs;

/* Aggregate statistics of the last reduce on a trackset (for metrics). */
typedef struct {
    int64_t n_runs;          /* emitted runs                                   */
    int64_t covered_bp;      /* sum(finish-start) of emitted runs              */
    int64_t n_intervals;     /* input intervals examined                       */
    int64_t n_windows;       /* alignment windows processed                    */
    int32_t window_bp;       /* window width used                              */
    int32_t lds_bytes;       /* LDS per workgroup                              */
    float index_ms;          /* last window-index kernel time (HIP events)     */
    float reduce_ms;         /* last multiplex+reduce kernel time (HIP events) */
    int32_t kernel;          /* kernel of the last reduction: 0 general bitmap multiplexer (wt_reduce_kernel),
                                1 exact difference array for Sum / Mean (wt_delta_kernel),
                                2 median by walking (wt_walk_kernel)                                  */
    int32_t patched_windows; /* difference-array windows whose values the general kernel rewrote (NaN, Inf, wide range) */
} wtamd_stats;

int wtamd_device_count(void);
int wtamd_set_device(int ordinal);
int wtamd_current_device(void);     /* the calling thread's device (-1: none); HIP devices are per thread */
/* Starts the HIP runtime (device discovery, code object, first hardware queues: ~0.2 s in a fresh process) on a helper thread,
 * once per process; returns at once.  wtamd_BigWiggleReaders calls it before it opens its files; the first pipe waits for it. */
void wtamd_warmup_async(void);
const char *wtamd_last_error(void);
const char *wtamd_version(void);

/* Largest run finish a track set may hold (the 32-bit window arithmetic keeps a margin below
 * INT32_MAX; creation fails with WTAMD_ERR_ARG above it).  The reference's coordinates are C `int`
 * as well (wiggleIterator.h:23-24). */
#define WTAMD_MAX_COORD (2147483647 - 65536)

/* Copies the SoA arrays from HOST memory into HBM. */
int wtamd_trackset_create_host(const wtamd_tracks *tracks, wtamd_trackset **out);
/* Zero-copy: start/finish/value are DEVICE pointers owned by the caller and
 * must stay valid until wtamd_trackset_destroy. */
int wtamd_trackset_create_device(const wtamd_tracks *tracks, wtamd_trackset **out);
void wtamd_trackset_destroy(wtamd_trackset *);

/* Input contract check, on device: inside every (chrom, track) segment the runs must be sorted,
 * non-overlapping and of positive length (finish > start).  Anything else is undefined behaviour
 * here as it is in the reference's Multiplexer (multiplexer.c:76-96; its text readers check it,
 * bedReader.c:46-49).  *n_bad = number of offending runs, *first_bad (optional) = global index of
 * the first one or -1.  Not called implicitly: one pass over start[] / finish[]. */
int wtamd_trackset_validate(wtamd_trackset *ts, int64_t *n_bad, int64_t *first_bad);

/* Upper bound on the number of runs any reduction over `ts` can emit. */
int64_t wtamd_trackset_max_runs(const wtamd_trackset *ts);

/* Build (or rebuild) the window index `op` (enum wtamd_op) would use, on `stream`
 * (hipStream_t as void*).  wtamd_reduce() calls this itself when the index is
 * missing; it is exposed so callers can time / amortise it separately, and it is what a caller of
 * a zero-copy track set must call after rewriting the run lists in place (same segment sizes):
 * it also voids what earlier reductions learnt about the values -- the next Sum / Mean verifies
 * again that its exact difference-array kernel applies and therefore waits for its launch. */
int wtamd_trackset_index(wtamd_trackset *ts, int op, void *stream);

/* Multiplex + reduce, everything on device. `runs` arrays are DEVICE memory.
 * *n_runs is written on the host after the stream has been synchronised iff
 * n_runs != NULL (otherwise the call is fully asynchronous and the count can be
 * read from chrom_run_off[n_chrom]). */
int wtamd_reduce(wtamd_trackset *ts, const wtamd_reduce_desc *desc, wtamd_runs *runs,
                 int64_t *n_runs, void *stream);

/* Convenience: same, but `runs` arrays are HOST memory (internally staged). */
int wtamd_reduce_host(wtamd_trackset *ts, const wtamd_reduce_desc *desc, wtamd_runs *runs,
                      int64_t *n_runs);

/* Multiplexer "materialise": emits the aligned tile the reference exposes per
 * popMultiplexer (multiplexer.h:21-36): for each run r, values[r*n_tracks+i]
 * and inplay[r*n_tracks+i]; runs->value[r] receives the run's inplay_count.  HOST output.
 * Used by the drop-in layer when a Multiplexer* escapes to code that reads its fields. */
int wtamd_multiplex_host(wtamd_trackset *ts, uint32_t flags, wtamd_runs *runs,
                         double *values, uint8_t *inplay, int64_t *n_runs);

/* Genome-wide scalar over the output of a reduction, computed on device:
 * AUC = sum over runs of (finish-start)*value skipping NaN
 * (reference statistics.c:103-120).  Result written to *auc (host). */
int wtamd_runs_auc(const wtamd_runs *runs, int64_t n_runs, double *auc, void *stream);
/* meanI: sum of (finish-start)*value over the non-NaN runs divided by their span; NaN when that
 * span is 0 (reference MeanIntegrator, statistics.c:62-100; its NonOverlapping wrapper is the
 * identity on reducer output, which never overlaps). */
int wtamd_runs_mean(const wtamd_runs *runs, int64_t n_runs, double *mean, void *stream);

/* Pearson correlation of the two tracks of `ts` over their Multiplexer tile, on device
 * (reference PearsonIntegrator over a 2-track Multiplexer: statistics.c:414-465,
 * commandParser.c:683-694).  Non-strict Multiplexer, absent tracks read as their defaults.
 * NaN when T_XX*T_YY == 0 (statistics.c:421-423).  Slices of runs are merged with the reference's
 * own update formula, so the result agrees to rounding (not bit-for-bit). */
int wtamd_pearson(wtamd_trackset *ts, double *result);
/* The same as six moments {n, sum_X, sum_Y, T_XX, T_XY, T_YY} of this track set's part of the genome,
 * so that shards (chromosomes on different GPUs) can be combined: gather the 6 doubles per shard
 * (RCCL all_gather), then merge IN GENOME ORDER with wtamd_pearson_merge -- the reference's update is
 * sequential (statistics.c:442-456), a pairwise merge agrees to rounding -- and finish. */
int wtamd_pearson_moments(wtamd_trackset *ts, double *moments6);
void wtamd_pearson_merge(double *a6, const double *b6);     /* a := a (+) b, b after a in genome order; HOST */
double wtamd_pearson_finish(const double *m6);               /* T_XY / sqrt(T_XX T_YY), NaN if 0 (statistics.c:421-423) */

/* The reference's `map`-able unary operators (src/unaryOps.c: scale :650-664, offset :722-734,
 * ln / log :760-813, exp :823-866, pow :873-899, abs :934-949; commandParser.c:115-211) applied to
 * whole run lists on device before they are multiplexed. */
enum wtamd_map_op {
    WTAMD_MAP_SCALE = 0,    /* param * value                                   */
    WTAMD_MAP_OFFSET = 1,   /* param + value                                   */
    WTAMD_MAP_LN = 2,       /* log(value); runs with value <= 0 are DROPPED    */
    WTAMD_MAP_LOG = 3,      /* log(value) / log(param); same                   */
    WTAMD_MAP_EXP = 4,      /* exp(value)                                      */
    WTAMD_MAP_EXPB = 5,     /* exp(value * log(param))                         */
    WTAMD_MAP_POW = 6,      /* pow(value, param); NaN if param < 0 && value <= 0 */
    WTAMD_MAP_ABS = 7,
    WTAMD_MAP_GT = 8,       /* runs with value > param, value 1; the others and NaN runs are DROPPED */
    WTAMD_MAP_GTE = 9,      /*   (HighPassFilterWiggleIterator, unaryOps.c:386-419; default 0)      */
    WTAMD_MAP_LT = 10,      /* value < param (the reference builds it as scale -1, gt -param)       */
    WTAMD_MAP_LTE = 11,
    WTAMD_MAP_COUNT_
};
/* start / finish / value and the o_* arrays are DEVICE memory (o_* sized for all input runs),
 * seg_off / o_seg_off HOST arrays of n_seg + 1 offsets (n_seg = n_chrom * n_tracks).  Output
 * values are f64.  For the operators that drop nothing o_start / o_finish may be NULL or alias
 * the inputs.  Synchronous on `stream`. */
int wtamd_runs_map(int map_op, double param, int64_t n_seg, const int64_t *seg_off, const int32_t *start,
                   const int32_t *finish, const void *value, int value_is_f64, int32_t *o_start, int32_t *o_finish,
                   double *o_value, int64_t *o_seg_off, void *stream);
/* default_value of the operator iterator the reference would build around a track whose default is
 * `default_value`, including the `float` truncation several constructors apply. */
double wtamd_map_default(int map_op, double param, double default_value);

/* The same operators INSIDE the streaming pipeline: one chain of up to WTAMD_MAP_CHAIN_MAX operators per
 * track (applied first to last), run on device on every batch between its arrival in HBM and the
 * Multiplexer -- `sum map ln a.bw b.bw` (README idiom; commandParser.c:115-211) without N host operator
 * iterators in front of the engine.  The pipe's default values must already be the mapped ones
 * (wtamd_map_default).  chains == NULL switches it off.  Mapped batches are f64. */
#define WTAMD_MAP_CHAIN_MAX 4
typedef struct wtamd_map_chain {
    int32_t n_ops;
    int32_t op[WTAMD_MAP_CHAIN_MAX];
    double param[WTAMD_MAP_CHAIN_MAX];
} wtamd_map_chain;
struct wtamd_pipe;
int wtamd_pipe_set_map(struct wtamd_pipe *p, const wtamd_map_chain *chains /* n_tracks entries */);
/* Drop-in side: the operator iterator the reference's parser builds around a track (ScaleWiggleIterator,
 * NaturalLogWiggleIterator, ... unaryOps.c:650-949, HighPassFilterWiggleIterator :386-419) as ONE
 * constructor.  Handed to newMultiplexer / newMultiset of this library it is unwrapped: the child is
 * drained raw (in blocks when it is bulk-capable) and the chain runs on device (wtamd_pipe_set_map);
 * popped by anything else it follows the reference's per-interval protocol on the host.
 * default_value = wtamd_map_default(op, param, child's).  Unknown operator or a chain deeper than
 * WTAMD_MAP_CHAIN_MAX: message and exit(1). */
WiggleIterator *wtamd_MapIterator(WiggleIterator *child, int map_op, double param);

/* Run compression on device (reference CompressionWiggleIterator, unaryOps.c:235-253, which the
 * default writer applies, wigWriter.c:263-267): adjacent runs of one chromosome merge while
 * start == previous finish and (both NaN or |value - value of the group's first run| < 1e-6).
 * `in` / `out` are DEVICE run lists (in->chrom_run_off required); *n_out is written on the host. */
int wtamd_runs_compress(const wtamd_runs *in, int64_t n_runs, int32_t n_chrom, wtamd_runs *out,
                        int64_t *n_out, void *stream);

int wtamd_get_stats(const wtamd_trackset *ts, wtamd_stats *out);

/* ---- Streaming pipeline (what the drop-in layer feeds; replaces the producer / consumer
 * overlap the reference gets from src/bufferedReader.c:41-55,99-109 -- 10 000-entry SoA blocks, a
 * producer up to 3 blocks ahead -- and the per-run evaluation behind it).
 *
 * A pipe owns `n_slots` batch slots.  Each slot has PINNED host staging for one batch of run lists
 * (one chromosome, run starts in [lo, hi)), device buffers allocated once, and pinned host output.
 * submit() enqueues  H2D (copy stream) -> window index + multiplex/reduce kernels (compute
 * stream) -> D2H of exactly the emitted runs (third stream)  and returns at once, so the caller
 * fills slot k+1 while slot k computes and slot k-1 is being read.  Values stay float32 end to end
 * when the source's values are float32-exact.  Nothing is allocated or freed per batch.
 *
 *   acquire -> fill the staging arrays (grow if needed) -> submit     (repeat, up to n_slots deep)
 *   collect -> read the result arrays -> release                       (in submission order)      */
#define WTAMD_OP_MULTIPLEX 12   /* pipe only: the aligned tile values[]/inplay[] per run (multiplexer.h:21-36) */
#define WTAMD_PIPE_COMPRESS 1u  /* apply CompressionWiggleIterator's merge rule on device before D2H (unaryOps.c:235-253) */

typedef struct wtamd_pipe wtamd_pipe;

typedef struct {
    int32_t n_tracks;
    int32_t n_slots;            /* 2..8 batches in flight; 0 = 3 */
    const double *defaults;     /* HOST, n_tracks default_values */
    wtamd_reduce_desc desc;     /* op may also be WTAMD_OP_MULTIPLEX */
    int64_t max_intervals;      /* initial input capacity of a slot (wtamd_pipe_grow enlarges it) */
    int64_t max_runs;           /* output capacity of a slot, fixed: keep hi - lo <= max_runs (a run is >= 1 bp) */
    uint32_t flags;             /* WTAMD_PIPE_* */
    int32_t reserved;
} wtamd_pipe_config;

typedef struct {                /* pinned staging of the slot being filled */
    int64_t capacity;           /* intervals the arrays below hold */
    int64_t *seg_off;           /* n_tracks + 1 offsets into the arrays (seg_off[0] = 0) */
    int32_t *start, *finish;
    float *value32;             /* always present */
    double *value64;            /* NULL until wtamd_pipe_grow(.., want_f64 = 1) */
} wtamd_pipe_batch;

typedef struct {                /* pinned output of the oldest submitted batch, valid until release */
    int64_t n_runs;
    const int32_t *start, *finish;
    const double *value;        /* reducer value; WTAMD_OP_MULTIPLEX: the run's inplay_count */
    const double *tile;         /* WTAMD_OP_MULTIPLEX: n_runs * n_tracks values, else NULL */
    const uint8_t *inplay;      /* WTAMD_OP_MULTIPLEX: n_runs * n_tracks flags, else NULL */
    int64_t covered_bp;         /* sum (finish - start) of the runs the kernels emitted (before compression) */
    int64_t n_intervals;        /* input intervals of the batch */
    int32_t integ_valid;        /* the batch was integrated on device (wtamd_pipe_set_integrate): start / finish / value are NULL */
    int32_t reserved;
    double integ[6];            /* reducers: {sum of (finish - start) * value, span} over the non-NaN runs (statistics.c:62-120);
                                   WTAMD_OP_MULTIPLEX over 2 tracks: the Pearson moments {n, sum_X, sum_Y, T_XX, T_XY, T_YY} (:414-465) */
} wtamd_pipe_result;

typedef struct {
    int64_t batches, intervals, runs, covered_bp;
    int64_t h2d_bytes, d2h_bytes;
    double kernel_ms;           /* sum over batches of index + reduce kernel time (HIP events on the compute stream) */
    double h2d_ms, d2h_ms;      /* same for the copies (events on their streams) */
    int32_t delta_batches;      /* batches the exact difference-array kernel evaluated */
    int32_t n_slots;
    double host_submit_ms;      /* host time inside wtamd_pipe_submit (enqueueing; nothing there waits for the GPU) */
    double host_wait_ms;        /* host time wtamd_pipe_collect spent waiting for a batch to finish */
    int64_t bw_sections;        /* BigWig sections inflated on device (wtamd_pipe_submit_bw) */
    double bw_decode_ms;        /* ... and the summed duration of their inflate / count / scan / scatter kernels (HIP events) */
} wtamd_pipe_stats;

int wtamd_pipe_create(const wtamd_pipe_config *cfg, wtamd_pipe **out);
void wtamd_pipe_destroy(wtamd_pipe *);
/* Staging of the next free slot.  WTAMD_ERR_ARG when every slot is in flight or unreleased. */
int wtamd_pipe_acquire(wtamd_pipe *, wtamd_pipe_batch *out);
/* Enlarges the acquired slot's staging to >= min_capacity intervals (and adds the float64 value
 * array if want_f64), preserving the first `used` entries of every array; *out is refreshed. */
int wtamd_pipe_grow(wtamd_pipe *, int64_t used, int64_t min_capacity, int want_f64, wtamd_pipe_batch *out);
/* Bulk side door: `count` float32-valued intervals of the acquired slot, at interval offset `at`
 * of its arrays, are NOT staged -- the pipe copies them to HBM straight from the caller's arrays at
 * submit (hipMemcpyAsync).  The arrays must stay valid and unchanged until the batch has been
 * collected and should be pinned (wtamd_host_alloc); pageable memory still works but the copy is
 * then staged by the runtime.  The staging arrays need not cover such ranges. */
int wtamd_pipe_put_direct(wtamd_pipe *, int64_t at, int64_t count, const int32_t *start, const int32_t *finish,
                          const float *value);
/* Ships the acquired slot: seg_off[n_tracks] intervals of one chromosome, runs whose start lies in
 * [range_lo, range_hi) are produced (same meaning as wtamd_tracks.range_lo/hi).  Asynchronous. */
int wtamd_pipe_submit(wtamd_pipe *, int value_is_f64, int32_t range_lo, int32_t range_hi);
/* Abandons the acquired slot without shipping it. */
int wtamd_pipe_cancel(wtamd_pipe *);
/* Result of the oldest submitted batch (waits for it). */
int wtamd_pipe_collect(wtamd_pipe *, wtamd_pipe_result *out);
/* Returns the oldest collected batch's slot to the pipe. */
int wtamd_pipe_release(wtamd_pipe *);
/* Turns the device-side run compression on / off for the batches submitted from now on: the
 * reference's CompressionWiggleIterator rule (unaryOps.c:235-253) inside every batch, from the batch's
 * first run that leads a group whatever came before it (the runs ahead of it may belong to the
 * previous batch's last group -- only a consumer that knows that group's leader can tell -- and are
 * passed through one by one).  Applying the reference's own wrapper to this output, as the default
 * writer always does (wigWriter.c:263-267), yields exactly what it yields on the uncompressed runs. */
int wtamd_pipe_set_compress(wtamd_pipe *, int on);
/* Genome-wide integrators fused into the pipeline (reference AUCIntegrator / MeanIntegrator over a reducer,
 * PearsonIntegrator over a 2-track Multiplexer: statistics.c:62-127,414-465): batches submitted from now on are
 * integrated ON THE DEVICE and their runs never cross PCIe -- a result carries 2 (6) doubles instead of 16 bytes
 * per run.  Sums are two-level (per lane slice, then ordered merge): agreement with the reference's sequential
 * accumulation to rounding.  Not together with WTAMD_PIPE_COMPRESS. */
int wtamd_pipe_set_integrate(wtamd_pipe *, int on);
/* The same integrals of the batch currently held (collected, not released) when it travelled the ordinary way
 * -- the batch a reducer's constructor primed with before an integrator took it over.  integ[6] as above. */
int wtamd_pipe_integrate_held(wtamd_pipe *, double *integ);
/* Submitted batches not yet collected. */
int wtamd_pipe_in_flight(const wtamd_pipe *);
int wtamd_pipe_get_stats(const wtamd_pipe *, wtamd_pipe_stats *out);

/* ---- BigWig sections decoded ON THE DEVICE (csrc/wt_bwdev.hip, csrc/wt_inflate.h).  A batch may be handed
 * over as the FILE BYTES of the BigWig data sections that overlap it instead of as run lists: the pipe
 * ships the compressed bytes (2.5 x fewer than the run lists), inflates every section on the GPU -- one
 * lane per zlib stream -- and expands the items into the slot's run lists with the reference reader's
 * conventions (1-based starts, 10 000-bp boxing, seek window: src/bigWiggleReader.c:36-83,125-145).  This
 * replaces the host-side inflate the reference gets from libBigWig; wtamd_BigWiggleReader children of a
 * reducer of this library take this route on their own (WTAMD_BW_DEVICE=0 keeps the host decoder).
 *
 *   acquire -> wtamd_pipe_bw_reserve -> fill bytes[] and sections[] -> wtamd_pipe_submit_bw -> collect ...
 *
 * Sections are listed track by track (track-major, ascending position inside a track); a track's
 * intervals must come out sorted and non-overlapping (checked on device: the batch fails otherwise). */
typedef struct {
    int64_t comp_off;               /* first byte of the section in bytes[] (any alignment) */
    uint32_t comp_size;             /* its size there */
    int32_t track;
    uint32_t leaf_start, leaf_end;  /* extents of the section in the file's R-tree leaf, 0-based half-open: every item must
                                       lie inside (checked on device) */
} wtamd_bw_section;

typedef struct {
    uint32_t chrom_id;              /* the file's id of the batch's chromosome (sections of other ids yield nothing) */
    uint32_t chrom_len;             /* its length in that file (bounds the boxing, bigWiggleReader.c:76) */
    int32_t box;                    /* != 0: cut intervals at the reader's 10 000-bp stretch edges (:42-44,73-83) */
    int32_t compressed;             /* != 0: sections are zlib streams (header uncompressBufSize != 0) */
    int32_t clip_lo, clip_hi;       /* pieces are clipped to [clip_lo, clip_hi) (1-based) and dropped when empty */
    int32_t first_section;          /* this track's slice of sections[]: [first_section, first_section + n_sections); */
    int32_t n_sections;             /*   first_section of a track without sections = that of the next track           */
    uint32_t plain_bytes;           /* upper bound of a section's inflated size (the file's uncompressBufSize; raw: its largest section) */
    int32_t reserved;
} wtamd_bw_track;

/* Pinned staging of the acquired slot for `n_bytes` file bytes and `n_sections` table entries (grown on
 * demand, contents undefined).  Pointers stay valid until the slot is submitted or cancelled. */
int wtamd_pipe_bw_reserve(wtamd_pipe *, int64_t n_bytes, int64_t n_sections, uint8_t **bytes, wtamd_bw_section **sections);
/* Ships the acquired slot as file bytes: tracks[n_tracks] describe the tracks, the tables / bytes are the
 * reserved ones; runs whose start lies in [range_lo, range_hi) are produced, as for wtamd_pipe_submit.
 * A malformed stream / section fails the batch at collect (WTAMD_ERR_INTERNAL with the cause). */
int wtamd_pipe_submit_bw(wtamd_pipe *, int64_t n_bytes, int64_t n_sections, const wtamd_bw_track *tracks,
                         int32_t range_lo, int32_t range_hi);
/* How many sections one launch of the inflate kernel keeps resident on the device (wavefronts the GPU holds at
 * once x 64 lanes): a batch of about that many sections fills the GPU exactly once -- fewer leave SIMDs idle, a
 * few more cost a whole second round. */
int64_t wtamd_pipe_bw_fill_sections(const wtamd_pipe *);
/* Why the last wtamd_pipe_collect of a file-byte batch failed: the device decoder's error bits (1 corrupt zlib stream or
 * Adler-32 mismatch, 2 malformed section, 4 items outside their index leaf's extents / out of order, 8 coordinate above
 * the maximum, 16 more intervals than the bound); 0: it did not fail there.  The drop-in layer answers 1 / 2 / 4 by going
 * back to the host decoder from that batch on -- libBigWig, which the reference reads through, checks none of these
 * extents (src/bigWiggleReader.c:52-83). */
unsigned wtamd_pipe_bw_error(const wtamd_pipe *);

/* Pinned (page-locked, DMA-able) host memory for bulk sources. */
void *wtamd_host_alloc(size_t bytes);
void wtamd_host_free(void *);
/* The process-wide pools behind the pipes: page-locked staging (also wtamd_host_alloc; WTAMD_PINNED_POOL_MB) and
 * device buffers (WTAMD_DEVICE_POOL_MB).  out[0..2]: pinned buffers of 1 MB and more that had to be page-locked afresh
 * so far (count, bytes) and the bytes resting in the pool now; out[3..5]: the same for device buffers (hipMalloc).  A
 * second run of the same job in a process should add no misses. */
void wtamd_pool_stats(int64_t out[6]);
/* Gives every buffer resting in the two pools back to the runtime (a process that wants its device memory or its
 * lockable pages for something else; the pools also do this on their own when an allocation of theirs fails). */
void wtamd_pool_trim(void);

/* ---- Bulk doors of the drop-in layer ------------------------------------------------------
 * The reference's iterator protocol moves ONE interval per indirect call (wiggleIterator.c:57-60);
 * its own readers soften that with 10 000-entry SoA blocks between threads (bufferedReader.c:21-28).
 * A child iterator built by this library exposes such blocks to the Multiplexer directly (producer
 * side), and a reducer hands its runs over in blocks (consumer side -- what TeeWiggleIterator copies
 * into its 10 000-entry blocks one pop at a time, wigWriter.c:164-203).  Both stay ordinary
 * WiggleIterators: pop() / seek() work as always, foreign iterators are drained with pop(). */

/* Array-backed reader: one track as SoA run lists in host memory (chromosomes in strcmp order,
 * chromosome c = slice [seg_off[c], seg_off[c+1]) of the arrays; 1-based start, exclusive finish,
 * float32 value).  The arrays are borrowed (and read by DMA when pinned).  Bulk-capable. */
WiggleIterator *wtamd_ArrayReader(int n_chrom, const char *const *chrom_names, const int64_t *seg_off,
                                  const int32_t *start, const int32_t *finish, const float *value,
                                  double default_value);
/* BigWig reader: the role of the reference's BigWiggleReader (src/bigWiggleReader.c:147-151) on this
 * library's own section decoder -- chromosomes in strcmp order, 1-based starts, box != 0: intervals
 * cut at the reference reader's 10 000-bp stretch edges (what `write_bg` parity needs); one producer
 * thread per file decodes the next chromosome while the current one is consumed.  Bulk-capable.
 * A file that is not BigWig: the reference's message and exit(1). */
WiggleIterator *wtamd_BigWiggleReader(const char *path, int box);
/* The same for n files at once, opened side by side on a few threads (every constructor reads its file's index and
 * primes: ~1-2 ms per file, which 100 tracks would otherwise pay one after the other before the first run). */
int wtamd_BigWiggleReaders(int n, const char *const *paths, int box, WiggleIterator **out);
/* Releases what destroyWiggleIterator (wiggleIterator.c:52-55) cannot know about: the file, the decode buffers and
 * the producer thread (joined).  The iterator stays the caller's to free; it must not be popped, sought or given to
 * a reducer afterwards.  A reader that is never closed keeps its file descriptor and, once its second part has been
 * asked for, one idle thread for the life of the process (the reference's reader thread is never joined either,
 * bigWiggleReader.c:147-151).  WTAMD_ERR_ARG for anything that is not an open wtamd_BigWiggleReader. */
int wtamd_BigWiggleReader_close(WiggleIterator *wi);
/* Fused integrators (reference AUCIntegrator / MeanIntegrator, statistics.c:62-127; PearsonIntegrator :414-465; built
 * by commandParser.c:653-704).  Same contract towards the consumer that prints the result: an iterator to be popped
 * to its end whose `data` starts with the double result and whose `append` is the source
 * (PrintStatisticsWiggleIteratorPop, statistics.c:574-590).  Handed a reducer / a 2-track Multiplexer of THIS library
 * that nothing else has popped yet, the integration happens on the device batch by batch and no per-position run
 * is exported: the iterator then yields ONE element per batch (the batch's window, value NaN) instead of one per run
 * -- use the reference's own integrators when something downstream reads the runs.  Anything else: the reference's
 * per-run pass-through on the host. */
WiggleIterator *wtamd_AUCIntegrator(WiggleIterator *wi);
WiggleIterator *wtamd_MeanIntegrator(WiggleIterator *wi);
WiggleIterator *wtamd_PearsonIntegrator(Multiplexer *multi);
/* Consumer door, for reducers built by this library: the runs from the iterator's current element
 * to the end of the batch it belongs to, as arrays valid until the next call on `wi`.  Returns the
 * number of runs (0 and wi->done at the end).  Mixes freely with pop(). */
int64_t wtamd_iterator_next_block(WiggleIterator *wi, const char **chrom, const int32_t **start,
                                  const int32_t **finish, const double **value);
/* Writer hand-off: asks a reducer of this library to merge its runs on the device before they
 * cross PCIe (wtamd_pipe_set_compress on its pipeline; batches already in flight are unaffected).
 * Meant for the iterator the default writer wraps in CompressionWiggleIterator anyway
 * (TeeWiggleIterator, wigWriter.c:261-267): the text written is byte-identical, the pops and the
 * D2H traffic shrink by the compression ratio.  Returns WTAMD_ERR_ARG for any other iterator. */
int wtamd_iterator_compress_output(WiggleIterator *wi, int on);
/* Counters of the pipeline behind a reducer of this library (bytes over PCIe, summed kernel / copy
 * durations from HIP events).  Returns WTAMD_ERR_ARG for any other iterator. */
int wtamd_iterator_pipe_stats(WiggleIterator *wi, wtamd_pipe_stats *out);
/* runWiggleIterator with counters: pops `wi` to the end one run at a time (the reference's own
 * protocol, wiggleIterator.c:62-65); returns the number of runs, their covered bp and value sum. */
int64_t wtamd_drain(WiggleIterator *wi, int64_t *covered_bp, double *value_sum);

/* ---- BigWig section decoder (bulk side door; replaces what the reference gets from libBigWig
 * through src/bigWiggleReader.c:52-83).  HOST only. ---- */
typedef struct wtamd_bw wtamd_bw;
int wtamd_bw_open(const char *path, wtamd_bw **out);    /* prints the reference's message on a non-BigWig file */
void wtamd_bw_close(wtamd_bw *);
int wtamd_bw_n_chrom(const wtamd_bw *);
const char *wtamd_bw_chrom_name(const wtamd_bw *, int i);
uint32_t wtamd_bw_chrom_length(const wtamd_bw *, int i);
/* All runs of one chromosome, 1-based start / exclusive finish, sorted.  box != 0 cuts runs at the
 * reference reader's 10 000-bp stretch edges (bigWiggleReader.c:42-44,73-83).  Returns the number
 * of runs; when that exceeds `capacity` nothing was written (call again with more room). */
int64_t wtamd_bw_read_chrom(wtamd_bw *, const char *chrom, int box, int64_t capacity,
                            int32_t *start, int32_t *finish, float *value);

/* Streaming form (what wtamd_BigWiggleReader's producer thread calls): the runs of the next `max_blocks`
 * data blocks of `chrom` from index-leaf position *cursor on (0: the chromosome's beginning), skipping
 * blocks that lie wholly outside 0-based [lo0, hi0).  Returns the number of runs and advances *cursor;
 * *last = 1 when no block of the chromosome (inside the window) is left.  More runs than `capacity`:
 * nothing written, *cursor unchanged, the decoded part is kept for the repeated call. */
int64_t wtamd_bw_read_part(wtamd_bw *, const char *chrom, int box, int64_t *cursor, int max_blocks, int32_t lo0, int32_t hi0,
                           int64_t capacity, int32_t *start, int32_t *finish, float *value, int *last);

/* ---- Drop-in for the reference's src/bufferedReader.c (same five functions as src/bufferedReader.h:27-31, the
 * struct opaque and free()-able as there): the producer / consumer block buffer of the binary-file readers
 * (bigWiggleReader.c, bamReader.c, bigBedReader.c, bcfReader.c).  Link this library INSTEAD of bufferedReader.o and
 * those readers, unchanged, hand their 10 000-entry blocks to a Multiplexer of this library whole (csrc/wt_bufreader.h). */
typedef struct bufferedReaderData_st BufferedReaderData;
void launchBufferedReader(void *(*readFileFunction)(void *), void *f_data, BufferedReaderData **buf_data);
wt_bool pushValuesToBuffer(BufferedReaderData *data, const char *chrom, int start, int finish, double value);
void endBufferedSignal(BufferedReaderData *data);
void killBufferedReader(BufferedReaderData *data);
void BufferedReaderPop(WiggleIterator *wi, BufferedReaderData *data);
int compare_chrom_lengths(const void *A, const void *B);
long long wtamd_bufreader_bulk_entries(void);      /* entries taken through the bulk door so far (tests) */
/* wtamd_ArrayReader's arrays behind a reader written like the reference's binary-file readers: a producer thread pushes
 * one interval at a time (pushValuesToBuffer), pop is BufferedReaderPop -- the buffered reader's protocol with a producer
 * that costs nothing else (tests, bench leg `e2e.buffered`). */
WiggleIterator *wtamd_BufferedArrayReader(int n_chrom, const char *const *chrom_names, const int64_t *seg_off,
                                          const int32_t *start, const int32_t *finish, const float *value,
                                          double default_value);

/* ---- BigWig WRITER (bench / test plumbing next to the synthetic generator; csrc/wt_bwwrite.cpp): bedGraph sections of
 * `items_per_block` records, one zlib stream each, an R-tree index of as many levels as needed.  Chromosome names in
 * strcmp order (= their ids).  Intervals use the engine's convention: 1-based start, exclusive finish. */
typedef struct wtamd_bw_writer wtamd_bw_writer;
int wtamd_bw_writer_open(const char *path, int n_chrom, const char *const *names, const uint32_t *lengths, int items_per_block,
                         int zlib_level, wtamd_bw_writer **out);
int wtamd_bw_writer_add(wtamd_bw_writer *, int chrom, int64_t n, const int32_t *start, const int32_t *finish, const float *value);
int64_t wtamd_bw_writer_close(wtamd_bw_writer *);      /* index + header; returns the number of sections or < 0 */
/* one chromosome of many files: track i = [seg_off[i], seg_off[i + 1]) of the arrays, dealt to `threads` workers */
int wtamd_bw_writers_add_chrom(wtamd_bw_writer *const *writers, int n_tracks, int chrom, const int64_t *seg_off, const int32_t *start,
                               const int32_t *finish, const float *value, int threads);

/* ---- Synthetic workload generator of SURVEY 8d, on device (bench / test plumbing; csrc/wt_synth.hip).
 * Counter-based: position x of (chromosome c, track t) is a breakpoint iff a hash of (seed, c, t, x)
 * falls below 2^32 / mean_run, the run starting there takes value k/8 (k < levels) and its gap flag
 * from a second hash -- any piece can be regenerated anywhere (wiggletools_amd/synthgen.py holds the
 * numpy mirror the CPU baseline uses).  Two passes with the caller's exclusive scan in between:
 * plan -> count (per 4096-position block) -> scan -> fill. */
int wtamd_synth_plan(int n_chrom, const int32_t *chrom_len, int n_tracks, int64_t *n_blocks, int64_t *seg_first_block);
int wtamd_synth_count(uint64_t seed, int n_chrom, const int32_t *chrom_len, int n_tracks, double mean_run, double gap_prob,
                      int levels, int64_t *counts, void *stream);
int wtamd_synth_fill(uint64_t seed, int n_chrom, const int32_t *chrom_len, int n_tracks, double mean_run, double gap_prob,
                     int levels, const int64_t *block_off, int32_t *start, int32_t *finish, float *value, void *stream);

/* Default value a reducer iterator advertises to its parent
 * (reference reducers.c ctor of each op, incl. float truncations). HOST only. */
double wtamd_reducer_default(int op, int n_tracks, const double *defaults);

#ifdef __cplusplus
}
#endif
#endif /* WIGGLETOOLS_AMD_H_ */
