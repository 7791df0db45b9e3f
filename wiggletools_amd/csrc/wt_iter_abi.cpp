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

namespace {

// One child iterator plus intervals that were popped from it but pushed back.
struct TrackSource {
    WiggleIterator *it = nullptr;
    BulkSource *bulk = nullptr;     // non-NULL: the child hands over whole blocks
    std::deque<Ivl> pending;        // pushed back (take-over); precede the iterator's current element
    std::deque<Ivl> log;            // consumed by batches not yet handed to the consumer (Multiplexer mode)
    wtamd_map_chain chain{};        // operators wrapped around `it` (wtamd_MapIterator), run on device
    bool drops = false;             // ... one of them drops runs (ln, log, gt, gte, lt, lte)
    const char *raw = nullptr;      // last chrom pointer seen on `it` ...
    const char *interned = nullptr; // ... and its interned name
    int32_t seen_finish = 0;        // finish of the last element it_chrom() looked at

    // A child may reuse ONE name buffer across chromosomes (same pointer, new content), so a pointer seen before is
    // compared by CONTENT every time, as the reference's multiplexer does (strcmp per pop, multiplexer.c:56).  (Round 3
    // compared only when a start fell below the previous finish: a sparse track whose next chromosome starts beyond
    // the last finish kept the stale name and had its intervals merged into the wrong chromosome -- the advisor's
    // finding.)
    const char *it_chrom(Interner &in) {
        if (it->chrom != raw || !interned || strcmp(raw, interned) != 0) { raw = it->chrom; interned = in.get(raw); }
        seen_finish = it->finish;
        return interned;
    }
    bool empty() const { return pending.empty() && it->done; }
};

// Batch seams under operators that drop runs.  A batch must hold, for every track, the first breakpoint
// at or beyond its cut (the interval that reaches the cut or the first one past it); when the device
// is going to DROP that interval the guarantee moves on to the next interval it keeps.  The host
// cannot see the device's decision, so it evaluates the chain itself -- for these seam intervals
// only -- and asks for certainty: kept, and not within rounding distance of a threshold when a
// transcendental operator (whose last bits differ between libm implementations) came before it.
bool wt_surely_kept(const wtamd_map_chain &c, double v) {
    bool fuzzy = false;
    for (int k = 0; k < c.n_ops; k++) {
        const int op = c.op[k];
        const double p = c.param[k];
        if (op == WTAMD_MAP_LN || op == WTAMD_MAP_LOG) {
            if (!(v != v) && (v <= 0 || (fuzzy && v < 1e-300))) return false;
        } else if (op >= WTAMD_MAP_GT && op <= WTAMD_MAP_LTE) {
            if (fuzzy && v == v) {
                const double d = v > p ? v - p : p - v, m = std::max(std::fabs(v), std::fabs(p));
                if (d <= 1e-9 * m) return false;
            }
        }
        bool keep;
        v = wm_apply(op, p, (op == WTAMD_MAP_LOG || op == WTAMD_MAP_EXPB) ? log(p) : 1.0, v, keep);
        if (!keep) return false;
        if (op == WTAMD_MAP_LN || op == WTAMD_MAP_LOG || op == WTAMD_MAP_EXP || op == WTAMD_MAP_EXPB || op == WTAMD_MAP_POW) fuzzy = true;
    }
    return true;
}

// ---------------------------------------------------------------------------
// Parallel draining of foreign children.  The reference's protocol is one indirect call per interval;
// one host thread sustains ~1.2e8 of them per second, which is what bounded the `pop` leg of the
// end-to-end path.  The protocol only demands that ONE iterator is never entered by two
// threads at once (its readers already run producer threads of their own: bufferedReader.c:118-134),
// so the N children of a Multiplexer are dealt to a few worker threads -- a child always to the same
// one -- which pop them into private buffers; the batch is then laid out in track order and the
// workers copy their tracks into the pinned staging.  WTAMD_DRAIN_THREADS=1 switches it off.
// ---------------------------------------------------------------------------
struct DrainOut {
    std::vector<int32_t> s, f;
    std::vector<double> v;
    std::vector<float> vf;          // f32: the values as float32 (children that hand over float blocks) -- else `v`
    bool f32 = false;
    bool more = false, carry = false, need64 = false;
    int32_t sentinel_lo = INT32_MAX;
    int64_t at = 0;
    void clear(bool as_f32 = false) {
        s.clear(); f.clear(); v.clear(); vf.clear();
        f32 = as_f32;
        more = carry = need64 = false; sentinel_lo = INT32_MAX; at = 0;
    }
    void push(int32_t st, int32_t fi, double x) {
        s.push_back(st); f.push_back(fi);
        const float fl = (float) x;
        const bool exact = !((double) fl != x && x == x);
        if (f32 && exact) { vf.push_back(fl); return; }
        if (f32) { v.assign(vf.begin(), vf.end()); vf.clear(); f32 = false; }
        v.push_back(x);
        if (!exact) need64 = true;
    }
    void append(const int32_t *bs, const int32_t *bf, const float *bv, int64_t k) {       // a block of float32 entries
        s.insert(s.end(), bs, bs + k); f.insert(f.end(), bf, bf + k);
        if (f32) vf.insert(vf.end(), bv, bv + k);
        else v.insert(v.end(), bv, bv + k);
    }
};

struct DrainPool {
    int T = 0;
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    uint64_t gen = 0;
    int pending = 0;
    bool quit = false;
    std::function<void(int)> job;

    void start(int t) {
        T = t;
        // HIP's current device is per thread: children that are reducers of this library issue HIP calls
        // from the worker that pops them, which must land on the device the caller selected
        const int dev = wtamd_current_device();
        for (int w = 0; w < T; w++) th.emplace_back([this, w, dev] { if (dev >= 0) (void) wtamd_set_device(dev); loop(w); });
    }
    void loop(int w) {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_go.wait(lk, [&] { return quit || gen != seen; });
            if (quit) return;
            seen = gen;
            lk.unlock();
            job(w);
            lk.lock();
            if (--pending == 0) cv_done.notify_all();
        }
    }
    void start_job(std::function<void(int)> j) {       // returns at once; wait() before the next job
        std::unique_lock<std::mutex> lk(mu);
        job = std::move(j);
        pending = T;
        gen++;
        cv_go.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    void run(std::function<void(int)> j) {
        start_job(std::move(j));
        wait();
    }
    ~DrainPool() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_go.notify_all();
        for (auto &t : th) t.join();
    }
};

int wt_usable_cores() {
    // WTAMD_HOST_THREADS: the number the library's thread pools are sized by (experiments: a cgroup quota of 16 cores
    // is exhausted by 16 busy workers + the feeder + the runtime's own threads, and the whole group is throttled)
    static const int forced = [] { const char *e = getenv("WTAMD_HOST_THREADS"); const int v = e ? atoi(e) : 0; return v > 0 ? (v > 64 ? 64 : v) : 0; }();
    if (forced) return forced;
    // container CPU quota first (the GPU box shows 256 logical CPUs and grants 16): "quota period" or "max period"
    if (FILE *fp = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64]; long long period = 0;
        const int got = fscanf(fp, "%63s %lld", q, &period);
        fclose(fp);
        if (got == 2 && period > 0 && strcmp(q, "max") != 0) {
            const long long c = atoll(q) / period;
            if (c >= 1) return (int) (c > 64 ? 64 : c);
        }
    }
    const unsigned h = std::thread::hardware_concurrency();
    return h ? (int) (h > 64 ? 64 : h) : 1;
}

const int64_t kDirectMin = 64;              // bulk blocks of at least this many intervals bypass the staging
const int64_t kFirstSpan = 2048;            // bp of a Multiplexer's priming batch
const int64_t kReducerFirstSpan = 65536;    // bp of a reducer's first batch

// ---- BigWig files decoded on the device (details with the BigWig reader further down) ----
struct BwReader;
struct Feeder;
struct BwDevTrack {
    BwReader *r = nullptr;
    int ci = 0;                     // chromosome (index into the reader's strcmp-sorted names); past the end: exhausted
    bool have = false;              // info / cursor / cname describe chromosome ci
    WtBwChromInfo info{};
    int64_t cursor = 0;             // first index leaf (relative to info.first) that can still hold an interval reaching the next batch
    int32_t clip_lo = 1, clip_hi = INT32_MAX;
    int box = 1;
    bool single = false;            // seek window: nothing after this chromosome
    const char *cname = nullptr;    // interned name of chromosome ci
};
void bw_seek(WiggleIterator *wi, const char *chrom, int start, int finish);
struct TrackSource;
BwReader *bwdev_reader(const TrackSource &s);
bool bwdev_eligible(const Feeder &F);
bool bwdev_drain_and_submit(Feeder &F);
void bwdev_fallback(Feeder &F, const char *chrom, int32_t lo, unsigned why);

// Drains the children into pipeline slots and keeps `depth` batches in flight.
struct Feeder {
    std::vector<TrackSource> src;
    std::vector<double> defaults;
    Interner names;
    // One pipe per GPU (WTAMD_DEVICES=all | k; default 1): batches -- (chromosome, run-start range) work items, the
    // reference's own sharding unit, python/wiggletools/parallelWiggleTools.py:63-68,103-113 -- are dealt to the pipes
    // round robin and collected in submission order, i.e. in (strcmp(chrom), start) order (multiplexer.c:56).
    // `pipe` is the pipe of the batch being filled / of the next collect.
    std::vector<wtamd_pipe *> pipes;
    std::vector<int> pipe_dev;          // device ordinal of every pipe
    int home_dev = -1;                  // the caller's device: restored after every call into another device's pipe
    int64_t dealt = 0;                  // batches submitted so far (round robin position)
    wtamd_pipe *pipe = nullptr;
    wtamd_pipe *held_pipe = nullptr;    // pipe of the batch being read (holding)
    int64_t max_runs = 0;               // output capacity of a slot = upper bound of hi - lo
    int64_t target = 0;                 // intervals per steady-state batch
    int depth = 1;                      // batches kept in flight (at most the pipe's slots - 1)
    int n_slots_open = 3;
    int n_pipes = 1;                    // depth counts batches in flight PER PIPE
    bool keep_log = false;              // Multiplexer mode: remember what was consumed (take-over pushes it back)
    bool f64_mode = false;              // a value that is not float32-exact was seen
    bool use_bulk = true;               // WTAMD_NO_BULK=1: children of this library are popped like foreign ones
    bool all_bulk = false;              // every child is a bulk source of this library (float32 SoA): unstaged DMA
    DrainPool *pool = nullptr;          // parallel draining: every child is foreign, nothing is dropped on device
    std::vector<DrainOut> outs;
    // every child is a wtamd_BigWiggleReader: the batches travel as FILE BYTES and are inflated / decoded on the
    // device (wtamd_pipe_submit_bw); the readers' own host decoders idle
    bool bw_mode = false, bw_dirty = true;
    std::vector<BwDevTrack> bwt;
    DrainPool *io_pool = nullptr;       // parallel pread() of the section bytes
    int64_t bw_target_bytes = 0, bw_target_sections = 0;
    // the NEXT file-byte batch: planned, its slot acquired and its bytes being read by the I/O threads while the batches
    // in flight compute (read-ahead: the read of 300 MB is 5 ms of the chain results -> read -> ship -> inflate)
    struct BwPlanned {
        bool valid = false, reading = false, failed = false;
        std::vector<wtamd_bw_section> secs;
        std::vector<wtamd_bw_track> tracks;
        struct ReadOp { int fd; int64_t off, len, dst; };
        std::vector<ReadOp> ops;
        uint8_t *bytes = nullptr;
        int64_t n_bytes = 0;
        int32_t lo = 0, hi = 0;
        const char *chrom = nullptr;
        wtamd_pipe *pipe = nullptr;     // the pipe whose slot was acquired for it
    } bwp;
    bool bw_readahead = true;
    // drain position
    const char *chrom = nullptr;        // chromosome of the batch being / last drained
    bool continuing = false;            // next batch continues `chrom` at next_lo
    int32_t next_lo = 0;
    int64_t span = kFirstSpan, min_span = kFirstSpan;
    // batches in flight, oldest first
    struct Flight { const char *chrom; std::vector<int32_t> consumed; int32_t lo = 0, hi = 0; wtamd_pipe *pipe = nullptr; };
    std::deque<Flight> flights;
    bool holding = false;               // front flight was collected and is being read
    wtamd_pipe_result res{};
    const char *res_chrom = nullptr;
    int32_t res_lo = 0, res_hi = 0;     // window of the batch being read

    int n_tracks() const { return (int) src.size(); }

    // what open() was called with: a pipe that was released at the end of the data is opened again by seek()
    wtamd_reduce_desc o_desc{};
    int64_t o_max_runs = 0, o_first_span = 0;
    int o_n_slots = 0;
    bool opened_once = false, compress_on = false;
    wtamd_pipe_stats last_stats{};      // of the pipe that was released

    void reopen() { if (!pipe && opened_once) open(o_desc, o_max_runs, o_n_slots, o_first_span); }

    // End of the data: the pipe's streams and buffers go back (the pinned ones into the process-wide pool, for
    // the next reducer) instead of idling until the process exits.
    void finish() {
        if (!pipe) return;
        sum_stats(&last_stats);
        close();
    }

    // counters of all pipes together
    void sum_stats(wtamd_pipe_stats *out) const {
        memset(out, 0, sizeof(*out));
        for (wtamd_pipe *q : pipes) {
            wtamd_pipe_stats t;
            if (wtamd_pipe_get_stats(q, &t) != WTAMD_OK) continue;
            out->batches += t.batches; out->intervals += t.intervals; out->runs += t.runs; out->covered_bp += t.covered_bp;
            out->h2d_bytes += t.h2d_bytes; out->d2h_bytes += t.d2h_bytes; out->kernel_ms += t.kernel_ms; out->h2d_ms += t.h2d_ms;
            out->d2h_ms += t.d2h_ms; out->delta_batches += t.delta_batches; out->n_slots += t.n_slots;
            out->host_submit_ms += t.host_submit_ms; out->host_wait_ms += t.host_wait_ms;
            out->bw_sections += t.bw_sections; out->bw_decode_ms += t.bw_decode_ms;
        }
    }

    // the pipe the next batch goes to (round robin), made current together with its device
    void next_fill_pipe() {
        const size_t k = (size_t) (dealt % (int64_t) pipes.size());
        pipe = pipes[k];
    }

    void open(const wtamd_reduce_desc &desc, int64_t max_runs_, int n_slots, int64_t first_span) {
        o_desc = desc; o_max_runs = max_runs_; o_n_slots = n_slots; o_first_span = first_span; opened_once = true;
        wtamd_pipe_config cfg;
        memset(&cfg, 0, sizeof(cfg));
        cfg.n_tracks = n_tracks();
        cfg.n_slots = n_slots;
        cfg.defaults = defaults.data();
        cfg.desc = desc;
        cfg.max_intervals = 1 << 16;
        cfg.max_runs = max_runs_;
        max_runs = max_runs_;
        min_span = env_i64("WTAMD_MIN_SPAN", kFirstSpan);       // tests cut every few bp to stress the seams
        use_bulk = !getenv("WTAMD_NO_BULK");
        all_bulk = !src.empty();
        for (const auto &s : src) all_bulk = all_bulk && s.bulk != nullptr && s.bulk->stable;
        // intervals per batch.  Stable bulk sources are read by the copy engine where they lie (no staging on the host):
        // three times the batch costs device memory only and takes the per-batch share of the link time from 17 % to 8 %
        // (MI355X, round 4: 100 tracks, steady 5.6e8 -> 6.3e8 bp/s; the run capacity of a slot is the other bound)
        target = env_i64("WTAMD_BATCH_INTERVALS", (all_bulk && use_bulk) ? (24 << 20) : (8 << 20));
        if (getenv("WTAMD_MIN_SPAN")) first_span = min_span;
        span = first_span < max_runs ? first_span : max_runs;
        bw_mode = desc.op != WTAMD_OP_MULTIPLEX && bwdev_eligible(*this);
        bw_dirty = true;
        bw_target_bytes = env_i64("WTAMD_BW_BATCH_BYTES", (int64_t) 1 << 30);
        bw_readahead = !(getenv("WTAMD_BW_READAHEAD") && atoi(getenv("WTAMD_BW_READAHEAD")) == 0);
        {
            // WTAMD_DEVICES: "all" or a count; the pipes sit on the devices following the caller's (modulo the
            // number of GPUs: a count above it -- a test aid -- puts several pipes on one device)
            const char *ed = getenv("WTAMD_DEVICES");
            const int n_dev = std::max(wtamd_device_count(), 1);
            int want = 1;
            if (ed && !strcmp(ed, "all")) want = n_dev;
            else if (ed && atoi(ed) > 0) want = std::min(atoi(ed), 64);
            if (desc.op == WTAMD_OP_MULTIPLEX) want = 1;        // (a Multiplexer that is popped run by run: one device)
            home_dev = wtamd_current_device();
            dealt = 0;
            for (int k = 0; k < want; k++) {
                const int dev = ((home_dev >= 0 ? home_dev : 0) + k) % n_dev;
                if (home_dev >= 0 && wtamd_set_device(dev) != WTAMD_OK) die("wtamd_set_device");
                wtamd_pipe *q = nullptr;
                if (wtamd_pipe_create(&cfg, &q) != WTAMD_OK) die("wtamd_pipe_create");
                pipes.push_back(q);
                pipe_dev.push_back(dev);
            }
            if (home_dev >= 0) (void) wtamd_set_device(home_dev);
            pipe = pipes[0];
        }
        for (wtamd_pipe *q : pipes)
            if (compress_on && wtamd_pipe_set_compress(q, 1) != WTAMD_OK) die("wtamd_pipe_set_compress");
        // a batch of file bytes should fill the GPU's inflate lanes once, never more (a second round for a few sections
        // costs half a launch again).  27/32 of the lanes: with two wavefronts per SIMD a launch's time grows with its
        // fill (12.5 ms at 80 %, 15 ms at 100 %: the sections per millisecond stay the same), and the smaller batches keep
        // less memory in flight -- measured (round 4, GRCh38 x 0.5): 80 / 88 / 94 / 100 % within noise of each other.
        bw_target_sections = env_i64("WTAMD_BW_BATCH_SECTIONS", bw_mode ? std::max<int64_t>(wtamd_pipe_bw_fill_sections(pipe) * 27 / 32, 64) : 0);
        n_slots_open = n_slots ? std::min(std::max(n_slots, 2), 8) : 3;     // (wtamd_pipe_create's own clamp)
        if (depth > n_slots_open - 1) depth = n_slots_open - 1;
        n_pipes = (int) pipes.size();
        bool any_map = false;
        std::vector<wtamd_map_chain> chains;
        for (const auto &s : src) { chains.push_back(s.chain); any_map = any_map || s.chain.n_ops > 0; }
        for (wtamd_pipe *q : pipes)
            if (any_map && wtamd_pipe_set_map(q, chains.data()) != WTAMD_OK) die("wtamd_pipe_set_map");
        // parallel draining when every child is popped through the reference's protocol
        bool eligible = !keep_log && !src.empty() && !bw_mode;
        for (const auto &s : src) eligible = eligible && !(s.bulk && use_bulk && s.bulk->peek != &wt_buf_peek) && !s.drops;
        if (bw_mode && !io_pool) {
            const int t = std::max(1, std::min({wt_usable_cores(), 16, n_tracks()}));
            io_pool = new DrainPool();
            io_pool->start(t);
        }
        const char *et = getenv("WTAMD_DRAIN_THREADS");
        int threads = et ? atoi(et) : (n_tracks() >= 16 ? std::min(wt_usable_cores(), 16) : 1);
        if (threads > n_tracks()) threads = n_tracks();
        if (eligible && threads >= 2 && !pool) {
            pool = new DrainPool();
            pool->start(threads);
            outs.resize(src.size());
        }
    }

    void close() {
        drop_planned();
        for (wtamd_pipe *q : pipes) wtamd_pipe_destroy(q);
        pipes.clear(); pipe_dev.clear();
        pipe = nullptr; held_pipe = nullptr;
        delete pool;
        pool = nullptr;
        delete io_pool;
        io_pool = nullptr;
    }

    // One foreign child, popped up to the cut `hi` of chromosome `chrom` (interned) into `o`.  Worker
    // threads run this: it must not intern (the table is not thread-safe) -- a raw name the source
    // has not seen interned yet is compared by content.
    void drain_foreign(TrackSource &s, const char *chrom, int32_t hi, DrainOut &o) {
        // a reader on this library's buffered reader (csrc/wt_bufreader.h): its blocks go over whole, by the worker
        // that owns the child (the door appears with the reader's first pop, which may be later than the constructor)
        if (!s.bulk && use_bulk && s.it->pop != &wt_bulk_pop) s.bulk = wt_bufreader_bulk(s.it);
        BulkSource *door = (s.bulk && use_bulk && s.bulk->peek == &wt_buf_peek) ? s.bulk : nullptr;
        o.clear(door != nullptr);
        while (!s.pending.empty()) {
            const Ivl h = s.pending.front();
            if (h.chrom != chrom) return;
            o.push(h.start, h.finish, h.value);
            if (h.start >= hi) { o.more = true; o.sentinel_lo = h.start; return; }
            if (h.finish >= hi) { o.more = o.carry = true; return; }     // reaches the cut: seen again
            s.pending.pop_front();
        }
        WiggleIterator *it = s.it;
        while (door && !it->done && strcmp(it->chrom, chrom) == 0) {
            const int32_t *bs, *bf;
            const float *bv;
            const int64_t cnt = door->peek(door, &bs, &bf, &bv);
            if (cnt <= 0) break;                                            // (a value that is no float: one pop at a time, below)
            const int64_t k1 = std::lower_bound(bs, bs + cnt, hi) - bs;     // starts below the cut
            const bool reach = k1 > 0 && bf[k1 - 1] >= hi;                 // the last of them reaches it: seen again
            const bool sentinel = !reach && k1 < cnt;
            o.append(bs, bf, bv, k1 + (sentinel ? 1 : 0));
            if (sentinel) { o.more = true; o.sentinel_lo = bs[k1]; }
            if (reach) o.more = o.carry = true;
            const int64_t consumed = reach ? k1 - 1 : k1;
            if (consumed > 0) door->advance(door, it, consumed);
            if (reach || sentinel) return;
        }
        while (!it->done) {
            const char *rc = it->chrom;
            const int32_t st = it->start, fi = it->finish;
            if (strcmp(rc, chrom) != 0) return;             // by content, every pop (multiplexer.c:56): see it_chrom
            s.seen_finish = fi;
            o.push(st, fi, it->value);
            if (st >= hi) { o.more = true; o.sentinel_lo = st; return; }    // sentinel: stays current
            if (fi >= hi) { o.more = o.carry = true; return; }              // reaches the cut: stays current
            it->pop(it);
        }
    }

    // A read-ahead batch that will not be shipped: wait for its reads, give the slot back.
    void drop_planned() {
        if (!bwp.valid) return;
        if (bwp.reading && io_pool) io_pool->wait();
        bwp.reading = false;
        bwp.valid = false;
        if (bwp.pipe) wtamd_pipe_cancel(bwp.pipe);
    }

    // Throws away everything in flight (results included).
    void drop_flights() {
        if (!pipe) return;
        drop_planned();
        if (holding) { wtamd_pipe_release(held_pipe); holding = false; flights.pop_front(); }
        while (!flights.empty()) {
            wtamd_pipe_result r;
            wtamd_pipe *q = flights.front().pipe;
            // (results nobody will read: a file-byte batch that failed to decode may be among them)
            if (wtamd_pipe_collect(q, &r) != WTAMD_OK && !wtamd_pipe_bw_error(q)) die("wtamd_pipe_collect");
            wtamd_pipe_release(q);
            flights.pop_front();
        }
    }

    // Take-over: everything drained but not yet consumed by a reducer goes back to the sources.
    // (A Multiplexer is taken over right after its constructor primed it, commandParser.c:500-569;
    // the reducer then starts from the Multiplexer's first run, as in the reference.)
    void rewind() {
        drop_flights();
        for (auto &s : src) {
            while (!s.log.empty()) { s.pending.push_front(s.log.back()); s.log.pop_back(); }
        }
        continuing = false;
    }

    void reset() {      // after seek: forget everything that was buffered
        drop_flights();
        for (auto &s : src) { s.pending.clear(); s.log.clear(); s.raw = nullptr; s.interned = nullptr; }
        continuing = false;
        bw_dirty = true;        // (device-decoded files: the tracks' positions are read off the re-positioned readers again)
    }

    // A batch was cut at INT32_MAX (an open-ended interval: finish == INT32_MAX always "reaches the cut"):
    // every interval of the chromosome that is still pending or current starts below the cut and was part
    // of the batch, so it is consumed here instead of being carried into an endless series of empty batches.
    void finish_open_ended(const char *c) {
        for (auto &s : src) {
            while (!s.pending.empty() && s.pending.front().chrom == c) s.pending.pop_front();
            if (!s.pending.empty()) continue;
            while (!s.it->done && s.it_chrom(names) == c) s.it->pop(s.it);
        }
        continuing = false;
    }

    // Fills one slot with the next batch and ships it.  False: the sources are exhausted.
    bool drain_and_submit() {
        if (bw_mode) return bwdev_drain_and_submit(*this);
        const int N = n_tracks();
        if (!pool)      // (readers held until their first seek register their buffer then: commandParser.c:615-624)
            for (auto &s : src)
                if (!s.bulk && s.it->pop != &wt_bulk_pop) s.bulk = wt_bufreader_bulk(s.it);
        int32_t lo;
        if (continuing) {
            lo = next_lo;
        } else {
            chrom = nullptr;
            for (int i = 0; i < N; i++) {
                TrackSource &s = src[i];
                if (s.empty()) continue;
                const char *c = s.pending.empty() ? s.it_chrom(names) : s.pending.front().chrom;
                if (!chrom || strcmp(c, chrom) < 0) chrom = c;     // multiplexer.c:56
            }
            if (!chrom) return false;
            int64_t m = INT32_MAX;
            for (int i = 0; i < N; i++) {
                TrackSource &s = src[i];
                if (s.empty()) continue;
                const char *c = s.pending.empty() ? s.it_chrom(names) : s.pending.front().chrom;
                const int32_t st = s.pending.empty() ? s.it->start : s.pending.front().start;
                if (c == chrom && st < m) m = st;
            }
            lo = (int32_t) m;
        }
        const int64_t hi64 = (int64_t) lo + span;
        const int32_t hi = hi64 >= INT32_MAX ? INT32_MAX : (int32_t) hi64;

        const double t_drain0 = g_trace ? now_ms() : 0;
        wtamd_pipe_batch b;
        next_fill_pipe();
        if (wtamd_pipe_acquire(pipe, &b) != WTAMD_OK) die("wtamd_pipe_acquire");
        if (f64_mode && !b.value64 && wtamd_pipe_grow(pipe, 0, b.capacity, 1, &b) != WTAMD_OK) die("wtamd_pipe_grow");
        Flight fl;
        fl.chrom = chrom;
        fl.lo = lo; fl.hi = hi;
        fl.pipe = pipe;
        if (keep_log) fl.consumed.assign((size_t) N, 0);
        int64_t n = 0;
        bool carry = false, more = false;
        int64_t sentinel_lo = INT32_MAX;

        auto put = [&](int32_t st, int32_t fi, double v) {
            if (n >= b.capacity) {      // (n may have jumped past the staging: direct ranges are not staged)
                const int64_t want = 2 * b.capacity > n + 1 ? 2 * b.capacity : n + 1;
                if (wtamd_pipe_grow(pipe, n < b.capacity ? n : b.capacity, want, f64_mode, &b) != WTAMD_OK) die("wtamd_pipe_grow");
            }
            b.start[n] = st;
            b.finish[n] = fi;
            if (f64_mode) {
                b.value64[n] = v;
            } else {
                const float f = (float) v;
                if ((double) f != v && v == v) {        // not float32-exact (NaN is): float64 from here on
                    if (wtamd_pipe_grow(pipe, n < b.capacity ? n : b.capacity, b.capacity, 1, &b) != WTAMD_OK) die("wtamd_pipe_grow");
                    for (int64_t k = 0; k < n; k++) b.value64[k] = (double) b.value32[k];
                    f64_mode = true;
                    b.value64[n] = v;
                } else {
                    b.value32[n] = f;
                }
            }
            n++;
        };

        // the stop interval X (first unconsumed element of the track: pending.front() or the iterator's
        // current one) may be dropped by the track's operators: extend the batch, without consuming
        // anything, to the first interval that surely is not (see wt_surely_kept)
        auto lookahead = [&](TrackSource &s, double xv) {
            if (!s.drops || wt_surely_kept(s.chain, xv)) return;
            WiggleIterator *it = s.it;
            if (s.pending.empty()) {
                Ivl x = { chrom, it->start, it->finish, it->value };
                s.pending.push_back(x);
                it->pop(it);
            }
            for (size_t idx = 1;; idx++) {
                Ivl h;
                if (idx < s.pending.size()) {
                    h = s.pending[idx];
                    if (h.chrom != chrom) return;
                } else {
                    if (it->done || s.it_chrom(names) != chrom) return;
                    h = Ivl{ chrom, it->start, it->finish, it->value };
                    s.pending.push_back(h);
                    it->pop(it);
                }
                put(h.start, h.finish, h.value);
                if (wt_surely_kept(s.chain, h.value)) return;
            }
        };

        if (pool) {
            const int T = pool->T;
            const char *cname = chrom;
            pool->run([&](int w) { for (int i = w; i < N; i += T) drain_foreign(src[(size_t) i], cname, hi, outs[(size_t) i]); });
            bool need64 = false;
            for (int i = 0; i < N; i++) {
                DrainOut &o = outs[(size_t) i];
                o.at = n;
                n += (int64_t) o.s.size();
                need64 = need64 || o.need64;
                more = more || o.more; carry = carry || o.carry;
                if (o.sentinel_lo < sentinel_lo) sentinel_lo = o.sentinel_lo;
            }
            if (need64) f64_mode = true;        // (nothing staged yet: no conversion of earlier entries needed)
            if (n > b.capacity || (f64_mode && !b.value64)) {
                const int64_t want = n > 2 * b.capacity ? n : 2 * b.capacity;
                if (wtamd_pipe_grow(pipe, 0, n > b.capacity ? want : b.capacity, f64_mode, &b) != WTAMD_OK) die("wtamd_pipe_grow");
            }
            for (int i = 0; i < N; i++) b.seg_off[i] = outs[(size_t) i].at;
            const bool w64 = f64_mode;
            pool->run([&](int w) {
                for (int i = w; i < N; i += T) {
                    const DrainOut &o = outs[(size_t) i];
                    const size_t k = o.s.size();
                    if (!k) continue;
                    memcpy(b.start + o.at, o.s.data(), sizeof(int32_t) * k);
                    memcpy(b.finish + o.at, o.f.data(), sizeof(int32_t) * k);
                    if (o.f32) {
                        if (w64) for (size_t q = 0; q < k; q++) b.value64[o.at + (int64_t) q] = (double) o.vf[q];
                        else memcpy(b.value32 + o.at, o.vf.data(), sizeof(float) * k);
                    } else if (w64) memcpy(b.value64 + o.at, o.v.data(), sizeof(double) * k);
                    else for (size_t q = 0; q < k; q++) b.value32[o.at + (int64_t) q] = (float) o.v[q];
                }
            });
        }
        for (int i = 0; i < N && !pool; i++) {
            TrackSource &s = src[i];
            b.seg_off[i] = n;
            bool stop = false;
            while (!s.pending.empty()) {
                const Ivl h = s.pending.front();
                if (h.chrom != chrom) { stop = true; break; }
                put(h.start, h.finish, h.value);
                if (h.start >= hi) { more = true; if (h.start < sentinel_lo) sentinel_lo = h.start; stop = true; lookahead(s, h.value); break; }
                if (h.finish >= hi) { more = carry = true; stop = true; lookahead(s, h.value); break; }     // reaches the cut: seen again
                if (keep_log) { s.log.push_back(h); fl.consumed[i]++; }
                s.pending.pop_front();
            }
            if (stop) continue;
            WiggleIterator *it = s.it;
            if (s.bulk && use_bulk && !keep_log) {
                // bulk side door: whole blocks, no per-interval call; big blocks are not even
                // staged -- the copy engine reads them where they lie
                bool per_interval = false;      // the door has nothing to offer for the current element: the reference's protocol
                while (!it->done && s.it_chrom(names) == chrom) {
                    const int32_t *bs, *bf;
                    const float *bv;
                    const int64_t cnt = s.bulk->peek(s.bulk, &bs, &bf, &bv);
                    // (a buffered reader's value that is no float32, wt_buf_peek: the batch turns float64 in put() below.
                    // Round 4 skipped to the next track here -- the child never advanced and the Feeder span for ever:
                    // the advisor's finding, tests/test_dropin.py::test_dropin_buffered_reader_non_float_values)
                    if (cnt <= 0) { per_interval = true; break; }
                    const int64_t k1 = std::lower_bound(bs, bs + cnt, hi) - bs;     // starts below the cut
                    const bool reach = k1 > 0 && bf[k1 - 1] >= hi;                 // the last of them reaches it: seen again
                    const bool sentinel = !reach && k1 < cnt;
                    const int64_t include = k1 + (sentinel ? 1 : 0);
                    if (all_bulk && include >= kDirectMin) {
                        // every track is float32 SoA of this library: no staging copy at all
                        if (wtamd_pipe_put_direct(pipe, n, include, bs, bf, bv) != WTAMD_OK) die("wtamd_pipe_put_direct");
                        n += include;
                    } else if (include >= kDirectMin) {
                        // mixed with foreign iterators (which may switch the batch to float64): block copy into the staging
                        if (n + include > b.capacity) {
                            const int64_t want = 2 * b.capacity > n + include ? 2 * b.capacity : n + include;
                            if (wtamd_pipe_grow(pipe, n < b.capacity ? n : b.capacity, want, f64_mode, &b) != WTAMD_OK) die("wtamd_pipe_grow");
                        }
                        memcpy(b.start + n, bs, sizeof(int32_t) * (size_t) include);
                        memcpy(b.finish + n, bf, sizeof(int32_t) * (size_t) include);
                        if (f64_mode) for (int64_t q = 0; q < include; q++) b.value64[n + q] = (double) bv[q];
                        else memcpy(b.value32 + n, bv, sizeof(float) * (size_t) include);
                        n += include;
                    } else {
                        for (int64_t q = 0; q < include; q++) put(bs[q], bf[q], (double) bv[q]);
                    }
                    if (sentinel) { more = true; if (bs[k1] < sentinel_lo) sentinel_lo = bs[k1]; }
                    if (reach) more = carry = true;
                    const int64_t consumed = reach ? k1 - 1 : k1;
                    const double xv = (reach || sentinel) ? (double) bv[reach ? k1 - 1 : k1] : 0.0;     // (before advance(): the block may be recycled)
                    if (consumed > 0) s.bulk->advance(s.bulk, it, consumed);
                    if (reach || sentinel) { lookahead(s, xv); break; }
                }
                if (!per_interval) continue;
            }
            while (!it->done) {
                if (s.it_chrom(names) != chrom) break;
                const int32_t st = it->start, fi = it->finish;
                put(st, fi, it->value);
                if (st >= hi) { more = true; if (st < sentinel_lo) sentinel_lo = st; lookahead(s, it->value); break; }   // sentinel: stays current
                if (fi >= hi) { more = carry = true; lookahead(s, it->value); break; }                                    // reaches the cut: stays current
                if (keep_log) { Ivl h = { chrom, st, fi, it->value }; s.log.push_back(h); fl.consumed[i]++; }
                it->pop(it);
            }
        }
        b.seg_off[N] = n;
        const double t_sub0 = g_trace ? now_ms() : 0;
        if (wtamd_pipe_submit(pipe, f64_mode ? 1 : 0, lo, hi) != WTAMD_OK) die("wtamd_pipe_submit");
        if (g_trace) fprintf(stderr, "[feeder] drain %.3f -> %.3f submit -> %.3f  (%lld intervals, [%d, %d))\n", t_drain0, t_sub0, now_ms(), (long long) n, lo, hi);
        flights.push_back(std::move(fl));
        dealt++;
        // where the next batch starts: at the cut if an interval reaches it, else at the first
        // interval beyond it (no track is in play in between: no run can start there)
        continuing = more;
        next_lo = carry ? hi : (int32_t) sentinel_lo;
        if (hi == INT32_MAX && more) finish_open_ended(chrom);    // no run can start at or beyond INT32_MAX: the chromosome is done
        // steer the span towards the interval budget, bounded by the slot's output capacity
        const int64_t max_span = max_runs < ((int64_t) 1 << 31) ? max_runs : ((int64_t) 1 << 31);
        int64_t want = span * 2;
        if (n > 0) {
            const double per_bp = (double) n / (double) std::max<int64_t>((int64_t) hi - lo, 1);
            const double w = (double) target / per_bp;
            want = w > 4e9 ? (int64_t) 4e9 : (int64_t) w;
            if (want > span * 8) want = span * 8;
        }
        if (want < min_span) want = min_span;
        span = want < max_span ? want : max_span;
        return true;
    }

    // Next non-empty batch result; false when everything has been delivered.
    bool next() {
        if (holding) {
            wtamd_pipe_release(held_pipe);
            holding = false;
            if (keep_log) {
                const Flight &f = flights.front();
                for (size_t i = 0; i < src.size(); i++)
                    for (int32_t k = 0; k < f.consumed[i]; k++) src[i].log.pop_front();
            }
            flights.pop_front();
        }
        for (;;) {
            if (depth > n_slots_open - 1) depth = n_slots_open - 1;
            while ((int) flights.size() < depth * n_pipes && drain_and_submit()) { }
            if (flights.empty()) return false;
            const double t_c0 = g_trace ? now_ms() : 0;
            held_pipe = flights.front().pipe;
            if (wtamd_pipe_collect(held_pipe, &res) != WTAMD_OK) {
                // A file-byte batch the device decoder rejected for something libBigWig -- what the reference reads
                // through, src/bigWiggleReader.c:52-83 -- never looks at (items beyond their index leaf's extents, a
                // section it cannot parse, a stream that does not inflate): the host decoder takes over from this
                // batch on.  A truly corrupt stream fails there too, with the reader's own message.
                const unsigned e = bw_mode ? wtamd_pipe_bw_error(held_pipe) : 0u;
                static const bool no_fallback = getenv("WTAMD_BW_NO_FALLBACK") != nullptr;
                if (!e || (e & ~7u) || no_fallback) die("wtamd_pipe_collect");
                const char *fc = flights.front().chrom;
                const int32_t flo = flights.front().lo;
                wtamd_pipe_release(held_pipe);
                flights.pop_front();
                bwdev_fallback(*this, fc, flo, e);
                continue;
            }
            if (g_trace) fprintf(stderr, "[feeder] collect %.3f -> %.3f (%lld runs, %d in flight)\n", t_c0, now_ms(), (long long) res.n_runs, (int) flights.size());
            res_chrom = flights.front().chrom;
            res_lo = flights.front().lo; res_hi = flights.front().hi;
            holding = true;
            if (res.n_runs > 0 || res.integ_valid) return true;
            wtamd_pipe_release(held_pipe);
            holding = false;
            if (keep_log) {
                const Flight &f = flights.front();
                for (size_t i = 0; i < src.size(); i++)
                    for (int32_t k = 0; k < f.consumed[i]; k++) src[i].log.pop_front();
            }
            flights.pop_front();
        }
    }
};

int pipe_depth() { return (int) env_i64("WTAMD_PIPE_DEPTH", 2); }

// ---------------------------------------------------------------------------
// Multiplexer
// ---------------------------------------------------------------------------
struct MuxState {
    Feeder fd;
    int64_t cur = 0;
    bool open = false;
    bool taken_over = false;        // a reducer owns the sources now
};

MuxState *mux_state(Multiplexer *m) { return (MuxState *) m->data; }

// Tile batches hold runs x tracks values: bound the runs per batch by a byte budget, so a
// Multiplexer that is popped directly (mWigWriter, Select / FillIn, Pearson through the C API)
// streams in bounded memory whatever its track count, like the reference does.
int64_t mux_max_runs(int n_tracks) {
    const int64_t budget = env_i64("WTAMD_TILE_BYTES", 64 << 20);
    int64_t r = budget / (9 * (int64_t) n_tracks + 16);
    if (r < kFirstSpan) r = kFirstSpan;
    if (r > (2 << 20)) r = 2 << 20;
    return r;
}

void mux_pop(Multiplexer *m) {
    MuxState *S = mux_state(m);
    if (S->taken_over) { m->done = 1; return; }
    Feeder &F = S->fd;
    if (!S->open) {
        wtamd_reduce_desc d = { WTAMD_OP_MULTIPLEX, m->strict ? WTAMD_STRICT_SET0 : 0u, 0, 0 };
        F.keep_log = true;
        F.depth = 1;                // priming batch only; deeper once the consumer keeps popping
        F.open(d, mux_max_runs(m->count), 3, kFirstSpan);
        S->open = true;
    }
    if (!F.holding || S->cur >= F.res.n_runs) {
        if (F.holding) F.depth = pipe_depth();
        if (!F.next()) { m->done = 1; F.finish(); S->open = false; return; }
        if (F.res.integ_valid) die("popMultiplexer: the batch was integrated on the device (no runs came home)");
        S->cur = 0;
    }
    const int N = m->count;
    const int64_t r = S->cur++;
    m->chrom = (char *) F.res_chrom;
    m->start = F.res.start[r];
    m->finish = F.res.finish[r];
    const double *tv = F.res.tile + (size_t) r * N;
    const uint8_t *ti = F.res.inplay + (size_t) r * N;
    for (int i = 0; i < N; i++) {
        m->values[i] = tv[i];
        m->inplay[i] = (wt_bool) ti[i];
    }
    m->inplay_count = (int) F.res.value[r];
}

void mux_seek(Multiplexer *m, const char *chrom, int start, int finish) {
    MuxState *S = mux_state(m);
    m->done = 0;
    for (int i = 0; i < m->count; i++) seek(m->iters[i], chrom, start, finish);   // multiplexer.c:133-134
    S->fd.reset();
    S->cur = 0;
    S->taken_over = false;
    m->inplay_count = 0;
    popMultiplexer(m);
}

// ---------------------------------------------------------------------------
// Reducers (one- and two-sample): iterate over the fused kernel's run list
// ---------------------------------------------------------------------------
struct RedState {
    Feeder fd;
    int64_t cur = 0;
    bool block_done = false;            // wtamd_iterator_next_block delivered the rest of the current batch
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
    Feeder &F = R->fd;
    R->block_done = false;
    if (!F.holding || R->cur >= F.res.n_runs) {
        if (!F.next()) {
            wi->done = 1;
            if (R->multi) R->multi->done = 1;
            if (R->multiset) R->multiset->done = 1;
            F.finish();
            return;
        }
        if (F.res.integ_valid) die("pop of a reducer whose batch was integrated on the device (no runs came home)");
        R->cur = 0;
    }
    const int64_t r = R->cur++;
    wi->chrom = (char *) F.res_chrom;
    wi->start = F.res.start[r];
    wi->finish = F.res.finish[r];
    wi->value = F.res.value[r];
}

void red_take_over(RedState *R, Multiplexer *m) {
    MuxState *S = mux_state(m);
    S->fd.rewind();
    for (auto &s : S->fd.src) R->fd.src.push_back(std::move(s));
    for (double d : S->fd.defaults) R->fd.defaults.push_back(d);
    for (char *n : S->fd.names.names) R->fd.names.names.push_back(n);   // interned pointers stay valid
    S->fd.names.names.clear();
    S->fd.src.clear();
    S->fd.close();
    S->taken_over = true;
}

void red_seek(WiggleIterator *wi, const char *chrom, int start, int finish) {
    // reference WiggleReducerSeek (reducers.c:25-29) / SetComparisonSeek (setComparisons.c:25-29):
    // seek the children, then pop once.
    RedState *R = red_state(wi);
    for (auto &s : R->fd.src) seek(s.it, chrom, start, finish);
    R->fd.reopen();
    R->fd.reset();
    R->cur = 0;
    if (R->multi) R->multi->done = 0;
    if (R->multiset) R->multiset->done = 0;
    wi->done = 0;
    pop(wi);
}

void red_open(RedState *R, int op, uint32_t flags, int n_set0) {
    // pending intervals of two Multiplexers may carry the same name interned twice: re-intern
    for (auto &s : R->fd.src) {
        for (auto &h : s.pending) h.chrom = R->fd.names.get(h.chrom);
        s.raw = nullptr; s.interned = nullptr;
    }
    wtamd_reduce_desc d = { op, flags, n_set0, 0 };
    R->fd.depth = pipe_depth();
    // (file-byte batches are sized to fill the GPU's inflate lanes: ~65 000 sections, ~11 Mbp at 100 dense tracks)
    bool all_bw = !R->fd.src.empty();
    for (const auto &s : R->fd.src) all_bw = all_bw && bwdev_reader(s) != nullptr;
    R->fd.open(d, env_i64("WTAMD_BATCH_RUNS", all_bw ? (16 << 20) : (4 << 20)), R->fd.depth + (all_bw ? 2 : 1), kReducerFirstSpan);
}

WiggleIterator *make_reducer(Multiplexer *m, int op) {
    RedState *R = new RedState();
    R->multi = m;
    if (g_trace) fprintf(stderr, "[reducer] take-over %.3f\n", now_ms());
    red_take_over(R, m);
    if (g_trace) fprintf(stderr, "[reducer] open %.3f\n", now_ms());
    red_open(R, op, m->strict ? WTAMD_STRICT_SET0 : 0u, 0);
    if (g_trace) fprintf(stderr, "[reducer] opened %.3f\n", now_ms());
    RedData *d = (RedData *) calloc(1, sizeof(RedData));
    d->state = R;
    const double dflt = wtamd_reducer_default(op, m->count, m->default_values);
    return newWiggleIterator(d, &red_pop, &red_seek, dflt, 0);
}

WiggleIterator *make_set_reducer(Multiset *ms, int op) {
    RedState *R = new RedState();
    R->multiset = ms;
    const int n_set0 = ms->multis[0]->count;
    const uint32_t flags = (ms->multis[0]->strict ? WTAMD_STRICT_SET0 : 0u) | (ms->multis[1]->strict ? WTAMD_STRICT_SET1 : 0u);
    red_take_over(R, ms->multis[0]);
    red_take_over(R, ms->multis[1]);
    red_open(R, op, flags, n_set0);
    RedData *d = (RedData *) calloc(1, sizeof(RedData));
    d->state = R;
    return newWiggleIterator(d, &red_pop, &red_seek, NAN, 0);     // setComparisons.c:130,389
}


// ---------------------------------------------------------------------------
// Array-backed reader (bulk-capable child iterator)
// ---------------------------------------------------------------------------
struct ArrReader {
    BulkSource hdr;                 // must stay first (see wt_bulk_pop)
    int n_chrom = 0;
    char **names = nullptr;         // own copies: stable for the process lifetime
    int64_t *seg_off = nullptr;     // own copy
    const int32_t *start = nullptr, *finish = nullptr;
    const float *value = nullptr;
    int c = 0;                      // current chromosome
    int64_t j = 0, end = 0;         // current interval, end of what this chromosome delivers
    bool windowed = false;          // after seek(): one chromosome, intervals clipped to [win_start, win_finish)
    int32_t win_start = 0, win_finish = 0;
    bool done = false;
    int32_t e_start = 0, e_finish = 0;      // the current element when it had to be clipped
    float e_value = 0;

    bool clipped(int64_t g) const { return windowed && (start[g] < win_start || finish[g] > win_finish); }

    void settle(WiggleIterator *wi) {       // skip exhausted chromosomes, refresh the visible fields
        while (!done && j >= end) {
            if (windowed) { done = true; break; }
            c++;
            if (c >= n_chrom) { done = true; break; }
            j = seg_off[c]; end = seg_off[c + 1];
        }
        if (done) { wi->done = 1; return; }
        wi->chrom = names[c];
        wi->start = start[j]; wi->finish = finish[j];
        if (clipped(j)) {
            if (wi->start < win_start) wi->start = win_start;
            if (wi->finish > win_finish) wi->finish = win_finish;
        }
        wi->value = (double) value[j];
    }
};

int64_t arr_peek(BulkSource *b, const int32_t **s, const int32_t **f, const float **v) {
    ArrReader *a = (ArrReader *) b;
    if (a->done || a->j >= a->end) return 0;
    if (a->clipped(a->j)) {                 // a window edge: one clipped copy
        a->e_start = a->start[a->j] < a->win_start ? a->win_start : a->start[a->j];
        a->e_finish = a->finish[a->j] > a->win_finish ? a->win_finish : a->finish[a->j];
        a->e_value = a->value[a->j];
        *s = &a->e_start; *f = &a->e_finish; *v = &a->e_value;
        return 1;
    }
    int64_t k = a->end;
    if (a->windowed && k - 1 > a->j && a->clipped(k - 1)) k--;      // the far edge is delivered on its own
    *s = a->start + a->j; *f = a->finish + a->j; *v = a->value + a->j;
    return k - a->j;
}

void arr_advance(BulkSource *b, WiggleIterator *wi, int64_t k) {
    ArrReader *a = (ArrReader *) b;
    if (a->done) { wi->done = 1; return; }
    a->j += k;
    a->settle(wi);
}

void arr_seek(WiggleIterator *wi, const char *chrom, int start, int finish) {
    // what the reference's readers deliver after seek (bigWiggleReader.c:125-145, wigReader /
    // bedReader likewise): only that chromosome, intervals overlapping [start, finish), clipped
    ArrReader *a = (ArrReader *) wi->data;
    a->windowed = true;
    a->win_start = start; a->win_finish = finish;
    a->done = true;
    for (int c = 0; c < a->n_chrom; c++)
        if (strcmp(a->names[c], chrom) == 0) {
            const int64_t lo = a->seg_off[c], hi = a->seg_off[c + 1];
            a->c = c;
            a->j = std::upper_bound(a->finish + lo, a->finish + hi, start) - a->finish;    // first finish > start
            a->end = std::lower_bound(a->start + lo, a->start + hi, finish) - a->start;    // first start >= finish
            a->done = a->j >= a->end;
            break;
        }
    wi->done = 0;
    if (a->done) { wi->done = 1; return; }
    a->settle(wi);
}

// ---------------------------------------------------------------------------
// BigWig reader (bulk-capable child iterator) -- what the reference gets from libBigWig through
// src/bigWiggleReader.c:52-123 + the producer thread of src/bufferedReader.c:118-134, on top of
// this library's own section decoder (wt_bigwig.cpp): chromosomes in strcmp order (:91-101),
// 1-based starts (:39-40), intervals boxed to 10 000-bp stretches (:42-44,73-83), float values.
// One producer thread per file decodes the NEXT part (a growing number of data blocks: 4, 16, 64,
// 256 -- the first one is small so that constructors, which must prime, return quickly) into the
// idle one of two SoA buffers while the current one is consumed: the reference's 10 000-entry
// blocks (bufferedReader.c:21-28), a few hundred thousand entries at a time.  The buffers are
// recycled, so the source is not `stable`: the Multiplexer copies each block into its pinned
// staging as it takes it (a memcpy, far cheaper than the zlib decode that produced it).
// ---------------------------------------------------------------------------
struct BwBuffer {
    int32_t *start = nullptr, *finish = nullptr;
    float *value = nullptr;
    int64_t cap = 0, n = 0;
    int chrom = -1;             // index into BwReader::names; -1: end of the data
};

struct BwReader {
    wtamd_bw *bw = nullptr;
    std::vector<std::string> names;     // chromosomes in strcmp order
    std::vector<char *> cnames;         // stable char* per chromosome
    int box = 1;
    BwBuffer buf[2];
    int cur = 0;                // buffer being consumed
    int64_t j = 0, end = 0;     // position / end inside it
    bool done = false;
    // window after seek(): one chromosome, clipped
    bool windowed = false;
    int32_t win_start = 0, win_finish = 0;
    int32_t e_start = 0, e_finish = 0;
    float e_value = 0;
    // producer: position in the file (touched by the producer thread only while a request is pending)
    int p_chrom = 0;            // next chromosome index
    int64_t p_cursor = 0;       // wtamd_bw_read_part cursor inside it
    int p_blocks = 4;
    bool p_single = false;      // stop after p_chrom (seek window)
    int p_box = 1;              // box of the parts being decoded: off inside a seek window (one region query, bigWiggleReader.c:91-92)
    int32_t p_lo0 = 0, p_hi0 = INT32_MAX;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    int want_buf = -1;          // buffer the producer should fill next (-1: idle)
    bool ready = false, quit = false, failed = false;
    bool started = false;       // the producer thread exists (it is created when the SECOND part is asked for: the
                                // constructor decodes the priming block itself, and a reducer that ships the file's
                                // sections to the device undecoded never needs the thread)

    bool clipped(int64_t g) const {
        const BwBuffer &b = buf[cur];
        return windowed && (b.start[g] < win_start || b.finish[g] > win_finish);
    }
};

// wi->data of a BigWig reader: free()-able like every iterator's data (wiggleIterator.c:52-55 frees it);
// the reader proper -- buffers, file, producer thread -- lives on (idle) if the iterator is destroyed.
struct BwHandle {
    BulkSource hdr;             // must stay first (see wt_bulk_pop)
    BwReader *r;
};

void bw_free(BwBuffer &b) {
    free(b.start); free(b.finish); free(b.value);
    b.start = b.finish = nullptr; b.value = nullptr; b.cap = 0;
}

bool bw_alloc(BwBuffer &b, int64_t cap) {
    bw_free(b);
    b.start = (int32_t *) malloc(sizeof(int32_t) * (size_t) cap);
    b.finish = (int32_t *) malloc(sizeof(int32_t) * (size_t) cap);
    b.value = (float *) malloc(sizeof(float) * (size_t) cap);
    if (!b.start || !b.finish || !b.value) return false;
    b.cap = cap;
    return true;
}

// the next non-empty part of the file into b (producer thread)
void bw_decode(BwReader *r, BwBuffer &b) {
    b.n = 0;
    b.chrom = -1;
    if (b.cap == 0 && !bw_alloc(b, 1 << 14)) { r->failed = true; return; }
    while (r->p_chrom < (int) r->names.size()) {
        int last = 0;
        const char *name = r->names[(size_t) r->p_chrom].c_str();
        int64_t n = wtamd_bw_read_part(r->bw, name, r->p_box, &r->p_cursor, r->p_blocks, r->p_lo0, r->p_hi0, b.cap, b.start, b.finish,
                                       b.value, &last);
        if (n > b.cap) {
            if (!bw_alloc(b, n + n / 8)) { r->failed = true; return; }
            n = wtamd_bw_read_part(r->bw, name, r->p_box, &r->p_cursor, r->p_blocks, r->p_lo0, r->p_hi0, b.cap, b.start, b.finish,
                                   b.value, &last);
        }
        if (n < 0) { r->failed = true; return; }
        const int ci = r->p_chrom;
        if (r->p_blocks < 256) r->p_blocks *= 4;
        if (last) {
            r->p_chrom = r->p_single ? (int) r->names.size() : r->p_chrom + 1;
            r->p_cursor = 0;
            if (!r->p_single) { r->p_lo0 = 0; r->p_hi0 = INT32_MAX; }      // (a restart position applies to its chromosome only)
        }
        if (n > 0) { b.n = n; b.chrom = ci; return; }
    }
}

void bw_producer(BwReader *r) {
    std::unique_lock<std::mutex> lk(r->mu);
    for (;;) {
        r->cv.wait(lk, [&] { return r->quit || r->want_buf >= 0; });
        if (r->quit) return;
        const int bi = r->want_buf;
        lk.unlock();
        bw_decode(r, r->buf[bi]);
        lk.lock();
        r->want_buf = -1;
        r->ready = true;
        r->cv.notify_all();
    }
}

// asks the producer for the next part in buffer bi (does not wait)
void bw_request(BwReader *r, int bi) {
    std::lock_guard<std::mutex> lk(r->mu);
    r->ready = false;
    r->want_buf = bi;
    r->cv.notify_all();
}

void bw_wait(BwReader *r) {
    std::unique_lock<std::mutex> lk(r->mu);
    r->cv.wait(lk, [&] { return r->ready; });
    if (r->failed) { fprintf(stderr, "wiggletools_amd: BigWig decode failed\n"); exit(1); }
}

// Switches to the part the producer has been decoding into the idle buffer and asks for the one after
// it, whose decode then overlaps the consumption of this one.
void bw_producer(BwReader *r);
void bw_start(BwReader *r) {
    if (r->started) return;
    r->started = true;
    r->th = std::thread(bw_producer, r);        // idles between requests; wtamd_BigWiggleReader_close ends and joins it
    bw_request(r, r->cur ^ 1);
}

void bw_next_part(BwReader *r, WiggleIterator *wi) {
    for (;;) {
        bw_start(r);
        bw_wait(r);
        r->cur ^= 1;
        const BwBuffer &b = r->buf[r->cur];
        if (b.chrom < 0) { r->done = true; wi->done = 1; return; }
        bw_request(r, r->cur ^ 1);
        r->j = 0; r->end = b.n;
        if (r->windowed) {
            r->j = std::upper_bound(b.finish, b.finish + b.n, r->win_start) - b.finish;       // first finish > start
            r->end = std::lower_bound(b.start, b.start + b.n, r->win_finish) - b.start;       // first start >= finish
        }
        if (r->j < r->end) return;
    }
}

void bw_settle(BwReader *r, WiggleIterator *wi) {
    if (!r->done && r->j >= r->end) bw_next_part(r, wi);
    if (r->done) { wi->done = 1; return; }
    const BwBuffer &b = r->buf[r->cur];
    wi->chrom = r->cnames[(size_t) b.chrom];
    wi->start = b.start[r->j]; wi->finish = b.finish[r->j];
    if (r->clipped(r->j)) {
        if (wi->start < r->win_start) wi->start = r->win_start;
        if (wi->finish > r->win_finish) wi->finish = r->win_finish;
    }
    wi->value = (double) b.value[r->j];
}

int64_t bw_peek(BulkSource *bs, const int32_t **s, const int32_t **f, const float **v) {
    BwReader *r = ((BwHandle *) bs)->r;
    if (r->done || r->j >= r->end) return 0;
    const BwBuffer &b = r->buf[r->cur];
    if (r->clipped(r->j)) {
        r->e_start = b.start[r->j] < r->win_start ? r->win_start : b.start[r->j];
        r->e_finish = b.finish[r->j] > r->win_finish ? r->win_finish : b.finish[r->j];
        r->e_value = b.value[r->j];
        *s = &r->e_start; *f = &r->e_finish; *v = &r->e_value;
        return 1;
    }
    int64_t k = r->end;
    if (r->windowed && k - 1 > r->j && r->clipped(k - 1)) k--;
    *s = b.start + r->j; *f = b.finish + r->j; *v = b.value + r->j;
    return k - r->j;
}

void bw_advance(BulkSource *bs, WiggleIterator *wi, int64_t k) {
    BwReader *r = ((BwHandle *) bs)->r;
    if (r->done) { wi->done = 1; return; }
    r->j += k;
    bw_settle(r, wi);
}

void bw_seek(WiggleIterator *wi, const char *chrom, int start, int finish) {
    // bigWiggleReader.c:125-145: the producer is restarted on ONE region query [start, finish) of that
    // chromosome (:91-92 -> readBigWiggleRegion): intervals are boxed into that window only (:42-44), not
    // into the 10 000-bp stretches of a whole-chromosome read (:73-83)
    BwReader *r = ((BwHandle *) wi->data)->r;
    if (r->started) bw_wait(r);             // whatever the producer is decoding lands first; it is idle afterwards
    r->windowed = true;
    r->win_start = start; r->win_finish = finish;
    r->done = false;
    wi->done = 0;
    int ci = (int) r->names.size();
    for (size_t c = 0; c < r->names.size(); c++)
        if (r->names[c] == chrom) ci = (int) c;
    r->p_chrom = ci;                        // unknown chromosome: the producer reports the end at once
    r->p_cursor = 0;
    r->p_blocks = 4;
    r->p_single = true;
    r->p_box = 0;
    r->p_lo0 = start > 0 ? start - 1 : 0;
    r->p_hi0 = finish > 0 ? finish - 1 : 0;
    r->j = r->end = 0;
    if (r->started) bw_request(r, r->cur ^ 1);      // (else bw_next_part starts the producer, which takes the request)
    bw_settle(r, wi);
}

// ---------------------------------------------------------------------------
// BigWig files decoded ON THE DEVICE.  When every child of a reducer is a wtamd_BigWiggleReader, the
// Feeder does not drain intervals at all: per batch it lists, for every file, the index leaves (data
// sections) overlapping the batch's window, pread()s their bytes -- still compressed -- into the slot's
// pinned staging on a few I/O threads and ships them with wtamd_pipe_submit_bw.  The GPU inflates
// (one lane per zlib stream), shifts to 1-based, boxes into the reference reader's 10 000-bp
// stretches, clips to the seek window (bigWiggleReader.c:36-83,125-145) and multiplexes.  What the
// host contributes is the R-tree arithmetic:
//   * a batch [lo, hi) needs, per track, every interval starting below hi that is not wholly before
//     lo, plus the first interval at or beyond hi (the sentinel that gives the last run its true
//     finish): all leaves from the track's cursor that start below hi, and one more;
//   * the cursor moves past a leaf once all of its intervals finish BELOW the next batch's start
//     (a leaf ending exactly at the cut is seen again: its last finish is a breakpoint there).
//     Leaves read twice are decoded twice -- one in ~85 at the default batch size.
// Runs come out exactly as from the host decoder: the device applies the same arithmetic to the same
// items (tests/test_bwdev.py: byte-for-byte the host path's output; WTAMD_BW_DEVICE=0 selects it).
// ---------------------------------------------------------------------------
BwReader *bwdev_reader(const TrackSource &s) {
    if (!s.it || s.it->seek != &bw_seek || s.it->pop != &wt_bulk_pop) return nullptr;
    return ((BwHandle *) s.it->data)->r;
}

bool bwdev_eligible(const Feeder &F) {
    const char *e = getenv("WTAMD_BW_DEVICE");
    if (e && atoi(e) == 0) return false;
    if (F.keep_log || !F.use_bulk || F.src.empty()) return false;
    for (const auto &s : F.src) {
        BwReader *r = bwdev_reader(s);
        if (!r || s.drops) return false;            // (operators that drop runs need the host's seam look-ahead)
        for (const std::string &n : r->names) {
            WtBwChromInfo ci;
            if (!wt_bw_chrom_info(r->bw, n.c_str(), &ci) || !ci.device_ok) return false;
        }
    }
    return true;
}

// Where every track stands: read off the readers (their current element, or what a Multiplexer had
// popped and pushed back), once after open / seek.
void bwdev_init(Feeder &F) {
    F.bwt.assign(F.src.size(), BwDevTrack());
    for (size_t i = 0; i < F.src.size(); i++) {
        TrackSource &s = F.src[i];
        BwDevTrack &t = F.bwt[i];
        BwReader *r = bwdev_reader(s);
        t.r = r;
        t.ci = (int) r->names.size();
        const char *rc = nullptr;
        int32_t rs = 1;
        if (!s.pending.empty()) { rc = s.pending.front().chrom; rs = s.pending.front().start; }
        else if (!s.it->done) { rc = s.it->chrom; rs = s.it->start; }
        s.pending.clear();
        if (!rc) continue;
        for (size_t c = 0; c < r->names.size(); c++)
            if (r->names[c] == rc) t.ci = (int) c;
        t.clip_lo = rs;
        if (r->windowed) { t.clip_hi = r->win_finish; t.box = 0; t.single = true; }
        else { t.clip_hi = INT32_MAX; t.box = r->box; t.single = false; }
    }
}

// Makes t.info / t.cursor describe the track's next chromosome that still has leaves to deliver.
void bwdev_settle(Feeder &F, BwDevTrack &t) {
    const int nc = (int) t.r->names.size();
    const WtBwLeaf *L = wt_bw_leaves(t.r->bw, nullptr);
    while (t.ci < nc) {
        if (!t.have) {
            if (!wt_bw_chrom_info(t.r->bw, t.r->names[(size_t) t.ci].c_str(), &t.info)) { t.info.count = 0; }
            // leaves that end at or before clip_lo hold nothing for this track (sorted, disjoint: binary search)
            int64_t lo = 0, hi = t.info.count;
            while (lo < hi) {
                const int64_t mid = (lo + hi) / 2;
                if ((int64_t) L[t.info.first + mid].end_base + 1 <= (int64_t) t.clip_lo) lo = mid + 1; else hi = mid;
            }
            t.cursor = lo;
            t.cname = F.names.get(t.r->cnames[(size_t) t.ci]);
            t.have = true;
        }
        if (t.cursor < t.info.count && (int64_t) L[t.info.first + t.cursor].start_base + 1 < (int64_t) t.clip_hi) return;
        // chromosome finished
        t.have = false;
        if (t.single) { t.ci = nc; return; }
        t.ci++;
        t.clip_lo = 1;
    }
}

// Plans the next batch (sections, window, cursors), acquires a slot for it and starts reading its bytes on the I/O
// threads.  False: the files are exhausted.
bool bwdev_plan(Feeder &F) {
    const int N = F.n_tracks();
    if (F.bw_dirty) { bwdev_init(F); F.bw_dirty = false; F.continuing = false; }
    for (auto &t : F.bwt) bwdev_settle(F, t);
    int32_t lo;
    if (F.continuing) {
        lo = F.next_lo;
    } else {
        F.chrom = nullptr;
        for (const auto &t : F.bwt)
            if (t.have && (!F.chrom || strcmp(t.cname, F.chrom) < 0)) F.chrom = t.cname;        // multiplexer.c:56
        if (!F.chrom) return false;
        int64_t m = INT32_MAX;
        for (const auto &t : F.bwt) {
            if (!t.have || t.cname != F.chrom) continue;
            const WtBwLeaf &l = wt_bw_leaves(t.r->bw, nullptr)[t.info.first + t.cursor];
            const int64_t st = std::max<int64_t>((int64_t) l.start_base + 1, t.clip_lo);
            if (st < m) m = st;
        }
        lo = (int32_t) m;
    }
    // The cut: as far as the span goes (it grows by 8 per batch from the priming 65 536 bp), but never so far that the
    // batch holds more sections than the GPU's inflate lanes -- every lane inflates one section, a launch takes ~13 ms
    // whether 80 % or 100 % of the lanes are busy, and the sections beyond the lanes wait for a second round (+6 ms:
    // 32 Mbp batches measured 19.5 ms against 12.5 ms for 16 Mbp ones).  The index tells how many sections a cut
    // takes, so the cut is found by bisection instead of being steered by the previous batch's density (round 3; its
    // batches wobbled around 80 % of the lanes because the span was also capped by the slots' output capacity).
    const double t_plan0 = g_trace ? now_ms() : 0;
    int64_t hi64 = std::min<int64_t>((int64_t) lo + F.span, INT32_MAX);
    {
        auto weigh = [&](int64_t cut, int64_t &secs, int64_t &bytes) {
            secs = bytes = 0;
            for (const auto &t : F.bwt) {
                if (!t.have || t.cname != F.chrom || t.cursor >= t.info.count) continue;
                const WtBwLeaf *L = wt_bw_leaves(t.r->bw, nullptr) + t.info.first;
                const int64_t stop = std::min<int64_t>(cut, t.clip_hi);
                // leaves from the cursor on that start below the cut, and the sentinel's
                int64_t a = t.cursor, b2 = t.info.count;
                while (a < b2) { const int64_t mid = (a + b2) / 2; if ((int64_t) L[mid].start_base + 1 < stop) a = mid + 1; else b2 = mid; }
                int64_t e = a;
                if (e < t.info.count && (int64_t) L[e].start_base + 1 < (int64_t) t.clip_hi) e++;
                secs += e - t.cursor;
                if (e > t.cursor) bytes += (int64_t) (L[e - 1].offset + L[e - 1].size - L[t.cursor].offset);     // (leaves of a chromosome lie one after the other)
            }
        };
        int64_t secs, bytes;
        weigh(hi64, secs, bytes);
        if ((secs > F.bw_target_sections || bytes > F.bw_target_bytes) && hi64 > (int64_t) lo + F.min_span) {
            int64_t good = (int64_t) lo + F.min_span, bad = hi64;        // the shortest cut always goes (progress)
            while (bad - good > 64) {
                const int64_t mid = good + (bad - good) / 2;
                weigh(mid, secs, bytes);
                if (secs > F.bw_target_sections || bytes > F.bw_target_bytes) bad = mid; else good = mid;
            }
            hi64 = good;
        }
    }
    const int32_t hi = hi64 >= INT32_MAX ? INT32_MAX : (int32_t) hi64;

    wtamd_pipe_batch b;
    F.next_fill_pipe();
    if (wtamd_pipe_acquire(F.pipe, &b) != WTAMD_OK) die("wtamd_pipe_acquire");
    Feeder::BwPlanned &P = F.bwp;
    P.pipe = F.pipe;
    F.dealt++;                  // (the slot is taken: the next plan goes to the next pipe)
    P.secs.clear(); P.ops.clear();
    P.tracks.assign((size_t) N, wtamd_bw_track());
    P.lo = lo; P.hi = hi; P.chrom = F.chrom; P.failed = false;
    int64_t n_bytes = 0;
    bool more = false;
    for (int i = 0; i < N; i++) {
        BwDevTrack &t = F.bwt[(size_t) i];
        wtamd_bw_track &k = P.tracks[(size_t) i];
        memset(&k, 0, sizeof(k));
        k.first_section = (int32_t) P.secs.size();
        k.clip_lo = 1; k.clip_hi = INT32_MAX;
        if (!t.have || t.cname != F.chrom) continue;
        const WtBwLeaf *L = wt_bw_leaves(t.r->bw, nullptr) + t.info.first;
        const uint32_t ub = wt_bw_uncompress_buf(t.r->bw);
        k.chrom_id = t.info.id; k.chrom_len = t.info.length;
        k.box = t.box; k.compressed = ub ? 1 : 0;
        k.clip_lo = t.clip_lo; k.clip_hi = t.clip_hi;
        k.plain_bytes = ub ? ub : t.info.max_size;
        const int64_t stop = std::min<int64_t>(hi, t.clip_hi);     // leaves starting at or beyond it hold nothing below the cut
        int64_t e = t.cursor;
        while (e < t.info.count && (int64_t) L[e].start_base + 1 < stop) e++;
        if (e < t.info.count && (int64_t) L[e].start_base + 1 < (int64_t) t.clip_hi) e++;        // the sentinel's leaf
        const int fd = wt_bw_fd(t.r->bw);
        for (int64_t q = t.cursor; q < e; q++) {
            const WtBwLeaf &l = L[q];
            if (!P.ops.empty() && P.ops.back().fd == fd && P.ops.back().off + P.ops.back().len == (int64_t) l.offset) P.ops.back().len += (int64_t) l.size;
            else P.ops.push_back(Feeder::BwPlanned::ReadOp{ fd, (int64_t) l.offset, (int64_t) l.size, n_bytes });
            wtamd_bw_section sc;
            sc.comp_off = n_bytes; sc.comp_size = (uint32_t) l.size; sc.track = i;
            sc.leaf_start = l.start_base; sc.leaf_end = l.end_base;
            P.secs.push_back(sc);
            n_bytes += (int64_t) l.size;
        }
        k.n_sections = (int32_t) (e - t.cursor);
        // retire the leaves no later batch can need: every interval finishes below the cut
        if (hi == INT32_MAX) t.cursor = t.info.count;
        else while (t.cursor < t.info.count && (int64_t) L[t.cursor].end_base + 1 < (int64_t) hi) t.cursor++;
        if (t.cursor < t.info.count && (int64_t) L[t.cursor].start_base + 1 < (int64_t) t.clip_hi && hi < t.clip_hi) more = true;
        else { t.cursor = t.info.count; }       // nothing of this chromosome is left for this track
    }
    P.n_bytes = n_bytes;
    wtamd_bw_section *tab = nullptr;
    if (wtamd_pipe_bw_reserve(F.pipe, n_bytes, (int64_t) P.secs.size(), &P.bytes, &tab) != WTAMD_OK) die("wtamd_pipe_bw_reserve");
    if (!P.secs.empty()) memcpy(tab, P.secs.data(), sizeof(wtamd_bw_section) * P.secs.size());
    {
        const int T = F.io_pool ? F.io_pool->T : 1;
        Feeder::BwPlanned *pp = &P;
        auto work = [pp, T](int w) {
            for (size_t q = (size_t) w; q < pp->ops.size(); q += (size_t) T) {
                int64_t done = 0;
                while (done < pp->ops[q].len) {
                    const ssize_t got = pread(pp->ops[q].fd, pp->bytes + pp->ops[q].dst + done, (size_t) (pp->ops[q].len - done), (off_t) (pp->ops[q].off + done));
                    if (got <= 0) { pp->failed = true; break; }
                    done += got;
                }
            }
        };
        if (F.io_pool && P.ops.size() > 1) { F.io_pool->start_job(work); P.reading = true; }
        else { work(0); P.reading = false; }
    }
    P.valid = true;
    if (g_trace) fprintf(stderr, "[feeder] bw plan %.3f -> %.3f  (%lld sections, %lld bytes, [%d, %d))\n", t_plan0, now_ms(),
                         (long long) P.secs.size(), (long long) n_bytes, lo, hi);
    F.continuing = more;
    F.next_lo = hi;
    if (more) {
        // a gap in every track beyond the cut: no run can start inside it, the next batch begins where data does
        int64_t first = INT32_MAX;
        for (const auto &t : F.bwt) {
            if (!t.have || t.cname != F.chrom || t.cursor >= t.info.count) continue;
            first = std::min<int64_t>(first, (int64_t) wt_bw_leaves(t.r->bw, nullptr)[t.info.first + t.cursor].start_base + 1);
        }
        if (first > hi && first < INT32_MAX) F.next_lo = (int32_t) first;
    }
    // the span grows by 8 per batch up to the slot's output capacity; the section / byte budget cuts it short (above)
    const int64_t max_span = F.max_runs < ((int64_t) 1 << 31) ? F.max_runs : ((int64_t) 1 << 31);
    int64_t want = std::max<int64_t>(F.span, (int64_t) hi - lo) * 8;
    if (want < F.min_span) want = F.min_span;
    F.span = want < max_span ? want : max_span;
    return true;
}

bool bwdev_drain_and_submit(Feeder &F) {
    Feeder::BwPlanned &P = F.bwp;
    if (!P.valid && !bwdev_plan(F)) return false;
    const double t_wait0 = g_trace ? now_ms() : 0;
    if (P.reading) { F.io_pool->wait(); P.reading = false; }
    if (P.failed) { fprintf(stderr, "wiggletools_amd: short read of BigWig data sections\n"); exit(1); }
    const double t_sub0 = g_trace ? now_ms() : 0;
    if (wtamd_pipe_submit_bw(P.pipe, P.n_bytes, (int64_t) P.secs.size(), P.tracks.data(), P.lo, P.hi) != WTAMD_OK) die("wtamd_pipe_submit_bw");
    if (g_trace) fprintf(stderr, "[feeder] bw read-wait %.3f submit %.3f -> %.3f  [%d, %d)\n", t_wait0, t_sub0, now_ms(), P.lo, P.hi);
    Feeder::Flight fl;
    fl.chrom = P.chrom;
    fl.lo = P.lo; fl.hi = P.hi;
    fl.pipe = P.pipe;
    F.flights.push_back(std::move(fl));
    P.valid = false;
    // read-ahead: the next batch's bytes are fetched while the consumer waits for results (needs a free slot:
    // the pipe was opened with two more slots than batches in flight)
    if (F.bw_readahead) {
        const wtamd_pipe *target = F.pipes[(size_t) (F.dealt % (int64_t) F.pipes.size())];
        int busy = 0;           // slots of that pipe in flight or being read by the consumer
        for (const auto &f : F.flights) busy += f.pipe == target ? 1 : 0;
        if (busy + 2 <= F.n_slots_open) (void) bwdev_plan(F);
    }
    return true;
}

// Repositions a reader on chromosome index ci (of its own, strcmp-sorted names) so that its current element is the
// first interval finishing at or beyond `lo` -- boxed and windowed as before.  (The producer skips the data blocks
// that end before lo; the caller pops past the few intervals of the first block kept that still finish below it.)
void bw_restart(WiggleIterator *wi, int ci, int32_t lo) {
    BwReader *r = ((BwHandle *) wi->data)->r;
    if (r->started) bw_wait(r);
    r->done = false;
    wi->done = 0;
    r->p_chrom = ci;
    r->p_cursor = 0;
    r->p_blocks = 4;
    const int32_t lo0 = lo > 2 ? lo - 2 : 0;
    if (r->windowed) {
        r->p_single = true; r->p_box = 0;
        r->p_lo0 = std::max<int32_t>(r->win_start > 0 ? r->win_start - 1 : 0, lo0);
        r->p_hi0 = r->win_finish > 0 ? r->win_finish - 1 : 0;
    } else {
        r->p_single = false; r->p_box = r->box;
        r->p_lo0 = lo0; r->p_hi0 = INT32_MAX;
    }
    r->j = r->end = 0;
    if (r->started) bw_request(r, r->cur ^ 1);
    bw_settle(r, wi);
}

// The device decoder gave up on the batch [lo, ...) of `chrom`: everything in flight is dropped, the readers are moved
// to that position and the Feeder goes on draining them through their host decoders (bw_mode off for good).
void bwdev_fallback(Feeder &F, const char *chrom, int32_t lo, unsigned why) {
    fprintf(stderr, "wiggletools_amd: note: the device BigWig decoder rejected a batch at %s:%d (%s%s%s); continuing with the host decoder\n",
            chrom, lo, (why & 1u) ? "zlib stream / checksum " : "", (why & 2u) ? "malformed section " : "",
            (why & 4u) ? "items outside their index leaf or out of order" : "");
    F.drop_flights();
    F.bw_mode = false;
    F.bw_dirty = true;
    delete F.io_pool;
    F.io_pool = nullptr;
    for (auto &s : F.src) {
        BwReader *r = bwdev_reader(s);
        int ci = (int) r->names.size();
        for (size_t c = r->names.size(); c-- > 0;)
            if (strcmp(r->names[c].c_str(), chrom) >= 0) ci = (int) c;      // first chromosome at or after `chrom` (sorted names)
        s.pending.clear(); s.log.clear(); s.raw = nullptr; s.interned = nullptr;
        const bool on_it = ci < (int) r->names.size() && r->names[(size_t) ci] == chrom;
        bw_restart(s.it, ci, on_it ? lo : 1);
        while (on_it && !s.it->done && !strcmp(s.it->chrom, chrom) && s.it->finish < lo) s.it->pop(s.it);
    }
    F.chrom = chrom;
    F.continuing = true;
    F.next_lo = lo;
}

// ---------------------------------------------------------------------------
// Operator iterator (wtamd_MapIterator): the reference's value maps around one track
// (unaryOps.c:650-949, :386-419).  newMultiplexer unwraps it (wt_unwrap_maps): the engine drains
// the raw child and runs the chain on device.  pop / seek below are the per-interval protocol for
// any other consumer -- one wm_apply per interval, runs the operator drops are skipped
// (LogWiggleIteratorPop :760-779, HighPassFilterWiggleIteratorPop :387-412).
// ---------------------------------------------------------------------------
struct MapIter {
    WiggleIterator *child;
    int op;
    double param, lg;
};

void map_settle(WiggleIterator *wi) {
    MapIter *m = (MapIter *) wi->data;
    WiggleIterator *c = m->child;
    while (!c->done) {
        bool keep;
        const double v = wm_apply(m->op, m->param, m->lg, c->value, keep);
        if (keep) {
            wi->chrom = c->chrom; wi->start = c->start; wi->finish = c->finish; wi->value = v;
            return;
        }
        c->pop(c);
    }
    wi->done = 1;
}

void map_pop(WiggleIterator *wi) {
    MapIter *m = (MapIter *) wi->data;
    if (wi->done) return;
    if (!m->child->done) m->child->pop(m->child);
    map_settle(wi);
}

void map_seek(WiggleIterator *wi, const char *chrom, int start, int finish) {
    MapIter *m = (MapIter *) wi->data;
    m->child->done = 0;
    m->child->seek(m->child, chrom, start, finish);
    wi->done = 0;
    map_settle(wi);
}

// Peels the wtamd_MapIterator layers off `wi`: returns the raw child, fills `chain` innermost first.
WiggleIterator *wt_unwrap_maps(WiggleIterator *wi, wtamd_map_chain &chain) {
    int ops[WTAMD_MAP_CHAIN_MAX];
    double params[WTAMD_MAP_CHAIN_MAX];
    int n = 0;
    while (wi->pop == &map_pop && n < WTAMD_MAP_CHAIN_MAX) {
        MapIter *m = (MapIter *) wi->data;
        ops[n] = m->op; params[n] = m->param; n++;
        wi = m->child;
    }
    chain.n_ops = n;
    for (int k = 0; k < n; k++) { chain.op[k] = ops[n - 1 - k]; chain.param[k] = params[n - 1 - k]; }
    return wi;
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

// ---------------------------------------------------------------------------
// Genome-wide integrators (reference statistics.c:62-127 AUC / mean, :414-465 Pearson).  Towards the
// consumer they are what the reference's are: an iterator popped to its end, `data` starting with the
// double result, `append` = the source (PrintStatisticsWiggleIteratorPop reads exactly that).  Fused: the
// source is a reducer (a 2-track Multiplexer) of this library that nothing has popped since its
// constructor primed it -- the integrals are computed on the device batch by batch
// (wtamd_pipe_set_integrate), one element per BATCH is handed on.  Otherwise: the reference's per-run
// pass-through, on the host (like Select / FillIn, this is glue around pop()).
// ---------------------------------------------------------------------------
struct IntegData {
    double res;                 // must stay first: the consumer prints *(double *) wi->data (statistics.c:579)
    WiggleIterator *source;
    Multiplexer *multi;
    int kind;                   // 0 AUC, 1 mean, 2 Pearson
    int fused;
    int primed;                 // the held batch of the source has been absorbed
    double sum, span;
    double mom[6];              // fused Pearson: moments so far
    int count;                  // host Pearson: the reference's `int count` (statistics.c:400), sums below
    double sum_X, sum_Y, T_XX, T_XY, T_YY;
};

void integ_finish(WiggleIterator *wi, IntegData *d) {
    if (d->kind == 0) d->res = d->sum;
    else if (d->kind == 1) { if (d->span > 0) d->res = d->sum / d->span; }
    else if (d->fused) d->res = wtamd_pearson_finish(d->mom);
    else if (d->T_XX * d->T_YY != 0.0) d->res = d->T_XY / sqrt(d->T_XX * d->T_YY);
    wi->done = 1;
}

void integ_absorb(IntegData *d, Feeder &F) {
    double g[6];
    if (F.res.integ_valid) memcpy(g, F.res.integ, sizeof g);
    else if (wtamd_pipe_integrate_held(F.held_pipe, g) != WTAMD_OK) die("wtamd_pipe_integrate_held");
    if (d->kind == 2) wtamd_pearson_merge(d->mom, g);
    else { d->sum += g[0]; d->span += g[1]; if (d->kind == 0) d->res = d->sum; }
}

void integ_fused_pop(WiggleIterator *wi) {
    if (wi->done) return;
    IntegData *d = (IntegData *) wi->data;
    Feeder &F = d->kind == 2 ? mux_state(d->multi)->fd : red_state(d->source)->fd;
    if (!d->primed) {
        d->primed = 1;
        const bool empty = d->kind == 2 ? d->multi->done != 0 : d->source->done != 0;
        if (empty || !F.pipe || !F.holding) { integ_finish(wi, d); return; }
        for (wtamd_pipe *q : F.pipes)
            if (wtamd_pipe_set_integrate(q, 1) != WTAMD_OK) die("wtamd_pipe_set_integrate");
    } else if (!F.next()) {
        if (d->kind == 2) d->multi->done = 1; else d->source->done = 1;
        F.finish();
        if (d->kind == 2) mux_state(d->multi)->open = false;
        integ_finish(wi, d);
        return;
    } else if (d->kind == 2) {
        F.depth = pipe_depth();             // (a Multiplexer primes with one batch in flight)
    }
    integ_absorb(d, F);
    wi->chrom = (char *) F.res_chrom;
    wi->start = F.res_lo; wi->finish = F.res_hi;
    wi->value = NAN;
}

void integ_fused_seek(WiggleIterator *wi, const char *chrom, int start, int finish) {
    // StatisticSeek / MeanSeek / PearsonSeek (statistics.c:38-43,84-88,406-410): seek the source, pop -- the sums go on
    IntegData *d = (IntegData *) wi->data;
    // The source's seek re-primes by popping RUNS: its pipes go back to shipping them (a pass that ended mid-stream left
    // them integrating: the priming pop would have read a batch without runs); the pop below switches them over again
    // and integrates the primed batches where they lie.
    Feeder &F = d->kind == 2 ? mux_state(d->multi)->fd : red_state(d->source)->fd;
    for (wtamd_pipe *q : F.pipes)
        if (wtamd_pipe_set_integrate(q, 0) != WTAMD_OK) die("wtamd_pipe_set_integrate");
    if (d->kind == 2) seekMultiplexer(d->multi, chrom, start, finish); else seek(d->source, chrom, start, finish);
    d->primed = 0;
    wi->done = 0;
    integ_fused_pop(wi);
}

void integ_host_pop(WiggleIterator *wi) {
    if (wi->done) return;
    IntegData *d = (IntegData *) wi->data;
    if (d->kind == 2) {                     // PearsonPop, statistics.c:414-458
        Multiplexer *m = d->multi;
        if (m->done) { integ_finish(wi, d); return; }
        wi->chrom = m->chrom; wi->start = m->start; wi->finish = m->finish; wi->value = m->values[1];
        const double X = m->inplay[0] ? m->values[0] : m->iters[0]->default_value;
        const double Y = m->inplay[1] ? m->values[1] : m->iters[1]->default_value;
        const int length = m->finish - m->start;
        if (d->count) {
            const double old_mean_X = d->sum_X / d->count, new_mean_X = d->sum_X / (d->count + length);
            const double old_mean_Y = d->sum_Y / d->count, new_mean_Y = d->sum_Y / (d->count + length);
            const double scaling_ratio = (double) d->count / (d->count + length);
            d->T_XY += (new_mean_X * old_mean_Y + scaling_ratio * X * Y - new_mean_X * Y - new_mean_Y * X) * length;
            d->T_XX += (new_mean_X * (old_mean_X - 2 * X) + scaling_ratio * X * X) * length;
            d->T_YY += (new_mean_Y * (old_mean_Y - 2 * Y) + scaling_ratio * Y * Y) * length;
        }
        d->count += length;
        d->sum_X += X * length;
        d->sum_Y += Y * length;
        popMultiplexer(m);
        return;
    }
    WiggleIterator *src = d->source;        // MeanPop / AUCPop, statistics.c:62-82,103-120
    if (src->done) { integ_finish(wi, d); return; }
    wi->chrom = src->chrom; wi->start = src->start; wi->finish = src->finish; wi->value = src->value;
    if (!(wi->value != wi->value)) {
        d->sum += (wi->finish - wi->start) * wi->value;
        d->span += (wi->finish - wi->start);
        if (d->kind == 0) d->res = d->sum;
    }
    pop(src);
}

void integ_host_seek(WiggleIterator *wi, const char *chrom, int start, int finish) {
    IntegData *d = (IntegData *) wi->data;
    if (d->kind == 2) seekMultiplexer(d->multi, chrom, start, finish); else seek(d->source, chrom, start, finish);
    wi->done = 0;
    pop(wi);
}

WiggleIterator *make_integrator(WiggleIterator *src, Multiplexer *multi, int kind) {
    IntegData *d = (IntegData *) calloc(1, sizeof(IntegData));
    d->kind = kind;
    d->multi = multi;
    d->res = kind == 0 ? 0.0 : NAN;          // statistics.c:98,125,463
    bool fused = !getenv("WTAMD_NO_FUSED_INTEGRATORS");
    WiggleIterator *tail;
    double dflt;
    if (kind == 2) {
        MuxState *S = multi->pop == &mux_pop ? mux_state(multi) : nullptr;
        fused = fused && S && !S->taken_over && multi->count == 2 && (multi->done || (S->open && S->cur == 1 && S->fd.holding));
        tail = multi->iters[1];
        dflt = multi->iters[1]->default_value;
    } else {
        d->source = NonOverlappingWiggleIterator(src);
        RedState *R = d->source->pop == &red_pop ? red_state(d->source) : nullptr;
        fused = fused && R && (d->source->done || (R->cur == 1 && !R->block_done && R->fd.holding && R->fd.pipe));
        tail = src;
        dflt = src->default_value;
    }
    d->fused = fused ? 1 : 0;
    WiggleIterator *wi = newWiggleIterator(d, fused ? &integ_fused_pop : &integ_host_pop, fused ? &integ_fused_seek : &integ_host_seek, dflt, 0);
    wi->append = tail;
    return wi;
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
