// wt_abi_common.h -- part of the DROP-IN LAYER (csrc/wt_iter_abi.cpp includes it; one translation unit, one anonymous namespace):
// what the pieces of the drop-in layer share: a child iterator with what was pushed back onto it (TrackSource), seam look-ahead under
// operators that drop runs, the worker pools that drain foreign children in parallel.
#ifndef WT_ABI_COMMON_H_
#define WT_ABI_COMMON_H_

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

}  // namespace

#endif  // WT_ABI_COMMON_H_
