// wt_abi_readers.h -- part of the DROP-IN LAYER (csrc/wt_iter_abi.cpp includes it; one translation unit, one anonymous namespace):
// bulk-capable child iterators of this library: wtamd_ArrayReader and the BigWig reader (host decoder, producer thread per file).
#ifndef WT_ABI_READERS_H_
#define WT_ABI_READERS_H_

namespace {

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

}  // namespace

#endif  // WT_ABI_READERS_H_
