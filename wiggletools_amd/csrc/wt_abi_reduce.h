// wt_abi_reduce.h -- part of the DROP-IN LAYER (csrc/wt_iter_abi.cpp includes it; one translation unit, one anonymous namespace):
// Multiplexer and reducer iterators over the pipeline's results (popMultiplexer's coherent fields; take-over by a reducer).
#ifndef WT_ABI_REDUCE_H_
#define WT_ABI_REDUCE_H_

namespace {

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

}  // namespace

#endif  // WT_ABI_REDUCE_H_
