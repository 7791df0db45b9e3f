// wt_abi_ops.h -- part of the DROP-IN LAYER (csrc/wt_iter_abi.cpp includes it; one translation unit, one anonymous namespace):
// wtamd_MapIterator handles, Select / FillIn pass-throughs, Multiset stepping.
#ifndef WT_ABI_OPS_H_
#define WT_ABI_OPS_H_

namespace {

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

}  // namespace

#endif  // WT_ABI_OPS_H_
