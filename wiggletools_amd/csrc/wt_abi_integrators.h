// wt_abi_integrators.h -- part of the DROP-IN LAYER (csrc/wt_iter_abi.cpp includes it; one translation unit, one anonymous namespace):
// genome-wide integrators (AUC, mean, Pearson): fused on the device or per run on the host.
#ifndef WT_ABI_INTEGRATORS_H_
#define WT_ABI_INTEGRATORS_H_

namespace {

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

#endif  // WT_ABI_INTEGRATORS_H_
