// wt_abi_bwdev.h -- part of the DROP-IN LAYER (csrc/wt_iter_abi.cpp includes it; one translation unit, one anonymous namespace):
// BigWig files decoded ON THE DEVICE: the Feeder's file-byte batches (index arithmetic, read-ahead, fallback to the host decoder).
#ifndef WT_ABI_BWDEV_H_
#define WT_ABI_BWDEV_H_

namespace {

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

}  // namespace

#endif  // WT_ABI_BWDEV_H_
