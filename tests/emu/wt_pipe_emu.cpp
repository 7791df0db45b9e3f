// wt_pipe_emu.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// The streaming-pipeline entry points of include/wiggletools_amd.h (wtamd_pipe_*) on top of the
// CPU emulator of the kernels (wt_emu.cpp), synchronously.  Linked with the product's own
// csrc/wt_iter_abi.cpp + csrc/wt_defaults.cpp into tests/emu/libwt_dropin_emu.so, so that the
// drop-in layer's host logic -- the Drainer's batch cuts, carried intervals and sentinels, seek,
// take-over, Multiset stepping, the Feeder's slot bookkeeping -- runs against the oracle and the
// compiled reference in the GPU-less container.  The product's pipe (csrc/wt_pipe.h: pinned
// staging, HIP streams, events) is exercised by the -m gpu tests.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/wiggletools_amd.h"
#include "../../wiggletools_amd/csrc/wt_mapop.h"
#include "../../wiggletools_amd/csrc/wt_inflate.h"
#include "../../wiggletools_amd/csrc/wt_bwdev_core.h"

extern "C" long long wtemu_reduce(int n_chrom, int n_tracks, const int64_t *seg_off, const int32_t *start,
                                  const int32_t *finish, const void *value, int value_is_f64, const double *defaults,
                                  int op, unsigned flags, int n_set0, long long capacity,
                                  int32_t *o_start, int32_t *o_finish, double *o_value, int64_t *chrom_run_off,
                                  double *o_tile, uint8_t *o_inplay, long long *info,
                                  const int32_t *range_lo, const int32_t *range_hi);

namespace {
std::string g_err;

struct Slot {
    int state = 0;      // 0 free, 1 acquired, 2 submitted, 3 collected
    std::vector<int64_t> seg_off;
    std::vector<int32_t> start, finish;
    std::vector<float> v32;
    std::vector<double> v64;
    bool has64 = false;
    int64_t cap = 0;
    // result
    std::vector<int32_t> os, of;
    std::vector<double> ov, tile;
    std::vector<uint8_t> ip;
    int64_t n_runs = 0, covered = 0, n_int = 0;
    struct Direct { int64_t at, count; const int32_t *start, *finish; const float *value; };
    std::vector<Direct> direct;
    // wtamd_pipe_bw_reserve: file bytes + section table of a batch that is decoded "on device"
    std::vector<uint8_t> bw_bytes;
    std::vector<wtamd_bw_section> bw_secs;
    int64_t bw_res_bytes = -1, bw_res_secs = -1;
    bool integrated = false;
    double integ[6] = {0, 0, 0, 0, 0, 0};
    unsigned bw_err = 0;        // the "device" decoder's error bits: reported when the batch is collected
};
}  // namespace

struct wtamd_pipe {
    wtamd_pipe_config cfg;
    std::vector<double> defaults;
    std::vector<Slot> slots;
    int head = 0;       // next slot to acquire
    int tail = 0;       // oldest submitted / collected slot
    int acquired = -1;
    int in_flight = 0;  // submitted, not collected
    int held = 0;       // collected, not released
    bool compress = false, integrate = false;
    unsigned last_bw_err = 0;
    std::vector<wtamd_map_chain> chains;    // wtamd_pipe_set_map (empty: off)
    wtamd_pipe_stats st{};
};

extern "C" {
static void emu_integrals(const wtamd_pipe *p, const Slot &s, int64_t n, double *g);


const char *wtamd_last_error(void) { return g_err.c_str(); }

int wtamd_pipe_create(const wtamd_pipe_config *cfg, wtamd_pipe **out) {
    if (!cfg || !out || cfg->n_tracks <= 0 || !cfg->defaults || cfg->max_runs <= 0) { g_err = "bad pipe config"; return WTAMD_ERR_ARG; }
    wtamd_pipe *p = new wtamd_pipe();
    p->cfg = *cfg;
    p->defaults.assign(cfg->defaults, cfg->defaults + cfg->n_tracks);
    p->cfg.defaults = p->defaults.data();
    int ns = cfg->n_slots ? cfg->n_slots : 3;
    if (ns < 2) ns = 2;
    if (ns > 8) ns = 8;
    p->slots.resize((size_t) ns);
    // tests: a tiny initial capacity so that wtamd_pipe_grow is exercised
    const char *e = getenv("WTEMU_PIPE_CAP");
    const int64_t cap = e ? atoll(e) : (cfg->max_intervals > 0 ? cfg->max_intervals : 1024);
    for (auto &s : p->slots) {
        s.cap = cap;
        s.seg_off.assign((size_t) cfg->n_tracks + 1, 0);
        s.start.resize((size_t) cap); s.finish.resize((size_t) cap); s.v32.resize((size_t) cap);
    }
    p->st.n_slots = ns;
    p->compress = (cfg->flags & WTAMD_PIPE_COMPRESS) != 0;
    *out = p;
    return WTAMD_OK;
}

int wtamd_pipe_set_compress(wtamd_pipe *p, int on) {
    if (!p) return WTAMD_ERR_ARG;
    p->compress = on != 0;
    return WTAMD_OK;
}

int wtamd_pipe_set_map(wtamd_pipe *p, const wtamd_map_chain *chains) {
    if (!p) return WTAMD_ERR_ARG;
    p->chains.clear();
    if (chains) p->chains.assign(chains, chains + p->cfg.n_tracks);
    return WTAMD_OK;
}

void wtamd_pipe_destroy(wtamd_pipe *p) { delete p; }

static void fill_batch(Slot &s, wtamd_pipe_batch *b) {
    b->capacity = s.cap;
    b->seg_off = s.seg_off.data();
    b->start = s.start.data(); b->finish = s.finish.data();
    b->value32 = s.v32.data();
    b->value64 = s.has64 ? s.v64.data() : nullptr;
}

int wtamd_pipe_acquire(wtamd_pipe *p, wtamd_pipe_batch *out) {
    if (!p || !out) { g_err = "NULL"; return WTAMD_ERR_ARG; }
    if (p->acquired >= 0) { g_err = "a slot is already acquired"; return WTAMD_ERR_ARG; }
    Slot &s = p->slots[(size_t) p->head];
    if (s.state != 0) { g_err = "every slot is in flight or unreleased"; return WTAMD_ERR_ARG; }
    s.state = 1;
    p->acquired = p->head;
    fill_batch(s, out);
    return WTAMD_OK;
}

int wtamd_pipe_grow(wtamd_pipe *p, int64_t used, int64_t min_capacity, int want_f64, wtamd_pipe_batch *out) {
    if (!p || p->acquired < 0) { g_err = "no acquired slot"; return WTAMD_ERR_ARG; }
    Slot &s = p->slots[(size_t) p->acquired];
    if (used > s.cap) { g_err = "used > capacity"; return WTAMD_ERR_ARG; }
    if (min_capacity > s.cap) {
        // a fresh allocation on purpose: the caller must not keep stale pointers
        std::vector<int32_t> a((size_t) min_capacity), b((size_t) min_capacity);
        std::vector<float> c((size_t) min_capacity);
        memcpy(a.data(), s.start.data(), sizeof(int32_t) * (size_t) used);
        memcpy(b.data(), s.finish.data(), sizeof(int32_t) * (size_t) used);
        memcpy(c.data(), s.v32.data(), sizeof(float) * (size_t) used);
        s.start.swap(a); s.finish.swap(b); s.v32.swap(c);
        if (s.has64) {
            std::vector<double> d((size_t) min_capacity);
            memcpy(d.data(), s.v64.data(), sizeof(double) * (size_t) used);
            s.v64.swap(d);
        }
        s.cap = min_capacity;
    }
    if (want_f64 && !s.has64) { s.v64.assign((size_t) s.cap, 0.0); s.has64 = true; }
    fill_batch(s, out);
    return WTAMD_OK;
}

// The emulated pipe has no DMA: a direct range is remembered by POINTER (as the product does) and
// read at submit, so a caller that changes or frees the arrays too early shows up in the tests too.
int wtamd_pipe_put_direct(wtamd_pipe *p, int64_t at, int64_t count, const int32_t *start, const int32_t *finish,
                          const float *value) {
    if (!p || p->acquired < 0) { g_err = "no acquired slot"; return WTAMD_ERR_ARG; }
    if (count <= 0) return WTAMD_OK;
    Slot &s = p->slots[(size_t) p->acquired];
    if (!s.direct.empty() && s.direct.back().at + s.direct.back().count > at) { g_err = "direct ranges out of order"; return WTAMD_ERR_ARG; }
    s.direct.push_back({at, count, start, finish, value});
    return WTAMD_OK;
}

// WTEMU_DEVICES: how many GPUs the emulated runtime reports (the drop-in layer's WTAMD_DEVICES dealing is tested with it)
int wtamd_device_count(void) { const char *e = getenv("WTEMU_DEVICES"); return e && atoi(e) > 0 ? atoi(e) : 1; }
int wtamd_current_device(void) { return 0; }
void wtamd_warmup_async(void) { }
int wtamd_set_device(int) { return WTAMD_OK; }
void *wtamd_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void wtamd_pool_trim(void) {}
void wtamd_pool_stats(int64_t out[6]) { if (out) for (int i = 0; i < 6; i++) out[i] = 0; }       // (no pools to emulate)
void wtamd_host_free(void *q) { free(q); }

int wtamd_pipe_cancel(wtamd_pipe *p) {
    if (!p || p->acquired < 0) { g_err = "no acquired slot"; return WTAMD_ERR_ARG; }
    p->slots[(size_t) p->acquired].direct.clear();
    p->slots[(size_t) p->acquired].state = 0;
    p->acquired = -1;
    return WTAMD_OK;
}

int wtamd_pipe_submit(wtamd_pipe *p, int value_is_f64, int32_t range_lo, int32_t range_hi) {
    if (!p || p->acquired < 0) { g_err = "no acquired slot"; return WTAMD_ERR_ARG; }
    Slot &s = p->slots[(size_t) p->acquired];
    const int N = p->cfg.n_tracks;
    const int64_t n = s.seg_off[(size_t) N];
    if (value_is_f64 && !s.has64) { g_err = "float64 values were never staged"; return WTAMD_ERR_ARG; }
    if ((int64_t) range_hi - range_lo > p->cfg.max_runs && range_hi != INT32_MAX) { g_err = "hi - lo above max_runs"; return WTAMD_ERR_ARG; }
    const bool tile = p->cfg.desc.op == WTAMD_OP_MULTIPLEX;
    if (!s.direct.empty()) {                // merge the direct ranges with the staged ones
        if (value_is_f64) { g_err = "direct ranges are float32"; return WTAMD_ERR_ARG; }
        std::vector<int32_t> a((size_t) n), b((size_t) n);
        std::vector<float> c((size_t) n);
        int64_t pos = 0;
        auto staged = [&](int64_t lo, int64_t hi) {
            if (hi <= lo) return true;
            if (hi > s.cap) return false;
            memcpy(a.data() + lo, s.start.data() + lo, sizeof(int32_t) * (size_t) (hi - lo));
            memcpy(b.data() + lo, s.finish.data() + lo, sizeof(int32_t) * (size_t) (hi - lo));
            memcpy(c.data() + lo, s.v32.data() + lo, sizeof(float) * (size_t) (hi - lo));
            return true;
        };
        for (const auto &d : s.direct) {
            if (d.at + d.count > n || !staged(pos, d.at)) { g_err = "direct range / staging mismatch"; return WTAMD_ERR_ARG; }
            memcpy(a.data() + d.at, d.start, sizeof(int32_t) * (size_t) d.count);
            memcpy(b.data() + d.at, d.finish, sizeof(int32_t) * (size_t) d.count);
            memcpy(c.data() + d.at, d.value, sizeof(float) * (size_t) d.count);
            pos = d.at + d.count;
        }
        if (!staged(pos, n)) { g_err = "staged intervals beyond the staging capacity"; return WTAMD_ERR_ARG; }
        s.start.swap(a); s.finish.swap(b); s.v32.swap(c);
        if (n > s.cap) s.cap = n;
        if ((int64_t) s.start.size() < s.cap) { s.start.resize((size_t) s.cap); s.finish.resize((size_t) s.cap); s.v32.resize((size_t) s.cap); }
        s.direct.clear();
    } else if (n > s.cap) { g_err = "staged intervals beyond the staging capacity"; return WTAMD_ERR_ARG; }
    // operator chains: the batch is mapped (and compacted) before it is multiplexed, values become f64
    std::vector<int64_t> mseg;
    std::vector<int32_t> ms, mf;
    std::vector<double> mv;
    const bool mapped = !p->chains.empty();
    if (mapped) {
        mseg.assign((size_t) N + 1, 0);
        for (int t = 0; t < N; t++) {
            const wtamd_map_chain &c = p->chains[(size_t) t];
            for (int64_t g = s.seg_off[(size_t) t]; g < s.seg_off[(size_t) t + 1]; g++) {
                double v = value_is_f64 ? s.v64[(size_t) g] : (double) s.v32[(size_t) g];
                bool keep = true;
                for (int k = 0; k < c.n_ops && keep; k++) {
                    const bool based = c.op[k] == WTAMD_MAP_LOG || c.op[k] == WTAMD_MAP_EXPB;
                    v = wm_apply(c.op[k], c.param[k], based ? log(c.param[k]) : 1.0, v, keep);
                }
                if (keep) { ms.push_back(s.start[(size_t) g]); mf.push_back(s.finish[(size_t) g]); mv.push_back(v); }
            }
            mseg[(size_t) t + 1] = (int64_t) ms.size();
        }
        ms.push_back(0); mf.push_back(0); mv.push_back(0);      // never empty
    }
    int64_t cap = 2 * n + 8;
    s.os.assign((size_t) cap, 0); s.of.assign((size_t) cap, 0); s.ov.assign((size_t) cap, 0.0);
    if (tile) { s.tile.assign((size_t) cap * N, 0.0); s.ip.assign((size_t) cap * N, 0); }
    int64_t cro[2] = {0, 0};
    long long info[16] = {0};
    const long long r = wtemu_reduce(1, N, mapped ? mseg.data() : s.seg_off.data(), mapped ? ms.data() : s.start.data(),
                                     mapped ? mf.data() : s.finish.data(),
                                     mapped ? (const void *) mv.data() : value_is_f64 ? (const void *) s.v64.data() : (const void *) s.v32.data(),
                                     mapped ? 1 : value_is_f64,
                                     p->defaults.data(), p->cfg.desc.op, p->cfg.desc.flags, p->cfg.desc.n_set0, cap,
                                     s.os.data(), s.of.data(), s.ov.data(), cro, tile ? s.tile.data() : nullptr,
                                     tile ? s.ip.data() : nullptr, info, &range_lo, &range_hi);
    if (r < 0) { g_err = "emulator error " + std::to_string(r); return WTAMD_ERR_INTERNAL; }
    if (r > p->cfg.max_runs) { g_err = "more runs than max_runs"; return WTAMD_ERR_CAPACITY; }
    long long rr = r;
    s.integrated = p->integrate;
    if (s.integrated) emu_integrals(p, s, r, s.integ);
    if (p->compress && !tile && r > 0 && !s.integrated) {
        // CompressionWiggleIterator's leader rule (reference unaryOps.c:235-253), sequentially -- with an
        // OPEN START: the batch continues a previous one whose last group may reach into it, so the
        // runs before the first run that leads a group whatever came before it (not contiguous,
        // NaN-ness changes, or more than 2e-6 away from its predecessor) are passed through one by one
        auto isnan_ = [](double x) { return x != x; };
        long long f = r;
        for (long long q = 1; q < r; q++) {
            const double v = s.ov[(size_t) q], pv = s.ov[(size_t) q - 1];
            const bool contiguous = s.os[(size_t) q] == s.of[(size_t) q - 1];
            bool sure = !contiguous || isnan_(v) != isnan_(pv);
            if (!sure && !isnan_(v)) { const double dd = v > pv ? v - pv : pv - v; sure = dd >= 2.000001e-6; }
            if (sure) { f = q; break; }
        }
        long long o = f > 0 ? f - 1 : 0;       // runs [0, f) unchanged; the group of run f starts at output f
        if (f < r) {
            o = f;
            s.os[(size_t) o] = s.os[(size_t) f]; s.ov[(size_t) o] = s.ov[(size_t) f];
            double leader = s.ov[(size_t) f];
            for (long long q = f + 1; q < r; q++) {
                const double v = s.ov[(size_t) q];
                const bool same = s.os[(size_t) q] == s.of[(size_t) q - 1] &&
                                  ((isnan_(v) && isnan_(leader)) || (v - leader < 0.000001 && leader - v < 0.000001));
                if (!same) {
                    s.of[(size_t) o] = s.of[(size_t) q - 1];
                    o++;
                    s.os[(size_t) o] = s.os[(size_t) q]; s.ov[(size_t) o] = v;
                    leader = v;
                }
            }
            s.of[(size_t) o] = s.of[(size_t) r - 1];
        }
        rr = o + 1;
    }
    s.n_runs = rr; s.covered = info[4]; s.n_int = n;
    s.state = 2;
    p->acquired = -1;
    p->head = (p->head + 1) % (int) p->slots.size();
    p->in_flight++;
    p->st.batches++; p->st.intervals += n; p->st.runs += rr; p->st.covered_bp += info[4];
    if (info[8]) p->st.delta_batches++;
    return WTAMD_OK;
}

// ---- fused integrators (wtamd_pipe_set_integrate): the reference's own sequential updates (statistics.c:62-120,414-465)
static void emu_integrals(const wtamd_pipe *p, const Slot &s, int64_t n, double *g) {
    for (int k = 0; k < 6; k++) g[k] = 0;
    if (p->cfg.desc.op == WTAMD_OP_MULTIPLEX) {
        double cnt = 0, sx = 0, sy = 0, txx = 0, txy = 0, tyy = 0;
        for (int64_t r = 0; r < n; r++) {
            const double X = s.ip[(size_t) (2 * r)] ? s.tile[(size_t) (2 * r)] : p->defaults[0];
            const double Y = s.ip[(size_t) (2 * r + 1)] ? s.tile[(size_t) (2 * r + 1)] : p->defaults[1];
            const double L = (double) (s.of[(size_t) r] - s.os[(size_t) r]);
            if (cnt > 0) {
                const double omx = sx / cnt, nmx = sx / (cnt + L), omy = sy / cnt, nmy = sy / (cnt + L), ratio = cnt / (cnt + L);
                txy += (nmx * omy + ratio * X * Y - nmx * Y - nmy * X) * L;
                txx += (nmx * (omx - 2 * X) + ratio * X * X) * L;
                tyy += (nmy * (omy - 2 * Y) + ratio * Y * Y) * L;
            }
            cnt += L; sx += X * L; sy += Y * L;
        }
        g[0] = cnt; g[1] = sx; g[2] = sy; g[3] = txx; g[4] = txy; g[5] = tyy;
    } else {
        for (int64_t r = 0; r < n; r++) {
            const double v = s.ov[(size_t) r];
            if (v == v) { const double L = (double) (s.of[(size_t) r] - s.os[(size_t) r]); g[0] += L * v; g[1] += L; }
        }
    }
}

int wtamd_pipe_set_integrate(wtamd_pipe *p, int on) {
    if (!p) return WTAMD_ERR_ARG;
    if (on && p->cfg.desc.op == WTAMD_OP_MULTIPLEX && p->cfg.n_tracks != 2) { g_err = "fused Pearson needs two tracks"; return WTAMD_ERR_ARG; }
    p->integrate = on != 0;
    return WTAMD_OK;
}

int wtamd_pipe_integrate_held(wtamd_pipe *p, double *integ) {
    if (!p || !integ || !p->held) { g_err = "no collected batch"; return WTAMD_ERR_ARG; }
    const Slot &s = p->slots[(size_t) p->tail];
    if (s.integrated) { for (int k = 0; k < 6; k++) integ[k] = s.integ[k]; return WTAMD_OK; }
    emu_integrals(p, s, s.n_runs, integ);
    return WTAMD_OK;
}

// ---- BigWig sections "on device": the kernels' own per-lane / per-item code (csrc/wt_inflate.h, csrc/wt_bwdev_core.h)
// run one section after the other -- what csrc/wt_bwdev.hip runs one lane / one wavefront per section.

// One zlib (or raw deflate) stream through the lane state machine; returns bytes produced or -(error).
}  // extern "C"
template <int RING>
static long long emu_inflate_ring(const uint8_t *src, long long n, uint8_t *dst, long long cap, int raw_deflate, long long *steps, uint32_t *end_byte = nullptr) {
    std::vector<uint8_t> perm(WT_INF_PERM + 8, 0);
    std::vector<uint32_t> ring(RING, 0);
    WtInfMem m{perm.data(), ring.data(), 1};
    // the decoder reads whole aligned 16-byte chunks around the stream and writes whole words: private padded copies
    std::vector<uint8_t> in((size_t) n + 96, 0), out(((size_t) cap + 3) / 4 * 4 + 8, 0);
    const int mis = (int) (n % 16);             // any alignment must work
    if (n > 0) memcpy(in.data() + 32 + mis, src, (size_t) n);
    WtInflateT<RING> z;
    wt_inf_begin(z, in.data() + 32 + mis, (uint32_t) n, out.data(), (uint32_t) cap, raw_deflate != 0);
    long long rounds = 0;
    while (wt_inf_land(z, m)) {
        for (int r = 0; r < WT_INF_ROUND; r++) wt_inf_step(z, m);
        rounds++;
    }
    if (steps) *steps = rounds * WT_INF_ROUND;
    const long long r = (long long) wt_inf_finish(z);
    if (end_byte) *end_byte = wt_inf_end_byte(z);
    if (r > 0) memcpy(dst, out.data(), (size_t) r);
    return r;
}

extern "C" {
// One zlib (or raw deflate) stream through the lane state machine; returns bytes produced or -(error).
long long wtemu_inflate(const uint8_t *src, long long n, uint8_t *dst, long long cap, int raw_deflate) {
    return emu_inflate_ring<WT_INF_RING>(src, n, dst, cap, raw_deflate, nullptr);
}

// The same with a ring of `ring` dwords (8 or 64: the two instantiations of the kernel); *steps = steps the lane took.
long long wtemu_inflate_ring(const uint8_t *src, long long n, uint8_t *dst, long long cap, int raw_deflate, int ring, long long *steps) {
    if (ring == 64) return emu_inflate_ring<64>(src, n, dst, cap, raw_deflate, steps);
    return emu_inflate_ring<8>(src, n, dst, cap, raw_deflate, steps);
}

// The sections of a batch -> run lists (o_* sized `capacity`), seg_off[n_tracks + 1].  Returns the error bits.
unsigned wtemu_bw_decode(const uint8_t *bytes, const wtamd_bw_section *secs, long long n_secs, const wtamd_bw_track *tracks,
                         int n_tracks, long long capacity, int32_t *o_start, int32_t *o_finish, float *o_value, int64_t *seg_off) {
    unsigned err = 0;
    std::vector<std::vector<uint8_t>> plain((size_t) n_secs);
    std::vector<long long> plen((size_t) n_secs, 0), cnt((size_t) n_secs, 0);
    for (long long i = 0; i < n_secs; i++) {
        const wtamd_bw_track &tk = tracks[secs[i].track];
        uint32_t stride = ((tk.plain_bytes + 16) + 15u) & ~15u;
        if (stride < 64) stride = 64;
        plain[(size_t) i].assign(stride + 8, 0);
        if (tk.compressed) {
            uint32_t at = 0;        // end of the final block: where the trailer sits (a leaf may carry padding behind its stream)
            plen[(size_t) i] = emu_inflate_ring<WT_INF_RING>(bytes + secs[i].comp_off, secs[i].comp_size, plain[(size_t) i].data(), stride, 0, nullptr, &at);
            if (plen[(size_t) i] >= 0 && secs[i].comp_size >= 6) {      // Adler-32 against the trailer (the count kernel's check)
                uint32_t a = 1, b = 0;
                for (long long q = 0; q < plen[(size_t) i]; q++) { a = (a + plain[(size_t) i][(size_t) q]) % 65521u; b = (b + a) % 65521u; }
                if (at + 4u > secs[i].comp_size) plen[(size_t) i] = -WT_INF_ERR_INPUT;
                else {
                    const uint8_t *t = bytes + secs[i].comp_off + at;
                    const uint32_t want = ((uint32_t) t[0] << 24) | ((uint32_t) t[1] << 16) | ((uint32_t) t[2] << 8) | (uint32_t) t[3];
                    if (((b << 16) | a) != want) plen[(size_t) i] = -WT_INF_ERR_INPUT;
                }
            }
        }
        else if (secs[i].comp_size > stride) plen[(size_t) i] = -WT_INF_ERR_SPACE;
        else { memcpy(plain[(size_t) i].data(), bytes + secs[i].comp_off, secs[i].comp_size); plen[(size_t) i] = secs[i].comp_size; }
    }
    auto walk = [&](long long i, bool write, long long at) -> long long {
        const wtamd_bw_section &sc = secs[i];
        const wtamd_bw_track &tk = tracks[sc.track];
        if (plen[(size_t) i] < 0) { err |= WT_BW_ERR_INFLATE; return 0; }
        const uint8_t *p = plain[(size_t) i].data();
        WtBwHdr h;
        if (!wt_bw_parse_hdr(p, (uint32_t) plen[(size_t) i], h)) { err |= WT_BW_ERR_SECTION; return 0; }
        if (h.chrom_id != tk.chrom_id) return 0;
        long long n = 0;
        for (uint32_t k = 0; k < h.count; k++) {
            uint32_t s0, e0, vb;
            wt_bw_item(p, h, k, s0, e0, vb);
            if (s0 < sc.leaf_start || e0 > sc.leaf_end || e0 <= s0) err |= WT_BW_ERR_EXTENT;
            if (e0 >= (uint32_t) WTAMD_MAX_COORD) err |= WT_BW_ERR_COORD;
            if (k > 0) { uint32_t ps, pe, pv; wt_bw_item(p, h, k - 1, ps, pe, pv); if (s0 < pe) err |= WT_BW_ERR_EXTENT; }
            float v;
            memcpy(&v, &vb, 4);
            n += wt_bw_pieces(s0, e0, tk, [&](int32_t a, int32_t b) {
                if (write) { if (at < capacity) { o_start[at] = a; o_finish[at] = b; o_value[at] = v; } at++; }
            });
        }
        return n;
    };
    long long total = 0;
    for (long long i = 0; i < n_secs; i++) { cnt[(size_t) i] = walk(i, false, 0); total += cnt[(size_t) i]; }
    if (total > capacity) err |= WT_BW_ERR_CAPACITY;
    if (err) { for (int t = 0; t <= n_tracks; t++) seg_off[t] = 0; return err; }
    long long at = 0;
    std::vector<long long> off((size_t) n_secs + 1, 0);
    for (long long i = 0; i < n_secs; i++) { off[(size_t) i] = at; walk(i, true, at); at += cnt[(size_t) i]; }
    off[(size_t) n_secs] = at;
    for (int t = 0; t < n_tracks; t++) seg_off[t] = tracks[t].first_section < n_secs ? off[(size_t) tracks[t].first_section] : at;
    seg_off[n_tracks] = at;
    return 0;
}

int64_t wtamd_pipe_bw_fill_sections(const wtamd_pipe *p) { return p ? 4096 : 0; }

unsigned wtamd_pipe_bw_error(const wtamd_pipe *p) { return p ? p->last_bw_err : 0u; }

int wtamd_pipe_bw_reserve(wtamd_pipe *p, int64_t n_bytes, int64_t n_sections, uint8_t **bytes, wtamd_bw_section **sections) {
    if (!p || p->acquired < 0 || n_bytes < 0 || n_sections < 0 || !bytes || !sections) { g_err = "wtamd_pipe_bw_reserve: bad arguments"; return WTAMD_ERR_ARG; }
    Slot &s = p->slots[(size_t) p->acquired];
    // fresh allocations on purpose (stale pointers must show up in the tests)
    std::vector<uint8_t>((size_t) n_bytes + 16, 0xA5).swap(s.bw_bytes);
    std::vector<wtamd_bw_section>((size_t) n_sections + 1).swap(s.bw_secs);
    s.bw_res_bytes = n_bytes; s.bw_res_secs = n_sections;
    *bytes = s.bw_bytes.data();
    *sections = s.bw_secs.data();
    return WTAMD_OK;
}

int wtamd_pipe_submit_bw(wtamd_pipe *p, int64_t n_bytes, int64_t n_sections, const wtamd_bw_track *tracks,
                         int32_t range_lo, int32_t range_hi) {
    if (!p || p->acquired < 0 || !tracks) { g_err = "wtamd_pipe_submit_bw: no acquired slot"; return WTAMD_ERR_ARG; }
    Slot &s = p->slots[(size_t) p->acquired];
    const int N = p->cfg.n_tracks;
    if (s.bw_res_bytes < 0 || n_bytes > s.bw_res_bytes || n_sections > s.bw_res_secs) { g_err = "wtamd_pipe_submit_bw: more than reserved"; return WTAMD_ERR_ARG; }
    if (p->cfg.desc.op == WTAMD_OP_MULTIPLEX) { g_err = "wtamd_pipe_submit_bw: not for the tile"; return WTAMD_ERR_ARG; }
    // the host's bound, as the product computes it (a decode producing more fails the batch)
    int64_t bound = 0, next = 0;
    for (int i = 0; i < N; i++) {
        if (tracks[i].first_section != next) { g_err = "wtamd_pipe_submit_bw: sections must be listed track by track"; return WTAMD_ERR_ARG; }
        for (int64_t q = next; q < next + tracks[i].n_sections; q++) {
            const wtamd_bw_section &c = s.bw_secs[(size_t) q];
            if (c.track != i || c.comp_off < 0 || c.comp_off + (int64_t) c.comp_size > n_bytes) { g_err = "wtamd_pipe_submit_bw: bad section entry"; return WTAMD_ERR_ARG; }
            if (q > next && c.leaf_start < s.bw_secs[(size_t) q - 1].leaf_end) { g_err = "wtamd_pipe_submit_bw: sections not sorted / disjoint"; return WTAMD_ERR_ARG; }
            bound += wt_bw_section_bound(tracks[i].plain_bytes, c.leaf_start, c.leaf_end, tracks[i].box);
        }
        next += tracks[i].n_sections;
    }
    if (next != n_sections) { g_err = "wtamd_pipe_submit_bw: section count mismatch"; return WTAMD_ERR_ARG; }
    const int64_t cap = bound > 0 ? bound : 1;
    s.start.assign((size_t) cap, 0); s.finish.assign((size_t) cap, 0); s.v32.assign((size_t) cap, 0.f);
    if (s.has64) s.v64.assign((size_t) cap, 0.0);
    s.cap = cap;
    const unsigned e = wtemu_bw_decode(s.bw_bytes.data(), s.bw_secs.data(), n_sections, tracks, N, cap, s.start.data(), s.finish.data(),
                                       s.v32.data(), s.seg_off.data());
    s.bw_res_bytes = s.bw_res_secs = -1;
    p->st.bw_sections += n_sections;
    const int rc = wtamd_pipe_submit(p, 0, range_lo, range_hi);     // (a failed decode left seg_off[] all zero: an empty batch)
    if (rc == WTAMD_OK) s.bw_err = e;
    return rc;
}

int wtamd_pipe_collect(wtamd_pipe *p, wtamd_pipe_result *out) {
    if (!p || !out) { g_err = "NULL"; return WTAMD_ERR_ARG; }
    if (p->in_flight <= 0) { g_err = "nothing in flight"; return WTAMD_ERR_ARG; }
    if (p->held) { g_err = "the previous result was not released"; return WTAMD_ERR_ARG; }
    Slot &s = p->slots[(size_t) p->tail];
    if (s.state != 2) { g_err = "slot order corrupted"; return WTAMD_ERR_INTERNAL; }
    s.state = 3;
    p->in_flight--;
    p->held = 1;
    p->last_bw_err = s.bw_err;
    if (s.bw_err) {
        g_err = "BigWig sections could not be decoded (bits " + std::to_string(s.bw_err) + ")";
        s.bw_err = 0;
        return WTAMD_ERR_INTERNAL;
    }
    const bool tile = p->cfg.desc.op == WTAMD_OP_MULTIPLEX;
    p->st.d2h_bytes += s.integrated ? 176 : s.n_runs * (16 + (tile ? 9 * (int64_t) p->cfg.n_tracks : 0));
    out->n_runs = s.n_runs;
    out->integ_valid = s.integrated ? 1 : 0;
    out->reserved = 0;
    for (int k = 0; k < 6; k++) out->integ[k] = s.integrated ? s.integ[k] : 0.0;
    out->start = s.integrated ? nullptr : s.os.data(); out->finish = s.integrated ? nullptr : s.of.data();
    out->value = s.integrated ? nullptr : s.ov.data();
    out->tile = (tile && !s.integrated) ? s.tile.data() : nullptr;
    out->inplay = (tile && !s.integrated) ? s.ip.data() : nullptr;
    out->covered_bp = s.covered;
    out->n_intervals = s.n_int;
    return WTAMD_OK;
}

int wtamd_pipe_release(wtamd_pipe *p) {
    if (!p || !p->held) { g_err = "nothing to release"; return WTAMD_ERR_ARG; }
    Slot &s = p->slots[(size_t) p->tail];
    s.state = 0;
    // poison: a consumer that keeps reading a released result must show up in the tests
    std::fill(s.os.begin(), s.os.end(), -1);
    std::fill(s.of.begin(), s.of.end(), -1);
    p->held = 0;
    p->tail = (p->tail + 1) % (int) p->slots.size();
    return WTAMD_OK;
}

int wtamd_pipe_in_flight(const wtamd_pipe *p) { return p ? p->in_flight : 0; }

int wtamd_pipe_get_stats(const wtamd_pipe *p, wtamd_pipe_stats *out) {
    if (!p || !out) return WTAMD_ERR_ARG;
    *out = p->st;
    return WTAMD_OK;
}

}  // extern "C"
