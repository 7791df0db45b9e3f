// wt_bigwig.cpp -- BigWig section decoder: file -> run lists (start, finish, float value),
// the bulk side door in front of the engine.  Host C++ + zlib.
//
// The reference reads BigWig through libBigWig (bwOverlappingIntervalsIterator, reference
// src/bigWiggleReader.c:52-83), which is not available here; this is an independent decoder of
// the published BigWig layout (Kent et al. 2010: 64-byte header, chromosome B+ tree, R-tree
// index over zlib-compressed sections of type 1 bedGraph / 2 variableStep / 3 fixedStep,
// float32 payload).  What it reproduces from the reference reader, because it changes what the
// Multiplexer sees:
//   * 0-based half-open -> 1-based start, exclusive finish        bigWiggleReader.c:39-40
//   * chromosomes in strcmp order (done by the caller)            bigWiggleReader.c:91-101
//   * intervals boxed into 10 000-bp stretches [1+10000k, 1+10000(k+1))  :42-44, :73-83
//     (optional; a run crossing a stretch edge is cut there, as the reference does)
//   * the stretch loop `for (start = 1; start < length; ...)` (:76) never visits a stretch that
//     would start at position `length`, so the last base of a chromosome whose length is
//     1 (mod 10000) is dropped -- reproduced when boxing is on.
// Pinned by the reference's own fixtures test/fixedStep.bw == fixedStep.wig and
// variableStep.bw == variableStep.wig (reference test/test.py:28,52).
#include <dirent.h>
#include <fcntl.h>
#include <sys/resource.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/wiggletools_amd.h"
#include "wt_bigwig_int.h"

namespace {

typedef WtBwLeaf BwBlock;

struct BwChrom {
    std::string name;
    uint32_t id, length;
};

}  // namespace

struct wtamd_bw {
    FILE *fp = nullptr;
    uint32_t uncompress_buf = 0;
    std::vector<BwChrom> chroms;
    std::vector<BwBlock> blocks;
    std::string error;
    std::vector<unsigned char> raw, plain;      // scratch of decode_block
    // wtamd_bw_read_part: the last decoded part, kept when the caller's arrays were too small
    void *part = nullptr;                       // std::vector<Triple> *
    int64_t part_from = -1, part_to = -1;
    int part_chrom = -1, part_blocks = 0, part_lo = 0, part_hi = 0;
    bool part_last = false;
    std::vector<WtBwChromInfo> infos;           // wt_bw_chrom_info, one per chroms[] entry (bw_build_infos, at open)
    std::unordered_map<std::string, size_t> by_name;
    // wtamd_bw_open only: a window of the file, so that the walk over the header, the chromosome tree and the R-tree
    // index costs a handful of pread() calls instead of two stdio seeks + reads per node (~500 system calls per file
    // of chromosome 1: 1.6 ms per file and NOT parallel on the hosts measured -- 100 files took 120-190 ms however
    // many threads opened them; round 3)
    std::vector<unsigned char> win;
    uint64_t win_off = 0;
};

namespace {

bool rd(FILE *fp, uint64_t off, void *dst, size_t n) {
    if (fseeko(fp, (off_t) off, SEEK_SET) != 0) return false;
    return fread(dst, 1, n, fp) == n;
}

// the same through the open-time window (refilled with 1 MB from `off` on when the bytes are not in it)
bool rdw(wtamd_bw *bw, uint64_t off, void *dst, size_t n) {
    if (!(off >= bw->win_off && off + n <= bw->win_off + bw->win.size())) {
        const size_t want = n > ((size_t) 1 << 20) ? n : ((size_t) 1 << 20);
        bw->win.resize(want);
        const ssize_t got = pread(fileno(bw->fp), bw->win.data(), want, (off_t) off);
        if (got < 0) { bw->win.clear(); return false; }
        bw->win.resize((size_t) got);
        bw->win_off = off;
        if ((size_t) got < n) return false;
    }
    memcpy(dst, bw->win.data() + (off - bw->win_off), n);
    return true;
}

template <class T>
T get(const unsigned char *p) {
    T v;
    memcpy(&v, p, sizeof(T));
    return v;
}

bool walk_chrom_tree(wtamd_bw *bw, uint64_t node_off, uint32_t key_size, uint32_t val_size) {
    unsigned char hdr[4];
    if (!rdw(bw, node_off, hdr, 4)) return false;
    const bool leaf = hdr[0] != 0;
    const uint16_t count = get<uint16_t>(hdr + 2);
    const size_t item = key_size + (leaf ? val_size : 8);
    std::vector<unsigned char> buf(item * count);
    if (count && !rdw(bw, node_off + 4, buf.data(), buf.size())) return false;
    for (uint16_t k = 0; k < count; k++) {
        const unsigned char *p = buf.data() + item * k;
        if (leaf) {
            BwChrom c;
            c.name.assign((const char *) p, strnlen((const char *) p, key_size));
            c.id = get<uint32_t>(p + key_size);
            c.length = get<uint32_t>(p + key_size + 4);
            bw->chroms.push_back(c);
        } else if (!walk_chrom_tree(bw, get<uint64_t>(p + key_size), key_size, val_size)) {
            return false;
        }
    }
    return true;
}

bool walk_rtree(wtamd_bw *bw, uint64_t node_off) {
    unsigned char hdr[4];
    if (!rdw(bw, node_off, hdr, 4)) return false;
    const bool leaf = hdr[0] != 0;
    const uint16_t count = get<uint16_t>(hdr + 2);
    const size_t item = leaf ? 32 : 24;
    std::vector<unsigned char> buf(item * count);
    if (count && !rdw(bw, node_off + 4, buf.data(), buf.size())) return false;
    for (uint16_t k = 0; k < count; k++) {
        const unsigned char *p = buf.data() + item * k;
        if (leaf) {
            BwBlock b = { get<uint32_t>(p), get<uint32_t>(p + 4), get<uint32_t>(p + 8), get<uint32_t>(p + 12),
                          get<uint64_t>(p + 16), get<uint64_t>(p + 24) };
            bw->blocks.push_back(b);
        } else if (!walk_rtree(bw, get<uint64_t>(p + 16))) {
            return false;
        }
    }
    return true;
}

struct Triple {
    uint32_t start, end;   // 0-based half-open
    float value;
};

bool decode_block(wtamd_bw *bw, const BwBlock &b, uint32_t chrom_id, std::vector<Triple> &out) {
    std::vector<unsigned char> &raw = bw->raw, &plain = bw->plain;
    raw.resize(b.size);
    if (!rd(bw->fp, b.offset, raw.data(), raw.size())) { bw->error = "short read of a data block"; return false; }
    const unsigned char *d = raw.data();
    size_t dn = raw.size();
    if (bw->uncompress_buf) {
        if (plain.size() < bw->uncompress_buf) plain.resize(bw->uncompress_buf);
        uLongf n = plain.size();
        if (uncompress(plain.data(), &n, raw.data(), raw.size()) != Z_OK) { bw->error = "zlib: bad data block"; return false; }
        d = plain.data();
        dn = n;
    }
    if (dn < 24) { bw->error = "data block too short"; return false; }
    const uint32_t cid = get<uint32_t>(d), cstart = get<uint32_t>(d + 4);
    const uint32_t step = get<uint32_t>(d + 12), span = get<uint32_t>(d + 16);
    const uint8_t type = d[20];
    const uint16_t n_items = get<uint16_t>(d + 22);
    if (cid != chrom_id) return true;       // block of another chromosome sharing the index leaf
    const unsigned char *p = d + 24;
    const size_t item = type == 1 ? 12 : type == 2 ? 8 : 4;
    if (24 + item * n_items > dn) { bw->error = "data block item count exceeds its size"; return false; }
    // one loop per section type (the type test used to sit inside the per-item loop)
    const size_t at = out.size();
    out.resize(at + n_items);
    Triple *o = out.data() + at;
    if (type == 1) {
        for (uint16_t k = 0; k < n_items; k++, p += 12) { o[k].start = get<uint32_t>(p); o[k].end = get<uint32_t>(p + 4); o[k].value = get<float>(p + 8); }
    } else if (type == 2) {
        for (uint16_t k = 0; k < n_items; k++, p += 8) { o[k].start = get<uint32_t>(p); o[k].end = o[k].start + span; o[k].value = get<float>(p + 4); }
    } else if (type == 3) {
        for (uint16_t k = 0; k < n_items; k++, p += 4) { o[k].start = cstart + k * step; o[k].end = o[k].start + span; o[k].value = get<float>(p); }
    } else {
        out.resize(at);
        bw->error = "unknown BigWig section type";
        return false;
    }
    return true;
}

}  // namespace

const WtBwLeaf *wt_bw_leaves(const wtamd_bw *bw, int64_t *n) {
    if (n) *n = (int64_t) bw->blocks.size();
    return bw->blocks.data();
}

int wt_bw_fd(const wtamd_bw *bw) { return fileno(bw->fp); }

uint32_t wt_bw_uncompress_buf(const wtamd_bw *bw) { return bw->uncompress_buf; }

// Per chromosome: its index leaves (a contiguous range of the R-tree's leaves in file order) and whether the device
// route can take them.  ONE pass over the leaves (round 3 scanned all of them once per chromosome: O(chromosomes x
// leaves), seconds per file for an assembly with thousands of contigs -- the advisor's finding); run by wtamd_bw_open,
// i.e. on the threads that open the files side by side.
static void bw_build_infos(wtamd_bw *bw) {
    const size_t nc = bw->chroms.size();
    bw->infos.resize(nc);
    bw->by_name.clear();
    std::vector<size_t> by_id(nc);
    std::vector<int64_t> last(nc, -1);
    for (size_t c = 0; c < nc; c++) {
        WtBwChromInfo &x = bw->infos[c];
        x.id = bw->chroms[c].id; x.length = bw->chroms[c].length;
        x.first = 0; x.count = 0; x.device_ok = true; x.max_size = 0;
        by_id[c] = c;
        bw->by_name.emplace(bw->chroms[c].name, c);
    }
    std::sort(by_id.begin(), by_id.end(), [&](size_t a, size_t b) { return bw->chroms[a].id < bw->chroms[b].id; });
    for (int64_t i = 0; i < (int64_t) bw->blocks.size(); i++) {
        const BwBlock &b = bw->blocks[(size_t) i];
        // the chromosomes this leaf touches: ids in [start_chrom, end_chrom] (one, in every file the usual tools write)
        size_t q = (size_t) (std::lower_bound(by_id.begin(), by_id.end(), b.start_chrom,
                                              [&](size_t a, uint32_t id) { return bw->chroms[a].id < id; }) - by_id.begin());
        for (; q < nc && bw->chroms[by_id[q]].id <= b.end_chrom; q++) {
            const size_t c = by_id[q];
            WtBwChromInfo &x = bw->infos[c];
            if (b.start_chrom != x.id || b.end_chrom != x.id) x.device_ok = false;         // a leaf spanning chromosomes
            if (x.count == 0) x.first = i;
            else if (i != last[c] + 1 || b.start_base < bw->blocks[(size_t) last[c]].end_base) x.device_ok = false;   // not contiguous / sorted / disjoint
            if (b.end_base < b.start_base || b.size > 0x7FFFFFFFull) x.device_ok = false;
            if (b.size > x.max_size) x.max_size = (uint32_t) std::min<uint64_t>(b.size, 0xFFFFFFFFull);
            last[c] = i;
            x.count++;
        }
    }
    for (size_t c = 0; c < nc; c++)
        if (!bw->infos[c].device_ok) bw->infos[c].count = last[c] >= 0 ? last[c] - bw->infos[c].first + 1 : 0;
}

bool wt_bw_chrom_info(wtamd_bw *bw, const char *chrom, WtBwChromInfo *out) {
    if (bw->infos.size() != bw->chroms.size()) bw_build_infos(bw);
    const auto it = bw->by_name.find(chrom);
    if (it == bw->by_name.end()) return false;
    *out = bw->infos[it->second];
    return true;
}

extern "C" {

static double bw_now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// The kernel grows a process's file-descriptor table by doubling (64 -> 128 -> 256 ...), and the table of a process with
// more than one thread is replaced behind synchronize_rcu(): on the 256-CPU hosts measured that is 140-180 ms, paid by the
// open() that crosses the boundary and by every other thread whose open needs the new table meanwhile -- the second job of
// 100 files in a process (round 4: 0.18 s instead of 0.04 s to open them) or, round 5's trace of a FRESH process, 16 of
// the 100 fopen() calls of the first job (138.8 ms each, the other 84: 0.02 ms).  A single-threaded process grows its
// table without any grace period, so:
//   * when this library is loaded into a process that has one thread (a C program linked against it: the reference's CLI),
//     its constructor grows the table to 4096 entries on the spot -- microseconds;
//   * otherwise (an interpreter that already runs worker threads) a helper thread starts growing it at load time -- the
//     grace period runs while the host program goes about its own start-up -- and the first open makes sure of it.
// F_DUPFD_CLOEXEC takes the lowest FREE descriptor >= target: it never closes a descriptor another thread of the host
// application opened in the meantime (round 4 used F_GETFD + dup2, which can: the advisor's finding).  WTAMD_NO_FD_GROW=1: off.
static std::atomic<int> g_fd_grown{0};       // 0 not yet, 1 under way or done

static void bw_grow_fd_table_now() {
    struct rlimit rl;
    int target = 4095;
    if (getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur != RLIM_INFINITY && (long long) rl.rlim_cur - 1 < target) target = (int) rl.rlim_cur - 1;
    const int fd = open("/dev/null", O_RDONLY | O_CLOEXEC);
    if (fd < 0) return;
    if (target > fd) {
        const int hi = fcntl(fd, F_DUPFD_CLOEXEC, target);
        if (hi >= 0) close(hi);
    }
    close(fd);
}

static int bw_thread_count() {
    int n = 0;
    if (DIR *d = opendir("/proc/self/task")) {
        while (struct dirent *e = readdir(d))
            if (e->d_name[0] != '.') n++;
        closedir(d);
    }
    return n;
}

// The helper thread (a multi-threaded host: the kernel's grace period, 140-180 ms, is kept off the caller's clock) is JOINED when the
// library is unloaded or the process exits -- detached, it could still be inside this library's code after a dlclose (the
// advisor's finding, round 5).  Load-time side effects of the library: INTEGRATION.md 5.
static std::mutex g_fd_mu;
static std::thread *g_fd_thread = nullptr;

static void bw_grow_fd_table(bool at_load) {
    if (getenv("WTAMD_NO_FD_GROW")) return;
    int expect = 0;
    if (!g_fd_grown.compare_exchange_strong(expect, 1)) return;
    if (at_load && bw_thread_count() == 1) { bw_grow_fd_table_now(); return; }     // no other thread: no grace period
    std::lock_guard<std::mutex> lk(g_fd_mu);
    g_fd_thread = new std::thread(bw_grow_fd_table_now);
}

__attribute__((constructor)) static void bw_library_loaded() { bw_grow_fd_table(true); }
__attribute__((destructor)) static void bw_library_unloaded() {
    std::thread *t = nullptr;
    { std::lock_guard<std::mutex> lk(g_fd_mu); t = g_fd_thread; g_fd_thread = nullptr; }
    if (t) { if (t->joinable()) t->join(); delete t; }
}

int wtamd_bw_open(const char *path, wtamd_bw **out) {
    if (!path || !out) return WTAMD_ERR_ARG;
    bw_grow_fd_table(false);        // (already done by the library's constructor unless that was switched off)
    static const bool trace = getenv("WTAMD_TRACE_OPEN") != nullptr;
    const double t0 = trace ? bw_now_ms() : 0;
    wtamd_bw *bw = new wtamd_bw();
    bw->fp = fopen(path, "rb");
    const double t1 = trace ? bw_now_ms() : 0;
    unsigned char h[64];
    if (!bw->fp || !rdw(bw, 0, h, 64) || get<uint32_t>(h) != 0x888FFC26u) {
        // message of the reference (bigWiggleReader.c:116-118) for a non-BigWig file
        fprintf(stderr, "File %s is not in BigWig format\n", path);
        if (bw->fp) fclose(bw->fp);
    delete (std::vector<Triple> *) bw->part;
        delete bw;
        return WTAMD_ERR_ARG;
    }
    const uint64_t chrom_tree = get<uint64_t>(h + 8), full_index = get<uint64_t>(h + 24);
    bw->uncompress_buf = get<uint32_t>(h + 52);
    unsigned char t[32];
    bool ok = rdw(bw, chrom_tree, t, 32) && get<uint32_t>(t) == 0x78CA8C91u;
    if (ok) ok = walk_chrom_tree(bw, chrom_tree + 32, get<uint32_t>(t + 8), get<uint32_t>(t + 12));
    unsigned char r[48];
    if (ok) ok = rdw(bw, full_index, r, 48) && get<uint32_t>(r) == 0x2468ACE0u;
    if (ok) ok = walk_rtree(bw, full_index + 48);
    if (!ok) {
        fprintf(stderr, "File %s: corrupt BigWig index\n", path);
        fclose(bw->fp);
        delete bw;
        return WTAMD_ERR_ARG;
    }
    std::vector<unsigned char>().swap(bw->win);
    bw_build_infos(bw);
    if (trace) fprintf(stderr, "[bw_open] fopen %.3f ms, header + chromosome tree + index walk (%zu leaves) %.3f ms\n", t1 - t0, bw->blocks.size(), bw_now_ms() - t1);
    *out = bw;
    return WTAMD_OK;
}

void wtamd_bw_close(wtamd_bw *bw) {
    if (!bw) return;
    if (bw->fp) fclose(bw->fp);
    delete (std::vector<Triple> *) bw->part;
    delete bw;
}

int wtamd_bw_n_chrom(const wtamd_bw *bw) { return bw ? (int) bw->chroms.size() : 0; }

const char *wtamd_bw_chrom_name(const wtamd_bw *bw, int i) {
    return (bw && i >= 0 && i < (int) bw->chroms.size()) ? bw->chroms[i].name.c_str() : nullptr;
}

uint32_t wtamd_bw_chrom_length(const wtamd_bw *bw, int i) {
    return (bw && i >= 0 && i < (int) bw->chroms.size()) ? bw->chroms[i].length : 0;
}

// Decodes every interval of chromosome `chrom` into start[]/finish[]/value[] (1-based start,
// exclusive finish, sorted).  box != 0: cut at the reference reader's 10 000-bp stretch edges.
// Returns the number of runs; if it exceeds `capacity` nothing is written and the caller
// retries with a bigger buffer; < 0 on error.
// Runs of `t` (0-based, sorted) as the reference's reader hands them over: 1-based, optionally cut at
// its 10 000-bp stretch edges.  write == false only counts.
// Never writes beyond `cap` entries (it keeps counting): the caller compares the result with its capacity.
static int64_t emit_runs(const std::vector<Triple> &t, int box, int64_t length, bool write, int32_t *start, int32_t *finish,
                         float *value, int64_t cap = INT64_MAX) {
    const int64_t stretch = 10000;
    int64_t n = 0;
    for (const Triple &x : t) {
        const int64_t s = (int64_t) x.start + 1, f = (int64_t) x.end + 1;      // bigWiggleReader.c:39-40
        if (!box) {
            if (write && n < cap) { start[n] = (int32_t) s; finish[n] = (int32_t) f; value[n] = x.value; }
            n++;
            continue;
        }
        // stretches [1+10000k, 1+10000(k+1)) for 1+10000k < length  (bigWiggleReader.c:73-83)
        for (int64_t k = (s - 1) / stretch; ; k++) {
            const int64_t a = 1 + k * stretch, b = a + stretch;
            if (a >= length || a >= f) break;
            const int64_t bs = std::max(s, a), bf = std::min(f, b);          // :42-44
            if (bs < bf) {
                if (write && n < cap) { start[n] = (int32_t) bs; finish[n] = (int32_t) bf; value[n] = x.value; }
                n++;
            }
        }
    }
    return n;
}

int64_t wtamd_bw_read_part(wtamd_bw *bw, const char *chrom, int box, int64_t *cursor, int max_blocks, int32_t lo0, int32_t hi0,
                           int64_t capacity, int32_t *start, int32_t *finish, float *value, int *last) {
    if (!bw || !chrom || !cursor || !last || max_blocks <= 0) return -1;
    const BwChrom *c = nullptr;
    for (const BwChrom &x : bw->chroms)
        if (x.name == chrom) c = &x;
    if (!c) { *last = 1; return 0; }
    if (!bw->part) bw->part = new std::vector<Triple>();
    std::vector<Triple> &t = *(std::vector<Triple> *) bw->part;
    const bool cached = bw->part_from == *cursor && bw->part_chrom == (int) c->id && bw->part_blocks == max_blocks &&
                        bw->part_lo == lo0 && bw->part_hi == hi0;
    if (!cached) {
        t.clear();
        int64_t i = *cursor;
        const int64_t nb = (int64_t) bw->blocks.size();
        int taken = 0;
        bool ended = false;
        for (; i < nb && taken < max_blocks; i++) {
            const BwBlock &b = bw->blocks[(size_t) i];
            if (b.end_chrom < c->id) continue;
            if (b.start_chrom > c->id) { ended = true; break; }
            if (b.end_chrom == c->id && (int64_t) b.end_base <= (int64_t) lo0) continue;             // wholly before the window
            if (b.start_chrom == c->id && (int64_t) b.start_base >= (int64_t) hi0) { ended = true; break; }
            if (!decode_block(bw, b, c->id, t)) { fprintf(stderr, "wiggletools_amd: %s\n", bw->error.c_str()); return -2; }
            taken++;
        }
        if (!ended) {                       // anything of this chromosome left?
            ended = true;
            for (int64_t k = i; k < nb; k++) {
                const BwBlock &b = bw->blocks[(size_t) k];
                if (b.end_chrom < c->id) continue;
                if (b.start_chrom > c->id) break;
                if (b.start_chrom == c->id && (int64_t) b.start_base >= (int64_t) hi0) break;
                ended = false;
                break;
            }
        }
        if (!std::is_sorted(t.begin(), t.end(), [](const Triple &a, const Triple &b) { return a.start < b.start; }))
            std::stable_sort(t.begin(), t.end(), [](const Triple &a, const Triple &b) { return a.start < b.start; });
        bw->part_from = *cursor; bw->part_to = i; bw->part_chrom = (int) c->id; bw->part_blocks = max_blocks;
        bw->part_lo = lo0; bw->part_hi = hi0; bw->part_last = ended;
    }
    // pieces <= runs + stretch edges crossed: when the caller's arrays surely hold that, write at once
    int64_t bound = (int64_t) t.size();
    if (box && !t.empty()) bound += ((int64_t) t.back().end - (int64_t) t.front().start) / 10000 + 2 + (int64_t) t.size() / 64;
    int64_t n;
    if (!box || bound <= capacity) {
        if ((int64_t) t.size() > capacity) return (int64_t) t.size();      // (box == 0: one piece per run)
        n = emit_runs(t, box, c->length, true, start, finish, value, capacity);
        if (n > capacity) return n;         // (overlapping runs in a malformed file can exceed the bound)
    } else {
        n = emit_runs(t, box, c->length, false, nullptr, nullptr, nullptr);
        if (n > capacity) return n;         // the part stays cached: the next call only writes
        emit_runs(t, box, c->length, true, start, finish, value);
    }
    *cursor = bw->part_to;
    *last = bw->part_last ? 1 : 0;
    bw->part_from = -1;
    return n;
}

int64_t wtamd_bw_read_chrom(wtamd_bw *bw, const char *chrom, int box, int64_t capacity,
                            int32_t *start, int32_t *finish, float *value) {
    if (!bw || !chrom) return -1;
    const BwChrom *c = nullptr;
    for (const BwChrom &x : bw->chroms)
        if (x.name == chrom) c = &x;
    if (!c) return 0;
    std::vector<Triple> t;
    for (const BwBlock &b : bw->blocks)
        if (b.start_chrom <= c->id && c->id <= b.end_chrom)
            if (!decode_block(bw, b, c->id, t)) { fprintf(stderr, "wiggletools_amd: %s\n", bw->error.c_str()); return -2; }
    if (!std::is_sorted(t.begin(), t.end(), [](const Triple &a, const Triple &b) { return a.start < b.start; }))
        std::stable_sort(t.begin(), t.end(), [](const Triple &a, const Triple &b) { return a.start < b.start; });
    const int64_t n = emit_runs(t, box, c->length, false, nullptr, nullptr, nullptr);
    if (n > capacity) return n;
    return emit_runs(t, box, c->length, true, start, finish, value, capacity);
}

}  // extern "C"
