// wt_bwwrite.cpp -- BigWig WRITER for the bench's and the tests' input files (plumbing next to the synthetic
// generator csrc/wt_synth.hip, NOT part of the path: the product reads BigWig, the reference never writes it either --
// wigWriter.c emits text).  Exists because the north-star workload is "mean over 100 whole-genome BigWig files":
// 100 files x 24 chromosomes at mean run 16 bp are 6e9 intervals, 6 million zlib streams -- wiggletools_amd/bwwrite.py
// (one Python call per section) needed minutes for chromosome 1 alone.  Here every file has its own writer, a
// chromosome's tracks are dealt to worker threads, and the index is an R-tree of as many levels as the sections need
// (bwwrite.py stopped at 65 536 sections).
//
// Layout written (Kent et al. 2010, what csrc/wt_bigwig.cpp and tests/bw_indep_reader.py read): 64-byte header,
// chromosome B+ tree with one leaf node, section count, bedGraph (type 1) sections of `items_per_block` records
// (start, end, float value; 0-based half-open) each a zlib stream, R-tree index with 256 entries per node.
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/wiggletools_amd.h"

struct wtamd_bw_writer {
    FILE *fp = nullptr;
    std::vector<std::string> names;     // in id order (= strcmp order: the caller sorts)
    std::vector<uint32_t> lengths;
    int items = 1024, level = 1;
    uint64_t data_off = 0, pos = 0;
    struct Leaf { uint32_t chrom, start, end_chrom, end; uint64_t off, size; };
    std::vector<Leaf> leaves;
    uint32_t ubuf = 0;
    int last_chrom = -1;
    std::vector<unsigned char> raw, comp;
    bool failed = false;
};

namespace {

template <class T>
void put(std::vector<unsigned char> &b, T v) {
    const unsigned char *p = (const unsigned char *) &v;
    b.insert(b.end(), p, p + sizeof(T));
}

bool wr(wtamd_bw_writer *w, const void *p, size_t n) {
    if (n && fwrite(p, 1, n, w->fp) != n) { w->failed = true; return false; }
    w->pos += n;
    return true;
}

}  // namespace

extern "C" {

int wtamd_bw_writer_open(const char *path, int n_chrom, const char *const *names, const uint32_t *lengths, int items_per_block,
                         int zlib_level, wtamd_bw_writer **out) {
    if (!path || !names || !lengths || !out || n_chrom <= 0 || n_chrom > 65535 || items_per_block <= 0 || items_per_block > 65535 ||
        zlib_level < 0 || zlib_level > 9)
        return WTAMD_ERR_ARG;
    for (int c = 1; c < n_chrom; c++)
        if (strcmp(names[c - 1], names[c]) >= 0) return WTAMD_ERR_ARG;    // ids follow the B+ tree's key order
    wtamd_bw_writer *w = new wtamd_bw_writer();
    w->fp = fopen(path, "wb");
    if (!w->fp) { delete w; return WTAMD_ERR_ARG; }
    setvbuf(w->fp, nullptr, _IOFBF, 4 << 20);
    w->items = items_per_block; w->level = zlib_level;
    size_t key = 1;
    for (int c = 0; c < n_chrom; c++) { w->names.push_back(names[c]); w->lengths.push_back(lengths[c]); key = std::max(key, strlen(names[c])); }
    std::vector<unsigned char> head(64, 0), tree;
    put<uint32_t>(tree, 0x78CA8C91u); put<uint32_t>(tree, (uint32_t) n_chrom); put<uint32_t>(tree, (uint32_t) key); put<uint32_t>(tree, 8u);
    put<uint64_t>(tree, (uint64_t) n_chrom); put<uint64_t>(tree, 0ull);
    put<uint8_t>(tree, 1); put<uint8_t>(tree, 0); put<uint16_t>(tree, (uint16_t) n_chrom);
    for (int c = 0; c < n_chrom; c++) {
        std::string k = w->names[(size_t) c];
        k.resize(key, '\0');
        tree.insert(tree.end(), k.begin(), k.end());
        put<uint32_t>(tree, (uint32_t) c); put<uint32_t>(tree, lengths[c]);
    }
    w->data_off = 64 + tree.size();
    const uint64_t zero = 0;
    if (!wr(w, head.data(), 64) || !wr(w, tree.data(), tree.size()) || !wr(w, &zero, 8)) { fclose(w->fp); delete w; return WTAMD_ERR_INTERNAL; }
    *out = w;
    return WTAMD_OK;
}

// The intervals of chromosome `chrom` (ids ascending from call to call; none twice): 1-based start, exclusive finish
// -- the engine's run-list convention -- sorted and disjoint.
int wtamd_bw_writer_add(wtamd_bw_writer *w, int chrom, int64_t n, const int32_t *start, const int32_t *finish, const float *value) {
    if (!w || chrom <= w->last_chrom || chrom >= (int) w->names.size() || n < 0 || (n > 0 && (!start || !finish || !value))) return WTAMD_ERR_ARG;
    w->last_chrom = chrom;
    for (int64_t k = 0; k < n; k += w->items) {
        const int64_t m = std::min<int64_t>(w->items, n - k);
        w->raw.resize(24 + 12 * (size_t) m);
        unsigned char *p = w->raw.data();
        const uint32_t s0 = (uint32_t) (start[k] - 1), e1 = (uint32_t) (finish[k + m - 1] - 1), z32 = 0, cid = (uint32_t) chrom;
        memcpy(p, &cid, 4); memcpy(p + 4, &s0, 4); memcpy(p + 8, &e1, 4); memcpy(p + 12, &z32, 4); memcpy(p + 16, &z32, 4);
        p[20] = 1; p[21] = 0;
        const uint16_t cnt = (uint16_t) m;
        memcpy(p + 22, &cnt, 2);
        p += 24;
        for (int64_t q = 0; q < m; q++, p += 12) {
            const uint32_t a = (uint32_t) (start[k + q] - 1), b = (uint32_t) (finish[k + q] - 1);
            memcpy(p, &a, 4); memcpy(p + 4, &b, 4); memcpy(p + 8, &value[k + q], 4);
        }
        uLongf cn = compressBound((uLong) w->raw.size());
        w->comp.resize(cn);
        if (compress2(w->comp.data(), &cn, w->raw.data(), (uLong) w->raw.size(), w->level) != Z_OK) { w->failed = true; return WTAMD_ERR_INTERNAL; }
        w->leaves.push_back({cid, s0, cid, e1, w->pos, (uint64_t) cn});
        if (!wr(w, w->comp.data(), cn)) return WTAMD_ERR_INTERNAL;
        w->ubuf = std::max<uint32_t>(w->ubuf, (uint32_t) w->raw.size());
    }
    return WTAMD_OK;
}

// Index + header; closes the file.  Returns the number of sections written or < 0.
int64_t wtamd_bw_writer_close(wtamd_bw_writer *w) {
    if (!w) return WTAMD_ERR_ARG;
    const uint64_t index_off = w->pos;
    const size_t n = w->leaves.size();
    // R-tree bottom-up: level 0 = leaf nodes over <= 256 sections, level k = nodes over <= 256 nodes of level k - 1
    struct Node { uint32_t c0, s0, c1, e1; uint64_t off; };
    std::vector<std::vector<Node>> levels;       // extents of the nodes of each level; offsets filled in below
    {
        std::vector<Node> cur;
        for (size_t i = 0; i < n || (n == 0 && i == 0); i += 256) {
            Node nd{0, 0, 0, 0, 0};
            if (n) {
                const size_t j = std::min(n, i + 256) - 1;
                nd = Node{w->leaves[i].chrom, w->leaves[i].start, w->leaves[j].end_chrom, w->leaves[j].end, 0};
            }
            cur.push_back(nd);
        }
        levels.push_back(cur);
        while (levels.back().size() > 1) {
            const std::vector<Node> &lo = levels.back();
            std::vector<Node> up;
            for (size_t i = 0; i < lo.size(); i += 256) {
                const size_t j = std::min(lo.size(), i + 256) - 1;
                up.push_back(Node{lo[i].c0, lo[i].s0, lo[j].c1, lo[j].e1, 0});
            }
            levels.push_back(up);
        }
    }
    // node sizes -> offsets, root first, then each level in order
    uint64_t at = index_off + 48;
    for (size_t L = levels.size(); L-- > 0;) {
        for (size_t i = 0; i < levels[L].size(); i++) {
            levels[L][i].off = at;
            const size_t below = L == 0 ? n : levels[L - 1].size();
            const size_t cnt = below ? std::min<size_t>(256, below - i * 256) : 0;
            at += 4 + (L == 0 ? 32 : 24) * cnt;
        }
    }
    std::vector<unsigned char> idx;
    put<uint32_t>(idx, 0x2468ACE0u); put<uint32_t>(idx, 256u); put<uint64_t>(idx, (uint64_t) n);
    put<uint32_t>(idx, n ? w->leaves.front().chrom : 0u); put<uint32_t>(idx, n ? w->leaves.front().start : 0u);
    put<uint32_t>(idx, n ? w->leaves.back().end_chrom : 0u); put<uint32_t>(idx, n ? w->leaves.back().end : 0u);
    put<uint64_t>(idx, index_off); put<uint32_t>(idx, 1u); put<uint32_t>(idx, 0u);
    for (size_t L = levels.size(); L-- > 0;) {
        for (size_t i = 0; i < levels[L].size(); i++) {
            const size_t below = L == 0 ? n : levels[L - 1].size();
            const size_t cnt = below ? std::min<size_t>(256, below - i * 256) : 0;
            put<uint8_t>(idx, L == 0 ? 1 : 0); put<uint8_t>(idx, 0); put<uint16_t>(idx, (uint16_t) cnt);
            for (size_t q = i * 256; q < i * 256 + cnt; q++) {
                if (L == 0) {
                    const auto &s = w->leaves[q];
                    put<uint32_t>(idx, s.chrom); put<uint32_t>(idx, s.start); put<uint32_t>(idx, s.end_chrom); put<uint32_t>(idx, s.end);
                    put<uint64_t>(idx, s.off); put<uint64_t>(idx, s.size);
                } else {
                    const Node &c = levels[L - 1][q];
                    put<uint32_t>(idx, c.c0); put<uint32_t>(idx, c.s0); put<uint32_t>(idx, c.c1); put<uint32_t>(idx, c.e1);
                    put<uint64_t>(idx, c.off);
                }
            }
        }
    }
    bool ok = !w->failed && wr(w, idx.data(), idx.size());
    std::vector<unsigned char> head;
    put<uint32_t>(head, 0x888FFC26u); put<uint16_t>(head, 4); put<uint16_t>(head, 0);
    put<uint64_t>(head, 64ull); put<uint64_t>(head, w->data_off); put<uint64_t>(head, index_off);
    put<uint16_t>(head, 0); put<uint16_t>(head, 0); put<uint64_t>(head, 0ull); put<uint64_t>(head, 0ull);
    put<uint32_t>(head, w->ubuf ? w->ubuf : 1u); put<uint64_t>(head, 0ull);
    ok = ok && fflush(w->fp) == 0;
    const uint64_t cnt = (uint64_t) n;
    ok = ok && pwrite(fileno(w->fp), head.data(), head.size(), 0) == (ssize_t) head.size();
    ok = ok && pwrite(fileno(w->fp), &cnt, 8, (off_t) w->data_off) == 8;
    ok = (fclose(w->fp) == 0) && ok;
    delete w;
    return ok ? (int64_t) n : (int64_t) WTAMD_ERR_INTERNAL;
}

// One chromosome of MANY files at once: track i's intervals are [seg_off[i], seg_off[i + 1]) of the SoA arrays (the
// engine's run-list layout for one chromosome); the tracks are dealt to `threads` workers.
int wtamd_bw_writers_add_chrom(wtamd_bw_writer *const *ws, int n_tracks, int chrom, const int64_t *seg_off, const int32_t *start,
                               const int32_t *finish, const float *value, int threads) {
    if (!ws || n_tracks <= 0 || !seg_off) return WTAMD_ERR_ARG;
    if (threads < 1) threads = 1;
    if (threads > n_tracks) threads = n_tracks;
    std::atomic<int> next(0), bad(0);
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n_tracks) return;
            const int64_t a = seg_off[i], b = seg_off[i + 1];
            if (wtamd_bw_writer_add(ws[i], chrom, b - a, start + a, finish + a, value + a) != WTAMD_OK) bad.store(1);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    return bad.load() ? WTAMD_ERR_INTERNAL : WTAMD_OK;
}

}  // extern "C"
