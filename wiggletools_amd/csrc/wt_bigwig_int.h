// wt_bigwig_int.h -- what the drop-in layer (wt_iter_abi.cpp) needs to know about an open BigWig file
// to hand its data sections to the device undecoded (wtamd_pipe_submit_bw): the R-tree leaves of a
// chromosome in file order and where their bytes lie.  Internal to the library (not part of the C ABI).
#ifndef WT_BIGWIG_INT_H_
#define WT_BIGWIG_INT_H_

#include <stdint.h>

#include "../../include/wiggletools_amd.h"

struct WtBwLeaf {               // one data section as the index describes it
    uint32_t start_chrom, start_base, end_chrom, end_base;     // 0-based half-open extents
    uint64_t offset, size;      // bytes in the file
};

struct WtBwChromInfo {
    uint32_t id, length;
    int64_t first, count;       // leaves [first, first + count) of wt_bw_leaves()
    bool device_ok;             // every leaf lies on this chromosome only, leaves sorted and disjoint
    uint32_t max_size;          // largest leaf in bytes
};

const WtBwLeaf *wt_bw_leaves(const wtamd_bw *bw, int64_t *n);
bool wt_bw_chrom_info(wtamd_bw *bw, const char *chrom, WtBwChromInfo *out);    // false: no such chromosome
int wt_bw_fd(const wtamd_bw *bw);                   // for pread(): thread-safe, independent of the FILE position
uint32_t wt_bw_uncompress_buf(const wtamd_bw *bw);  // 0: sections are stored raw

#endif  // WT_BIGWIG_INT_H_
