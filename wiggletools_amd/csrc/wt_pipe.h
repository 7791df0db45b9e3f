// wt_pipe.h -- the streaming pipeline (wtamd_pipe_*, include/wiggletools_amd.h).  Included at the
// end of wt_engine.hip (it launches that file's kernels and rebinds its track sets).
//
// The reference overlaps its readers with the evaluation through producer threads and
// 10 000-entry SoA blocks, at most 3 blocks ahead (bufferedReader.c:17-28,41-55,99-109).  Here
// the same role is played across host, PCIe and GPU by `n_slots` batch slots and three HIP streams:
//
//      host (Drainer)     fill k+2 | fill k+3 | ...
//      copy stream        H2D k+1  | H2D k+2  | ...          hipMemcpyAsync from PINNED staging
//      compute stream     index + multiplex/reduce kernels k | k+1 | ...
//      result stream      D2H k-1 (exactly the emitted runs) | D2H k | ...   into PINNED output
//
// Events order the three streams per slot; nothing is allocated, freed or synchronised per batch
// (staging, device buffers and window tables only ever grow) and the host never has to learn a
// run count before the result can travel: the export kernel reads it on the device and writes
// exactly the emitted runs (and the counters) into the slot's pinned output through the link.
// All three legs are KERNELS -- gather (reads the page-locked run lists over PCIe), the
// multiplex/reduce kernels, export (writes over PCIe) -- so consecutive batches overlap on the
// GPU's queues without SDMA hand-offs in between (measured: with hipMemcpyAsync legs the next
// batch's H2D did not start before the previous batch's D2H had finished).  A batch whose
// difference-array launch reported windows it could not prove exact (rare: NaN, Inf, huge dynamic
// range) is patched and exported again when it is collected.
#ifndef WT_PIPE_H_
#define WT_PIPE_H_

// wt_compress.hip
long long wt_map_scratch_words(long long capacity);                             // wt_map.hip
int wt_map_upload_chains(const wtamd_map_chain *chains, int n_tracks, void **d_out, bool *drops, bool *f32_exact);
int wt_map_chain_async(const void *d_chains, int n_tracks, bool drops, const int64_t *d_seg_in, long long n, const int32_t *start,
                       const int32_t *finish, const void *value, bool value_is_f64, unsigned long long *scratch,
                       int32_t *o_start, int32_t *o_finish, double *o_value, int64_t *d_seg_out, hipStream_t stream, bool out_f32);
long long wt_bw_scratch_bytes(long long n_sec, long long plain_stride);           // wt_bwdev.hip
long long wt_bw_fill_sections(int num_cu);
int wt_bw_decode_async(const void *h_bytes, void *d_bytes, long long n_bytes, const void *d_comp, const void *d_secs, const void *d_tracks, int n_tracks,
                       long long n_sec, long long plain_stride, void *scratch, long long capacity, int32_t *o_start, int32_t *o_finish,
                       float *o_value, int64_t *d_seg_off, unsigned long long *h_status, int copy_blocks, hipStream_t s_copy,
                       hipEvent_t e_copied, hipStream_t s_dec);
long long wt_compress_scratch_words(long long capacity);
int wt_compress_async(const int32_t *start, const int32_t *finish, const double *value, const unsigned long long *d_n,
                      long long capacity, unsigned long long *scratch, int32_t *o_start, int32_t *o_finish, double *o_value,
                      unsigned long long *d_n_out, hipStream_t s);

// Gather: ONE kernel pulls every range of a batch -- the caller's pinned SoA blocks (bulk side
// door) and the staged ranges alike -- from host memory into the slot's device arrays.  The copy
// engine needs three hipMemcpyAsync per track and batch (~10 us of launch overhead each: 300 calls
// for 100 tracks, more than the transfer itself at 4 M intervals per batch); a kernel reading the
// page-locked host arrays through the PCIe link has no per-range cost and keeps thousands of reads
// in flight.  Host bandwidth is the bound either way (12 B per interval).
struct WtGatherSeg {
    const int32_t *start, *finish;
    const float *value;
    long long dst;                  // first interval of the range in the device arrays
    long long count;
    long long chunk_first;          // prefix sum of ceil(count / WT_GATHER_CHUNK)
};
#define WT_GATHER_CHUNK 4096

#define WT_GATHER_MAX_SEGS 1024      // table entries one launch caches in LDS (48 KB)

__global__ void __launch_bounds__(256) wt_gather_kernel(const WtGatherSeg *segs, int n_segs, long long n_chunks,
                                                         int32_t *d_start, int32_t *d_finish, float *d_value) {
    // the table lies in pinned HOST memory: every block pulls it into LDS once (one coalesced read
    // through the link) instead of a separate H2D copy ahead of the launch
    __shared__ WtGatherSeg tab[WT_GATHER_MAX_SEGS];
    {
        const long long *src = (const long long *) segs;
        long long *dst = (long long *) tab;
        const int words = n_segs * (int) (sizeof(WtGatherSeg) / 8);
        for (int i = threadIdx.x; i < words; i += 256) dst[i] = __builtin_nontemporal_load(src + i);
    }
    __syncthreads();
    for (long long ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        int lo = 0, hi = n_segs - 1;                    // last segment with chunk_first <= ch
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tab[mid].chunk_first <= ch) lo = mid; else hi = mid - 1;
        }
        const WtGatherSeg g = tab[lo];
        const long long a = (ch - g.chunk_first) * WT_GATHER_CHUNK;
        long long b = a + WT_GATHER_CHUNK;
        if (b > g.count) b = g.count;
        // issue every load of the chunk before the first store: the link's latency is microseconds
        int32_t vs[WT_GATHER_CHUNK / 256], vf[WT_GATHER_CHUNK / 256];
        float vv[WT_GATHER_CHUNK / 256];
#pragma unroll
        for (int q = 0; q < WT_GATHER_CHUNK / 256; q++) {
            const long long i = a + threadIdx.x + 256ll * q;
            if (i < b) { vs[q] = __builtin_nontemporal_load(g.start + i); vf[q] = __builtin_nontemporal_load(g.finish + i); vv[q] = __builtin_nontemporal_load(g.value + i); }
        }
#pragma unroll
        for (int q = 0; q < WT_GATHER_CHUNK / 256; q++) {
            const long long i = a + threadIdx.x + 256ll * q;
            if (i < b) { d_start[g.dst + i] = vs[q]; d_finish[g.dst + i] = vf[q]; d_value[g.dst + i] = vv[q]; }
        }
    }
}

// Export: the emitted runs (their count is read on the device) and the launch counters go to the
// slot's PINNED host output, written through the link by the kernel itself.
#define WT_CTR_EXPORTED 6            // h_counters slot: runs the export kernel shipped (== WT_CTR_RUNS unless compressed)
__global__ void __launch_bounds__(256) wt_export_kernel(const unsigned long long *d_counters, unsigned long long *h_counters,
                                                         const unsigned long long *n_src, long long capacity, int n_tracks,
                                                         const int32_t *d_os, const int32_t *d_of, const double *d_ov,
                                                         const double *d_tile, const uint8_t *d_ip,
                                                         int32_t *h_os, int32_t *h_of, double *h_ov, double *h_tile, uint8_t *h_ip) {
    long long n = (long long) *n_src;
    if (n > capacity) n = capacity;
    const long long stride = (long long) gridDim.x * 256, t = (long long) blockIdx.x * 256 + threadIdx.x;
    {   // coordinates: 16 bytes per lane
        const long long n4 = n >> 2;
        const int4 *a = (const int4 *) d_os, *b = (const int4 *) d_of;
        int4 *ha = (int4 *) h_os, *hb = (int4 *) h_of;
        for (long long i = t; i < n4; i += stride) { ha[i] = a[i]; hb[i] = b[i]; }
        for (long long i = (n4 << 2) + t; i < n; i += stride) { h_os[i] = d_os[i]; h_of[i] = d_of[i]; }
        const long long n2 = n >> 1;
        const double2 *v = (const double2 *) d_ov;
        double2 *hv = (double2 *) h_ov;
        for (long long i = t; i < n2; i += stride) hv[i] = v[i];
        if (t == 0 && (n & 1)) h_ov[n - 1] = d_ov[n - 1];
    }
    if (d_tile) {
        const long long m = n * n_tracks;
        for (long long i = t; i < m; i += stride) { h_tile[i] = d_tile[i]; h_ip[i] = d_ip[i]; }
    }
    if (t < WT_CTR_N) h_counters[t] = t == WT_CTR_EXPORTED ? (unsigned long long) n : d_counters[t];
}

// Page-locked (hipHostMalloc / hipHostRegister) host memory is readable by kernels; pageable memory
// is not -- such ranges go through hipMemcpyAsync, which stages them.
static bool wt_is_registered(const void *q);
static bool wt_is_pinned(const void *q) {
    if (wt_is_registered(q)) return true;       // (this library's own mmap + hipHostRegister buffers, below)
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void) hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// Page-locked host memory is expensive to get (hipHostMalloc pins pages at ~6 GB/s: the three slots of a pipe that
// streams 300 MB batches cost ~0.25 s) and to give back (hipHostFree waits for the device).  Buffers of 1 MB and more
// are therefore kept in a process-wide pool when a pipe lets go of them and handed to the next pipe that asks for
// the same size -- the Multiplexer a reducer takes over, the next reducer of a long-lived process.  Bounded by
// WTAMD_PINNED_POOL_MB (default 8192: a pipe of 100 BigWig tracks holds 3.7 GB; 0 switches the pool off).
struct WtPinnedPool {
    std::mutex mu;
    std::multimap<size_t, void *> free_list;        // by (rounded) size
    std::map<void *, size_t> size_of;               // every live buffer that came through here
    size_t pooled = 0;
    size_t misses = 0, miss_bytes = 0;              // buffers of 1 MB and more that had to be page-locked afresh
    size_t limit() const {
        const char *e = getenv("WTAMD_PINNED_POOL_MB");
        return (size_t) (e ? atoll(e) : 8192) << 20;
    }
};
static WtPinnedPool g_pinned_pool;

// Sizes of 1 MB and more are rounded up to eighths of their power of two before they reach the pool or the runtime:
// the staging of a file-byte batch is sized by the batch (306 995 195 bytes, then 308 322 053, ...), so the next run
// of the same job never asked for exactly what the previous one had returned and page-locked everything afresh --
// 0.8 s of hipHostMalloc on hosts where that runs at 1.5 GB/s (round 3; seen as a second run SLOWER than the first).
static size_t wt_pool_round(size_t bytes) {
    if (bytes < (1u << 20)) return bytes;
    int lg = 63;
    while (!((bytes >> lg) & 1u)) lg--;
    const size_t step = (size_t) 1 << (lg - 3);
    return (bytes + step - 1) / step * step;
}

// Page-locking by the page.  hipHostMalloc allocates AND faults AND pins from one thread: 176-229 ms per GiB on the
// MI355X hosts measured (tools/probes/cold_probe.hip; 4.3 GB of staging = 0.3-0.8 s of a cold file-byte run, round 4's
// "pinned_afresh").  The same GiB as an anonymous mapping with transparent huge pages, faulted in by 16 threads
// (4 ms) and then registered (hipHostRegister: 2 ms -- 512 huge pages to pin instead of 262 144 small ones) costs 6 ms,
// and the copy engine reads it at the same 57 GB/s.  Buffers of 2 MB and more take that route (WTAMD_PIN=malloc: the
// old one); anything the runtime refuses falls back to hipHostMalloc.
struct WtRegistered { void *base; size_t map_len; size_t len; };   // the mapping (for munmap) and the page-locked bytes from the pointer handed out
static std::mutex g_reg_mu;
static std::map<void *, WtRegistered> g_registered;        // registered mappings, by the pointer handed out
static std::atomic<int> g_reg_state{0};                      // 0 untried, 1 works, -1 does not (hipHostMalloc from then on)
static std::atomic<int> g_reg_failures{0};                   // hipHostRegister refusals in a row (a transient one -- RLIMIT_MEMLOCK on one large buffer -- does not end the route)

static bool wt_is_registered(const void *q) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = g_registered.upper_bound((void *) q);
    if (it == g_registered.begin()) return false;
    --it;
    return (const char *) q < (const char *) it->first + it->second.len;       // (the mapping's alignment slack behind it is NOT page-locked)
}

static int wt_pin_threads() {
    static const int n = [] {
        int c = (int) std::thread::hardware_concurrency();
        if (FILE *fp = fopen("/sys/fs/cgroup/cpu.max", "r")) {       // (the container's CPU quota: the GPU boxes show 256 CPUs and grant 16)
            char q[64]; long long period = 0;
            if (fscanf(fp, "%63s %lld", q, &period) == 2 && period > 0 && strcmp(q, "max") != 0) {
                const long long k = atoll(q) / period;
                if (k >= 1 && k < c) c = (int) k;
            }
            fclose(fp);
        }
        return c < 1 ? 1 : (c > 16 ? 16 : c);
    }();
    return n;
}

static bool wt_pin_by_register(void **out, size_t bytes) {
    static const bool off = getenv("WTAMD_PIN") && !strcmp(getenv("WTAMD_PIN"), "malloc");
    if (off || g_reg_state.load() < 0 || bytes < ((size_t) 2 << 20)) return false;
    const size_t huge = (size_t) 2 << 20;
    const size_t len = (bytes + huge - 1) / huge * huge;
    void *base = mmap(nullptr, len + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base == MAP_FAILED) return false;
    char *p = (char *) (((uintptr_t) base + huge - 1) & ~(uintptr_t) (huge - 1));
#ifdef MADV_HUGEPAGE
    (void) madvise(p, len, MADV_HUGEPAGE);
#endif
    // fault the pages in from several threads (one touch per 4 KB: right with and without huge pages)
    int T = wt_pin_threads();
    const size_t per_thread_min = (size_t) 32 << 20;
    if ((size_t) T > len / per_thread_min) T = (int) (len / per_thread_min);
    if (T < 1) T = 1;
    const size_t slice = (len / (size_t) T + huge - 1) / huge * huge;
    auto touch = [p, len, slice](int t) {
        const size_t a = slice * (size_t) t, b = a + slice < len ? a + slice : len;
        for (size_t q = a; q < b; q += 4096) ((volatile char *) p)[q] = 0;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(touch, t);
    touch(0);
    for (auto &t : th) t.join();
    void *dp = nullptr;
    const bool registered = hipHostRegister(p, len, hipHostRegisterDefault) == hipSuccess;
    if (!registered || hipHostGetDevicePointer(&dp, p, 0) != hipSuccess || dp != (void *) p) {
        // (kernels of the pipe read and write the staging through the HOST address: it must be the device's too)
        (void) hipGetLastError();
        // a host pointer that is not the device's: this runtime cannot do it, ever; a refused registration: maybe just this size, now
        const bool never = registered;
        if (registered) (void) hipHostUnregister(p);
        munmap(base, len + huge);
        if (never || g_reg_failures.fetch_add(1) + 1 >= 3) g_reg_state.store(-1);
        return false;
    }
    g_reg_state.store(1);
    g_reg_failures.store(0);
    { std::lock_guard<std::mutex> lk(g_reg_mu); g_registered[p] = WtRegistered{base, len + huge, len}; }
    *out = p;
    return true;
}

static hipError_t wt_pin_raw_alloc(void **out, size_t bytes) {
    if (wt_pin_by_register(out, bytes)) return hipSuccess;
    return hipHostMalloc(out, bytes, hipHostMallocDefault);
}

static void wt_pin_raw_free(void *q) {
    WtRegistered r{nullptr, 0};
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        auto it = g_registered.find(q);
        if (it != g_registered.end()) { r = it->second; g_registered.erase(it); }
    }
    if (r.base) { (void) hipHostUnregister(q); munmap(r.base, r.map_len); }
    else (void) hipHostFree(q);
}

static hipError_t wt_host_alloc(void **out, size_t bytes) {
    if (bytes < 1) bytes = 1;
    bytes = wt_pool_round(bytes);
    if (bytes >= (1u << 20)) {
        std::lock_guard<std::mutex> lk(g_pinned_pool.mu);
        // the smallest resting buffer that is large enough and at most a quarter larger (slot capacities grow by
        // doubling from whatever the first batches needed, so two runs of one job rarely end on identical sizes)
        auto it = g_pinned_pool.free_list.lower_bound(bytes);
        if (it != g_pinned_pool.free_list.end() && it->first <= bytes + bytes / 4) {
            *out = it->second;
            g_pinned_pool.pooled -= it->first;
            g_pinned_pool.free_list.erase(it);
            return hipSuccess;
        }
    }
    const auto t_alloc0 = std::chrono::steady_clock::now();
    hipError_t e = wt_pin_raw_alloc(out, bytes);
    if (e != hipSuccess) {
        // the host refuses to page-lock more while buffers rest in the pool: give them all back and try once more
        std::vector<void *> idle;
        {
            std::lock_guard<std::mutex> lk(g_pinned_pool.mu);
            for (auto &kv : g_pinned_pool.free_list) { idle.push_back(kv.second); g_pinned_pool.size_of.erase(kv.second); }
            g_pinned_pool.free_list.clear();
            g_pinned_pool.pooled = 0;
        }
        if (!idle.empty()) {
            (void) hipGetLastError();
            for (void *x : idle) wt_pin_raw_free(x);
            e = wt_pin_raw_alloc(out, bytes);
        }
    }
    if (e == hipSuccess && bytes >= (1u << 20)) {
        std::lock_guard<std::mutex> lk(g_pinned_pool.mu);
        g_pinned_pool.size_of[*out] = bytes;
        g_pinned_pool.misses++;
        g_pinned_pool.miss_bytes += bytes;
        static const bool trace = getenv("WTAMD_TRACE_POOL") != nullptr;
        bool by_register = false;
        if (trace) { std::lock_guard<std::mutex> lk2(g_reg_mu); by_register = g_registered.count(*out) != 0; }
        if (trace) fprintf(stderr, "[pool] page-locked %.1f MB (%s) in %.1f ms\n", bytes / 1048576.0, by_register ? "mmap + hipHostRegister" : "hipHostMalloc",
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_alloc0).count());
    }
    return e;
}

static void wt_host_free(void *q) {
    if (!q) return;
    {
        std::lock_guard<std::mutex> lk(g_pinned_pool.mu);
        auto it = g_pinned_pool.size_of.find(q);
        if (it != g_pinned_pool.size_of.end()) {
            if (g_pinned_pool.pooled + it->second <= g_pinned_pool.limit()) {
                g_pinned_pool.free_list.emplace(it->second, q);
                g_pinned_pool.pooled += it->second;
                return;
            }
            g_pinned_pool.size_of.erase(it);
        }
    }
    wt_pin_raw_free(q);
}

// Device buffers of a pipe, the same way: a pipe frees everything it holds when its reducer reaches the end of the data
// (35 hipFree calls, each of which synchronises the device and unmaps gigabytes), and the next reducer of the process
// maps it all again -- on some hosts that made the SECOND run of a job 2 x slower than the first (0.9 s inside
// wtamd_pipe_submit_bw for 27 batches; round 3).  Released buffers rest in a process-wide pool keyed by (device,
// rounded size); a pipe is destroyed only after its streams have been synchronised, so nothing in the pool is still
// in use.  Bounded by WTAMD_DEVICE_POOL_MB per process (default 65536 -- a pipe of 100 tracks holds 38 GB; 0 switches the pool off).
struct WtDevPool {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void *> free_list;
    std::map<void *, std::pair<int, size_t>> size_of;
    size_t pooled = 0, misses = 0, miss_bytes = 0;
    size_t limit() const {
        const char *e = getenv("WTAMD_DEVICE_POOL_MB");
        return (size_t) (e ? atoll(e) : 65536) << 20;
    }
};
static WtDevPool g_dev_pool;

template <class T>
static hipError_t wt_dev_alloc(T **out, size_t bytes, int line = __builtin_LINE()) {
    if (bytes < 1) bytes = 1;
    bytes = wt_pool_round(bytes);
    int dev = 0;
    (void) hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(g_dev_pool.mu);
        auto it = g_dev_pool.free_list.lower_bound({dev, bytes});       // (same rule as the pinned pool)
        if (it != g_dev_pool.free_list.end() && it->first.first == dev && it->first.second <= bytes + bytes / 4) {
            *out = (T *) it->second;
            g_dev_pool.pooled -= it->first.second;
            g_dev_pool.free_list.erase(it);
            return hipSuccess;
        }
    }
    void *q = nullptr;
    const auto t_alloc0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {
        // out of device memory with buffers resting in the pool: give them all back and try once more
        std::vector<void *> idle;
        {
            std::lock_guard<std::mutex> lk(g_dev_pool.mu);
            for (auto &kv : g_dev_pool.free_list) { idle.push_back(kv.second); g_dev_pool.size_of.erase(kv.second); }
            g_dev_pool.free_list.clear();
            g_dev_pool.pooled = 0;
        }
        if (!idle.empty()) {
            (void) hipGetLastError();
            for (void *x : idle) (void) hipFree(x);
            e = hipMalloc(&q, bytes);
        }
    }
    *out = (T *) q;
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_dev_pool.mu);
        g_dev_pool.size_of[q] = {dev, bytes};
        g_dev_pool.misses++;
        g_dev_pool.miss_bytes += bytes;
        static const bool trace = getenv("WTAMD_TRACE_POOL") != nullptr;
        if (trace && bytes >= (1u << 20)) fprintf(stderr, "[pool] hipMalloc %.1f MB in %.1f ms (device %d, wt_pipe.h:%d)\n", bytes / 1048576.0,
                                                  std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_alloc0).count(), dev, line);
    }
    return e;
}

static hipError_t wt_dev_free(void *q) {
    if (!q) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_dev_pool.mu);
        auto it = g_dev_pool.size_of.find(q);
        if (it != g_dev_pool.size_of.end()) {
            if (g_dev_pool.pooled + it->second.second <= g_dev_pool.limit()) {
                g_dev_pool.free_list.emplace(it->second, q);
                g_dev_pool.pooled += it->second.second;
                return hipSuccess;
            }
            g_dev_pool.size_of.erase(it);
        }
    }
    return hipFree(q);
}

struct WtSlot {
    int state = 0;                  // 0 free, 1 acquired, 2 submitted, 3 collected
    // input staging (pinned) and its device twin
    int64_t cap = 0;
    bool has64 = false;
    int64_t *h_seg = nullptr;
    int32_t *h_start = nullptr, *h_finish = nullptr;
    float *h_v32 = nullptr;
    double *h_v64 = nullptr;
    int64_t dcap = 0;
    bool d_has64 = false;
    int32_t *d_start = nullptr, *d_finish = nullptr;
    void *d_value = nullptr;
    // output: device + pinned
    int64_t ocap = 0;
    int32_t *d_os = nullptr, *d_of = nullptr;
    double *d_ov = nullptr, *d_tile = nullptr;
    uint8_t *d_ip = nullptr;
    int64_t *d_cro = nullptr;
    int32_t *h_os = nullptr, *h_of = nullptr;
    double *h_ov = nullptr, *h_tile = nullptr;
    uint8_t *h_ip = nullptr;
    wtamd_trackset *ts = nullptr;
    hipEvent_t e_h0 = nullptr, e_h1 = nullptr, e_k0 = nullptr, e_cnt = nullptr, e_patch = nullptr, e_d0 = nullptr, e_d1 = nullptr;
    // the batch in flight
    int64_t n_int = 0, n_runs = 0, covered = 0;
    bool f64 = false, used_delta = false, patched = false;
    int err = WTAMD_OK;
    std::string err_msg;
    int delta_W = 0;
    // bulk side door: ranges of the batch that are copied to HBM straight from the caller's arrays
    struct Direct { int64_t at, count; const int32_t *start, *finish; const float *value; };
    std::vector<Direct> direct;
    WtGatherSeg *h_segs = nullptr;  // gather table, pinned (the kernel reads it where it lies)
    // device-side run compression (WTAMD_PIPE_COMPRESS): merged runs + scratch bitmaps, sized with the output
    int32_t *d_cs = nullptr, *d_cf = nullptr;
    double *d_cv = nullptr;
    unsigned long long *d_cscratch = nullptr, *d_cn = nullptr;
    bool compressed = false;        // this batch's output went through the compression
    // operator chains (wtamd_pipe_set_map): mapped f64 values, compacted coordinates, raw offsets, scratch
    int64_t mcap = 0;
    bool m_has_coords = false;
    int32_t *d_mstart = nullptr, *d_mfinish = nullptr;
    double *d_mvalue = nullptr;
    int64_t *d_mseg = nullptr;
    unsigned long long *d_mscratch = nullptr;
    int64_t seg_cap = 0;
    bool direct_pinned = true;      // every direct range of this batch lies in page-locked memory
    // BigWig sections decoded on device (wtamd_pipe_submit_bw): pinned staging [track table | section table | file
    // bytes] and its device twin (ONE copy kernel moves all three), decode scratch, status words
    uint8_t *h_bw = nullptr, *d_bw = nullptr;
    int64_t h_bw_cap = 0, d_bw_cap = 0;
    int64_t bw_off_sec = 0, bw_off_bytes = 0;       // layout of the reserved staging
    int64_t bw_res_bytes = -1, bw_res_secs = -1;     // what wtamd_pipe_bw_reserve was asked for (-1: nothing reserved)
    int64_t bw_bytes = 0, bw_stride = 0;             // of the batch in flight (a batch that overflowed its run lists is decoded again)
    int64_t bw_bound = 0;                            // the host's upper bound of its intervals
    int bw_dec = 0;                                  // decode stream (and scratch) of the batch
    unsigned long long *h_bw_status = nullptr;       // pinned: error bits, pieces
    hipEvent_t e_bwc = nullptr, e_bw0 = nullptr, e_bw1 = nullptr;
    bool bw = false;                                 // the batch in flight came as file bytes
    int64_t bw_secs = 0;
    bool export_pending = false;                     // the runs travel by copy engine once the host knows their count (collect)
    // fused integrators (wtamd_pipe_set_integrate): partial sums / moments on device, the batch's integrals in pinned memory
    char *d_integ = nullptr;
    double *h_integ = nullptr;
    bool integrated = false;
};

// HIP's current device is per thread; a pipe lives on the device that was current when it was created, and the drop-in
// layer drives several pipes (one per GPU: WTAMD_DEVICES) from one thread.
struct WtDevGuard {
    int prev = -1;
    explicit WtDevGuard(int dev) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != dev && dev >= 0) { prev = cur; (void) hipSetDevice(dev); }
    }
    ~WtDevGuard() { if (prev >= 0) (void) hipSetDevice(prev); }
};

struct wtamd_pipe {
    wtamd_pipe_config cfg;
    int device = -1;
    std::vector<double> defaults;
    std::vector<WtSlot> slots;
    int head = 0, tail = 0, acquired = -1, in_flight = 0, held = 0;
    hipStream_t s_copy = nullptr, s_comp = nullptr, s_out = nullptr, s_dec = nullptr;
    // File-byte batches are inflated / decoded on the compute stream.  WTAMD_BW_DECODE_STREAMS=2: on two streams of their
    // own taking turns, so that the next batch's inflate kernel takes the lanes the part-filled last batch of a
    // chromosome leaves idle (a launch costs ~12 ms however few sections it holds).  Measured (round 4, 100 files x 24
    // chromosomes at GRCh38 x 0.2): SLOWER, 0.75 s against 0.68 s -- the workgroups of inflate(k + 1) hold their CUs for
    // 12 ms and the reduce kernels of batch k, which need whole CUs, wait behind them; the host then waits longer for
    // batch k and, with two batches in flight, submits k + 2 later.  Kept as a switch, off.
    hipStream_t s_decs[2] = {nullptr, nullptr};
    int n_decs = 0;
    int64_t bw_batches = 0;
    bool delta_failed = false;      // a batch had many inexact windows: Sum / Mean stay on the general kernel
    bool tile = false;
    bool compress = false;          // WTAMD_PIPE_COMPRESS: batches submitted from now on are merged on device before they travel
    bool integrate = false;         // wtamd_pipe_set_integrate: batches submitted from now on are integrated on device, no runs travel
    bool gather = true;             // WTAMD_PIPE_GATHER=0: hipMemcpyAsync per range instead of the gather kernel
    int gather_blocks = 64;         // WTAMD_GATHER_BLOCKS
    // buffers a slot outgrew: released when the pipe is destroyed -- hipFree / hipHostFree wait for the
    // whole device, i.e. for the batches in flight on the other slots (measured: 6-8 ms per growing
    // submit while the pipeline ramps up)
    std::vector<void *> dead_dev, dead_host;
    void *d_chains = nullptr;       // wtamd_pipe_set_map: per-track operator chains on device
    bool map_drops = false;         // ... some operator drops runs: batches are compacted
    bool map_f32 = false;           // ... every operator is float32-exact: float32 batches stay float32 (and on the exact kernels)
    int num_cu = 256;
    // File-byte batches: the inflate scratch is ONE per pipe -- every decode runs on the compute stream, in order, and
    // is through with the scratch before the next one starts (round 3 kept 0.8 GB of it in each of four slots).
    void *d_bw_scratch = nullptr;   // (the scratch of the decode stream in use: d_bw_scratches[k])
    int64_t bw_scratch_cap = 0;
    void *d_bw_scratches[2] = {nullptr, nullptr};
    int64_t bw_scratch_caps[2] = {0, 0};
    // ... and their run lists are sized from the densest batch seen so far (intervals / host bound), not from the bound:
    // the bound must assume 4-byte fixedStep items because the item type is inside the compressed stream, three times
    // what bedGraph sections hold.  A batch that does not fit reports WT_BW_ERR_CAPACITY and is decoded again at full
    // size when it is collected (wt_pipe_bw_redo); from then on the pipe sizes by the bound.  < 0: nothing seen yet.
    double bw_density = -1.0;
    int64_t bw_redone = 0;
    unsigned last_bw_err = 0;       // wtamd_pipe_bw_error
    wtamd_pipe_stats st{};
};

static void wt_slot_free(WtSlot &s) {
    if (s.h_seg) wt_host_free(s.h_seg);
    if (s.h_start) wt_host_free(s.h_start);
    if (s.h_finish) wt_host_free(s.h_finish);
    if (s.h_v32) wt_host_free(s.h_v32);
    if (s.h_v64) wt_host_free(s.h_v64);
    (void) wt_dev_free(s.d_start); (void) wt_dev_free(s.d_finish); (void) wt_dev_free(s.d_value);
    (void) wt_dev_free(s.d_os); (void) wt_dev_free(s.d_of); (void) wt_dev_free(s.d_ov); (void) wt_dev_free(s.d_tile); (void) wt_dev_free(s.d_ip);
    (void) wt_dev_free(s.d_cro);
    (void) wt_dev_free(s.d_cs); (void) wt_dev_free(s.d_cf); (void) wt_dev_free(s.d_cv); (void) wt_dev_free(s.d_cscratch); (void) wt_dev_free(s.d_cn);
    if (s.h_segs) wt_host_free(s.h_segs);
    (void) wt_dev_free(s.d_mstart); (void) wt_dev_free(s.d_mfinish); (void) wt_dev_free(s.d_mvalue); (void) wt_dev_free(s.d_mseg); (void) wt_dev_free(s.d_mscratch);
    if (s.h_os) wt_host_free(s.h_os);
    if (s.h_of) wt_host_free(s.h_of);
    if (s.h_ov) wt_host_free(s.h_ov);
    if (s.h_tile) wt_host_free(s.h_tile);
    if (s.h_ip) wt_host_free(s.h_ip);
    if (s.h_bw) wt_host_free(s.h_bw);
    if (s.h_integ) wt_host_free(s.h_integ);
    (void) wt_dev_free(s.d_integ);
    if (s.h_bw_status) wt_host_free(s.h_bw_status);
    (void) wt_dev_free(s.d_bw);
    for (hipEvent_t e : {s.e_bwc, s.e_bw0, s.e_bw1})
        if (e) (void) hipEventDestroy(e);
    if (s.ts) {
        s.ts->d_start = s.ts->d_finish = nullptr; s.ts->d_value = nullptr;      // the slot's, freed above
        wtamd_trackset_destroy(s.ts);
    }
    for (hipEvent_t e : {s.e_h0, s.e_h1, s.e_k0, s.e_cnt, s.e_patch, s.e_d0, s.e_d1})
        if (e) (void) hipEventDestroy(e);
    s = WtSlot();
}

template <class T>
static hipError_t wt_pinned_grow(T **p, int64_t old_n, int64_t used, int64_t new_n) {
    T *q = nullptr;
    const hipError_t e = wt_host_alloc((void **) &q, sizeof(T) * (size_t) (new_n > 0 ? new_n : 1));
    if (e != hipSuccess) return e;
    if (*p) {
        if (used > 0) memcpy(q, *p, sizeof(T) * (size_t) (used < old_n ? used : old_n));
        wt_host_free(*p);
    }
    *p = q;
    return hipSuccess;
}

static int wt_slot_grow_input(WtSlot &s, int64_t used, int64_t min_cap, bool want64) {
    if (min_cap > s.cap) {
        WT_HIP(wt_pinned_grow(&s.h_start, s.cap, used, min_cap));
        WT_HIP(wt_pinned_grow(&s.h_finish, s.cap, used, min_cap));
        WT_HIP(wt_pinned_grow(&s.h_v32, s.cap, used, min_cap));
        if (s.has64) WT_HIP(wt_pinned_grow(&s.h_v64, s.cap, used, min_cap));
        s.cap = min_cap;
    }
    if (want64 && !s.has64) {
        WT_HIP(wt_pinned_grow(&s.h_v64, 0, 0, s.cap));
        s.has64 = true;
    }
    return WTAMD_OK;
}

// Bounded wait: a kernel that does not finish is reported, never waited for forever (a hung
// kernel cannot be cancelled and every later HIP call would block behind it).
static int wt_wait_event(hipEvent_t ev, const char *what) {
    const double limit_s = getenv("WTAMD_TIMEOUT_S") ? atof(getenv("WTAMD_TIMEOUT_S")) : 120.0;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return WTAMD_OK;
        if (q != hipErrorNotReady) return wt_fail(WTAMD_ERR_HIP, std::string("hipEventQuery (") + what + "): " + hipGetErrorString(q));
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el > limit_s) {
            fprintf(stderr, "wiggletools_amd: FATAL: pipeline %s did not finish within %.0f s\n", what, limit_s);
            fflush(stderr);
            _exit(70);
        }
        if (el > 0.0005) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

static int wt_pipe_enqueue_export(wtamd_pipe *p, WtSlot &s, hipEvent_t after) {
    WT_HIP(hipStreamWaitEvent(p->s_out, after, 0));
    WT_HIP(hipEventRecord(s.e_d0, p->s_out));
    long long blocks = (s.ocap + 256 * 16 - 1) / (256 * 16);
    if (blocks > 2ll * s.ts->num_cu) blocks = 2ll * s.ts->num_cu;
    if (blocks < 1) blocks = 1;
    const bool cz = s.compressed;
    hipLaunchKernelGGL(wt_export_kernel, dim3((unsigned) blocks), dim3(256), 0, p->s_out, s.ts->d_counters, s.ts->h_counters,
                       cz ? (const unsigned long long *) s.d_cn : (const unsigned long long *) (s.ts->d_counters + WT_CTR_RUNS),
                       (long long) s.ocap, p->cfg.n_tracks, cz ? s.d_cs : s.d_os, cz ? s.d_cf : s.d_of, cz ? s.d_cv : s.d_ov, p->tile ? s.d_tile : nullptr,
                       p->tile ? s.d_ip : nullptr, s.h_os, s.h_of, s.h_ov, p->tile ? s.h_tile : nullptr, p->tile ? s.h_ip : nullptr);
    WT_HIP(hipGetLastError());
    WT_HIP(hipEventRecord(s.e_d1, p->s_out));
    return WTAMD_OK;
}

// Integrals of the slot's (uncompressed) device runs -> s.h_integ, on `st`: {sum len * value, span} over the non-NaN
// runs (statistics.c:62-120), or the Pearson moments of the 2-track tile (:414-465).  The run count is read on
// the device.
#define WT_INTEG_BLOCKS 256
static int wt_pipe_enqueue_integ(wtamd_pipe *p, WtSlot &s, hipStream_t st) {
    const size_t need = sizeof(WtMoments) * WT_INTEG_BLOCKS + sizeof(double) * 16;
    if (!s.d_integ) WT_HIP(wt_dev_alloc((void **) &s.d_integ, need));
    if (!s.h_integ) { WT_HIP(wt_host_alloc((void **) &s.h_integ, sizeof(double) * 8)); }
    const unsigned long long *n_dev = s.ts->d_counters + WT_CTR_RUNS;
    double *d_out = (double *) (s.d_integ + sizeof(WtMoments) * WT_INTEG_BLOCKS);
    if (p->tile) {
        if (p->cfg.n_tracks != 2) return wt_fail(WTAMD_ERR_ARG, "the fused Pearson integrator needs a Multiplexer of exactly two tracks");
        hipLaunchKernelGGL(wt_pearson_kernel, dim3(WT_INTEG_BLOCKS), dim3(256), 0, st, s.d_os, s.d_of, s.d_tile, s.d_ip, p->defaults[0],
                           p->defaults[1], (long long) s.ocap, (WtMoments *) s.d_integ, n_dev);
        hipLaunchKernelGGL(wt_pearson_final_kernel, dim3(1), dim3(64), 0, st, (const WtMoments *) s.d_integ, WT_INTEG_BLOCKS, d_out);
        WT_HIP(hipGetLastError());
        WT_HIP(hipMemcpyAsync(s.h_integ, d_out + 1, sizeof(double) * 6, hipMemcpyDeviceToHost, st));
    } else {
        double *part = (double *) s.d_integ;
        hipLaunchKernelGGL(wt_auc_kernel, dim3(WT_INTEG_BLOCKS), dim3(256), 0, st, s.d_os, s.d_of, s.d_ov, (long long) s.ocap, part,
                           part + WT_INTEG_BLOCKS, n_dev);
        hipLaunchKernelGGL(wt_auc_final_kernel, dim3(1), dim3(64), 0, st, part, WT_INTEG_BLOCKS, d_out);
        hipLaunchKernelGGL(wt_auc_final_kernel, dim3(1), dim3(64), 0, st, part + WT_INTEG_BLOCKS, WT_INTEG_BLOCKS, d_out + 1);
        WT_HIP(hipGetLastError());
        WT_HIP(hipMemcpyAsync(s.h_integ, d_out, sizeof(double) * 2, hipMemcpyDeviceToHost, st));
    }
    return WTAMD_OK;
}

// The batch's export has landed: read the counters; patch + export again if the difference-array
// launch left windows it could not prove exact.
static int wt_pipe_finish(wtamd_pipe *p, WtSlot &s) {
    wtamd_trackset *ts = s.ts;
    const unsigned long long *hc = ts->h_counters;
    if (hc[WT_CTR_ERROR] & WT_ERR_LOOKBACK) return wt_fail(WTAMD_ERR_INTERNAL, "look-back timed out");
    if (hc[WT_CTR_ERROR] & WT_ERR_CAPACITY) return wt_fail(WTAMD_ERR_CAPACITY, "batch emitted more runs than the slot's output capacity (max_runs)");
    const long long n_bad = (long long) hc[WT_CTR_DELTA_BAD];
    if (s.used_delta && n_bad > 0) {
        wtamd_runs runs{};
        runs.capacity = s.ocap; runs.start = s.d_os; runs.finish = s.d_of; runs.value = s.d_ov; runs.chrom_run_off = s.d_cro;
        int rc = wt_launch_patch(ts, s.delta_W, p->cfg.desc.op, p->cfg.desc.flags, p->cfg.desc.n_set0, &runs, n_bad, p->s_comp);
        if (rc != WTAMD_OK) return rc;
        if (s.compressed) {
            rc = wt_compress_async(s.d_os, s.d_of, s.d_ov, ts->d_counters + WT_CTR_RUNS, (long long) s.ocap, s.d_cscratch, s.d_cs, s.d_cf,
                                   s.d_cv, s.d_cn, p->s_comp);
            if (rc != WTAMD_OK) return wt_fail(rc, "run compression launch failed");
        }
        if (s.integrated) {
            rc = wt_pipe_enqueue_integ(p, s, p->s_comp);
            if (rc != WTAMD_OK) return rc;
            WT_HIP(hipEventRecord(s.e_patch, p->s_comp));
            rc = wt_wait_event(s.e_patch, "patched integrals");
        } else {
            WT_HIP(hipEventRecord(s.e_patch, p->s_comp));
            rc = wt_pipe_enqueue_export(p, s, s.e_patch);
            if (rc != WTAMD_OK) return rc;
            rc = wt_wait_event(s.e_d1, "patched result");
        }
        if (rc != WTAMD_OK) return rc;
        s.patched = true;
        if (n_bad * 4 > (long long) ts->stats.n_windows) p->delta_failed = true;     // this data: general kernel from now on
    }
    s.n_runs = (int64_t) hc[WT_CTR_EXPORTED];
    s.covered = (int64_t) hc[WT_CTR_BP];
    return WTAMD_OK;
}

extern "C" {

int wtamd_pipe_create(const wtamd_pipe_config *cfg, wtamd_pipe **out) {
    if (!cfg || !out || cfg->n_tracks <= 0 || !cfg->defaults || cfg->max_runs <= 0)
        return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_create: bad configuration");
    if (cfg->flags & ~0u & ~WTAMD_PIPE_COMPRESS) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_create: unknown flag");
    const bool tile = cfg->desc.op == WTAMD_OP_MULTIPLEX;
    if ((cfg->flags & WTAMD_PIPE_COMPRESS) && tile) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_create: the Multiplexer tile cannot be compressed");
    wt_warmup_join();           // (wtamd_warmup_async: the runtime's start-up, if a helper thread is at it)
    if (wtamd_device_count() <= 0) return wt_fail(WTAMD_ERR_NODEVICE, "no HIP device visible");
    wtamd_pipe *p = new wtamd_pipe();
    (void) hipGetDevice(&p->device);
    p->cfg = *cfg;
    p->defaults.assign(cfg->defaults, cfg->defaults + cfg->n_tracks);
    p->cfg.defaults = p->defaults.data();
    p->tile = tile;
    p->compress = (cfg->flags & WTAMD_PIPE_COMPRESS) != 0;
    if (getenv("WTAMD_PIPE_GATHER")) p->gather = atoi(getenv("WTAMD_PIPE_GATHER")) != 0;
    if (getenv("WTAMD_GATHER_BLOCKS") && atoi(getenv("WTAMD_GATHER_BLOCKS")) > 0) p->gather_blocks = atoi(getenv("WTAMD_GATHER_BLOCKS"));
    int ns = cfg->n_slots ? cfg->n_slots : 3;
    if (ns < 2) ns = 2;
    if (ns > 8) ns = 8;
    p->st.n_slots = ns;
    auto fail = [&](int rc) { wtamd_pipe_destroy(p); return rc; };
#define WT_PIPE_HIP(expr)                                                                                         \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return fail(wt_fail(WTAMD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_))); \
    } while (0)
    if (!tile) {
        wtamd_trackset probe;       // argument check of the descriptor (same messages as wtamd_reduce)
        probe.n_tracks = cfg->n_tracks;
        const int rc = wt_check_desc(&probe, &cfg->desc);
        if (rc != WTAMD_OK) return fail(rc);
    }
    // (Confining the PCIe-facing kernels to a few CUs with hipExtStreamCreateWithCUMask was tried: the inflate kernel
    // got slower -- fewer CUs, 21 ms against 13.5 ms per batch -- and the masked streams crashed the process in the
    // drop-in tests; tools/probes/cumask_probe.hip shows how the mask bits map to CUs on this GPU.)
    WT_PIPE_HIP(hipStreamCreateWithFlags(&p->s_copy, hipStreamNonBlocking));
    WT_PIPE_HIP(hipStreamCreateWithFlags(&p->s_comp, hipStreamNonBlocking));
    WT_PIPE_HIP(hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
    p->slots.resize((size_t) ns);
    const int64_t cap0 = cfg->max_intervals > 0 ? cfg->max_intervals : 4096;
    const int N = cfg->n_tracks;
    std::vector<int64_t> seg0((size_t) N + 1, 0);
    for (auto &s : p->slots) {
        WT_PIPE_HIP(wt_host_alloc((void **) &s.h_seg, sizeof(int64_t) * ((size_t) N + 1)));
        memset(s.h_seg, 0, sizeof(int64_t) * ((size_t) N + 1));
        int rc = wt_slot_grow_input(s, 0, cap0, false);
        if (rc != WTAMD_OK) return fail(rc);
        for (hipEvent_t *e : {&s.e_h0, &s.e_h1, &s.e_k0, &s.e_cnt, &s.e_patch, &s.e_d0, &s.e_d1}) WT_PIPE_HIP(hipEventCreate(e));
        // the slot's track set: one chromosome, device arrays bound per batch
        wtamd_tracks t;
        memset(&t, 0, sizeof(t));
        t.n_chrom = 1; t.n_tracks = N; t.seg_off = seg0.data(); t.defaults = p->defaults.data();
        s.ts = new wtamd_trackset();
        rc = wt_trackset_common(&t, s.ts);
        if (rc != WTAMD_OK) return fail(rc);
        s.ts->pipe_mode = true;
        s.ts->owns = false;
        s.ts->first_start.assign((size_t) N, 0);
        s.ts->last_finish.assign((size_t) N, 0);
        s.ts->range_lo.assign(1, 0);
        s.ts->range_hi.assign(1, INT32_MAX);
        WT_PIPE_HIP(wt_dev_alloc(&s.d_cro, sizeof(int64_t) * 2));
    }
#undef WT_PIPE_HIP
    *out = p;
    return WTAMD_OK;
}

void wtamd_pipe_destroy(wtamd_pipe *p) {
    WtDevGuard dev_guard_(p ? p->device : -1);
    if (!p) return;
    // everything still in flight must have left the buffers before they are freed
    if (p->s_copy) (void) hipStreamSynchronize(p->s_copy);
    if (p->s_comp) (void) hipStreamSynchronize(p->s_comp);
    for (int k = 0; k < 2; k++)
        if (p->s_decs[k]) (void) hipStreamSynchronize(p->s_decs[k]);
    if (p->s_out) (void) hipStreamSynchronize(p->s_out);
    for (auto &s : p->slots) wt_slot_free(s);
    if (p->s_copy) (void) hipStreamDestroy(p->s_copy);
    if (p->s_comp) (void) hipStreamDestroy(p->s_comp);
    if (p->s_out) (void) hipStreamDestroy(p->s_out);
    if (p->d_chains) (void) wt_dev_free(p->d_chains);
    for (int k = 0; k < 2; k++) {
        (void) wt_dev_free(p->d_bw_scratches[k]);
        if (p->s_decs[k]) (void) hipStreamDestroy(p->s_decs[k]);
    }
    for (void *q : p->dead_dev) (void) wt_dev_free(q);
    for (void *q : p->dead_host) wt_host_free(q);
    delete p;
}

static void wt_fill_batch(const WtSlot &s, wtamd_pipe_batch *b) {
    b->capacity = s.cap;
    b->seg_off = s.h_seg;
    b->start = s.h_start; b->finish = s.h_finish;
    b->value32 = s.h_v32;
    b->value64 = s.has64 ? s.h_v64 : nullptr;
}

int wtamd_pipe_acquire(wtamd_pipe *p, wtamd_pipe_batch *out) {
    if (!p || !out) return wt_fail(WTAMD_ERR_ARG, "NULL argument");
    if (p->acquired >= 0) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_acquire: a slot is already acquired");
    WtSlot &s = p->slots[(size_t) p->head];
    if (s.state != 0) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_acquire: every slot is in flight or unreleased");
    s.state = 1;
    p->acquired = p->head;
    wt_fill_batch(s, out);
    return WTAMD_OK;
}

int wtamd_pipe_grow(wtamd_pipe *p, int64_t used, int64_t min_capacity, int want_f64, wtamd_pipe_batch *out) {
    WtDevGuard dev_guard_(p ? p->device : -1);
    if (!p || !out || p->acquired < 0) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_grow: no acquired slot");
    WtSlot &s = p->slots[(size_t) p->acquired];
    if (used > s.cap || used < 0) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_grow: used > capacity");
    const int rc = wt_slot_grow_input(s, used, min_capacity, want_f64 != 0);
    if (rc != WTAMD_OK) return rc;
    wt_fill_batch(s, out);
    return WTAMD_OK;
}

int wtamd_pipe_put_direct(wtamd_pipe *p, int64_t at, int64_t count, const int32_t *start, const int32_t *finish,
                          const float *value) {
    WtDevGuard dev_guard_(p ? p->device : -1);
    if (!p || p->acquired < 0) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_put_direct: no acquired slot");
    if (count <= 0) return WTAMD_OK;
    if (at < 0 || !start || !finish || !value) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_put_direct: bad arguments");
    WtSlot &s = p->slots[(size_t) p->acquired];
    if (!s.direct.empty() && s.direct.back().at + s.direct.back().count > at)
        return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_put_direct: ranges must be added in ascending order");
    if (s.direct.empty()) s.direct_pinned = true;
    if (p->gather && s.direct_pinned && !(wt_is_pinned(start) && wt_is_pinned(finish) && wt_is_pinned(value))) s.direct_pinned = false;
    s.direct.push_back({at, count, start, finish, value});
    return WTAMD_OK;
}

int wtamd_pipe_cancel(wtamd_pipe *p) {
    if (!p || p->acquired < 0) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_cancel: no acquired slot");
    p->slots[(size_t) p->acquired].direct.clear();
    p->slots[(size_t) p->acquired].bw_res_bytes = p->slots[(size_t) p->acquired].bw_res_secs = -1;
    p->slots[(size_t) p->acquired].state = 0;
    p->acquired = -1;
    return WTAMD_OK;
}

// bw_tracks != NULL: the batch came as BigWig file bytes (wtamd_pipe_submit_bw) -- the run lists are produced
// on the device, the host only knows upper bounds of their sizes and extents.
static int wt_pipe_submit_impl(wtamd_pipe *p, int value_is_f64, int32_t range_lo, int32_t range_hi,
                               const wtamd_bw_track *bw_tracks = nullptr, int64_t bw_bytes = 0, int64_t bw_secs = 0, WtSlot *redo = nullptr);

int wtamd_pipe_submit(wtamd_pipe *p, int value_is_f64, int32_t range_lo, int32_t range_hi) {
    WtDevGuard dev_guard_(p ? p->device : -1);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = wt_pipe_submit_impl(p, value_is_f64, range_lo, range_hi);
    if (p) p->st.host_submit_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

static int64_t wt_align256(int64_t x) { return (x + 255) & ~(int64_t) 255; }

unsigned wtamd_pipe_bw_error(const wtamd_pipe *p) { return p ? p->last_bw_err : 0u; }

int64_t wtamd_pipe_bw_fill_sections(const wtamd_pipe *p) {
    if (!p || p->slots.empty() || !p->slots[0].ts) return 0;
    return (int64_t) wt_bw_fill_sections(p->slots[0].ts->num_cu);
}

int wtamd_pipe_bw_reserve(wtamd_pipe *p, int64_t n_bytes, int64_t n_sections, uint8_t **bytes, wtamd_bw_section **sections) {
    WtDevGuard dev_guard_(p ? p->device : -1);
    if (!p || p->acquired < 0) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_bw_reserve: no acquired slot");
    if (n_bytes < 0 || n_sections < 0 || !bytes || !sections) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_bw_reserve: bad arguments");
    WtSlot &s = p->slots[(size_t) p->acquired];
    const int N = p->cfg.n_tracks;
    s.bw_off_sec = wt_align256((int64_t) sizeof(wtamd_bw_track) * N);
    s.bw_off_bytes = s.bw_off_sec + wt_align256((int64_t) sizeof(wtamd_bw_section) * n_sections);
    const int64_t need = s.bw_off_bytes + wt_align256(n_bytes + 64);
    if (s.h_bw_cap < need) {
        if (s.h_bw) p->dead_host.push_back(s.h_bw);
        s.h_bw = nullptr; s.h_bw_cap = 0;
        const int64_t c = need + need / 4;
        WT_HIP(wt_host_alloc((void **) &s.h_bw, (size_t) c));
        s.h_bw_cap = c;
    }
    s.bw_res_bytes = n_bytes; s.bw_res_secs = n_sections;
    *bytes = s.h_bw + s.bw_off_bytes;
    *sections = (wtamd_bw_section *) (s.h_bw + s.bw_off_sec);
    return WTAMD_OK;
}

int wtamd_pipe_submit_bw(wtamd_pipe *p, int64_t n_bytes, int64_t n_sections, const wtamd_bw_track *tracks,
                         int32_t range_lo, int32_t range_hi) {
    WtDevGuard dev_guard_(p ? p->device : -1);
    const auto t0 = std::chrono::steady_clock::now();
    if (!p || p->acquired < 0) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit_bw: no acquired slot");
    if (!tracks) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit_bw: tracks == NULL");
    const int rc = wt_pipe_submit_impl(p, 0, range_lo, range_hi, tracks, n_bytes, n_sections);
    p->st.host_submit_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

// Host-side bounds of a batch that arrives as file bytes: seg_off[] (piece counts), per-track extents.
static int wt_pipe_bw_bounds(wtamd_pipe *p, WtSlot &s, const wtamd_bw_track *tk, int64_t n_bytes, int64_t n_secs, int64_t *plain_stride) {
    const int N = p->cfg.n_tracks;
    if (s.bw_res_bytes < 0 || n_bytes > s.bw_res_bytes || n_secs > s.bw_res_secs || n_bytes < 0 || n_secs < 0)
        return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit_bw: more bytes / sections than reserved");
    if (p->tile) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit_bw: not available for the Multiplexer tile");
    const wtamd_bw_section *sec = (const wtamd_bw_section *) (s.h_bw + s.bw_off_sec);
    int64_t at = 0, next_sec = 0, stride = 64;
    for (int i = 0; i < N; i++) {
        const wtamd_bw_track &t = tk[i];
        if (t.first_section != next_sec || t.n_sections < 0 || (int64_t) t.first_section + t.n_sections > n_secs)
            return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit_bw: sections must be listed track by track");
        s.h_seg[i] = at;
        int32_t fs = 0, lf = 0;
        for (int64_t q = t.first_section; q < (int64_t) t.first_section + t.n_sections; q++) {
            const wtamd_bw_section &c = sec[q];
            if (c.track != i || c.comp_off < 0 || c.comp_off + (int64_t) c.comp_size > n_bytes || c.leaf_end < c.leaf_start)
                return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit_bw: bad section entry");
            if (q > t.first_section && c.leaf_start < sec[q - 1].leaf_end)
                return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit_bw: a track's sections must be sorted and disjoint");
            if (!t.compressed && c.comp_size > t.plain_bytes) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit_bw: raw section larger than plain_bytes");
            at += wt_bw_section_bound(t.plain_bytes, c.leaf_start, c.leaf_end, t.box);
        }
        if (t.n_sections > 0) {
            const int64_t a = (int64_t) sec[t.first_section].leaf_start + 1, b = (int64_t) sec[t.first_section + t.n_sections - 1].leaf_end + 1;
            fs = (int32_t) std::max<int64_t>(a, t.clip_lo);
            lf = (int32_t) std::min<int64_t>(std::min<int64_t>(b, t.clip_hi), INT32_MAX);
            if (lf <= fs) lf = fs + 1;
            if ((int64_t) t.plain_bytes + 16 > stride) stride = (int64_t) t.plain_bytes + 16;
        }
        s.ts->first_start[(size_t) i] = fs;
        s.ts->last_finish[(size_t) i] = lf;
        next_sec += t.n_sections;
    }
    if (next_sec != n_secs) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit_bw: section count does not match the tracks");
    s.h_seg[N] = at;
    *plain_stride = (stride + 15) & ~(int64_t) 15;
    return WTAMD_OK;
}

// redo != NULL: the file-byte batch of that (submitted) slot once more, its run lists at the size of the host's bound
// -- everything the first submit staged (tables, file bytes, seg_off bounds) is still in place.
static int wt_pipe_submit_impl(wtamd_pipe *p, int value_is_f64, int32_t range_lo, int32_t range_hi,
                               const wtamd_bw_track *bw_tracks, int64_t bw_bytes, int64_t bw_secs, WtSlot *redo) {
    if (!p || (!redo && p->acquired < 0)) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit: no acquired slot");
    WtSlot &s = redo ? *redo : p->slots[(size_t) p->acquired];
    const int N = p->cfg.n_tracks;
    const bool bw = bw_tracks != nullptr;
    int64_t bw_stride = redo ? s.bw_stride : 0;
    if (bw && !redo) {
        if (!s.direct.empty()) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit_bw: the slot holds direct ranges");
        const int rcb = wt_pipe_bw_bounds(p, s, bw_tracks, bw_bytes, bw_secs, &bw_stride);
        if (rcb != WTAMD_OK) return rcb;
    }
    const int64_t n = s.h_seg[N];
    if (s.h_seg[0] != 0 || n < 0) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit: bad seg_off");
    // staged ranges = [0, n) minus the direct ranges; they must lie inside the staging arrays
    if (!bw) {
        int64_t staged_end = 0, pos = 0;
        for (const auto &d : s.direct) {
            if (d.at + d.count > n) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit: a direct range lies beyond seg_off[n_tracks]");
            if (d.at > pos) staged_end = d.at;
            pos = d.at + d.count;
        }
        if (pos < n) staged_end = n;
        if (staged_end > s.cap) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit: staged intervals beyond the staging capacity");
        if (value_is_f64 && !s.direct.empty()) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit: direct ranges are float32");
    }
    for (int i = 0; i < N; i++)
        if (s.h_seg[i + 1] < s.h_seg[i]) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit: seg_off not monotone");
    if (value_is_f64 && !s.has64) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_submit: float64 values were never staged");
    wtamd_trackset *ts = s.ts;
    const bool f64 = value_is_f64 != 0;
    // device twin of the staging (grow-only)
    int64_t need_in = n > 0 ? n : 1;
    if (bw && !redo && p->bw_density >= 0.0 && p->bw_density < 1.0) {
        const int64_t by_density = (int64_t) ((double) n * p->bw_density * 1.125) + 65536;
        if (by_density < need_in) need_in = by_density;
    }
    if (s.dcap < need_in || (f64 && !s.d_has64)) {
        // an eighth of slack, not a doubling: the batches of a run settle on one size and wobble by a fraction of a
        // percent around it (63 220 sections, then 63 502), and "twice the old capacity" answered the first batch that
        // was a hair larger with three more arrays of 1.5 GB per slot -- 27.6 of the 37.6 GB a pipe of 100 BigWig tracks
        // held, and most of the time its first run spent in hipMalloc (round 3, WTAMD_TRACE_POOL=1).  The ramp at the
        // start of a run grows by factors anyway.
        int64_t c = need_in + need_in / 8;
        if (c < s.cap) c = s.cap;
        for (void *q : {(void *) s.d_start, (void *) s.d_finish, s.d_value}) if (q) p->dead_dev.push_back(q);
        s.d_start = s.d_finish = nullptr; s.d_value = nullptr; s.dcap = 0;
        const bool w64 = f64 || s.d_has64 || s.has64;
        WT_HIP(wt_dev_alloc(&s.d_start, sizeof(int32_t) * c));
        WT_HIP(wt_dev_alloc(&s.d_finish, sizeof(int32_t) * c));
        WT_HIP(wt_dev_alloc(&s.d_value, (w64 ? 8 : 4) * (size_t) c));
        s.dcap = c; s.d_has64 = w64;
    }
    const bool mapped = p->d_chains != nullptr;
    if (mapped && (s.mcap < s.dcap || (p->map_drops && !s.m_has_coords))) {
        for (void *q : {(void *) s.d_mstart, (void *) s.d_mfinish, (void *) s.d_mvalue, (void *) s.d_mscratch}) if (q) p->dead_dev.push_back(q);
        s.d_mstart = s.d_mfinish = nullptr; s.d_mvalue = nullptr; s.d_mscratch = nullptr; s.mcap = 0; s.m_has_coords = false;
        WT_HIP(wt_dev_alloc(&s.d_mvalue, sizeof(double) * (size_t) s.dcap));
        if (p->map_drops) {
            WT_HIP(wt_dev_alloc(&s.d_mstart, sizeof(int32_t) * (size_t) s.dcap));
            WT_HIP(wt_dev_alloc(&s.d_mfinish, sizeof(int32_t) * (size_t) s.dcap));
            WT_HIP(wt_dev_alloc(&s.d_mscratch, sizeof(unsigned long long) * (size_t) wt_map_scratch_words((long long) s.dcap)));
            if (!s.d_mseg) WT_HIP(wt_dev_alloc(&s.d_mseg, sizeof(int64_t) * ((size_t) N + 1)));
            s.m_has_coords = true;
        }
        s.mcap = s.dcap;
    }
    // output (grow-only, bounded by max_runs): a run is at least 1 bp and starts at an interval edge
    int64_t need_out = 2 * n;
    const int64_t span = (int64_t) range_hi - (int64_t) range_lo;
    if (range_hi != INT32_MAX && span < need_out) need_out = span > 0 ? span : 0;
    if (need_out > p->cfg.max_runs) need_out = p->cfg.max_runs;
    if (need_out < 1) need_out = 1;
    if (s.ocap < need_out) {
        int64_t c = need_out + need_out / 8;        // (slack, not a doubling: see the device twins above)
        if (c > p->cfg.max_runs) c = p->cfg.max_runs;
        if (c < need_out) c = need_out;
        for (void *q : {(void *) s.d_os, (void *) s.d_of, (void *) s.d_ov, (void *) s.d_tile, (void *) s.d_ip}) if (q) p->dead_dev.push_back(q);
        s.d_os = s.d_of = nullptr; s.d_ov = s.d_tile = nullptr; s.d_ip = nullptr;
        for (void *q : {(void *) s.h_os, (void *) s.h_of, (void *) s.h_ov, (void *) s.h_tile, (void *) s.h_ip}) if (q) p->dead_host.push_back(q);
        s.h_os = s.h_of = nullptr; s.h_ov = s.h_tile = nullptr; s.h_ip = nullptr; s.ocap = 0;
        WT_HIP(wt_dev_alloc(&s.d_os, sizeof(int32_t) * c));
        WT_HIP(wt_dev_alloc(&s.d_of, sizeof(int32_t) * c));
        WT_HIP(wt_dev_alloc(&s.d_ov, sizeof(double) * c));
        WT_HIP(wt_host_alloc((void **) &s.h_os, sizeof(int32_t) * c));
        WT_HIP(wt_host_alloc((void **) &s.h_of, sizeof(int32_t) * c));
        WT_HIP(wt_host_alloc((void **) &s.h_ov, sizeof(double) * c));
        for (void *q : {(void *) s.d_cs, (void *) s.d_cf, (void *) s.d_cv, (void *) s.d_cscratch}) if (q) p->dead_dev.push_back(q);
        s.d_cs = s.d_cf = nullptr; s.d_cv = nullptr; s.d_cscratch = nullptr;
        if (p->tile) {
            WT_HIP(wt_dev_alloc(&s.d_tile, sizeof(double) * c * N));
            WT_HIP(wt_dev_alloc(&s.d_ip, sizeof(uint8_t) * c * N));
            WT_HIP(wt_host_alloc((void **) &s.h_tile, sizeof(double) * c * N));
            WT_HIP(wt_host_alloc((void **) &s.h_ip, sizeof(uint8_t) * c * N));
        }
        s.ocap = c;
    }

    // rebind the slot's track set to this batch
    ts->n_intervals = n;
    ts->seg_off.assign(s.h_seg, s.h_seg + N + 1);
    if (!bw) {
        // entry g of the batch: in the staging arrays or in a direct range
        auto start_at = [&](int64_t g) -> int32_t {
            size_t lo = 0, hi = s.direct.size();
            while (lo < hi) { const size_t m = (lo + hi) / 2; if (s.direct[m].at + s.direct[m].count <= g) lo = m + 1; else hi = m; }
            return (lo < s.direct.size() && s.direct[lo].at <= g) ? s.direct[lo].start[g - s.direct[lo].at] : s.h_start[g];
        };
        auto finish_at = [&](int64_t g) -> int32_t {
            size_t lo = 0, hi = s.direct.size();
            while (lo < hi) { const size_t m = (lo + hi) / 2; if (s.direct[m].at + s.direct[m].count <= g) lo = m + 1; else hi = m; }
            return (lo < s.direct.size() && s.direct[lo].at <= g) ? s.direct[lo].finish[g - s.direct[lo].at] : s.h_finish[g];
        };
        for (int i = 0; i < N; i++) {
            const int64_t a = s.h_seg[i], b = s.h_seg[i + 1];
            ts->first_start[(size_t) i] = b > a ? start_at(a) : 0;
            ts->last_finish[(size_t) i] = b > a ? finish_at(b - 1) : 0;
        }
    }
    ts->range_lo[0] = range_lo;
    ts->range_hi[0] = range_hi;
    const bool map_f32 = mapped && p->map_f32 && !f64;
    ts->value_f64 = f64 || (mapped && !map_f32);
    ts->scratch_f32 = !ts->value_f64 && wt_defaults_fit_f32(ts->defaults.data(), N);
    // mapped batches: the kernels read the operator chains' output (the host-side seg_off[] / extents stay those
    // of the raw lists: upper bounds, which is all the planning needs)
    const bool compacted = mapped && p->map_drops;
    ts->d_start = compacted ? s.d_mstart : s.d_start;
    ts->d_finish = compacted ? s.d_mfinish : s.d_finish;
    ts->d_value = mapped ? (void *) s.d_mvalue : s.d_value;
    for (int q = 0; q < 3; q++) { ts->delta_failed_[q] = p->delta_failed; ts->delta_verified_[q] = false; ts->delta_n_bad_[q] = 0; }
    for (auto &kv : ts->windows) { kv.second.tab_valid = false; kv.second.indexed = false; }
    int rc = wt_check_extents(ts);
    if (rc != WTAMD_OK) return rc;

    // copy stream: pinned staging -> HBM
    WT_HIP(hipEventRecord(s.e_h0, p->s_copy));
    s.bw = bw;
    if (bw) {
        // file bytes + tables: one copy kernel; inflate / count / scan / scatter on the decode stream write the
        // run lists and the device-side seg_off[] (the authority downstream: the host's are upper bounds)
        // (on the COMPUTE stream: HIP maps its streams onto 4 hardware queues, and a fourth stream of the pipe landed
        // on the copy stream's queue -- the next batch's copy then waited behind this batch's inflate kernel)
        {
            if (p->n_decs == 0) {
                const char *e = getenv("WTAMD_BW_DECODE_STREAMS");
                p->n_decs = (e && atoi(e) == 2) ? 2 : -1;
                for (int k = 0; k < 2 && p->n_decs == 2; k++) WT_HIP(hipStreamCreateWithFlags(&p->s_decs[k], hipStreamNonBlocking));
            }
            const int k = p->n_decs == 2 ? (int) (p->bw_batches & 1) : 0;
            p->bw_batches++;
            p->s_dec = p->n_decs == 2 ? p->s_decs[k] : p->s_comp;
            // the slot's run lists and outputs were last touched by the kernels of its previous batch on the compute
            // stream (long collected); the scratch is per decode stream
            p->d_bw_scratch = p->d_bw_scratches[k]; p->bw_scratch_cap = p->bw_scratch_caps[k];
            s.bw_dec = k;
        }
        if (!s.e_bwc) { WT_HIP(hipEventCreate(&s.e_bwc)); WT_HIP(hipEventCreate(&s.e_bw0)); WT_HIP(hipEventCreate(&s.e_bw1)); }
        if (!s.h_bw_status) WT_HIP(wt_host_alloc((void **) &s.h_bw_status, 64));
        const int64_t total = s.bw_off_bytes + wt_align256(bw_bytes + 64);
        if (s.d_bw_cap < total) {
            if (s.d_bw) p->dead_dev.push_back(s.d_bw);
            s.d_bw = nullptr; s.d_bw_cap = 0;
            const int64_t c = total + total / 4;
            WT_HIP(wt_dev_alloc((void **) &s.d_bw, (size_t) c));
            s.d_bw_cap = c;
        }
        const int64_t need_scr = wt_bw_scratch_bytes(bw_secs, bw_stride);
        if (p->bw_scratch_cap < need_scr) {
            if (p->d_bw_scratch) p->dead_dev.push_back(p->d_bw_scratch);
            p->d_bw_scratch = nullptr; p->bw_scratch_cap = 0;
            const int64_t c = need_scr + need_scr / 4;
            WT_HIP(wt_dev_alloc(&p->d_bw_scratch, (size_t) c));
            p->bw_scratch_cap = c;
            p->d_bw_scratches[s.bw_dec] = p->d_bw_scratch; p->bw_scratch_caps[s.bw_dec] = c;
        }
        if (!redo) memcpy(s.h_bw, bw_tracks, sizeof(wtamd_bw_track) * (size_t) N);
        s.h_bw_status[0] = ~0ull; s.h_bw_status[1] = 0;
        WT_HIP(hipEventRecord(s.e_bw0, p->s_dec));
        rc = wt_bw_decode_async(s.h_bw, s.d_bw, total, s.d_bw + s.bw_off_bytes, s.d_bw + s.bw_off_sec, s.d_bw, N, bw_secs, bw_stride, p->d_bw_scratch,
                                (long long) s.dcap, s.d_start, s.d_finish, (float *) s.d_value, compacted ? s.d_mseg : ts->d_seg_off,
                                s.h_bw_status, p->gather_blocks, p->s_copy, s.e_bwc, p->s_dec);
        if (rc != WTAMD_OK) return rc;
        WT_HIP(hipEventRecord(s.e_bw1, p->s_dec));
        s.bw_secs = bw_secs; s.bw_bytes = bw_bytes; s.bw_stride = bw_stride; s.bw_bound = n;
        s.bw_res_bytes = s.bw_res_secs = -1;
    } else {
    WT_HIP(hipMemcpyAsync(compacted ? s.d_mseg : ts->d_seg_off, s.h_seg, sizeof(int64_t) * ((size_t) N + 1), hipMemcpyHostToDevice, p->s_copy));
    }
    if (bw) {
    } else if (n > 0 && p->gather && !f64 && !s.direct.empty() && s.direct_pinned && 2 * (int64_t) s.direct.size() + 1 <= WT_GATHER_MAX_SEGS) {
        // one table, one small copy, one kernel for the whole batch
        const int64_t max_segs = 2 * (int64_t) s.direct.size() + 1;
        if (s.seg_cap < max_segs) {
                    if (s.h_segs) wt_host_free(s.h_segs);
            s.h_segs = nullptr; s.seg_cap = 0;
            const int64_t c = 2 * max_segs;
            WT_HIP(wt_host_alloc((void **) &s.h_segs, sizeof(WtGatherSeg) * c));
            s.seg_cap = c;
        }
        int ns = 0;
        long long chunks = 0;
        auto add = [&](const int32_t *ps, const int32_t *pf, const float *pv, int64_t dst, int64_t count) {
            if (count <= 0) return;
            s.h_segs[ns++] = WtGatherSeg{ps, pf, pv, dst, count, chunks};
            chunks += (count + WT_GATHER_CHUNK - 1) / WT_GATHER_CHUNK;
        };
        int64_t pos = 0;
        for (const auto &d : s.direct) {
            add(s.h_start + pos, s.h_finish + pos, s.h_v32 + pos, pos, d.at - pos);       // staged gap before it
            add(d.start, d.finish, d.value, d.at, d.count);
            pos = d.at + d.count;
        }
        add(s.h_start + pos, s.h_finish + pos, s.h_v32 + pos, pos, n - pos);
        // Few blocks on purpose: every block keeps 48 KB of reads in flight, and whatever is queued
        // on the link delays every OTHER host read by queue / bandwidth -- kernel arguments and the
        // small tables of the compute kernels of the previous batch included (measured with 768
        // blocks = 37 MB in flight: those kernels started ~1.7 ms late, right at the gather's tail).
        // The bandwidth-delay product of the link is well below 1 MB.
        long long grid = p->gather_blocks;
        if (grid > chunks) grid = chunks;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL(wt_gather_kernel, dim3((unsigned) grid), dim3(256), 0, p->s_copy, s.h_segs, ns, chunks,
                           s.d_start, s.d_finish, (float *) s.d_value);
        WT_HIP(hipGetLastError());
    } else if (n > 0) {
        auto staged = [&](int64_t a, int64_t b) -> int {        // staging [a, b) -> HBM
            if (b <= a) return WTAMD_OK;
            WT_HIP(hipMemcpyAsync(s.d_start + a, s.h_start + a, sizeof(int32_t) * (b - a), hipMemcpyHostToDevice, p->s_copy));
            WT_HIP(hipMemcpyAsync(s.d_finish + a, s.h_finish + a, sizeof(int32_t) * (b - a), hipMemcpyHostToDevice, p->s_copy));
            if (f64) WT_HIP(hipMemcpyAsync((double *) s.d_value + a, s.h_v64 + a, sizeof(double) * (b - a), hipMemcpyHostToDevice, p->s_copy));
            else WT_HIP(hipMemcpyAsync((float *) s.d_value + a, s.h_v32 + a, sizeof(float) * (b - a), hipMemcpyHostToDevice, p->s_copy));
            return WTAMD_OK;
        };
        int64_t pos = 0;
        for (const auto &d : s.direct) {                        // the caller's arrays -> HBM, no staging copy
            rc = staged(pos, d.at);
            if (rc != WTAMD_OK) return rc;
            WT_HIP(hipMemcpyAsync(s.d_start + d.at, d.start, sizeof(int32_t) * d.count, hipMemcpyHostToDevice, p->s_copy));
            WT_HIP(hipMemcpyAsync(s.d_finish + d.at, d.finish, sizeof(int32_t) * d.count, hipMemcpyHostToDevice, p->s_copy));
            WT_HIP(hipMemcpyAsync((float *) s.d_value + d.at, d.value, sizeof(float) * d.count, hipMemcpyHostToDevice, p->s_copy));
            pos = d.at + d.count;
        }
        rc = staged(pos, n);
        if (rc != WTAMD_OK) return rc;
    }
    s.direct.clear();
    WT_HIP(hipEventRecord(s.e_h1, bw ? p->s_dec : p->s_copy));    // (file bytes: the run lists exist once the decode stream is through)
    p->st.h2d_bytes += bw ? s.bw_off_bytes + bw_bytes : (int64_t) sizeof(int64_t) * (N + 1) + n * (f64 ? 16 : 12);

    // compute stream: window index + fused multiplex / reduce, then the counters travel back
    WT_HIP(hipStreamWaitEvent(p->s_comp, s.e_h1, 0));
    WT_HIP(hipEventRecord(s.e_k0, p->s_comp));
    if (mapped) {
        rc = wt_map_chain_async(p->d_chains, N, p->map_drops, compacted ? s.d_mseg : ts->d_seg_off, (long long) n, s.d_start, s.d_finish,
                                s.d_value, f64, s.d_mscratch, s.d_mstart, s.d_mfinish, s.d_mvalue, ts->d_seg_off, p->s_comp, map_f32);
        if (rc != WTAMD_OK) return wt_fail(rc, "operator chain launch failed");
    }
    wtamd_runs runs{};
    runs.capacity = s.ocap; runs.start = s.d_os; runs.finish = s.d_of; runs.value = s.d_ov; runs.chrom_run_off = s.d_cro;
    const int op = p->cfg.desc.op;
    WtPlan plan;
    std::string err;
    s.used_delta = !p->tile && wt_wants_delta(ts, op);
    s.patched = false;
    if (s.used_delta) {
        wt_make_delta_plan_for(plan, N, op);
        s.delta_W = plan.W;
    } else if (!wt_pick_plan(ts, op, p->cfg.desc.n_set0, plan, err, p->s_comp)) {
        return wt_fail(WTAMD_ERR_ARG, err);
    }
    rc = wt_reduce_plan(ts, plan, op, p->cfg.desc.flags, p->cfg.desc.n_set0, &runs, p->tile ? s.d_tile : nullptr,
                        p->tile ? s.d_ip : nullptr, nullptr, p->s_comp);
    if (rc != WTAMD_OK) return rc;
    s.integrated = p->integrate;
    s.compressed = p->compress && !s.integrated;
    if (s.compressed) {
        if (!s.d_cs) {          // (grow-only, with the output buffers)
            WT_HIP(wt_dev_alloc(&s.d_cs, sizeof(int32_t) * s.ocap));
            WT_HIP(wt_dev_alloc(&s.d_cf, sizeof(int32_t) * s.ocap));
            WT_HIP(wt_dev_alloc(&s.d_cv, sizeof(double) * s.ocap));
            WT_HIP(wt_dev_alloc(&s.d_cscratch, sizeof(unsigned long long) * (size_t) wt_compress_scratch_words((long long) s.ocap)));
        }
        if (!s.d_cn) WT_HIP(wt_dev_alloc(&s.d_cn, sizeof(unsigned long long)));
        rc = wt_compress_async(s.d_os, s.d_of, s.d_ov, ts->d_counters + WT_CTR_RUNS, (long long) s.ocap, s.d_cscratch, s.d_cs, s.d_cf,
                               s.d_cv, s.d_cn, p->s_comp);
        if (rc != WTAMD_OK) return wt_fail(rc, "run compression launch failed");
    }
    // File-byte batches: the runs go home through the COPY ENGINE, not the export kernel.  Next to a kernel whose
    // wavefronts wait on the PCIe link the per-lane inflate kernel of the following batch (a serial, latency-bound
    // lane per stream) took 13.5 ms instead of 9.5; the copy engine costs no CU anything.  It needs the run count on
    // the host: the counters travel first (128 bytes), the runs are requested when the batch is collected.
    static const bool sdma_out = !(getenv("WTAMD_BW_EXPORT") && !strcmp(getenv("WTAMD_BW_EXPORT"), "kernel"));
    s.export_pending = bw && sdma_out && !p->tile && !s.integrated;
    if (s.integrated) {
        // fused integrator: two (six) doubles and the counters go home, the runs stay
        rc = wt_pipe_enqueue_integ(p, s, p->s_comp);
        if (rc != WTAMD_OK) return rc;
        WT_HIP(hipMemcpyAsync(ts->h_counters, ts->d_counters, sizeof(unsigned long long) * WT_CTR_N, hipMemcpyDeviceToHost, p->s_comp));
        WT_HIP(hipEventRecord(s.e_cnt, p->s_comp));
    } else if (s.export_pending) {
        WT_HIP(hipMemcpyAsync(ts->h_counters, ts->d_counters, sizeof(unsigned long long) * WT_CTR_N, hipMemcpyDeviceToHost, p->s_comp));
        if (s.compressed)
            WT_HIP(hipMemcpyAsync(ts->h_counters + WT_CTR_EXPORTED, s.d_cn, sizeof(unsigned long long), hipMemcpyDeviceToHost, p->s_comp));
        WT_HIP(hipEventRecord(s.e_cnt, p->s_comp));
    } else {
        WT_HIP(hipEventRecord(s.e_cnt, p->s_comp));
        rc = wt_pipe_enqueue_export(p, s, s.e_cnt);
        if (rc != WTAMD_OK) return rc;
    }

    s.n_int = n; s.f64 = f64; s.err = WTAMD_OK;
    if (redo) return WTAMD_OK;
    s.state = 2;
    p->acquired = -1;
    p->head = (p->head + 1) % (int) p->slots.size();
    p->in_flight++;
    p->st.batches++;
    if (!bw) p->st.intervals += n;      // (file-byte batches: counted when collected, the device knows)
    if (s.used_delta) p->st.delta_batches++;
    return WTAMD_OK;
}

// Waits for the submitted batch of slot s (and, for file-byte batches whose runs travel by copy engine, asks for them
// once their count is known).
static int wt_pipe_wait_slot(wtamd_pipe *p, WtSlot &s) {
    int rc = WTAMD_OK;
    if (s.integrated) {
        rc = wt_wait_event(s.e_cnt, "batch kernels");
        if (rc == WTAMD_OK) s.ts->h_counters[WT_CTR_EXPORTED] = s.ts->h_counters[WT_CTR_RUNS];
    } else if (s.export_pending) {
        s.export_pending = false;
        rc = wt_wait_event(s.e_cnt, "batch kernels");
        if (rc == WTAMD_OK) {
            unsigned long long *hc = s.ts->h_counters;
            if (!s.compressed) hc[WT_CTR_EXPORTED] = hc[WT_CTR_RUNS];
            if ((int64_t) hc[WT_CTR_EXPORTED] > s.ocap) hc[WT_CTR_EXPORTED] = (unsigned long long) s.ocap;
            const size_t nr = (size_t) hc[WT_CTR_EXPORTED];
            const bool cz = s.compressed;
            hipError_t e = hipEventRecord(s.e_d0, p->s_out);
            if (e == hipSuccess && nr > 0) {
                e = hipMemcpyAsync(s.h_os, cz ? s.d_cs : s.d_os, sizeof(int32_t) * nr, hipMemcpyDeviceToHost, p->s_out);
                if (e == hipSuccess) e = hipMemcpyAsync(s.h_of, cz ? s.d_cf : s.d_of, sizeof(int32_t) * nr, hipMemcpyDeviceToHost, p->s_out);
                if (e == hipSuccess) e = hipMemcpyAsync(s.h_ov, cz ? s.d_cv : s.d_ov, sizeof(double) * nr, hipMemcpyDeviceToHost, p->s_out);
            }
            if (e == hipSuccess) e = hipEventRecord(s.e_d1, p->s_out);
            if (e != hipSuccess) rc = wt_fail(WTAMD_ERR_HIP, std::string("copy-engine export: ") + hipGetErrorString(e));
        }
    }
    if (rc == WTAMD_OK && !s.integrated) rc = wt_wait_event(s.e_d1, "batch");
    return rc;
}

int wtamd_pipe_collect(wtamd_pipe *p, wtamd_pipe_result *out) {
    WtDevGuard dev_guard_(p ? p->device : -1);
    if (!p || !out) return wt_fail(WTAMD_ERR_ARG, "NULL argument");
    if (p->in_flight <= 0) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_collect: nothing in flight");
    if (p->held) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_collect: the previous result was not released");
    WtSlot &s = p->slots[(size_t) p->tail];
    if (s.state != 2) return wt_fail(WTAMD_ERR_INTERNAL, "wtamd_pipe_collect: slot order corrupted");
    const auto t_wait0 = std::chrono::steady_clock::now();
    int rc = wt_pipe_wait_slot(p, s);
    if (rc == WTAMD_OK && s.bw && s.h_bw_status[0] == WT_BW_ERR_CAPACITY && s.dcap < s.bw_bound) {
        // more intervals than the run lists sized by density hold (the device wrote nothing): once more, at the bound
        p->bw_density = 2.0;
        p->bw_redone++;
        rc = wt_pipe_submit_impl(p, 0, s.ts->range_lo[0], s.ts->range_hi[0], (const wtamd_bw_track *) s.h_bw, s.bw_bytes, s.bw_secs, &s);
        if (rc == WTAMD_OK) rc = wt_pipe_wait_slot(p, s);
    }
    s.state = 3;
    p->in_flight--;
    p->held = 1;
    if (rc != WTAMD_OK) return rc;
    p->last_bw_err = 0;
    if (s.bw) {
        const unsigned long long e = s.h_bw_status[0];
        if (e) {
            p->last_bw_err = e == ~0ull ? ~0u : (unsigned) e;
            std::string why = "BigWig sections could not be decoded on the device:";
            if (e == ~0ull) why += " decode kernels did not report";
            else {
                if (e & WT_BW_ERR_INFLATE) why += " corrupt zlib stream;";
                if (e & WT_BW_ERR_SECTION) why += " malformed section;";
                if (e & WT_BW_ERR_EXTENT) why += " items outside their index leaf / out of order (WTAMD_BW_DEVICE=0 selects the host decoder);";
                if (e & WT_BW_ERR_COORD) why += " coordinate above the supported maximum;";
                if (e & WT_BW_ERR_CAPACITY) why += " more intervals than the host's bound;";
            }
            return wt_fail(WTAMD_ERR_INTERNAL, why);
        }
        s.n_int = (int64_t) s.h_bw_status[1];
        if (s.bw_bound > 0 && p->bw_density < 1.0) {
            const double d = (double) s.n_int / (double) s.bw_bound;
            if (d > p->bw_density) p->bw_density = d;
        }
        p->st.intervals += s.n_int;
        p->st.bw_sections += s.bw_secs;
        float msb = 0;
        if (hipEventElapsedTime(&msb, s.e_bw0, s.e_bw1) == hipSuccess) p->st.bw_decode_ms += msb;
    }
    rc = wt_pipe_finish(p, s);
    if (rc != WTAMD_OK) return rc;
    p->st.host_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_wait0).count();
    float ms = 0;
    if (hipEventElapsedTime(&ms, s.e_h0, s.e_h1) == hipSuccess) p->st.h2d_ms += ms;
    if (hipEventElapsedTime(&ms, s.e_k0, s.e_cnt) == hipSuccess) p->st.kernel_ms += ms;
    if (!s.integrated && hipEventElapsedTime(&ms, s.e_d0, s.e_d1) == hipSuccess) p->st.d2h_ms += ms;
    p->st.runs += s.n_runs;
    p->st.covered_bp += s.covered;
    p->st.d2h_bytes += s.integrated ? (int64_t) (sizeof(unsigned long long) * WT_CTR_N + 48) : s.n_runs * (16 + (p->tile ? 9 * (int64_t) p->cfg.n_tracks : 0));
    out->n_runs = s.n_runs;
    out->integ_valid = s.integrated ? 1 : 0;
    out->reserved = 0;
    for (int k = 0; k < 6; k++) out->integ[k] = s.integrated ? s.h_integ[k] : 0.0;
    if (s.integrated && !p->tile) { out->integ[2] = out->integ[3] = out->integ[4] = out->integ[5] = 0.0; }
    out->start = s.integrated ? nullptr : s.h_os; out->finish = s.integrated ? nullptr : s.h_of; out->value = s.integrated ? nullptr : s.h_ov;
    out->tile = (p->tile && !s.integrated) ? s.h_tile : nullptr;
    out->inplay = (p->tile && !s.integrated) ? s.h_ip : nullptr;
    out->covered_bp = s.covered;
    out->n_intervals = s.n_int;
    return WTAMD_OK;
}

int wtamd_pipe_release(wtamd_pipe *p) {
    if (!p || !p->held) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_release: nothing to release");
    p->slots[(size_t) p->tail].state = 0;
    p->held = 0;
    p->tail = (p->tail + 1) % (int) p->slots.size();
    return WTAMD_OK;
}

int wtamd_pipe_in_flight(const wtamd_pipe *p) { return p ? p->in_flight : 0; }

int wtamd_pipe_set_compress(wtamd_pipe *p, int on) {
    if (!p) return wt_fail(WTAMD_ERR_ARG, "NULL argument");
    if (on && p->tile) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_set_compress: the Multiplexer tile cannot be compressed");
    p->compress = on != 0;
    return WTAMD_OK;
}

int wtamd_pipe_set_integrate(wtamd_pipe *p, int on) {
    if (!p) return wt_fail(WTAMD_ERR_ARG, "NULL argument");
    if (on && p->tile && p->cfg.n_tracks != 2) return wt_fail(WTAMD_ERR_ARG, "the fused Pearson integrator needs a Multiplexer of exactly two tracks");
    p->integrate = on != 0;
    return WTAMD_OK;
}

int wtamd_pipe_integrate_held(wtamd_pipe *p, double *integ) {
    WtDevGuard dev_guard_(p ? p->device : -1);
    if (!p || !integ) return wt_fail(WTAMD_ERR_ARG, "NULL argument");
    if (!p->held) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_integrate_held: no collected batch");
    WtSlot &s = p->slots[(size_t) p->tail];
    for (int k = 0; k < 6; k++) integ[k] = 0.0;
    if (!s.integrated) {
        // the device still holds the batch's runs (d_os / d_of / d_ov, the tile): integrate them there, now
        int rc = wt_pipe_enqueue_integ(p, s, p->s_comp);
        if (rc != WTAMD_OK) return rc;
        WT_HIP(hipEventRecord(s.e_patch, p->s_comp));
        rc = wt_wait_event(s.e_patch, "integrals of the held batch");
        if (rc != WTAMD_OK) return rc;
    }
    for (int k = 0; k < (p->tile ? 6 : 2); k++) integ[k] = s.h_integ[k];
    return WTAMD_OK;
}

int wtamd_pipe_set_map(wtamd_pipe *p, const wtamd_map_chain *chains) {
    WtDevGuard dev_guard_(p ? p->device : -1);
    if (!p) return wt_fail(WTAMD_ERR_ARG, "NULL argument");
    if (p->in_flight > 0 || p->acquired >= 0) return wt_fail(WTAMD_ERR_ARG, "wtamd_pipe_set_map: batches in flight");
    if (p->d_chains) { (void) wt_dev_free(p->d_chains); p->d_chains = nullptr; p->map_drops = false; p->map_f32 = false; }
    if (!chains) return WTAMD_OK;
    bool any = false;
    for (int t = 0; t < p->cfg.n_tracks; t++) any = any || chains[t].n_ops != 0;
    if (!any) return WTAMD_OK;
    return wt_map_upload_chains(chains, p->cfg.n_tracks, &p->d_chains, &p->map_drops, &p->map_f32);
}

void *wtamd_host_alloc(size_t bytes) {
    void *q = nullptr;
    if (wt_host_alloc(&q, bytes ? bytes : 1) != hipSuccess) return nullptr;
    return q;
}

void wtamd_host_free(void *q) {
    if (q) wt_host_free(q);
}

void wtamd_pool_trim(void) {
    std::vector<void *> host, dev;
    {
        std::lock_guard<std::mutex> lk(g_pinned_pool.mu);
        for (auto &kv : g_pinned_pool.free_list) { host.push_back(kv.second); g_pinned_pool.size_of.erase(kv.second); }
        g_pinned_pool.free_list.clear();
        g_pinned_pool.pooled = 0;
    }
    {
        std::lock_guard<std::mutex> lk(g_dev_pool.mu);
        for (auto &kv : g_dev_pool.free_list) { dev.push_back(kv.second); g_dev_pool.size_of.erase(kv.second); }
        g_dev_pool.free_list.clear();
        g_dev_pool.pooled = 0;
    }
    for (void *x : host) wt_pin_raw_free(x);
    for (void *x : dev) (void) hipFree(x);
}

void wtamd_pool_stats(int64_t out[6]) {
    if (!out) return;
    {
        std::lock_guard<std::mutex> lk(g_pinned_pool.mu);
        out[0] = (int64_t) g_pinned_pool.misses; out[1] = (int64_t) g_pinned_pool.miss_bytes; out[2] = (int64_t) g_pinned_pool.pooled;
    }
    std::lock_guard<std::mutex> lk(g_dev_pool.mu);
    out[3] = (int64_t) g_dev_pool.misses; out[4] = (int64_t) g_dev_pool.miss_bytes; out[5] = (int64_t) g_dev_pool.pooled;
}

int wtamd_pipe_get_stats(const wtamd_pipe *p, wtamd_pipe_stats *out) {
    if (!p || !out) return wt_fail(WTAMD_ERR_ARG, "NULL argument");
    *out = p->st;
    return WTAMD_OK;
}

}  // extern "C"

#endif  // WT_PIPE_H_
